/*
 * irrl_oracle.h -- CPU restatement ("oracle") of the FlexibleRobotRaisimGym env.step() hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may build, load or call anything in oracle/.  The product path
 * (high_speed_quadrupedal_locomotion_by_irrl_amd/csrc, HIP/gfx950) never links or calls it.
 *
 * Parity status
 *   - task math (PD law, torque-speed clamp, gait generator, IK, observation, 8 reward terms,
 *     termination, reset order): restated from the reference's open C++ and pinned against
 *     golden vectors captured from the reference's importable Python twins (tests/golden/).
 *   - rigid-body physics + contact (the reference calls closed-source RaiSim, ENV:768):
 *     NO STATE-LEVEL PIN (RaiSim cannot be built or run here, no oracle/_ref).  The formulation below
 *     is this build's own (documented in DESIGN.md); it is pinned against first-principles
 *     invariants (tests/test_oracle_physics.py), against an independent implementation of RaiSim's
 *     published per-contact rule (tests/test_contact_model_gap.py) and -- round 6 -- against the
 *     reference repository's own RaiSim recordings of the bp5_155 policy (Exp_Raw_Data/
 *     body-center-*.bin, decoded into tests/golden/raisim_body_logs.json by
 *     tools/gen_raisim_log_fixture.py): eleven closed-loop trajectories whose steady-state speed,
 *     height, attitude and ripple this oracle reproduces within the bounds of
 *     tests/parity_lib.py RAISIM_LOG_TOL (tests/test_raisim_logs.py).  A statistical pin, not a
 *     trajectory one: the harness that made the recordings is not in the repository.
 *     It is the f64 checker the f32 HIP kernels are compared with.
 *
 * Reference aliases (paths under /root/reference/IRRL/FlexibleRobotRaisimGym/flex_gym/env):
 *   ENV = env/BlackPanther_V55/Environment.hpp, VEC = VectorizedEnvironment.hpp,
 *   BASE = RaisimGymEnv.hpp, URDF = env/BlackPanther_V55/urdf/black_panther.urdf.
 *
 * The library is compiled twice: -DORC_REAL=double (liborc_f64.so, the reference's numeric
 * model: ENV:1917-1934 uses Eigen::VectorXd everywhere) and -DORC_REAL=float (liborc_f32.so,
 * used to separate "fp32 rounding" from "algorithm differs" when a GPU parity test fails).
 */
#ifndef IRRL_ORACLE_H
#define IRRL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Flat copy of the `environment:` YAML sub-tree (ENV:1594-1659 + VEC:136-171), filled by the
 * Python test harness (PyYAML) -- always doubles/ints regardless of ORC_REAL. */
typedef struct orc_cfg {
  /* vectorizer keys (VEC:146-171) */
  int32_t num_envs;
  int32_t num_threads;
  double simulation_dt;
  double control_dt;
  int32_t seedd;
  /* gait (ENV:1598-1613) */
  double abad, period, lam, stand_height, up_height, down_height, gait_step;
  double Vx, Vy, Omega, LeanFront, LeanHind;
  /* mode flags (ENV:1616-1629) */
  int32_t Terrain, Manual, Crutial, Filter, Camera, StochasticDynamics, HeightVariable;
  int32_t TimeBasedContact, ManualTraj, MotorDynamics, ObsFilter, WILDCAT, ForceDisturbance;
  int32_t Convert2Torque;
  /* reward (ENV:1632-1639) */
  double terminalRewardCoeff, EndEffectorRewardCoeff, BodyPosRewardCoeff, BodyAttitudeRewardCoeff;
  double JointRewardCoeff, VelRewardCoeff, TorqueCoeff, ContactCoeff;
  /* control (ENV:1643-1658) */
  double Stiffness, Stiffness_Low, AbadRatio, Damping, Freq, max_time;
  int32_t CubeNum;
  double FPS, ActionNoise, ObsNoise;
  int32_t GaitType;
  double MotorMaxTorque, MotorCriticalSpeed, MotorMaxSpeed;
  /* build-defined extensions (not reference keys; defaults documented in DESIGN.md) */
  int32_t ContactIterations;    /* Gauss-Seidel sweeps per substep (default 6) */
  int32_t SharedNoiseScalar;    /* 1: Eigen 12x1*12x1 pitfall => one shared factor (ENV:584,586,705) */
  int32_t RandomizePerEpisode;  /* 1: redo the ctor domain randomisation at every reset (config 5) */
  int32_t EnvIdOffset;          /* global id of env 0 of this pool: random draws are addressed by the GLOBAL env id (multi-GPU shards) */
  double ContactTolerance;      /* stop the sweeps once sum|dlambda|^2 <= tol^2 sum|lambda|^2 (0: always ContactIterations sweeps) */
  int32_t ContactSolver;        /* bit 1 = ORDER of the toe updates inside one sweep: set = the four toes at once (the kernels' one-solve-per-sweep),
                                 * clear = Gauss-Seidel over FR, FL, HR, HL;  bit 0 = per-contact RULE for a sliding contact: set = the published
                                 * rule of RaiSim's solver (Hwangbo, Lee, Hutter, RA-L 2018: the minimiser of the post-impact kinetic energy on
                                 * {cone boundary} x {v_n = target}, solve_contact_md below), clear = the build's first rule (slide along the sticking
                                 * impulse's tangential direction).  So 0 = GS + build rule, 1 = GS + published rule (the published method),
                                 * 2 = simultaneous + build rule, 3 = simultaneous + published rule (default of the shipped configs).
                                 * Trunk-box corners / meteorite: one pass of sequential impulses behind the toe iteration with the same rule. */
  int32_t ContactExit;          /* how ContactTolerance ends the simultaneous sweeps (ContactSolver bit 1 set; Gauss-Seidel keeps rule 0):
                                 * 0 = CONFIRMED: stop after a sweep whose OWN change was sum|dlambda|^2 <= tol^2 sum|lambda|^2 (rounds 1-3: the last
                                 *     sweep of every substep only confirms what the one before it already reached);
                                 * 1 = PREDICTED (default): stop BEFORE a sweep whose change is predicted to be that small -- the impulses' last
                                 *     change dlambda moves contact l's velocity by dc_l = sum_{p != l} G_lp dlambda_p, to which a contact answers with
                                 *     -G_ll^-1 dc_l (exactly, while it sticks; the cone only shortens the answer): stop once
                                 *     sum_l |G_ll^-1 dc_l|^2 <= tol^2 sum|lambda|^2.  One solve less per substep for the same stated tolerance. */
} orc_cfg;

typedef struct orc_env orc_env; /* opaque vector-env handle */

/* lifecycle (VEC:132-194) */
orc_env *orc_create(const orc_cfg *cfg);
void orc_destroy(orc_env *h);
/* reference-trajectory table of a ManualTraj: False pool (Environment.hpp:1895 set_ref), [rows, cols >= 30] f32 */
int orc_set_ref(orc_env *h, const float *table, int rows, int cols);
void orc_init(orc_env *h); /* ctor randomisation + first reset of every env (VEC:145-194) */
int orc_num_envs(const orc_env *h);
int orc_real_bytes(void); /* sizeof(ORC_REAL) of this build */

/* VEC:201-207, 209-212, 268-278 */
void orc_reset(orc_env *h, float *ob /* [N,35] */);
void orc_observe(orc_env *h, float *ob /* [N,35] */);
void orc_step(orc_env *h, const float *action /* [N,12] */, float *ob /* [N,35] */,
              float *reward /* [N] */, uint8_t *done /* [N] */, float *extra /* [N,6] */);
void orc_is_terminal(orc_env *h, uint8_t *done);
void orc_set_seed(orc_env *h, int seed);
void orc_set_control_dt(orc_env *h, double dt);

/* diagnostics getters (ENV:1317-1418), f32 rows like the reference's Eigen::Ref<EigenVec> */
void orc_origin_state(orc_env *h, float *out /* [N,41] */);
void orc_reference_state(orc_env *h, float *out /* [N,24] */);
void orc_joint_effort(orc_env *h, float *out /* [N,12] */);
void orc_generalized_force(orc_env *h, float *out /* [N,18] */);
void orc_inverse_mass_matrix(orc_env *h, float *out /* [N,324] column-major */);
void orc_nonlinear(orc_env *h, float *out /* [N,18] */);
void orc_set_contact_coeff(orc_env *h, const float *in /* [N,3] */);

/* full-precision state exchange for parity tests: the same flat layout the product exposes
 * through irrl_env_get_state/irrl_env_set_state (include/irrl_env.h).  ORC_STATE_DIM doubles. */
#define ORC_STATE_DIM 288
void orc_get_state(orc_env *h, double *out /* [N,ORC_STATE_DIM] */);
void orc_set_state(orc_env *h, const double *in /* [N,ORC_STATE_DIM] */);

/* --- unit-level entry points (always double in/out) used by the golden-vector tests --- */
void orc_solve_contact(int rule, const double G[9], const double c[3], const double n[3], double vstar, double mu, double lam[3]);
void orc_cubic_bezier(const double p0[3], const double pf[3], double s, double out[3]);
double orc_gauss(double x, double width, double height);
void orc_bezier2(const double p0[3], const double pf[3], double s, double h, double out[3]);
double orc_smooth_function(double phase, double slope, double lam);
double orc_smooth_function2(double phase, double slope, double lam);
double orc_sampling_reshape(double ratio);
/* theta is in/out (stale-value semantics of ENV:1703-1750); returns error bitmask */
int orc_inverse_kinematics(double x, double y, double z, double l_hip, double l_thigh, double l_calf,
                           double max_len, int is_right, double theta[3]);
/* one torque_clamp pass (ENV:1273-1312) */
void orc_torque_clamp(const double tau_in[12], const double qd[12], double tau_max, double w_crit,
                      double w_max, double tau_out[12], double upper[12], double lower[12]);
/* gait_generator_manual on explicit inputs (ENV:1756-1890); jointRefLast is in/out */
void orc_gait_reference(const orc_cfg *cfg, const double cmd_f[3], double t, int is_first,
                        double jointRefLast[12], double jointRef[12], double jointDotRef[12],
                        double eeRef[12], double *up_height_io);
/* obs scaling constants (ENV:375-393) */
void orc_obs_scaling(const orc_cfg *cfg, double mean[35], double std[35]);
/* GAE reverse scan (ppo2.py:554-568), [T,N] row-major float */
void orc_gae(int T, int N, const float *rewards, const float *values, const uint8_t *dones,
             const float *last_values, const uint8_t *last_dones, float gamma, float lam,
             float *adv, float *returns);
/* physics probes on explicit (gc, gv) with the nominal (un-randomised) model, world-frame gv:
 * M (18x18 row-major), nonlinear term h (18) such that M*gvdot + h = tau (+ J^T f). */
void orc_mass_matrix_world(const double gc[19], double M[324]);
void orc_nonlinear_world(const double gc[19], const double gv[18], double h[18]);
/* toe-frame world positions / velocities (4x3) */
void orc_toe_kinematics(const double gc[19], const double gv[18], double pos[12], double vel[12]);
/* counter-based RNG probe: 4 uniforms in [0,1) for (seed, env, episode, step, purpose) */
void orc_rng_u01(uint32_t seed, uint32_t env, uint32_t episode, uint32_t step, uint32_t purpose,
                 double out[4]);
/* algorithmic flop count of the last orc_step (per env, averaged); 0 unless built -DORC_COUNT_FLOPS */
double orc_last_step_flops(const orc_env *h);
/* height field of a Terrain: True pool (returns 0 and leaves `out` alone on flat ground); [nx, ny] row-major */
int orc_heightfield(const orc_env *h, float *out, int *nx, int *ny);
/* (height, nx, ny, nz) at world (x, y) */
void orc_terrain_sample(const orc_env *h, double x, double y, double out[4]);
/* mean number of contact sweeps per substep since creation (statistics for DESIGN.md) */
double orc_mean_contact_sweeps(const orc_env *h);
/* (trunk-box corner, substep) pairs in contact since orc_init, summed over the envs (ENV:242 collision body "body/0") */
long orc_box_hits(const orc_env *h);
/* Crutial: True -- ENV:1423-1436 GetSphereInfo: [N,4] = centre of the meteorite (world) and its radius; substeps with sphere-trunk contact */
void orc_sphere_info(orc_env *h, float *out /* [N,4] */);
long orc_sphere_hits(const orc_env *h);
/* tests: capture the toe contact problem of env `env_id` in every substep (-1: off) and read the last one back:
 * G [12,12] Delassus blocks (base components), cfree [4,3], unit normals [4,3], target normal speeds [4], the solved impulses
 * [4,3] and the active flags [4]; rows / columns of inactive toes are zero */
void orc_set_probe(orc_env *h, int env_id);
void orc_get_probe(const orc_env *h, double *G, double *cfree, double *n, double *vstar, double *lam, int *active);

#ifdef __cplusplus
}
#endif
#endif
