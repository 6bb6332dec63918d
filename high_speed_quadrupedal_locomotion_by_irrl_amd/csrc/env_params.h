// env_params.h -- plain-old-data shared by the host side (C-ABI, config parser) and the kernels.
#pragma once
#include <stdint.h>

// Configuration scalars the kernels need, derived from the `environment:` YAML sub-tree
// (reference keys: Environment.hpp:1594-1659, VectorizedEnvironment.hpp:136-171).  Passed BY VALUE as
// a kernel argument; the env kernels name it in the kernarg segment and read a field with a scalar load
// where it is used (IRRL_BIND_ARGS, env_kernels.hip) -- 92 words + EnvState's 26 pointers do not fit the SGPRs.
struct EnvParams {
  int32_t n_envs;
  int32_t loop_count;      // int(control_dt / simulation_dt + 1e-10), Environment.hpp:711
  uint32_t seed;           // seedd
  uint32_t env_id_offset;  // [ext] EnvIdOffset: global id of this pool's env 0 -- every random draw is addressed by (seed, GLOBAL env id,
                           //       episode, step, purpose), so rank r of an N-GPU job (offset r * num_envs) owns exactly the robots r * num_envs ..
                           //       of the one big pool: results do not depend on how the envs are sharded over GPUs
  int32_t contact_iters;   // [ext] ContactIterations
  float sim_dt, control_dt, max_time;
  float abad, period, lam, stand_height, up_height_max;
  float Vx, Vy, Omega, lean_front, lean_hind;
  int32_t manual, height_variable, time_based_contact, wildcat, stochastic, obs_filter;
  int32_t shared_noise, randomize_per_episode;
  float c_term, c_ee, c_pos, c_att, c_joint, c_vel, c_torque, c_contact;
  float kp[3], kd[3];      // abad / hip / knee gains (AbadRatio folded in), Environment.hpp:338-350
  float filter_para;       // Environment.hpp:396 (evaluated with the base-class control_dt_ = 0.01)
  float obs_filter_alpha;  // Environment.hpp:423-427 (same quirk)
  float action_noise, obs_noise;
  float tau_max, w_crit, w_max;
  float phase[4];          // Environment.hpp:398-409
  float max_len;           // Environment.hpp:395
  float contact_tol;       // [ext] ContactTolerance: early exit of the contact sweeps (0 = fixed sweep count)
  int32_t contact_jacobi;  // [ext] ContactSolver bit 1: set = all contacts of a robot update simultaneously in a sweep, clear = Gauss-Seidel FR, FL, HR, HL
  int32_t contact_rule;    // [ext] ContactSolver bit 0: set = the published per-contact rule of RaiSim's solver (maximum dissipation on the cone
                           //       boundary, Hwangbo et al. 2018), clear = the build's first sliding rule (along the sticking impulse)
  int32_t contact_exit;    // [ext] ContactExit (simultaneous sweeps only): 1 = leave the sweep loop BEFORE a sweep whose change is predicted to be
                           //       below ContactTolerance (the last change dlambda moves contact l's velocity by dc_l = sum_{p != l} G_lp dlambda_p,
                           //       answered by -G_ll^-1 dc_l), 0 = after a sweep whose own change was (the confirming sweep of rounds 1-3)
  float clamp_r;           // tau_max / (w_max - w_crit)            (Environment.hpp:1279)
  float clamp_inv_den;     // 1 / (-w_max + w_crit)                 (Environment.hpp:1296-1297)
  // height field (Terrain: True, Environment.hpp:254-264); height == nullptr / terrain == 0 means the plane z = 0
  int32_t terrain, hf_nx, hf_ny;
  float hf_x0, hf_y0, hf_inv_dx, hf_inv_dy;
  // cell coordinate of world x: x * hf_inv_dx + hf_fx, plus hf_ix0 WHOLE cells added to the integer index (-x0 / dx = hf_ix0 + hf_fx, 0 <= hf_fx < 1);
  // hf_xlo / hf_xhi: the table's edge in that shifted coordinate (irrl_host::derive_terrain_params)
  int32_t hf_ix0, hf_iy0;
  float hf_fx, hf_fy, hf_xlo, hf_xhi, hf_ylo, hf_yhi;
  float hf_max;            // highest sample of the height field (0 on the plane): pre-test of the trunk-box corner contacts
  const float *height;     // [hf_nx, hf_ny] row-major, shared by every robot of the pool
  // reference-trajectory mode (ManualTraj: False, Manual: False; Environment.hpp:17-21, 565-573, 972, 1100-1107, 1667-1671):
  // table [ref_rows, 30] f32 = theta 12 | theta_dot 12 | z | phase 2 | cmd 3, one row per control step
  int32_t ref_traj, ref_rows;
  int32_t state_disturbance;   // Manual + ForceDisturbance: periodic kick of the base state (Environment.hpp:912-940)
  int32_t disturb_every;       // int(period / control_dt * 10) evaluated in double like ENV:747 (in f32, 0.02 / 0.002 * 10 truncates to 99)
  const float *ref;
  // Crutial: True -- the meteorite (Environment.hpp:273-284, 731-740, 815-861)
  int32_t crutial;
  int32_t attack_every;        // int(5 * period / control_dt) evaluated in double like ENV:733
  float cube_num;              // CubeNum coincident spheres (cube_place_radius = 0, ENV:1976) carried as one of CubeNum x the mass
  // derived scalars (irrl_host::derive_params): reciprocals and products of the configuration that the per-step epilogue would
  // otherwise divide by in every lane (an IEEE f32 division is ~10 VALU instructions on gfx950)
  float inv_control_dt, inv_period, inv_lam, inv_one_minus_lam;
  float cmd_resample_p;    // 0.5 / (max_time / control_dt), Environment.hpp:1031
  float two_pi_over_period;
};

// Device-resident state pool, structure of arrays in the reference's natural row-major shapes so the
// diagnostics getters are plain copies.  N = n_envs.
struct EnvState {
  float *gc;            // [N,19]  x y z, quat wxyz, 12 joint angles            (RaiSim gc)
  float *gv;            // [N,18]  world lin vel, world ang vel, 12 joint rates   (RaiSim gv)
  float *ptarget_last;  // [N,12]
  float *torque_last;   // [N,12]  NORMALISED torque of the previous reward evaluation
  float *torque;        // [N,12]  last applied joint torque
  float *joint_ref, *joint_ref_last, *joint_dot_ref, *ee_ref;  // [N,12] each
  float *lam_w;         // [N,12]  contact impulses of the last substep, world components (warm start)
  int32_t *in_contact;  // [N,4]
  float *contact;       // [N,4]
  float *command, *command_filtered;  // [N,3]
  float *t0;            // [N]
  int32_t *frame_idx;   // [N]   control steps since reset; time = t0 + frame_idx * control_dt
  uint32_t *episode;    // [N]
  float *up_height;     // [N]
  float *material;      // [N,3]  mu, restitution, restitution threshold
  float *mass;          // [N,13]
  float *com;           // [N,39]
  float *thigh_dz;      // [N]
  float *ob;            // [N,35] unscaled observation (obDouble_)
  float *ob_last;       // [N,35]
  float *sphere;        // [N,9]  Crutial: True -- meteorite centre 3, velocity 3 (world), radius, mass, body type (0 STATIC / 1 DYNAMIC)
  // diagnostic counter, NOT part of the env state (own allocation, may be NULL): toe-substeps spent in the contact list
  // since the pool was created, per (env, leg); bench.py reads it around the timed region to prove the region was not free flight
  uint32_t *contact_count;  // [N,4]
};

// ContactIterations of every shipped configuration (and the default when the key is absent): a compile-time constant of the default pool's kernels
// (env_core.hpp IRRL_SOLVER_FIXED; the launcher checks the pool's value against it, irrl_env_abi.hip shipped_solver)
#define IRRL_SHIPPED_SWEEP_CAP 6
#define IRRL_SHIPPED_SUBSTEPS 8    /* control_dt / simulation_dt of every shipped configuration (Environment.hpp:711) */

// RNG purposes -- (purpose, slot) addresses every random draw; identical table in the oracle.
enum {
  IRRL_P_DR_MATERIAL = 1, IRRL_P_DR_MASS = 2, IRRL_P_DR_COM = 6, IRRL_P_DR_THIGH = 16,
  IRRL_P_RESET_TIME = 20, IRRL_P_RESET_CMD = 21, IRRL_P_RESET_JOINT = 22, IRRL_P_RESET_JOINT_IND = 23,
  IRRL_P_RESET_BASE = 29, IRRL_P_RESET_XY = 30,
  IRRL_P_ACTION_NOISE = 40, IRRL_P_OBS_JOINT = 44, IRRL_P_OBS_JVEL = 47, IRRL_P_OBS_NORMAL = 50, IRRL_P_CMD = 56, IRRL_P_DISTURB = 60 /* ..61 */
};
