/*
 * irrl_env.h -- C-ABI of the MI355X-native BlackPanther vector environment (libirrl_env.so).
 *
 * This is the drop-in boundary for the hot path  FlexibleGymEnv.step()  of
 * WoodenJin/High_Speed_Quadrupedal_Locomotion_by_IRRL.  Every entry point names the reference
 * interface it replaces; paths are under IRRL/FlexibleRobotRaisimGym/flex_gym/env/ :
 *   PYB = raisim_gym.cpp (pybind11 module `_flexible_robot`, class `FlexibleGymEnv`)
 *   VEC = VectorizedEnvironment.hpp,  ENV = env/BlackPanther_V55/Environment.hpp
 *
 * Conventions
 *   - plain C: opaque handle, pointers and sizes only; no torch / pybind / Eigen types.
 *   - every function returns 0 on success, non-zero on failure (irrl_last_error() has the text);
 *     the reference has no error channel: it aborts on a missing YAML key (RaisimGymEnv.hpp:41-42).
 *   - "device pointer" entry points (suffix-less) take HIP device pointers and are stream-ordered on the
 *     stream given to irrl_env_set_stream (default: the null stream); they never synchronise.
 *     "_host" entry points take host pointers like the reference's numpy arrays
 *     (Eigen::Ref<RowMajor float/bool>, RaisimGymEnv.hpp:46-49), copy through a pinned staging buffer and
 *     return after the outputs are valid on the host.
 *   - the library needs a gfx950 GPU: irrl_env_create fails (returns NULL) when none is usable.
 *     There is no CPU fallback of any kind behind this ABI.
 *   - array shapes: ob [N,35] f32, action [N,12] f32, reward [N] f32, done [N] u8 (numpy bool),
 *     extraInfo [N,6] f32, all C-contiguous, caller-owned, written in place (RaisimGymVecEnv.py:16-20).
 */
#ifndef IRRL_ENV_H
#define IRRL_ENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct irrl_env irrl_env;

#define IRRL_OB_DIM 35      /* ENV:360 */
#define IRRL_ACTION_DIM 12  /* ENV:361 */
#define IRRL_EXTRA_DIM 6    /* ENV:942-950 */
#define IRRL_ORIGIN_STATE_DIM 41 /* ENV:1330-1334: gc 19 + gv 18 + contact 4 */

const char *irrl_last_error(void);
/* build identification: "gfx950;<git-describe or date>" */
const char *irrl_version(void);

/* PYB:16 ctor (std::string resourceDir, std::string cfg) -> VEC:132-138.  cfg_yaml is the dumped
 * `environment:` mapping (run_bp_v5.py:205-207).  device = HIP device ordinal.  NULL on failure. */
irrl_env *irrl_env_create(const char *resource_dir, const char *cfg_yaml, int device);
/* VEC:140-143 destructor */
void irrl_env_destroy(irrl_env *h);
/* PYB:17 init -> VEC:145-194 (constructor randomisation ENV:435-477 + first reset of every env) */
int irrl_env_init(irrl_env *h);
/* use `hip_stream` (a hipStream_t) for all later launches / async copies of this handle */
int irrl_env_set_stream(irrl_env *h, void *hip_stream);

/* PYB:28-31 */
int irrl_env_num_envs(const irrl_env *h);
/* lane layout of this pool's kernels: 16 (one DPP row per robot: while the pool's waves of four robots fit the device's SIMDs, <= 4096 envs
 * on an MI355X) or 4 (larger pools: one DPP quad per robot); build-defined, overridable with the environment variable IRRL_LANES_PER_ROBOT */
int irrl_env_lanes_per_robot(const irrl_env *h);
/* resident waves per SIMD the pool's kernels are compiled for: 1 (the whole register file of a SIMD for one wave), or 2 for 4-lane pools
 * with more waves than the device has SIMDs (> 16 384 envs on an MI355X); build-defined, overridable with IRRL_L4_WAVES=1|2 */
int irrl_env_waves_per_simd(const irrl_env *h);
int irrl_env_ob_dim(const irrl_env *h);
int irrl_env_action_dim(const irrl_env *h);
int irrl_env_extra_dim(const irrl_env *h);
/* PYB:18 getExtraInfoNames (VEC:191-193): name j in [0, 6); order fixed by this build */
const char *irrl_env_extra_name(const irrl_env *h, int j);

/* PYB:21,23 step -> VEC:268-278 (+ perAgentStep VEC:352-372, ENVIRONMENT::step ENV:692-809) */
int irrl_env_step(irrl_env *h, const float *action, float *ob, float *reward, uint8_t *done, float *extra);
int irrl_env_step_host(irrl_env *h, const float *action, float *ob, float *reward, uint8_t *done, float *extra);
/* build-defined: `count` consecutive steps from a device-resident action table [n_rows, N, 12] (step k takes row
 * (first_row + k) % n_rows), launched back to back on the env's stream by one call -- open-loop playback / benchmarking without a
 * host round trip per step.  Outputs as irrl_env_step: [N, .] arrays that every step overwrites (those of the last step survive). */
int irrl_env_step_rows(irrl_env *h, int count, const float *action_rows, int n_rows, int first_row, float *ob, float *reward,
                       uint8_t *done, float *extra);
/* the same `count` steps as ONE launch: a wave walks its own robots through all of them (robots never interact, VEC:273), so there is no
 * grid-wide boundary between steps and no launch per step.  States and outputs bit-identical to irrl_env_step_rows (which it falls back
 * to for pools with the meteorite or the build's first contact rule). */
int irrl_env_step_rows_persistent(irrl_env *h, int count, const float *action_rows, int n_rows, int first_row, float *ob, float *reward,
                                  uint8_t *done, float *extra);
/* the two calls above WITH EVERY STEP'S OUTPUTS KEPT: ob_rows [count, N, 35], reward_rows [count, N], done_rows [count, N] u8,
 * extra_rows [count, N, 6] (device); step k fills row k -- exactly what `count` calls of the reference's step() hand back one after the
 * other (VEC:268-278 fills ob / reward / done / extraInfo on every control step; RaisimGymVecEnv.py:26-52 copies them out), so the caller
 * of the K-step entry point gets the trajectory it simulated.  Rows bit-identical to `count` irrl_env_step calls. */
int irrl_env_step_rows_out(irrl_env *h, int count, const float *action_rows, int n_rows, int first_row, float *ob_rows, float *reward_rows,
                           uint8_t *done_rows, float *extra_rows);
int irrl_env_step_rows_persistent_out(irrl_env *h, int count, const float *action_rows, int n_rows, int first_row, float *ob_rows,
                                      float *reward_rows, uint8_t *done_rows, float *extra_rows);
/* PYB:24 testStep -> VEC:280-290: env 0 only in the reference (visual eval); here it steps env 0 only and
 * leaves rows 1.. of the outputs untouched (headless: no rendering). */
int irrl_env_test_step_host(irrl_env *h, const float *action, float *ob, float *reward, uint8_t *done, float *extra);
/* PYB:19 reset -> VEC:201-207 (all envs, then observe) */
int irrl_env_reset(irrl_env *h, float *ob);
int irrl_env_reset_host(irrl_env *h, float *ob);
/* PYB:20 observe -> VEC:209-212 */
int irrl_env_observe(irrl_env *h, float *ob);
int irrl_env_observe_host(irrl_env *h, float *ob);
/* PYB:26 isTerminalState -> VEC:314-319 */
int irrl_env_is_terminal(irrl_env *h, uint8_t *done);
int irrl_env_is_terminal_host(irrl_env *h, uint8_t *done);
/* PYB:22 setSeed -> VEC:308-312.  The reference seeds the process-global std::srand per env
 * (last call wins); here it re-keys the counter RNG for all later draws. */
int irrl_env_set_seed(irrl_env *h, int seed);
/* PYB:27-28 setSimulationTimeStep / setControlTimeStep -> VEC:321-329 */
int irrl_env_set_simulation_dt(irrl_env *h, double dt);
int irrl_env_set_control_dt(irrl_env *h, double dt);
/* PYB:25 close (no-op, ENV:1585), PYB:36 curriculumUpdate (no-op, RaisimGymEnv.hpp:76) */
int irrl_env_close(irrl_env *h);
int irrl_env_curriculum_update(irrl_env *h);

/* diagnostics getters, host pointers (PYB:37-45; RaisimGymVecEnv.py:54-93) */
int irrl_env_origin_state_host(irrl_env *h, float *out /* [N,41]  ENV:1317-1325 */);
int irrl_env_reference_state_host(irrl_env *h, float *out /* [N,24] ENV:1339-1345 jointRef, jointDotRef */);
int irrl_env_joint_effort_host(irrl_env *h, float *out /* [N,12] ENV:1350-1358 */);
int irrl_env_generalized_force_host(irrl_env *h, float *out /* [N,18] ENV:1363-1370 */);
int irrl_env_inverse_mass_matrix_host(irrl_env *h, float *out /* [N,324] column-major, ENV:1375-1391 */);
int irrl_env_nonlinear_host(irrl_env *h, float *out /* [N,18] ENV:1396-1402 */);
int irrl_env_set_contact_coeff_host(irrl_env *h, const float *in /* [N,3] mu, e, thr; ENV:1407-1418 */);
/* PYB:46 / VEC:253-256 GetSphereInfo (ENV:1423-1436): centre (world) and radius of the meteorite of a `Crutial: True` pool;
 * an error on a pool without it (the reference prints "Please make sure the [Flag_Crucial] is True") */
int irrl_env_sphere_info_host(irrl_env *h, float *out /* [N,4] */);

/* Full state exchange (build-defined, for checkpoint/parity): flat [N, IRRL_STATE_DIM] doubles per env,
 * fields at the IRRL_S_* offsets below. */
#define IRRL_STATE_DIM 288
enum {
  IRRL_S_GC = 0, IRRL_S_GV = 19, IRRL_S_PTARGET_LAST = 37, IRRL_S_TORQUE_LAST = 49, IRRL_S_TORQUE = 61,
  IRRL_S_JOINT_REF = 73, IRRL_S_JOINT_REF_LAST = 85, IRRL_S_JOINT_DOT_REF = 97, IRRL_S_EE_REF = 109,
  IRRL_S_COMMAND = 121, IRRL_S_COMMAND_FILTERED = 124, IRRL_S_T0 = 127, IRRL_S_FRAME = 128, IRRL_S_EPISODE = 129,
  IRRL_S_UP_HEIGHT = 130, IRRL_S_CONTACT = 131, IRRL_S_LAMBDA_W = 135, IRRL_S_IN_CONTACT = 147,
  IRRL_S_MATERIAL = 151, IRRL_S_MASS = 154, IRRL_S_COM = 167, IRRL_S_THIGH_DZ = 206, IRRL_S_OB = 207,
  IRRL_S_OB_LAST = 242,
  IRRL_S_SPHERE = 277 /* Crutial: meteorite centre 3, velocity 3, radius, mass, body type (0 static / 1 dynamic) */, IRRL_S_END = 286
};
int irrl_env_get_state_host(irrl_env *h, double *out);
int irrl_env_set_state_host(irrl_env *h, const double *in);
/* ENV:1895 set_ref / VEC:158-176: the reference-trajectory table of a `ManualTraj: False` pool, [rows, cols >= 30] row-major
 * f32 (theta 12 | theta_dot 12 | z | phase 2 | cmd 3 per control step, Environment.hpp:17-21).  create() loads the CSV named
 * by cfg["RefTraj"] when it is readable (VectorizedEnvironment.hpp:33-76 format); otherwise call this before init(). */
int irrl_env_set_ref_host(irrl_env *h, const float *table, int rows, int cols);
/* the shared height field of a `Terrain: True` pool (Environment.hpp:254-264), [nx, ny] row-major f32; out may be NULL
 * to query the shape only; returns non-zero on flat ground */
int irrl_env_heightfield_host(irrl_env *h, float *out, int *nx, int *ny);
/* device-side snapshot of the whole state pool / its restoration, stream-ordered on the pool's stream: a caller can run
 * throw-away steps (the warm-up in front of a hipGraph capture of the rollout) and continue from where it was */
int irrl_env_snapshot(irrl_env *h);
int irrl_env_restore(irrl_env *h);
/* diagnostic counters summed over the pool (the role of the per-env members itera / contact list sizes a reference user would
 * print, Environment.hpp:554, 1199-1243): out[0] = episodes started (init + every reset), out[1] = toe-substeps spent in the
 * contact list since create(), out[2] = sum of frame_idx.  Synchronises the pool's stream.  bench.py differences them around
 * its timed region to show the region was not free flight. */
int irrl_env_counters_host(irrl_env *h, unsigned long long *out);
/* the same three sums written to d_out[3] (device memory), stream-ordered on the pool's stream, no synchronisation */
int irrl_env_counters(irrl_env *h, unsigned long long *d_out);
/* value of a numeric/bool config key as parsed by the library (tests the YAML reader); NaN if absent */
double irrl_env_cfg_value(const irrl_env *h, const char *key);

/* ---- learner-side device ops on the same path (ppo2.py:554-568): GAE reverse scan ---- */
/* rewards/values/adv/returns [T,N] f32 row-major, dones [T,N] u8 (flag stored BEFORE step t, ppo2.py:526),
 * last_values [N], last_dones [N]; device pointers, stream-ordered on `hip_stream`. */
int irrl_gae(int T, int N, const float *rewards, const float *values, const uint8_t *dones, const float *last_values,
             const uint8_t *last_dones, float gamma, float lam, float *adv, float *returns, void *hip_stream);

/* PPO2 clipped-surrogate loss of a diagonal-Gaussian policy, forward and backward in one pass (ppo2.py:152-175): M samples,
 * mean / actions [M, act], logstd [act], vpred / returns / old_values / old_neglogp [M], adv_stats = (mean, std) of the raw
 * advantages returns - old_values (device; all-reduced first when there are several ranks).  Writes d loss / d mean [M, act] and
 * d loss / d vpred [M] (loss = pg - ent_coef * entropy + vf_coef * vf; the constant entropy term is the caller's) and per-block
 * partial sums [n_blocks, 4 + act] = pg loss, vf loss, approx KL, clip fraction, d loss / d logstd (pg part); act = 12. */
int irrl_ppo_loss(size_t M, int act_dim, const float *mean, const float *logstd, const float *vpred, const float *actions,
                  const float *returns, const float *old_values, const float *old_neglogp, const float *adv_stats, float cliprange,
                  float vf_coef, float *d_mean, float *d_vpred, float *partials, int n_blocks, void *hip_stream);

/* the policy / value heads of CustomLSTMPolicy (run_bp_v5.py:169-176: mean = h_pi W_pi + b_pi, v = h_v w_v + b_v) TOGETHER with the
 * loss above, forward and backward in one pass: h_pi / h_v [M, hid] are the two stacks' last-layer outputs over the rollout.
 * Writes d loss / d h_pi, d loss / d h_v [M, hid] (and, if not NULL, mean [M, act] / value [M]) and per-block partial sums
 * [n_blocks, 4 + act + act + 1 + hid + hid * act] = (pg, vf, kl, clipfrac) | d logstd | d b_pi | d b_v | d w_v | d W_pi [hid][act].
 * hid = 48, act = 12. */
int irrl_ppo_heads_loss(size_t M, int act_dim, int hid, const float *h_pi, const float *h_v, const float *pi_w, const float *pi_b, const float *vf_w,
                        const float *vf_b, const float *logstd, const float *actions, const float *returns, const float *old_values,
                        const float *old_neglogp, const float *adv_stats, float cliprange, float vf_coef, float *d_hpi, float *d_hv, float *mean_out,
                        float *value_out, float *partials, int n_blocks, void *hip_stream);

/* MlpPolicy (archi/policies.py:430-446: [64, 64] tanh stacks for policy and value over the 35 observations; the learner of BASELINE
 * config 2) under the PPO2 loss above: forward AND backward of one network over one minibatch in one launch -- what ppo2.py:243-298
 * (`_train_step`) evaluates through the TensorFlow graph.  kind 0 = policy network (w3 [64, 12]; uses actions, old_neglogp, logstd and
 * the normalised advantages), kind 1 = value network (w3 [64, 1]; clipped value loss times vf_coef).  idx [n] (int64, device) selects
 * the minibatch's rows of the flat rollout arrays (NULL: rows 0..n-1); nothing is gathered.  Weights are [in, out] row-major.
 * Writes per-workgroup partial sums [n_blocks, irrl_mlp_ppo_partial_len()], to be added up by the caller (fixed order); a row is
 * scalars[4] (kind 0: pg loss, approx KL, clip fraction; kind 1: value loss -- sums over samples, divide by n) | d logstd[16] (pg part) |
 * d b1[64] | d b2[64] | d b3[16] | d W1[48][64] (rows >= 35 are zero) | d W2[64][64] | d W3[64][16] (columns >= act / 1 are zero),
 * gradients of loss = pg - ent_coef * entropy + vf_coef * vf with the means taken over the n samples. */
int irrl_mlp_ppo_grads(int kind, size_t n, const int64_t *idx, int ob_dim, int hid, int act_dim, const float *obs, const float *actions,
                       const float *returns, const float *old_values, const float *old_neglogp, const float *w1, const float *b1, const float *w2,
                       const float *b2, const float *w3, const float *b3, const float *logstd, const float *adv_stats, float cliprange, float vf_coef,
                       float *partials, int n_blocks, void *hip_stream);
int irrl_mlp_ppo_partial_len(void);
/* The same gradients with every matrix product formed as THREE bf16 plane products on the matrix cores (each operand split into two
 * bf16 planes, f32 accumulation: ~2^-16 relative per product, csrc/mlp_bf16.hpp): same arguments, same partial-sum rows, ~3x faster.
 * What the learner uses by default; irrl_mlp_ppo_grads stays the exact-f32 form. */
int irrl_mlp_ppo_grads_bf16(int kind, size_t n, const int64_t *idx, int ob_dim, int hid, int act_dim, const float *obs, const float *actions,
                            const float *returns, const float *old_values, const float *old_neglogp, const float *w1, const float *b1, const float *w2,
                            const float *b2, const float *w3, const float *b3, const float *logstd, const float *adv_stats, float cliprange, float vf_coef,
                            float *partials, int n_blocks, void *hip_stream);

/* PACKED SAMPLE RECORDS (round 5).  A minibatch row is a random sample of the flat rollout (ppo2.py:364-380), so each of the five arrays above
 * costs whole 128-byte lines per sample (3.8x / 3.3x the payload, PMC).  irrl_mlp_pack_records builds, once per update, one 256-byte record per
 * sample -- rec [n, irrl_mlp_record_floats() = 64]: words [0, 35) observation | [36, 48) action | 48 return | 49 old value | 50 old neglogp |
 * 51 return - old value | the rest zero; rec 256-byte aligned -- and irrl_mlp_ppo_grads_bf16_rec / irrl_adv_moments_rec read the minibatch's rows
 * out of it: two lines per sample, same values, bit-identical results. */
int irrl_mlp_pack_records(size_t n, const float *obs, const float *actions, const float *returns, const float *old_values, const float *old_neglogp,
                          float *rec, void *hip_stream);
int irrl_mlp_record_floats(void);
int irrl_mlp_ppo_grads_bf16_rec(int kind, size_t n, const int64_t *idx, const float *rec, const float *w1, const float *b1, const float *w2,
                                const float *b2, const float *w3, const float *b3, const float *logstd, const float *adv_stats, float cliprange,
                                float vf_coef, float *partials, int n_blocks, void *hip_stream);
int irrl_adv_moments_rec(size_t n, const int64_t *idx, const float *rec, double *scratch, int n_blocks, double *sums, float *stats, void *hip_stream);

/* moments of the raw advantages a = returns[r] - old_values[r] of one minibatch (ppo2.py:262-263 normalises them per minibatch):
 * sums[3] = (sum a, sum a^2, n) as doubles, rows through idx [n] (int64, device) or 0..n-1 (NULL), fixed summation order.
 * old_values may be NULL: `returns` then holds the advantages themselves (formed once per update: one gathered array per minibatch instead of two).
 * scratch: 2 * n_blocks doubles (device).  stats (device, may be NULL): also (mean, population std) of a as two floats -- the adv_stats
 * argument of the loss kernels when the job has one rank (several ranks all-reduce `sums` first). */
int irrl_adv_moments(size_t n, const int64_t *idx, const float *returns, const float *old_values, double *scratch, int n_blocks, double *sums,
                     float *stats, void *hip_stream);

/* ---- tail of one PPO2 optimizer step on FLAT buffers (ppo2.py:182-197: tf.clip_by_global_norm(max_grad_norm) then
 * tf.train.AdamOptimizer(lr, epsilon).apply_gradients; kernels csrc/ppo_optim.hpp).  theta / grad / m / v: [n] floats (device,
 * 16-byte aligned), the parameters, their gradient and the Adam moments; every parameter tensor of the policy is a view of theta.
 * g = grad_scale * grad (1 / world after the all-reduce sum), clipped to max_norm by its global 2-norm (max_norm <= 0: no clip),
 * then Adam step number `step` (1-based; bias corrections evaluated in double on the host).  ONE launch, fixed summation order:
 * the same inputs give the same bits on every rank.  norm_out (device, may be NULL) receives the unclipped norm of g. */
int irrl_clip_adam(int n, float *theta, const float *grad, float *m, float *v, float grad_scale, float max_norm, float lr, float beta1, float beta2,
                   float eps, long long step, float *norm_out, void *hip_stream);

/* out[map[m][c]] = (add ? add[m][c] : 0) + sum over r of part[m][r][c]  for the nmat matrices part [nmat, rows, cols] of per-workgroup
 * partial sums (fixed order); map [nmat, cols] int32 (device), entries < 0 skip the column.  Used to drop the MlpPolicy gradient kernels'
 * partial rows straight into the flat gradient buffer in parameter layout. */
int irrl_sum_rows_scatter(const float *part, int nmat, int rows, int cols, const int *map, const float *add, float *out, void *hip_stream);

/* out[i] = E(i), i < n: a keyed random permutation of 0 .. n-1 (int64, device) in ONE launch and without a sort -- the shuffled sample order of
 * an optimisation epoch (ppo2.py:364-380).  E = 4-round Feistel network on the bits of n - 1 with cycle walking; depends on (n, seed, counter)
 * only; `ppo2.feistel_permutation` is the numpy twin. */
int irrl_random_permutation(long long n, unsigned seed, unsigned counter, long long *out, void *hip_stream);

/* synthetic action stream of the benchmark (a = clip(sigma N(0,1), -1, 1) from Philox(seed, stream = env,
 * counter = step)): fills out[n_steps][n_envs][12] (device) for envs env0 .. and steps step0 ..; values depend only on
 * (seed, global env id, step), not on the shape of the request.  tests/ hold the numpy twin. */
int irrl_bench_actions(unsigned seed, int env0, int n_envs, long long step0, int n_steps, float sigma, float *out, void *hip_stream);

/* PMC calibration helper: copies n floats with one dword per lane (the env kernels' access width) so that
 * FETCH_SIZE / WRITE_SIZE can be calibrated on a known byte count (MI355X_MICROARCH.md, HBM section). */
int irrl_calib_copy_dword(const float *src, float *dst, size_t n, void *hip_stream);

/* ---- persistent LSTM sequence kernels for the PPO2 update (stable-baselines lstm of CustomLSTMPolicy,
 * run_bp_v5.py:143-176; the train graph unrolls all n_steps, ppo2.py:132-134).  Device pointers, f32.
 * Gate columns are in [unit][gate] order (gate = i,f,o,g): zx/gates/dz [T,N,hid,4], cseq/hseq/dh_in [T,N,hid],
 * masks [T,N] (1.0 = done before step t), state0/state_out [N,2*hid] = [c|h], wh_p [hid][hid][4].
 * hid in {32,48,64}, N % 16 == 0.  Returns 0, or 1 for an unsupported shape, 2 for a launch error. */
int irrl_lstm_seq_forward(int hid, int T, int N, const float *zx, const float *wh_p, const float *masks, const float *state0,
                          float *gates, float *cseq, float *hseq, float *state_out, void *hip_stream);
/* the same forward pass with the input projection fused in (no zx tensor): x [T,N,n_in] is the layer input,
 * wx_p [n_in][hid][4], b_p [hid][4]; n_in <= 48 */
int irrl_lstm_seq_forward_x(int hid, int T, int N, int n_in, const float *x, const float *wx_p, const float *b_p, const float *wh_p,
                            const float *masks, const float *state0, float *gates, float *cseq, float *hseq, float *state_out,
                            void *hip_stream);
int irrl_lstm_seq_backward(int hid, int T, int N, const float *gates, const float *cseq, const float *masks, const float *state0,
                           const float *dh_in, const float *wh_p, float *dz, void *hip_stream);

/* out[c] = sum over rows of part[rows, cols] in one fixed order (the per-workgroup partial sums the gradient kernels leave);
 * unit_gate_hid > 0 also undoes the [unit][gate] column permutation inside each group of 4 * hid columns (out column g * hid + u
 * <- partial column 4 u + g).  Returns 0, 1 = bad shape, 2 = launch error. */
int irrl_sum_rows(const float *part, int rows, int cols, int unit_gate_hid, float *out, void *hip_stream);

/* backward pass with everything that consumes dz fused in (no dz tensor, no separate weight-gradient GEMMs):
 * hseq / x are the forward outputs / inputs; dx [T,N,n_in] or NULL; dwx_part [N/16, n_in, 4 hid], dwh_part [N/16, hid,
 * 4 hid], db_part [N/16 * 4, 4 hid] receive per-workgroup partial sums (permuted gate columns) that the caller adds up
 * over their first axis.  n_in <= 48. */
int irrl_lstm_seq_backward_x(int hid, int T, int N, int n_in, const float *gates, const float *cseq, const float *hseq, const float *x,
                             const float *masks, const float *state0, const float *dh_in, const float *wh_p, const float *wx_p,
                             float *dx, float *dwx_part, float *dwh_part, float *db_part, void *hip_stream);

/* the same two sequence kernels on the bf16 matrix cores with COMPENSATED OPERAND SPLITS and f32 accumulation (csrc/lstm_bf16.hpp; hid 48,
 * n_in <= 48): every operand is split into nsplit bf16 planes (2: ~2^-16 relative product error, 3: ~2^-24, the f32 level) and a product is the
 * sum of the plane products above that weight (3 resp. 6 MFMAs); v_mfma_f32_16x16x32_bf16 covers 32 units of K in half the time the exact-f32
 * v_mfma_f32_16x16x4_f32 needs for 4.  Tensors, layouts and return codes as irrl_lstm_seq_forward_x / irrl_lstm_seq_backward_x. */
/* (forward: gates == NULL and cseq == NULL selects the INFERENCE form -- only hseq and state_out are written: the critic pass behind an actor-only rollout;
 * gates == NULL with cseq given: c and h are kept, the gates are not -- the forward half of a backward pass that RECOMPUTES them.
 * backward: gates == NULL (nsplit 2 only; b_p = the layer's permuted bias, otherwise unused and may be NULL) runs the kernel that forms
 * z_t = b + [h_{t-1} keep_t | x_t] [wh ; wx] itself from the h / x tiles it stages for the weight gradients anyway -- the forward kernel's products,
 * plane for plane and in its order, so every output equals the gate-loading kernel's bit for bit; round 6) */
int irrl_lstm_seq_forward_bf16(int nsplit, int hid, int T, int N, int n_in, const float *x, const float *wx_p, const float *b_p, const float *wh_p,
                               const float *masks, const float *state0, float *gates, float *cseq, float *hseq, float *state_out, void *hip_stream);
int irrl_lstm_seq_backward_bf16(int nsplit, int hid, int T, int N, int n_in, const float *gates, const float *cseq, const float *hseq, const float *x,
                                const float *masks, const float *state0, const float *dh_in, const float *wh_p, const float *wx_p, const float *b_p,
                                float *dx, float *dwx_part, float *dwh_part, float *db_part, void *hip_stream);

/* ---- one ROLLOUT step of CustomLSTMPolicy in a single launch (run_bp_v5.py:178-185 `step`; the runner's clip and
 * buffer rows, ppo2.py:521-535).  Two stacks (actor, critic) of two LSTM layers of `hid` units, heads pi [hid,act],
 * vf [hid,1], logstd [act].  lstm_w is a HOST array of 12 device pointers: for layer in (pi0, pi1, v0, v1):
 * wx_p [n_in][hid][4], wh_p [hid][hid][4], b_p [hid][4] (gate columns in [unit][gate] order).  states [N, 8 hid] =
 * pi0 [c|h], pi1 [c|h], v0 [c|h], v1 [c|h]; states_out may alias states_in.
 * Sampling noise: `noise` [N,act] ~ N(0,1) if not NULL; else rng_on = 1 draws it in the kernel from the engine's
 * counter RNG (Philox4x32-10, key (rng_seed,'IRR1'), counter (env, step >> 32, step, 0x50 + a/4), Box-Muller on the
 * pairs (u0,u1), (u2,u3)) with step = rng_step + *rng_base (rng_base: device int64 scalar or NULL); else
 * deterministic (action = mean).
 * Outputs action (unclipped sample), clipped (to [-1,1]), value [N], neglogp [N].  With row >= 0 also row `row` of
 * mb_obs [T,N,ob], mb_actions [T,N,act], mb_values / mb_neglogp [T,N], mb_dones [T,N] u8, and, if mb_rewards and
 * prev_reward [N] are given and row > 0, row-1 of mb_rewards (the reward of the previous env step).
 * Any N (workgroups of 16 envs, the last one partially filled), hid in {32,48,64}, 16 act + 16 <= 8 hid. */
int irrl_lstm_policy_step(int hid, int ob_dim, int act_dim, int N, const float *obs, const uint8_t *dones, const float *states_in,
                          float *states_out, const float *const *lstm_w, const float *pi_w, const float *pi_b, const float *vf_w,
                          const float *vf_b, const float *logstd, const float *noise, int rng_on, unsigned rng_seed, long long rng_step,
                          const long long *rng_base, int env_id_offset, float *action, float *clipped, float *value, float *neglogp, long long row, float *mb_obs,
                          float *mb_actions, float *mb_values, float *mb_neglogp, uint8_t *mb_dones, float *mb_rewards,
                          const float *prev_reward, void *hip_stream);

/* (env_id_offset in the three policy entry points: global id of env 0 -- the in-kernel sampling noise of env e is the counter RNG's
 * stream e + env_id_offset, so an N-GPU job whose ranks pass rank * N draws exactly the noise of the one big pool.)
 * `steps` consecutive rollout steps (policy step t, then env.step on its clipped action) launched back to back by ONE call on
 * `hip_stream`: step k runs irrl_lstm_policy_step with rng_step + k, row + k and noise + k N act (if noise is given: a
 * [steps, N, act] table), states updated in place (states_out may equal states_in), and then the env step of pool `env`
 * (N = its num_envs) that writes obs / dones IN PLACE (the arrays the next policy step reads), the reward to `env_reward` [N]
 * (= prev_reward of the next policy step) and the extras to `env_extra` [N,6].  What ppo2.Runner used to record as a hipGraph of
 * 2 x steps kernel nodes (same speed, no capture).  fuse != 0: env.step k and the policy step k + 1 run as ONE launch (a workgroup =
 * the four env waves of 16 robots = one MFMA M-tile; 16-lane layout, hid 48, no Crutial; otherwise ignored) -- bit-identical
 * results, measured slower than the two-launch sequence on MI355X.  fuse == 2: the whole rollout as ONE persistent launch (a workgroup
 * loops over all `steps` for its 16 robots: no grid-wide boundary between steps; same conditions, otherwise the two-launch sequence) --
 * bit-identical results again.  fuse == 3 (round 5): that persistent launch with the CRITIC OFF THE PER-STEP PATH -- V(s_t) depends on the observation
 * history only and nothing in the rollout needs it before GAE, so the per-step part runs the actor stack alone (all of its operands resident in LDS) and
 * writes everything EXCEPT `value` / `mb_values` and the critic's half of `states`, which the caller obtains for the whole rollout afterwards with the
 * sequence kernels over the recorded observations (ppo2.Runner does); actor-side buffers bit-identical to the other modes; an error where the persistent
 * kernel is not instantiated. */
int irrl_lstm_rollout(irrl_env *env, int steps, int hid, int ob_dim, int act_dim, float *obs, uint8_t *dones, const float *states_in,
                      float *states_out, const float *const *lstm_w, const float *pi_w, const float *pi_b, const float *vf_w,
                      const float *vf_b, const float *logstd, const float *noise, int rng_on, unsigned rng_seed, long long rng_step,
                      const long long *rng_base, int env_id_offset, float *action, float *clipped, float *value, float *neglogp, long long row, float *mb_obs,
                      float *mb_actions, float *mb_values, float *mb_neglogp, uint8_t *mb_dones, float *mb_rewards,
                      float *env_reward, float *env_extra, int fuse, void *hip_stream);

/* Which `fuse` modes of irrl_lstm_rollout exist for THIS pool and a network of `hid` units per LSTM layer: 1 = the mode runs as described above,
 * 0 = it does not (fuse 1 / 2 then run as two launches per step inside the call; fuse 3 is refused with an error, because its caller has to
 * evaluate the critic itself and must know beforehand), -1 = bad handle.  Callers branch on this, not on the text of irrl_last_error(). */
int irrl_lstm_rollout_supports(irrl_env *env, int hid, int fuse);

/* the same single-launch rollout step for MlpPolicy (flex_gym/archi/policies.py:430-446: separate pi / vf nets of two tanh
 * layers of `hid` = 64 units).  mlp_w: HOST array of 8 device pointers pi_w1 [ob][hid], pi_b1, pi_w2 [hid][hid], pi_b2,
 * vf_w1, vf_b1, vf_w2, vf_b2; heads, sampling, outputs and rollout rows as above; act <= 15. */
int irrl_mlp_policy_step(int hid, int ob_dim, int act_dim, int N, const float *obs, const uint8_t *dones, const float *const *mlp_w,
                         const float *pi_w, const float *pi_b, const float *vf_w, const float *vf_b, const float *logstd, const float *noise,
                         int rng_on, unsigned rng_seed, long long rng_step, const long long *rng_base, int env_id_offset, float *action, float *clipped,
                         float *value, float *neglogp, long long row, float *mb_obs, float *mb_actions, float *mb_values, float *mb_neglogp,
                         uint8_t *mb_dones, float *mb_rewards, const float *prev_reward, void *hip_stream);

/* `steps` consecutive rollout steps of MlpPolicy (policy step t, then env.step on its clipped action) from ONE call -- the MlpPolicy twin of
 * irrl_lstm_rollout (no recurrent state; mlp_w as in irrl_mlp_policy_step; ob 35, act 12, hid 64).  fuse == 2 and a pool the combined
 * kernel exists for (16 lanes per robot, Crutial off, published contact rule): the whole rollout as ONE persistent launch -- a workgroup
 * keeps its 16 robots and the policy's weights (in LDS) for all `steps`, no grid-wide boundary between steps; otherwise 2 x steps launches
 * back to back.  Bit-identical buffers either way (what ppo2.Runner records as a hipGraph of 2 x steps nodes otherwise). */
int irrl_mlp_rollout(irrl_env *env, int steps, int hid, int ob_dim, int act_dim, float *obs, uint8_t *dones, const float *const *mlp_w, const float *pi_w,
                     const float *pi_b, const float *vf_w, const float *vf_b, const float *logstd, const float *noise, int rng_on, unsigned rng_seed,
                     long long rng_step, const long long *rng_base, int env_id_offset, float *action, float *clipped, float *value, float *neglogp,
                     long long row, float *mb_obs, float *mb_actions, float *mb_values, float *mb_neglogp, uint8_t *mb_dones, float *mb_rewards,
                     float *env_reward, float *env_extra, int fuse, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
