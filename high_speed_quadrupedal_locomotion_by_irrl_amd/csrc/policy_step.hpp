// policy_step.hpp -- device code of the single-launch rollout step of the reference's CustomLSTMPolicy, shared by
//   * lstm_policy_step_kernel (lstm_kernels.hip): the step alone, a workgroup of 2 HID/16 waves per 16 envs, and
//   * irrl_step_policy_kernel (env_kernels.hip): env.step of 16 robots and, behind a barrier, the policy step on the
//     observations those robots just produced -- one launch per rollout step instead of two.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef IRRL_LSTM_COMMON
#define IRRL_LSTM_COMMON
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte vector at a 4-byte aligned address
#define LSTM_DEV __device__ __forceinline__
// (This device code is compiled into TWO translation units -- the stand-alone policy-step kernels beside the update kernels, the rollout
// kernels beside the env kernels -- and the rollout modes promise bit-identical buffers: the whole library is built with -ffp-contract=on
// (build.py), i.e. a multiply-add fuses where the source writes a * b + c in one expression and nowhere else, whatever surrounds it.)
LSTM_DEV float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
LSTM_DEV float fast_tanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }
#endif

// ---------------------------------------------------------------------------------------------------------------
// One ROLLOUT step of the whole CustomLSTMPolicy in a single launch (run_bp_v5.py:178-185 `step`): actor stack and
// critic stack (two LSTM layers each), the action / value heads, the Gaussian sample, its neglogp, the [-1, 1] clip the
// runner applies (ppo2.py:533-535) and the rollout-buffer rows of step t (ppo2.py:521-531), including the reward row
// of the PREVIOUS step, so that a rollout step is exactly two launches (this + env step); the row index is a launch
// argument (the runner captures the whole rollout, one node pair per step, into a hipGraph).
// A workgroup owns 16 envs; waves [0, NW) run the actor stack, waves [NW, 2 NW) the critic stack, each wave 16 hidden
// units with their four gates (same MFMA mapping as the sequence kernels).  Weights are read once per workgroup from
// L2 (all workgroups read the same ~260 KB), the LSTM state [N, 8 HID] is updated in place.
struct PolicyStepArgs {
  const float *obs;        // [N, ob_dim]
  const uint8_t *dones;    // [N] episode ended before this step (mask of the state)
  const float *states_in;  // [N, 8 HID]: pi0 [c|h], pi1 [c|h], v0 [c|h], v1 [c|h]  (run_bp_v5.py:136-140)
  float *states_out;       // may alias states_in
  const float *w[12];      // layer (pi0, pi1, v0, v1) x (wx_p [n_in][HID][4], wh_p [HID][HID][4], b_p [HID][4])
  const float *pi_w, *pi_b, *vf_w, *vf_b, *logstd;
  const float *noise;      // [N, act_dim] standard normal, or NULL
  float *action, *clipped, *value, *neglogp;
  long long row;           // rollout row t written in the mb_* buffers, or -1: none
  const long long *rng_base;  // device scalar added to rng_step (e.g. steps of all earlier rollouts), or NULL
  float *mb_obs, *mb_actions, *mb_values, *mb_neglogp, *mb_rewards;
  uint8_t *mb_dones;
  const float *prev_reward;  // [N] reward of the previous env step -> mb_rewards[t-1] (t > 0)
  long long rng_step;
  unsigned rng_seed;
  unsigned env_id_offset;  // global id of env 0 (the sampling noise of env e is addressed by e + env_id_offset: multi-GPU shards)
  int rng_on;              // noise == NULL: 1 = counter-RNG sample (Philox keyed like the env's), 0 = deterministic
  int N, ob_dim, act_dim;
};

// Philox4x32-10, key (seed, 'IRR1') -- the env engine's generator (env_core.hpp philox_u01): 4 uniforms in [0, 1)
LSTM_DEV void policy_philox(unsigned seed, unsigned c0, unsigned c1, unsigned c2, unsigned c3, float out[4]) {
  unsigned k0 = seed, k1 = 0x49525231u;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned hi0 = __umulhi(c0, 0xD2511F53u), lo0 = c0 * 0xD2511F53u;
    const unsigned hi1 = __umulhi(c2, 0xCD9E8D57u), lo1 = c2 * 0xCD9E8D57u;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const float s = 1.0f / 16777216.0f;
  out[0] = (float)(c0 >> 8) * s; out[1] = (float)(c1 >> 8) * s; out[2] = (float)(c2 >> 8) * s; out[3] = (float)(c3 >> 8) * s;
}
#define IRRL_P_POLICY_NOISE 0x50u  /* purpose word of the sampling noise: block q = action index / 4 uses purpose 0x50 + q */

// Heads, sample, neglogp, clip and rollout-buffer rows shared by the LSTM and MLP policy-step kernels: thread (env, action)
// for the mean / sample, 16 more threads for the value and the neglogp sum.  hpi / hv: the two nets' last hidden
// activations [16 envs][LD] in LDS; head_w: pi_w [HID][act] then vf_w [HID] staged in LDS.
// NO_VALUE: the critic is not part of this step (the actor-only rollout kernel: values are computed for the whole rollout afterwards) --
// `value` / `mb_values` are not written, everything else is.
template <int HID, bool NO_VALUE = false>
LSTM_DEV void policy_heads(const PolicyStepArgs &a, const float *hpi, const float *hv, int LD, const float *head_w, float (*terms)[17],
                           int e0, int tid, long long t, long long gstep) {
  // heads: thread (env, action) for the mean / sample, 16 more threads for the value and the neglogp sum
  const int A = a.act_dim;
  if (tid < 16 * A && e0 + tid / A < a.N) {
    const int env = tid / A, ai = tid - env * A;
    float mean = a.pi_b[ai];
#pragma unroll
    for (int k = 0; k < HID; k++) mean = __builtin_fmaf(hpi[env * LD + k], head_w[k * A + ai], mean);
    const float ls = a.logstd[ai];
    const float sd = __expf(ls);
    const size_t o = (size_t)(e0 + env) * A + ai;
    float z = 0.0f;
    if (a.noise) {
      z = a.noise[o];
    } else if (a.rng_on) {
      float r[4];
      policy_philox(a.rng_seed, (unsigned)(e0 + env) + a.env_id_offset, (unsigned)((unsigned long long)gstep >> 32), (unsigned)gstep, IRRL_P_POLICY_NOISE + (unsigned)(ai >> 2), r);
      // Box-Muller on the pair (r0, r1) for slots 0/1 and (r2, r3) for slots 2/3; 1 - u is in (0, 1]
      const int pair = (ai >> 1) & 1;
      const float ua = pair ? r[2] : r[0], ub = pair ? r[3] : r[1];
      const float rad = __builtin_sqrtf(-2.0f * __logf(1.0f - ua));
      const float ang = 6.283185307179586f * ub;
      z = rad * ((ai & 1) ? __sinf(ang) : __cosf(ang));
    }
    // explicit FMAs: the contraction of these three statements must not depend on the kernel this function is inlined into (the
    // stand-alone, the one-launch-per-step and the persistent rollout kernels promise bit-identical buffers)
    const float act = __builtin_fmaf(sd, z, mean);
    const float d = (act - mean) / sd;
    terms[env][ai] = __builtin_fmaf(0.5f * d, d, ls);
    const float cl = fminf(fmaxf(act, -1.0f), 1.0f);
    a.action[o] = act;
    a.clipped[o] = cl;
    if (a.mb_actions) a.mb_actions[(size_t)t * a.N * A + o] = act;
  }
  float val = 0.0f;
  const int vt = tid - 16 * A;
  const bool vok = vt >= 0 && vt < 16 && e0 + vt < a.N;
  if (vok && !NO_VALUE) {
    val = a.vf_b[0];
#pragma unroll
    for (int k = 0; k < HID; k++) val = __builtin_fmaf(hv[vt * LD + k], head_w[HID * A + k], val);
  }
  __syncthreads();
  if (vok) {
    float nl = 0.0f;
    for (int ai = 0; ai < A; ai++) nl += terms[vt][ai];
    nl = __builtin_fmaf((float)A, 0.918938533204672742f, nl);   // 0.5 log(2 pi) per action dimension (an explicit FMA: inside a step loop the
                                                                  // product is loop-invariant and would otherwise be rounded on its own)
    const int e = e0 + vt;
    if (!NO_VALUE) a.value[e] = val;
    a.neglogp[e] = nl;
    if (a.mb_values) {
      if (!NO_VALUE) a.mb_values[(size_t)t * a.N + e] = val;
      a.mb_neglogp[(size_t)t * a.N + e] = nl;
      a.mb_dones[(size_t)t * a.N + e] = a.dones[e];
      if (a.prev_reward && t > 0) a.mb_rewards[(size_t)(t - 1) * a.N + e] = a.prev_reward[e];
    }
  }
  if (a.mb_obs) {
    const int n = ((a.N - e0 < 16) ? a.N - e0 : 16) * a.ob_dim;
    const float *src = a.obs + (size_t)e0 * a.ob_dim;
    float *dst = a.mb_obs + ((size_t)t * a.N + e0) * a.ob_dim;
    for (int i = tid; i < n; i += blockDim.x) dst[i] = src[i];
  }
}

#define PS_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b_, c_, 0, 0, 0)

// ---------------------------------------------------------------------------------------------------------------
// One rollout step of MlpPolicy (policies.py:430-446: separate pi / vf nets of two tanh layers of H units).
// a.w[] = pi_w1 [ob][H], pi_b1, pi_w2 [H][H], pi_b2, vf_w1, vf_b1, vf_w2, vf_b2 (plain row-major).
// LDSW: the eight arrays were copied to LDS once (mlp_policy_stage_lds: the persistent rollout kernel, whose workgroup keeps them for all
// steps), TRANSPOSED: per network [hidden unit][K = 36 rows of w1 (beyond ob_dim: the last row again, which is what the other form multiplies its x = 0 with) | 64 rows of w2] -- a lane owns a hidden unit, so its B
// operands for four consecutive k are one ds_read_b128 (rows of 100 floats: 16 consecutive lanes' reads fall into 16 different bank quads) --
// then b1 | b2; the head weights sit in head_w.  Otherwise everything is read from global memory as it lies there (the stand-alone kernel).
// Same values, same order of operations: the two agree bit for bit.
template <int H>
struct MlpLdsImage { static constexpr int OBMAX = 36, KW = OBMAX + H, B1 = H * KW, B2 = B1 + H, NET = B2 + H, FLOATS = 2 * NET; };
template <int H>
LSTM_DEV void mlp_policy_stage_lds(const PolicyStepArgs &a, float *wl, float *head_w, int tid, int nthr) {
  typedef MlpLdsImage<H> IMG;
  for (int net = 0; net < 2; net++) {
    float *dst = wl + net * IMG::NET;
    const float *w1 = a.w[4 * net], *b1 = a.w[4 * net + 1], *w2 = a.w[4 * net + 2], *b2 = a.w[4 * net + 3];
    for (int i = tid; i < IMG::OBMAX * H; i += nthr) { const int k = i / H, u = i - k * H; dst[u * IMG::KW + k] = w1[(k < a.ob_dim ? k : a.ob_dim - 1) * H + u]; }
    for (int i = tid; i < H * H; i += nthr) { const int k = i / H, u = i - k * H; dst[u * IMG::KW + IMG::OBMAX + k] = w2[i]; }
    for (int i = tid; i < H; i += nthr) { dst[IMG::B1 + i] = b1[i]; dst[IMG::B2 + i] = b2[i]; }
  }
  for (int i = tid; i < H * a.act_dim; i += nthr) head_w[i] = a.pi_w[i];
  if (tid < H) head_w[H * a.act_dim + tid] = a.vf_w[tid];
}
// ---------------------------------------------------------------------------------------------------------------
// THE STEP, BY ONE WAVE FOR ITS OWN FOUR ENVS (round 5; rounds 2-4: a workgroup of four waves for 16 envs on v_mfma_f32_16x16x4_f32, two waves
// per network).  MlpPolicy is per-robot arithmetic: nothing tied the 16 robots of a workgroup together except that 16-row MFMA tile -- and with
// it two workgroup barriers per step and, inside the persistent rollout kernel, the wait for the slowest of the workgroup's four env waves in
// EVERY step.  Here a wave (= the four robots its env step integrates, 16 lanes each) runs both networks for those four robots on
// v_mfma_f32_4x4x1_16b_f32: 16 blocks of a 4 x 4 outer product per instruction,
//   D[block b][robot i][unit 4 b + j] += x[robot i][k] * W[k][unit 4 b + j]        (lane = unit 4 b + j, accumulator register = robot i)
// so one instruction per k covers all 64 hidden units of the four robots -- no idle rows -- with the A operand the robot's activation
// (lane & 3 picks the robot: four distinct LDS words per read, broadcast) and the B operand row k of the weight matrix as it lies in memory
// (64 consecutive floats).  A network's layer is a chain of K dependent 2-pass MFMAs; the two networks' chains alternate.  Heads, sample,
// neglogp, clip and buffer rows: policy_heads' arithmetic, element for element, on 4 x act_dim + 4 lanes.  No workgroup barrier anywhere:
// the wave's LDS scratch is its own, and in the persistent rollout kernel a robot's next env step starts when ITS wave's policy is done.
// LDSW as above (the weight image of mlp_policy_stage_lds; head weights in `head_w`), else weights and head weights from global memory.
#define PS_MFMA4(a_, b_, c_) __builtin_amdgcn_mfma_f32_4x4x1f32(a_, b_, c_, 0, 0, 0)
// the wave's own LDS stores are ordered before its own later LDS loads (other lanes' words): compiler ordering; the LDS itself serves a wave in order
#define PS_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// a wave's LDS scratch (floats): the four robots' observations [4][ob_dim <= 64] | clipped actions [4][act_dim <= 15] | last reward [4] |
// last done flag [4] | h1 (pi, vf) [2][4][LD] | h2 (pi, vf) | neglogp terms [4][17]
template <int H>
struct MlpWaveLds {
  static constexpr int LD = H + 4, X = 0, ACT = 4 * 64, REW = ACT + 64, DON = REW + 4, H1 = DON + 4, H2 = H1 + 8 * LD, TERMS = H2 + 8 * LD, FLOATS = TERMS + 4 * 17;
};

// XLDS: the persistent rollout kernel -- observations, reward and done flag of the wave's robots are in its scratch already (its own env step
// left them there next to the stores to memory), and the clipped actions are left there for that env step: inside the step loop nothing a wave
// reads comes back from memory, so nothing waits for a store to complete
// LAY: the scratch layout (offsets X, ACT, REW, DON); NO_VALUE: the critic is not part of the step (the actor-only LSTM rollout)
template <int HID, bool XLDS, class LAY, bool NO_VALUE = false>
LSTM_DEV void policy_heads_wave(const PolicyStepArgs &a, float *ws, const float *hpi, const float *hv, int LD, const float *pw, const float *vw, float (*terms)[17],
                                int e4, int l, long long t, long long gstep) {
  // lane (env, action) for the mean / sample, four more lanes for the value and the neglogp sum (policy_heads with 4 envs instead of 16)
  const int A = a.act_dim;
  if (l < 4 * A && e4 + l / A < a.N) {
    const int env = l / A, ai = l - env * A;
    float mean = a.pi_b[ai];
#pragma unroll
    for (int k = 0; k < HID; k++) mean = __builtin_fmaf(hpi[env * LD + k], pw[k * A + ai], mean);
    const float ls = a.logstd[ai];
    const float sd = __expf(ls);
    const size_t o = (size_t)(e4 + env) * A + ai;
    float z = 0.0f;
    if (a.noise) {
      z = a.noise[o];
    } else if (a.rng_on) {
      float r[4];
      policy_philox(a.rng_seed, (unsigned)(e4 + env) + a.env_id_offset, (unsigned)((unsigned long long)gstep >> 32), (unsigned)gstep, IRRL_P_POLICY_NOISE + (unsigned)(ai >> 2), r);
      const int pair = (ai >> 1) & 1;
      const float ua = pair ? r[2] : r[0], ub = pair ? r[3] : r[1];
      const float rad = __builtin_sqrtf(-2.0f * __logf(1.0f - ua));
      const float ang = 6.283185307179586f * ub;
      z = rad * ((ai & 1) ? __sinf(ang) : __cosf(ang));
    }
    const float act = __builtin_fmaf(sd, z, mean);
    const float d = (act - mean) / sd;
    terms[env][ai] = __builtin_fmaf(0.5f * d, d, ls);
    const float cl = fminf(fmaxf(act, -1.0f), 1.0f);
    a.action[o] = act;
    a.clipped[o] = cl;
    if (XLDS) ws[LAY::ACT + env * A + ai] = cl;
    if (a.mb_actions) a.mb_actions[(size_t)t * a.N * A + o] = act;
  }
  float val = 0.0f;
  const int vt = l - 4 * A;
  const bool vok = vt >= 0 && vt < 4 && e4 + vt < a.N;
  if (vok && !NO_VALUE) {
    val = a.vf_b[0];
#pragma unroll
    for (int k = 0; k < HID; k++) val = __builtin_fmaf(hv[vt * LD + k], vw[k], val);
  }
  PS_WAVE_SYNC();
  if (vok) {
    float nl = 0.0f;
    for (int ai = 0; ai < A; ai++) nl += terms[vt][ai];
    nl = __builtin_fmaf((float)A, 0.918938533204672742f, nl);
    const int e = e4 + vt;
    if (!NO_VALUE) a.value[e] = val;
    a.neglogp[e] = nl;
    if (a.mb_values) {
      if (!NO_VALUE) a.mb_values[(size_t)t * a.N + e] = val;
      a.mb_neglogp[(size_t)t * a.N + e] = nl;
      a.mb_dones[(size_t)t * a.N + e] = XLDS ? (uint8_t)(ws[LAY::DON + vt] != 0.0f) : a.dones[e];
      if (a.prev_reward && t > 0) a.mb_rewards[(size_t)(t - 1) * a.N + e] = XLDS ? ws[LAY::REW + vt] : a.prev_reward[e];
    }
  }
  if (a.mb_obs && e4 < a.N) {      // the row of the observations the step was computed from: out of the scratch (the values a.obs holds)
    const int n = ((a.N - e4 < 4) ? a.N - e4 : 4) * a.ob_dim;
    float *dst = a.mb_obs + ((size_t)t * a.N + e4) * a.ob_dim;
    for (int i = l; i < n; i += 64) dst[i] = ws[LAY::X + i];
  }
}

template <int H, bool LDSW, bool XLDS = false>
LSTM_DEV void mlp_policy_wave_body(const PolicyStepArgs &a, const int e4, float *ws, const float *wl, const float *head_w, const int l) {
  static_assert(H == 64, "one lane per hidden unit");
  constexpr int LD = MlpWaveLds<H>::LD;
  float *xs = ws + MlpWaveLds<H>::X, *h1 = ws + MlpWaveLds<H>::H1, *h2 = ws + MlpWaveLds<H>::H2;
  float (*terms)[17] = (float (*)[17])(ws + MlpWaveLds<H>::TERMS);
  const long long t = a.row;
  const long long gstep = a.rng_step + (a.rng_base ? *a.rng_base : 0ll);
  const int r4 = l & 3, OB = a.ob_dim;
  const float *w1p, *b1p, *w2p, *b2p, *w1v, *b1v, *w2v, *b2v;
  typedef MlpLdsImage<H> IMG;
  const float *tp = wl + (size_t)l * IMG::KW, *tv = tp + IMG::NET;      // (LDSW) this lane's unit in the transposed images
  if (LDSW) {
    w1p = w2p = w1v = w2v = nullptr;
    b1p = wl + IMG::B1; b2p = wl + IMG::B2; b1v = b1p + IMG::NET; b2v = b2p + IMG::NET;
  } else {
    w1p = a.w[0]; b1p = a.w[1]; w2p = a.w[2]; b2p = a.w[3]; w1v = a.w[4]; b1v = a.w[5]; w2v = a.w[6]; b2v = a.w[7];
  }
  if (!XLDS) {      // the four robots' observations -> xs[robot][k] (rows of ob_dim words, as in memory)
    const int n = ((a.N - e4 < 4) ? a.N - e4 : 4) * OB;
    const float *src = a.obs + (size_t)e4 * OB;
    for (int i = l; i < 4 * OB; i += 64) xs[i] = (i < n) ? src[i] : 0.0f;
    PS_WAVE_SYNC();
  }
  f32x4 ap, av;
  { const float bp = b1p[l], bv = b1v[l]; ap = (f32x4){bp, bp, bp, bp}; av = (f32x4){bv, bv, bv, bv}; }
  // (groups of four k with the operands of a group requested together; beyond ob_dim: x = 0 against a clamped weight row -- adds nothing)
#pragma unroll 3
  for (int k = 0; k < OB; k += 4) {
    float x[4], bp[4], bv[4];
    f32x4 p4 = {0.0f, 0.0f, 0.0f, 0.0f}, v4 = p4;
    if (LDSW) { p4 = *(const f32x4 *)&tp[k]; v4 = *(const f32x4 *)&tv[k]; }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const int kc = (k + kk < OB) ? k + kk : OB - 1;
      const float xv = xs[r4 * OB + kc];
      x[kk] = (k + kk < OB) ? xv : 0.0f;
      bp[kk] = LDSW ? p4[kk] : w1p[(size_t)kc * H + l];
      bv[kk] = LDSW ? v4[kk] : w1v[(size_t)kc * H + l];
    }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      ap = PS_MFMA4(x[kk], bp[kk], ap);
      av = PS_MFMA4(x[kk], bv[kk], av);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) { h1[r * LD + l] = fast_tanh(ap[r]); h1[4 * LD + r * LD + l] = fast_tanh(av[r]); }
  PS_WAVE_SYNC();
  { const float bp = b2p[l], bv = b2v[l]; ap = (f32x4){bp, bp, bp, bp}; av = (f32x4){bv, bv, bv, bv}; }
#pragma unroll 4
  for (int k = 0; k < H; k += 4) {
    const f32x4 p4 = *(const f32x4 *)&h1[r4 * LD + k], v4 = *(const f32x4 *)&h1[4 * LD + r4 * LD + k];
    f32x4 wp = {0.0f, 0.0f, 0.0f, 0.0f}, wv = wp;
    if (LDSW) { wp = *(const f32x4 *)&tp[IMG::OBMAX + k]; wv = *(const f32x4 *)&tv[IMG::OBMAX + k]; }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      ap = PS_MFMA4(p4[kk], LDSW ? wp[kk] : w2p[(size_t)(k + kk) * H + l], ap);
      av = PS_MFMA4(v4[kk], LDSW ? wv[kk] : w2v[(size_t)(k + kk) * H + l], av);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) { h2[r * LD + l] = fast_tanh(ap[r]); h2[4 * LD + r * LD + l] = fast_tanh(av[r]); }
  PS_WAVE_SYNC();
  policy_heads_wave<H, XLDS, MlpWaveLds<H>>(a, ws, h2, h2 + 4 * LD, LD, LDSW ? head_w : a.pi_w, LDSW ? head_w + H * a.act_dim : a.vf_w, terms, e4, l, t, gstep);
}
#ifdef IRRL_PROFILE_POLICY   /* diagnostic build (tools/policy_phases.py): 100 MHz time stamps of the phases of one workgroup */
#define IRRL_PS_STAMP() do { __builtin_amdgcn_sched_barrier(0); ts_[tsn_++] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define IRRL_PS_STAMP() do { } while (0)
#endif
// The step for the 16 envs e0 .. e0 + 15 by one workgroup of NTHR threads.  The work is cut into 2 NW "virtual waves" (stack
// s = actor / critic, ws = 16 hidden units with their four gates: the MFMA mapping of the sequence kernels):
//   VPW = 1: NTHR = 2 NW x 64, every wave is one virtual wave (the stand-alone kernel);
//   VPW = 2: NTHR = 256, waves 0 .. NW-1 run (actor, ws = w) AND (critic, ws = w), two independent accumulation streams per
//            wave; a wave beyond NW only takes part in the barriers, the head weights, the heads and the row copies (the
//            fused env + policy kernel, whose workgroup is the four env waves of these 16 robots).
// Per output element the arithmetic and its order are the same in both: the two kernels agree bit for bit.
// OBK = k-steps of the observation projection ((ob_dim + 3) / 4) when known at compile time, 0 = runtime loop.
//
// LDSW: the layer-0 operands wh0 / wx0 of both stacks were copied to LDS (policy_prefetch_lds, same [k][unit][gate] image as in
// global memory: a wave's 16-byte reads of one k-row are 256 contiguous bytes, conflict free) while the workgroup was busy with
// something else -- the fused kernel's env part; the first MFMA block then starts without waiting for L2.
template <int HID>
struct PolicyLdsImage {
  static constexpr int WH0 = HID * HID * 4;                       // floats, a multiple of 256 (1 KiB pieces) for HID % 8 == 0
  static constexpr int WX0 = ((48 * HID * 4 + 255) / 256) * 256;  // room for ob_dim <= 48 rows, rounded up to whole 1 KiB pieces
  static constexpr int STACK = WH0 + WX0;
  static constexpr int FLOATS = 2 * STACK;
};
// all NTHR threads: one global_load_lds_dwordx4 per 1 KiB piece and wave (destination = wave-uniform base + lane x 16 bytes);
// no registers, completion on the VM counter -- the caller's next __syncthreads() drains it
template <int HID, int NTHR>
LSTM_DEV void policy_prefetch_lds(const PolicyStepArgs &a, float *lds_w) {
  typedef PolicyLdsImage<HID> IMG;
  constexpr int NWAVES = NTHR / 64;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int wx_floats = a.ob_dim * HID * 4;
  const int wx_pieces = (wx_floats + 255) / 256;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const float *wh0 = s ? a.w[7] : a.w[1], *wx0 = s ? a.w[6] : a.w[0];
    float *dst = lds_w + s * IMG::STACK;
    for (int c = w; c < IMG::WH0 / 256; c += NWAVES)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wh0 + c * 256 + l * 4),
                                       (__attribute__((address_space(3))) void *)(dst + c * 256), 16, 0, 0);
    for (int c = w; c < wx_pieces; c += NWAVES) {
      int idx = c * 256 + l * 4;
      idx = idx < wx_floats - 4 ? idx : wx_floats - 4;            // the last piece is ragged: lanes past the end re-read the last 16 bytes
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wx0 + idx),
                                       (__attribute__((address_space(3))) void *)(dst + IMG::WH0 + c * 256), 16, 0, 0);
    }
  }
}

// the ACTOR-ONLY image (round 5: irrl_rollout_persistent_actor_kernel): with the critic off the per-step path the same LDS holds ALL of the
// actor's LSTM operands -- [wh0 | wx0 | wh1 | wx1], 4 x 9216 floats for HID 48 -- so a step of the policy part fetches no weight from L2 at all
template <int HID, int NTHR>
LSTM_DEV void policy_prefetch_lds_actor(const PolicyStepArgs &a, float *lds_w) {
  typedef PolicyLdsImage<HID> IMG;
  static_assert(IMG::WX0 == HID * HID * 4, "the layer-1 input weights [HID][HID][4] fill the slot of the padded layer-0 input weights");
  constexpr int NWAVES = NTHR / 64;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int wx_floats = a.ob_dim * HID * 4;
  const int wx_pieces = (wx_floats + 255) / 256;
  const float *src[3] = {a.w[1], a.w[4], a.w[3]};                      // wh0, wh1, wx1: whole 1 KiB pieces
  const int dsto[3] = {0, IMG::STACK, IMG::STACK + IMG::WH0};
#pragma unroll
  for (int i = 0; i < 3; i++)
    for (int c = w; c < IMG::WH0 / 256; c += NWAVES)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[i] + c * 256 + l * 4),
                                       (__attribute__((address_space(3))) void *)(lds_w + dsto[i] + c * 256), 16, 0, 0);
  for (int c = w; c < wx_pieces; c += NWAVES) {                         // wx0: ob_dim rows, the last piece ragged
    int idx = c * 256 + l * 4;
    idx = idx < wx_floats - 4 ? idx : wx_floats - 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.w[0] + idx),
                                     (__attribute__((address_space(3))) void *)(lds_w + IMG::WH0 + c * 256), 16, 0, 0);
  }
}

// ACTOR_ONLY (needs VPW == 1, NTHR == 256, LDSW): waves 0 .. NW-1 run the actor stack's virtual waves, nobody runs the critic; all four
// weight matrices of the actor come from the LDS image of policy_prefetch_lds_actor; `value` is not produced.  Per output element of the
// actor the arithmetic and its order are those of every other instantiation: actions, neglogp and the actor's states agree bit for bit.
template <int HID, int OBK, int VPW, int NTHR, bool LDSW = false, bool ACTOR_ONLY = false>
LSTM_DEV void policy_step_body(const PolicyStepArgs &a, const int e0, float (*hbuf)[16 * (HID + 1)], float (*terms)[17], float *head_w,
                               const float *lds_w = nullptr, unsigned long long prof_t0 = 0, unsigned long long prof_t1 = 0, const int tid_opaque = -1) {
  constexpr int NW = HID / 16;
  constexpr int KS = HID / 4;
  constexpr int LD = HID + 1;
  constexpr int SD = 8 * HID;
  static_assert(ACTOR_ONLY ? (VPW == 1 && NTHR >= NW * 64 && LDSW) : (VPW == 1 ? NTHR == 2 * NW * 64 : (VPW == 2 && NTHR >= NW * 64)), "workgroup shape");
  // tid_opaque: the persistent rollout kernel hands in threadIdx.x through an empty asm every iteration, so that the per-lane
  // address arithmetic below stays INSIDE its step loop (hoisted, it is hundreds of live 64-bit registers: 2 KB of scratch)
  const int tid = tid_opaque >= 0 ? tid_opaque : (int)threadIdx.x;
  const int w = tid >> 6, l = tid & 63;
  const int col = l & 15, rq = l >> 4;
  const bool mine = (VPW == 1 && !ACTOR_ONLY) ? true : (w < NW);                 // wave-uniform: this wave owns virtual waves
  const int ws = (VPW == 1 && !ACTOR_ONLY) ? w % NW : (w < NW ? w : 0);
#ifdef IRRL_PROFILE_POLICY
  unsigned long long ts_[8];
  int tsn_ = 0;
#endif
  IRRL_PS_STAMP();   // 0: start
  // N need not be a multiple of 16: rows past the pool read the last env (clamped index) and store nothing
  const int eA = (e0 + col < a.N) ? e0 + col : a.N - 1;
  int eC[4];
  bool okC[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { okC[j] = e0 + 4 * rq + j < a.N; eC[j] = okC[j] ? e0 + 4 * rq + j : a.N - 1; }
  const int u = 16 * ws + col;
  const long long t = a.row;
  const long long gstep = a.rng_step + (a.rng_base ? *a.rng_base : 0ll);
  const float keepA = a.dones[eA] ? 0.0f : 1.0f;
  float keepC[4];
#pragma unroll
  for (int j = 0; j < 4; j++) keepC[j] = a.dones[eC[j]] ? 0.0f : 1.0f;
  // (the head weights are needed last: they are fetched into registers BEHIND the LSTM operands below and parked in LDS
  // after the first MFMA block -- staging them first made every wave wait for a global load before it issued the ~60
  // operand loads: 5 us from kernel start to "loads issued" and 3.4 us at the first barrier, tools/policy_phases.py)
  // Everything that does not depend on layer 0's output is requested up front, in program order, so that the L2 / HBM
  // latency of the independent loads overlaps instead of being paid once per k-step: both layers' previous h and c,
  // the observation slice, wh of both layers and wx of layer 0.  The recurrent half of layer 1 is accumulated before
  // layer 0's cell math; only h0 wx1 has to wait for it.
  constexpr int OBKC = OBK > 0 ? OBK : 1;
  const float *wx0[VPW], *wh0[VPW], *b0[VPW], *wx1[VPW], *wh1[VPW], *b1[VPW];
  size_t soff0[VPW], soff1[VPW];
  int stk[VPW];
#pragma unroll
  for (int v = 0; v < VPW; v++) {
    const int stack = ACTOR_ONLY ? 0 : (VPW == 1 ? w / NW : v);
    stk[v] = stack;
    // selects between kernel arguments (scalar registers), not an indexed load of the argument block
    wx0[v] = stack ? a.w[6] : a.w[0]; wh0[v] = stack ? a.w[7] : a.w[1]; b0[v] = stack ? a.w[8] : a.w[2];
    wx1[v] = stack ? a.w[9] : a.w[3]; wh1[v] = stack ? a.w[10] : a.w[4]; b1[v] = stack ? a.w[11] : a.w[5];
    soff0[v] = (size_t)(stack * 2 + 0) * 2 * HID; soff1[v] = (size_t)(stack * 2 + 1) * 2 * HID;
  }
  // previous h of both layers as 16-byte vectors: lane (env, rq) holds elements 16 m + 4 rq + j (m < KS / 4, j < 4), so the
  // k-step (m, j) of the recurrent products pairs A = h[16 m + 4 rq + j] with the wh row of the same index -- the k order of the
  // sum differs from 4 kk + rq (rounding only), and the 2 x KS scattered dword loads of a wave become 2 x KS / 4 vector loads
  // that use every byte of the 64-byte lines they touch
  constexpr int KV = KS / 4;
  constexpr int OBV = OBK > 0 ? (OBK - 1) / 4 : 0;   // whole 16-element groups of the observation row that are valid for every ob_dim with this OBK
  static_assert(KS % 4 == 0, "HID must be a multiple of 16");
  f32x4 hp0[VPW][KV], hp1[VPW][KV];
  float cp0[VPW][4], cp1[VPW][4], ob[OBKC];
  f32x4 Wh0[VPW][KS], Wh1[VPW][KS], Wx0[VPW][OBKC], Wx1[VPW][KS], bias0[VPW], bias1[VPW];
  f32x4 acc0[VPW][4], acc1[VPW][4];
  if (mine) {
#pragma unroll
    for (int v = 0; v < VPW; v++)
#pragma unroll
      for (int m = 0; m < KV; m++) hp0[v][m] = *(const f32x4 *)&a.states_in[(size_t)eA * SD + soff0[v] + HID + 16 * m + 4 * rq];
    if (OBK > 0) {
      // the observation row the same way: OBV 16-byte vectors (elements 16 m + 4 rq + j, all below ob_dim > 4 OBK - 4; rows of
      // 35 floats are only 4-byte aligned) and the k-steps behind them as single words in the plain 4 kk + rq order
#pragma unroll
      for (int m = 0; m < OBV; m++) {
        const f32x4u v4 = *(const f32x4u *)&a.obs[(size_t)eA * a.ob_dim + 16 * m + 4 * rq];
        ob[4 * m + 0] = v4[0]; ob[4 * m + 1] = v4[1]; ob[4 * m + 2] = v4[2]; ob[4 * m + 3] = v4[3];
      }
#pragma unroll
      for (int kk = 4 * OBV; kk < OBKC; kk++) {
        const int k = 4 * kk + rq, kc = k < a.ob_dim ? k : a.ob_dim - 1;   // clamped: the load is unconditional, the value masked
        ob[kk] = a.obs[(size_t)eA * a.ob_dim + kc];
      }
    }
#pragma unroll
    for (int v = 0; v < VPW; v++) {
#pragma unroll
      for (int kk = 0; kk < KS; kk++)
        Wh0[v][kk] = LDSW ? *(const f32x4 *)&lds_w[stk[v] * PolicyLdsImage<HID>::STACK + ((16 * (kk / 4) + 4 * rq + (kk % 4)) * HID + u) * 4]
                          : *(const f32x4 *)&wh0[v][((size_t)(16 * (kk / 4) + 4 * rq + (kk % 4)) * HID + u) * 4];
      if (OBK > 0) {
#pragma unroll
        for (int kk = 0; kk < OBKC; kk++) {
          const int k = kk < 4 * OBV ? 16 * (kk / 4) + 4 * rq + (kk % 4) : 4 * kk + rq, kc = k < a.ob_dim ? k : a.ob_dim - 1;
          Wx0[v][kk] = LDSW ? *(const f32x4 *)&lds_w[stk[v] * PolicyLdsImage<HID>::STACK + PolicyLdsImage<HID>::WH0 + (kc * HID + u) * 4]
                            : *(const f32x4 *)&wx0[v][((size_t)kc * HID + u) * 4];
        }
      }
    }
#pragma unroll
    for (int v = 0; v < VPW; v++) {
#pragma unroll
      for (int m = 0; m < KV; m++) hp1[v][m] = *(const f32x4 *)&a.states_in[(size_t)eA * SD + soff1[v] + HID + 16 * m + 4 * rq];
#pragma unroll
      for (int kk = 0; kk < KS; kk++)
        Wh1[v][kk] = ACTOR_ONLY ? *(const f32x4 *)&lds_w[PolicyLdsImage<HID>::STACK + ((16 * (kk / 4) + 4 * rq + (kk % 4)) * HID + u) * 4]
                                : *(const f32x4 *)&wh1[v][((size_t)(16 * (kk / 4) + 4 * rq + (kk % 4)) * HID + u) * 4];
    }
#pragma unroll
    for (int v = 0; v < VPW; v++) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        cp0[v][j] = a.states_in[(size_t)eC[j] * SD + soff0[v] + u];
        cp1[v][j] = a.states_in[(size_t)eC[j] * SD + soff1[v] + u];
      }
      bias0[v] = *(const f32x4 *)&b0[v][u * 4]; bias1[v] = *(const f32x4 *)&b1[v][u * 4];
    }
  }
  // head weights: pi_w [HID][act] then vf_w [HID] = HID * (act + 1) <= HID * 17 floats over the workgroup's threads
  // (ACTOR_ONLY: the rollout kernel staged pi_w in `head_w` once, in front of its step loop)
  constexpr int NHW = ACTOR_ONLY ? 1 : (HID * 17 + NTHR - 1) / NTHR;
  const int n_head = ACTOR_ONLY ? 0 : HID * (a.act_dim + 1);
  float hw[NHW];
#pragma unroll
  for (int i = 0; i < NHW; i++) {
    const int hi = tid + i * NTHR;
    hw[i] = hi < n_head ? (hi < HID * a.act_dim ? a.pi_w[hi] : a.vf_w[hi - HID * a.act_dim]) : 0.0f;
  }
  __builtin_amdgcn_sched_barrier(0);   // keep the loads above clustered: the scheduler must not sink them between the MFMAs
  IRRL_PS_STAMP();   // 1: loads issued
  if (mine) {
#pragma unroll
    for (int v = 0; v < VPW; v++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        acc0[v][g] = (f32x4){bias0[v][g], bias0[v][g], bias0[v][g], bias0[v][g]};
        acc1[v][g] = (f32x4){bias1[v][g], bias1[v][g], bias1[v][g], bias1[v][g]};
      }
#pragma unroll
    for (int kk = 0; kk < KS; kk++) {
#pragma unroll
      for (int v = 0; v < VPW; v++) {
        const float av = hp0[v][kk / 4][kk % 4] * keepA;
#pragma unroll
        for (int g = 0; g < 4; g++) acc0[v][g] = PS_MFMA(av, Wh0[v][kk][g], acc0[v][g]);
      }
    }
    if (OBK > 0) {
#pragma unroll
      for (int kk = 0; kk < OBKC; kk++) {
        const float av = (kk < 4 * OBV || 4 * kk + rq < a.ob_dim) ? ob[kk] : 0.0f;
#pragma unroll
        for (int v = 0; v < VPW; v++)
#pragma unroll
          for (int g = 0; g < 4; g++) acc0[v][g] = PS_MFMA(av, Wx0[v][kk][g], acc0[v][g]);
      }
    } else {
      const int ksx = (a.ob_dim + 3) >> 2;
      for (int kk = 0; kk < ksx; kk++) {
        const int k = 4 * kk + rq, kc = k < a.ob_dim ? k : a.ob_dim - 1;
        const float av = (k < a.ob_dim) ? a.obs[(size_t)eA * a.ob_dim + kc] : 0.0f;
#pragma unroll
        for (int v = 0; v < VPW; v++) {
          const f32x4 bw = *(const f32x4 *)&wx0[v][((size_t)kc * HID + u) * 4];
#pragma unroll
          for (int g = 0; g < 4; g++) acc0[v][g] = PS_MFMA(av, bw[g], acc0[v][g]);
        }
      }
    }
    // layer 1's input weights: requested now, consumed after layer 0's cell
#pragma unroll
    for (int v = 0; v < VPW; v++)
#pragma unroll
      for (int kk = 0; kk < KS; kk++)
        Wx1[v][kk] = ACTOR_ONLY ? *(const f32x4 *)&lds_w[PolicyLdsImage<HID>::STACK + PolicyLdsImage<HID>::WH0 + ((4 * kk + rq) * HID + u) * 4]
                                : *(const f32x4 *)&wx1[v][((size_t)(4 * kk + rq) * HID + u) * 4];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < KS; kk++) {
#pragma unroll
      for (int v = 0; v < VPW; v++) {
        const float av = hp1[v][kk / 4][kk % 4] * keepA;
#pragma unroll
        for (int g = 0; g < 4; g++) acc1[v][g] = PS_MFMA(av, Wh1[v][kk][g], acc1[v][g]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NHW; i++) {
    const int hi = tid + i * NTHR;
    if (hi < n_head) head_w[hi] = hw[i];
  }
  IRRL_PS_STAMP();   // 2: layer-0 and recurrent layer-1 MFMAs issued (the loads have landed)
  // every wave has read the previous h of both layers before anyone overwrites them: states_out may alias states_in
  __syncthreads();
  IRRL_PS_STAMP();   // 3: barrier
  if (mine) {
#pragma unroll
    for (int v = 0; v < VPW; v++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float ig = fast_sigmoid(acc0[v][0][j]), fg = fast_sigmoid(acc0[v][1][j]), og = fast_sigmoid(acc0[v][2][j]), gg = fast_tanh(acc0[v][3][j]);
        const float cn = fg * (cp0[v][j] * keepC[j]) + ig * gg;
        const float hn = og * fast_tanh(cn);
        const size_t row = (size_t)eC[j] * SD + soff0[v];
        if (okC[j]) { a.states_out[row + u] = cn; a.states_out[row + HID + u] = hn; }
        hbuf[stk[v]][(4 * rq + j) * LD + u] = hn;
      }
  }
  __syncthreads();
  if (mine) {
#pragma unroll
    for (int kk = 0; kk < KS; kk++) {
#pragma unroll
      for (int v = 0; v < VPW; v++) {
        const float av = hbuf[stk[v]][col * LD + 4 * kk + rq];
#pragma unroll
        for (int g = 0; g < 4; g++) acc1[v][g] = PS_MFMA(av, Wx1[v][kk][g], acc1[v][g]);
      }
    }
  }
  IRRL_PS_STAMP();   // 4: layer-0 cell + layer-1 input MFMAs
  __syncthreads();   // all reads of layer 0's h are done before hbuf is reused for layer 1's h
  if (mine) {
#pragma unroll
    for (int v = 0; v < VPW; v++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float ig = fast_sigmoid(acc1[v][0][j]), fg = fast_sigmoid(acc1[v][1][j]), og = fast_sigmoid(acc1[v][2][j]), gg = fast_tanh(acc1[v][3][j]);
        const float cn = fg * (cp1[v][j] * keepC[j]) + ig * gg;
        const float hn = og * fast_tanh(cn);
        const size_t row = (size_t)eC[j] * SD + soff1[v];
        if (okC[j]) { a.states_out[row + u] = cn; a.states_out[row + HID + u] = hn; }
        hbuf[stk[v]][(4 * rq + j) * LD + u] = hn;
      }
  }
  __syncthreads();
  IRRL_PS_STAMP();   // 5: layer-1 cell
  policy_heads<HID, ACTOR_ONLY>(a, hbuf[0], hbuf[ACTOR_ONLY ? 0 : 1], LD, head_w, terms, e0, tid, t, gstep);
  IRRL_PS_STAMP();   // 6: heads, sample, buffer rows
#ifdef IRRL_PROFILE_POLICY
  __syncthreads();   // (this workgroup's own neglogp entries are written: the stamps go over them)
  if (blockIdx.x == gridDim.x / 2 && tid == 0) {
    for (int k = 0; k < 7; k++) a.neglogp[e0 + k] = (float)(ts_[k] - ts_[0]);
    // fused kernel: kernel start -> this wave's env part done -> policy part entered (behind the workgroup barrier)
    a.neglogp[e0 + 7] = prof_t0 ? (float)(prof_t1 - prof_t0) : 0.0f;
    a.neglogp[e0 + 8] = prof_t0 ? (float)(ts_[0] - prof_t0) : 0.0f;
  }
#else
  (void)prof_t0; (void)prof_t1;
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// THE ACTOR STACK OF CustomLSTMPolicy BY ONE WAVE FOR ITS OWN FOUR ENVS (round 5; the LSTM twin of mlp_policy_wave_body, for the actor-only
// persistent rollout).  Same idea: v_mfma_f32_4x4x1_16b_f32 with lane = gate column, accumulator register = robot -- the 192 gate columns of a
// layer are three column groups, i.e. three independent chains over K -- the operand rows out of the LDS image of policy_prefetch_lds_actor as
// they lie there ([k][unit][gate]: a row is 192 consecutive floats), h of both layers in the wave's scratch, c in registers for the whole
// rollout.  BIT-IDENTICAL to policy_step_body's actor (so to every other rollout mode): a 16 x 16 x 4 MFMA adds its four products to the
// accumulator one after the other in k order, which is what four 4 x 4 x 1 instructions do -- so the chains below walk K in policy_step_body's
// order (recurrent part first, in its permuted order 16 m + 4 rq + j; then the input part), and the cell is its cell, statement for statement.
// A gate column's four robots sit in one lane's accumulator registers and a unit's four gates in the four lanes of a quad: a 4 x 4 transpose
// inside the quad (two DPP butterflies) hands lane (unit, j) the four gates of robot j, and every lane runs ONE cell per column group.
template <int HID>
struct LstmWaveLds { static constexpr int X = 0, ACT = 144, REW = 192, DON = 196, H0 = 200, H1 = H0 + 4 * HID, TERMS = H1 + 4 * HID, FLOATS = TERMS + 4 * 17; };

LSTM_DEV float ps_quad_xor1(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true)); }
LSTM_DEV float ps_quad_xor2(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true)); }
// M[lane of the quad][register] -> its transpose
LSTM_DEV void ps_quad_transpose(f32x4 &m, int l) {
  const bool o1 = l & 1, o2 = l & 2;
  {
    const float r01 = ps_quad_xor1(o1 ? m[0] : m[1]), r23 = ps_quad_xor1(o1 ? m[2] : m[3]);
    if (o1) { m[0] = r01; m[2] = r23; } else { m[1] = r01; m[3] = r23; }
  }
  {
    const float r02 = ps_quad_xor2(o2 ? m[0] : m[2]), r13 = ps_quad_xor2(o2 ? m[1] : m[3]);
    if (o2) { m[0] = r02; m[1] = r13; } else { m[2] = r02; m[3] = r13; }
  }
}

// The actor's operands for the wave body: TRANSPOSED, [gate column][K] with K = wh0 (HID) | wx0 (36: ob_dim rows, the rest copies of the last
// row -- what policy_step_body's masked k-step multiplies by zero) | wh1 (HID) | wx1 (HID) -- a lane owns a gate column per column group, so its
// B operands for FOUR consecutive k are one ds_read_b128 (a quarter of the LDS instructions of row-wise reads, which is what the step waited
// for: 13.6 -> see DESIGN 3.2); rows of 180 floats: 16 consecutive lanes' 16-byte reads fall into 16 different bank quads.
template <int HID>
struct LstmWaveImage { static constexpr int XK = 36, K = 3 * HID + XK, OWH0 = 0, OWX0 = HID, OWH1 = HID + XK, OWX1 = 2 * HID + XK, FLOATS = 4 * HID * K; };
template <int HID, int NTHR>
LSTM_DEV void lstm_wave_image_stage(const PolicyStepArgs &a, float *wt) {
  typedef LstmWaveImage<HID> IMG;
  constexpr int GC = 4 * HID;
  const int tid = threadIdx.x;
  for (int i = tid; i < HID * GC; i += NTHR) {
    const int k = i / GC, c = i - k * GC;
    wt[c * IMG::K + IMG::OWH0 + k] = a.w[1][i];
    wt[c * IMG::K + IMG::OWH1 + k] = a.w[4][i];
    wt[c * IMG::K + IMG::OWX1 + k] = a.w[3][i];
  }
  for (int i = tid; i < IMG::XK * GC; i += NTHR) {
    const int k = i / GC, c = i - k * GC, kc = k < a.ob_dim ? k : a.ob_dim - 1;
    wt[c * IMG::K + IMG::OWX0 + k] = a.w[0][kc * GC + c];
  }
}

// cst: c of layer 0 / layer 1 for (robot l & 3, unit 16 G + (l >> 2)), G = 0 .. HID / 16 - 1; bias: the lane's gate columns 64 G + l of both layers
template <int HID>
LSTM_DEV void lstm_actor_wave_body(const PolicyStepArgs &a, const int e4, float *ws, const float *wt, const float *head_w, const int l,
                                   float (&cst)[2][HID / 16], const float (&bias)[2][HID / 16]) {
  typedef LstmWaveLds<HID> LAY;
  typedef LstmWaveImage<HID> IMG;
  constexpr int NG = HID / 16, GC = 4 * HID;
  static_assert(GC == 64 * NG, "three column groups of 64 lanes");
  float *xs = ws + LAY::X, *h0 = ws + LAY::H0, *h1 = ws + LAY::H1;
  float (*terms)[17] = (float (*)[17])(ws + LAY::TERMS);
  const long long t = a.row;
  const long long gstep = a.rng_step + (a.rng_base ? *a.rng_base : 0ll);
  const int r4 = l & 3, q = l >> 2, OB = a.ob_dim;
  const float keep = ws[LAY::DON + r4] != 0.0f ? 0.0f : 1.0f;
  const float *wl = wt + (size_t)l * IMG::K;      // this lane's column of group 0; group G: + 64 G columns
  f32x4 acc[NG];
  // sixteen k of a chain in policy_step_body's order k = 16 m + 4 rq + j (j outer, rq inner: one of its MFMAs is the four rq); av[rq][j]: the A operand
  auto group16 = [&](const f32x4 (&av)[4], int off) {
    f32x4 wv[NG][4];
#pragma unroll
    for (int G = 0; G < NG; G++)
#pragma unroll
      for (int rq = 0; rq < 4; rq++) wv[G][rq] = *(const f32x4 *)&wl[64 * G * IMG::K + off + 4 * rq];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int rq = 0; rq < 4; rq++)
#pragma unroll
        for (int G = 0; G < NG; G++) acc[G] = PS_MFMA4(av[rq][j], wv[G][rq][j], acc[G]);
  };
  auto recurrent = [&](const float *hprev, int off) {
#pragma unroll
    for (int m = 0; m < HID / 16; m++) {
      f32x4 hv[4];
#pragma unroll
      for (int rq = 0; rq < 4; rq++) hv[rq] = *(const f32x4 *)&hprev[r4 * HID + 16 * m + 4 * rq] * keep;
      group16(hv, off + 16 * m);
    }
  };
  // the cell of (robot r4, unit 16 G + q) for every column group; h into `hout`
  auto cell = [&](float (&c)[NG], float *hout) {
#pragma unroll
    for (int G = 0; G < NG; G++) {
      ps_quad_transpose(acc[G], l);
      const float ig = fast_sigmoid(acc[G][0]), fg = fast_sigmoid(acc[G][1]), og = fast_sigmoid(acc[G][2]), gg = fast_tanh(acc[G][3]);
      const float cn = fg * (c[G] * keep) + ig * gg;
      const float hn = og * fast_tanh(cn);
      c[G] = cn;
      hout[r4 * HID + 16 * G + q] = hn;
    }
  };
  // ---- layer 0 ----
#pragma unroll
  for (int G = 0; G < NG; G++) acc[G] = (f32x4){bias[0][G], bias[0][G], bias[0][G], bias[0][G]};
  recurrent(h0, IMG::OWH0);
  {
    // the observation part in policy_step_body's order: whole groups of 16 permuted like the recurrent part, the rest in plain order, padded
    // to a multiple of four with x = 0 against the last row
    const int obv = ((OB + 3) / 4 - 1) / 4;          // whole 16-element groups (its OBV)
    for (int m = 0; m < obv; m++) {
      f32x4 xv[4];
#pragma unroll
      for (int rq = 0; rq < 4; rq++)
#pragma unroll
        for (int j = 0; j < 4; j++) xv[rq][j] = xs[r4 * OB + 16 * m + 4 * rq + j];
      group16(xv, IMG::OWX0 + 16 * m);
    }
    for (int k = 16 * obv; k < ((OB + 3) & ~3); k += 4) {
      f32x4 wv[NG];
#pragma unroll
      for (int G = 0; G < NG; G++) wv[G] = *(const f32x4 *)&wl[64 * G * IMG::K + IMG::OWX0 + k];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const int kc = k + kk < OB ? k + kk : OB - 1;
        const float xv = xs[r4 * OB + kc];
        const float av = k + kk < OB ? xv : 0.0f;
#pragma unroll
        for (int G = 0; G < NG; G++) acc[G] = PS_MFMA4(av, wv[G][kk], acc[G]);
      }
    }
  }
  PS_WAVE_SYNC();      // every lane has read the previous h of layer 0
  cell(cst[0], h0);
  PS_WAVE_SYNC();
  // ---- layer 1: the recurrent part first, then the input part (layer 0's new h, plain k order, no mask) ----
#pragma unroll
  for (int G = 0; G < NG; G++) acc[G] = (f32x4){bias[1][G], bias[1][G], bias[1][G], bias[1][G]};
  recurrent(h1, IMG::OWH1);
#pragma unroll
  for (int k = 0; k < HID; k += 4) {
    const f32x4 x4 = *(const f32x4 *)&h0[r4 * HID + k];
    f32x4 wv[NG];
#pragma unroll
    for (int G = 0; G < NG; G++) wv[G] = *(const f32x4 *)&wl[64 * G * IMG::K + IMG::OWX1 + k];
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
#pragma unroll
      for (int G = 0; G < NG; G++) acc[G] = PS_MFMA4(x4[kk], wv[G][kk], acc[G]);
  }
  PS_WAVE_SYNC();
  cell(cst[1], h1);
  PS_WAVE_SYNC();
  policy_heads_wave<HID, true, LAY, true>(a, ws, h1, h1, HID, head_w, head_w, terms, e4, l, t, gstep);
}
