// mlp_update.hpp -- one optimizer step's worth of gradients of the reference's MlpPolicy (archi/policies.py:430-446: separate
// [64, 64] tanh stacks for the policy and the value function over the 35 observations; BASELINE config 2's learner) under the PPO2
// loss (ppo2.py:152-175), forward AND backward in one launch per network.  The eager graph of this step is ~70 kernels over
// [minibatch, 64] tensors (skinny library GEMMs, tanh / tanh-backward passes, the gathers of the shuffled minibatch): 3.7 ms per
// minibatch of 768 k samples, 150 ms per 10-epoch update at 4096 x 750, every activation making several trips through HBM.  Here
// a sample's activations never leave the CU.
//
// Mapping (gfx950): a workgroup is four INDEPENDENT waves; a wave walks over tiles of 16 samples.  Everything is computed
// transposed -- Out^T[n][s] = sum_k W[k][n] In^T[k][s] -- with v_mfma_f32_16x16x4_f32 (exact f32), so that the sample index is the
// C/D column (lane & 15) in every layer and a layer's output, as it sits in the accumulators (lane (s, g) holds the features
// 16 nt + 4 g + r), IS the B operand of the next layer once the weight fragments are fetched in that permuted k order: the forward
// and the dx chain need no data movement at all.  The weight fragments of both directions stay in registers for the whole launch
// (196 per lane), next to the weight-gradient accumulators (128): the contraction of dW = In^T dOut runs over the SAMPLES, so
// its operands go through a per-wave LDS tile (written in C layout, read back sample-major).  Rows of the shuffled minibatch are
// read in place through the index vector (no gathered copies).  Per tile: 324 MFMAs (policy) / 276 (value: its 1-wide head is
// lane-local arithmetic), 4.4 / 3.7 us at the f32 MFMA rate.  Per-workgroup partial sums (fixed order: deterministic) go to `partials`; the caller adds
// the workgroups up.
#pragma once
#include "policy_step.hpp"

#define IRRL_MLP_OB 35
#define IRRL_MLP_H 64
// partial-sum row: scalars[4] | d logstd[16] | d b1[64] | d b2[64] | d b3[16] | d W1[48][64] | d W2[64][64] | d W3[64][16]
#define IRRL_MLP_P_DLS 4
#define IRRL_MLP_P_DB1 20
#define IRRL_MLP_P_DB2 84
#define IRRL_MLP_P_DB3 148
#define IRRL_MLP_P_DW1 164
#define IRRL_MLP_P_DW2 (164 + 48 * 64)
#define IRRL_MLP_P_DW3 (164 + 48 * 64 + 64 * 64)
#define IRRL_MLP_P (164 + 48 * 64 + 64 * 64 + 64 * 16)

struct MlpUpdateArgs {
  size_t n;                       // samples of this minibatch
  const int64_t *idx;             // [n] rows of the flat rollout arrays, or NULL (rows 0..n-1)
  const float *obs, *actions, *returns, *old_values, *old_neglogp;   // [rows, 35], [rows, 12], [rows] x 3
  const float *rec;               // or (bf16 kernels, REC instantiation): the same five arrays as ONE packed record per sample, see IRRL_MLP_REC below
  const float *w1, *b1, *w2, *b2, *w3, *b3;                          // [35,64] [64] [64,64] [64] [64,OUT] [OUT]
  const float *logstd, *adv_stats;                                    // [12]; (mean, std) of the raw advantages
  float cliprange, vf_coef, inv_n;
  float *partials;                // [gridDim.x, IRRL_MLP_P]
};

// PACKED SAMPLE RECORD (round 5): a minibatch row is a random sample of the flat rollout, so every tensor it touches costs whole 128-byte
// lines -- 140 B of observations over 2-3 lines, 48 B of actions over 1-2, and a line EACH for the 4-byte return, old value and old neglogp:
// 614 / 391 MB fetched per minibatch of 768 k samples by the policy / value kernel against 161 / 119 MB of payload (PMC, rounds 3-4).  The
// record holds everything a sample needs in 256 aligned bytes = exactly two lines:
//   [0, 35) observation | 35 zero | [36, 48) action | 48 return | 49 old value | 50 old neglogp | 51 raw advantage (return - old value) | zeros
// built ONCE per update by irrl_mlp_pack_kernel (the rollout does not change over the update's 10 epochs x 4 minibatches) and read by the
// REC instantiations of the bf16 gradient kernels with 16-byte loads (5 per lane and tile instead of 15 dword loads).
#define IRRL_MLP_REC 64
__global__ void __launch_bounds__(256)
irrl_mlp_pack_kernel(size_t n, const float *__restrict__ obs, const float *__restrict__ actions, const float *__restrict__ returns,
                     const float *__restrict__ old_values, const float *__restrict__ old_neglogp, float *__restrict__ rec) {
  // one lane per record word: 64 consecutive lanes write one record (256 contiguous bytes)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n * IRRL_MLP_REC; i += (size_t)gridDim.x * 256) {
    const size_t r = i / IRRL_MLP_REC;
    const int k = (int)(i % IRRL_MLP_REC);
    float v = 0.0f;
    if (k < IRRL_MLP_OB) v = obs[r * IRRL_MLP_OB + k];
    else if (k >= 36 && k < 48) v = actions[r * 12 + (k - 36)];
    else if (k == 48) v = returns[r];
    else if (k == 49) v = old_values[r];
    else if (k == 50) v = old_neglogp[r];
    else if (k == 51) v = returns[r] - old_values[r];
    rec[i] = v;
  }
}

#ifndef IRRL_MLP_EXP
#define IRRL_MLP_EXP 0   // diagnostics (wrong results): 1 = forward + loss only, 2 = no weight-gradient MFMAs; 3 = dx block before the dW block (A/B: slower, 10.7 vs 8.5 us per tile)
#endif
#define MU_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b_, c_, 0, 0, 0)
#define MU_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

struct MlpTileIn {
  float x[9];      // x[s][4 ks + g]
  f32x4 act;       // actions[s][4 g ..] (policy net)
  float ret, ov, onlp;
};

// KIND 0: policy network (DiagGaussian mean, clipped surrogate); KIND 1: value network (clipped value loss)
template <int KIND>
__global__ void __launch_bounds__(256)
irrl_mlp_ppo_kernel(const MlpUpdateArgs a) {
  constexpr int OB = IRRL_MLP_OB, H = IRRL_MLP_H, OUT = KIND == 0 ? 12 : 1;
  constexpr int XLD = 48, LD = 80;                     // row strides = 16 mod 32: the sample-major fragment reads hit 32 banks
  constexpr int WREG = 16 * XLD + 2 * 16 * LD;         // per-wave tile space: X | A (activations) | B (deltas)
  __shared__ float lds[4 * WREG];
  __shared__ float bias[H + H + 16];
  static_assert(4 * WREG >= IRRL_MLP_P, "the block reduction reuses the tile space");
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, c = l & 15, g = l >> 4;
  float *X = lds + wv * WREG, *TA = X + 16 * XLD, *TB = TA + 16 * LD;

  // ---- weight fragments (A operands: lane holds M[i = c][k = g]) ----
  float wa1[9][4], wa2[4][4][4], wa3[4][4], wt3[4][4], wt2[4][4][4];
#pragma unroll
  for (int ks = 0; ks < 9; ks++)
#pragma unroll
    for (int nt = 0; nt < 4; nt++) wa1[ks][nt] = (4 * ks + g < OB) ? a.w1[(4 * ks + g) * H + 16 * nt + c] : 0.0f;
#pragma unroll
  for (int n1 = 0; n1 < 4; n1++)
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int n2 = 0; n2 < 4; n2++) {
        wa2[n1][r][n2] = a.w2[(16 * n1 + 4 * g + r) * H + 16 * n2 + c];
        wt2[n1][n2][r] = a.w2[(16 * n1 + c) * H + 16 * n2 + 4 * g + r];     // [kt = n1][nt2 = n2][r]
      }
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (KIND == 0) {
        wa3[n2][r] = (c < OUT) ? a.w3[(16 * n2 + 4 * g + r) * OUT + c] : 0.0f;
        wt3[n2][r] = (4 * g + r < OUT) ? a.w3[(16 * n2 + c) * OUT + 4 * g + r] : 0.0f;   // [kt = n2][r]
      } else {
        wa3[n2][r] = a.w3[16 * n2 + 4 * g + r];   // the value head is one column: lane-local products, no MFMA
        wt3[n2][r] = 0.0f;
      }
    }
  if (threadIdx.x < H) { bias[threadIdx.x] = a.b1[threadIdx.x]; bias[H + threadIdx.x] = a.b2[threadIdx.x]; }
  if (threadIdx.x < 16) bias[2 * H + threadIdx.x] = (threadIdx.x < OUT) ? a.b3[threadIdx.x] : 0.0f;
  for (int i = l; i < 16 * XLD; i += 64) X[i] = 0.0f;      // columns 35..47 of the observation tile stay zero
  __syncthreads();

  // per-lane constants of the loss
  float sd_inv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ls_sum = 0.0f;
  if (KIND == 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) if (4 * g + r < OUT) sd_inv[r] = __expf(-a.logstd[4 * g + r]);
    for (int i = 0; i < OUT; i++) ls_sum += a.logstd[i];
  }
  const float a_mean = a.adv_stats[0], a_istd = 1.0f / (a.adv_stats[1] + 1e-8f);
  const float clip = a.cliprange;

  // ---- accumulators ----
  f32x4 gw1[3][4], gw2[4][4], gw3[4];
  f32x4 gb1[4], gb2[4], gb3, gls, sc;
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) { gw2[i][j] = zero4; if (i < 3) gw1[i][j] = zero4; }
    gw3[i] = zero4; gb1[i] = zero4; gb2[i] = zero4;
  }
  gb3 = zero4; gls = zero4; sc = zero4;

  const size_t ntiles = (a.n + 15) / 16, stride = (size_t)gridDim.x * 4;
  size_t tile = (size_t)blockIdx.x * 4 + wv;
  auto row_of = [&](size_t t) -> size_t {
    size_t j = t * 16 + c;
    if (j >= a.n) j = a.n - 1;
    return a.idx ? (size_t)a.idx[j] : j;
  };
  auto load_tile = [&](size_t row, MlpTileIn &in) {
    // no lane-dependent branches around the loads (a load under a branch makes the compiler drain every load in flight first: the
    // next tile's rows would be waited for as soon as they are requested); lanes beyond the row read a valid neighbour and the value is
    // masked where it is used
    const float *xr = a.obs + row * OB;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) in.x[ks] = xr[4 * ks + g];
    in.x[8] = xr[32 + (g < 3 ? g : 2)];
    if (KIND == 0) {
      in.act = *(const f32x4 *)(a.actions + row * 12 + 4 * (g < 3 ? g : 2));
      in.onlp = a.old_neglogp[row];
    }
    in.ret = a.returns[row];
    in.ov = a.old_values[row];
  };
  MlpTileIn cur, nxt;
  size_t row_next = 0;
  if (tile < ntiles) {
    load_tile(row_of(tile), cur);
    if (tile + stride < ntiles) row_next = row_of(tile + stride);
  }
  // every load of the prologue lands before the loop: with one of them still in flight at the loop's entry the compiler's wait in front of
  // the first use of `cur` (vmcnt(5), counted on the path that skips the prefetch) also waits, in every later iteration, for the
  // observations of the NEXT tile that were requested a few instructions earlier -- the whole gather latency, once per tile
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  for (; tile < ntiles; tile += stride) {
    const bool more = tile + stride < ntiles;
    if (more) {
      load_tile(row_next, nxt);
      if (tile + 2 * stride < ntiles) row_next = row_of(tile + 2 * stride);
    }
    const bool valid = tile * 16 + c < a.n;

    // ---- forward ----
    cur.x[8] = (g < 3) ? cur.x[8] : 0.0f;            // observation 35 does not exist
#pragma unroll
    for (int ks = 0; ks < 9; ks++) X[c * XLD + 4 * ks + g] = cur.x[ks];
    f32x4 h1[4], h2[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) h1[nt] = *(const f32x4 *)&bias[16 * nt + 4 * g];
#pragma unroll
    for (int ks = 0; ks < 9; ks++)
#pragma unroll
      for (int nt = 0; nt < 4; nt++) h1[nt] = MU_MFMA(wa1[ks][nt], cur.x[ks], h1[nt]);
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
      for (int r = 0; r < 4; r++) h1[nt][r] = fast_tanh(h1[nt][r]);
#pragma unroll
    for (int nt = 0; nt < 4; nt++) h2[nt] = *(const f32x4 *)&bias[H + 16 * nt + 4 * g];
#pragma unroll
    for (int n1 = 0; n1 < 4; n1++)
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int n2 = 0; n2 < 4; n2++) h2[n2] = MU_MFMA(wa2[n1][r][n2], h1[n1][r], h2[n2]);
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
      for (int r = 0; r < 4; r++) h2[nt][r] = fast_tanh(h2[nt][r]);
    // head: four independent chains (one per 16 inputs), then added
    f32x4 out;                                                 // out[r] = head output 4 g + r of sample c
    if (KIND == 0) {
      f32x4 o4[4];
      o4[0] = *(const f32x4 *)&bias[2 * H + 4 * g]; o4[1] = zero4; o4[2] = zero4; o4[3] = zero4;
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int n2 = 0; n2 < 4; n2++) o4[n2] = MU_MFMA(wa3[n2][r], h2[n2][r], o4[n2]);
      out = (o4[0] + o4[1]) + (o4[2] + o4[3]);
    } else {
      // value head: the lane's 16 features times their weights, then the four lane groups of the sample
      f32x4 pv = h2[0] * *(const f32x4 *)wa3[0];
#pragma unroll
      for (int n2 = 1; n2 < 4; n2++) pv += h2[n2] * *(const f32x4 *)wa3[n2];
      float v = (pv[0] + pv[1]) + (pv[2] + pv[3]);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      out = zero4;
      out[0] = v + bias[2 * H];
    }

    // ---- loss and d loss / d out (same arithmetic as irrl_ppo_loss_kernel) ----
    f32x4 dz3 = zero4;
    if (KIND == 0) {
      f32x4 diff;
      float q = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; r++) { diff[r] = (cur.act[r] - out[r]) * sd_inv[r]; q += diff[r] * diff[r]; }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      const float nlp = 0.5f * q + 0.918938533204672742f * (float)OUT + ls_sum;
      const float adv = (cur.ret - cur.ov - a_mean) * a_istd;
      const float ratio = __expf(cur.onlp - nlp);
      const float rc = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
      const float pg1 = -adv * ratio, pg2 = -adv * rc;
      const bool inside = (ratio >= 1.0f - clip) && (ratio <= 1.0f + clip);
      const float dpg_dratio = inside ? -adv : ((pg1 > pg2) ? -adv : ((pg1 == pg2) ? -0.5f * adv : 0.0f));
      const float dl_dnlp = valid ? a.inv_n * dpg_dratio * (-ratio) : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        dz3[r] = dl_dnlp * (-diff[r] * sd_inv[r]);
        gls[r] += (4 * g + r < OUT) ? dl_dnlp * (1.0f - diff[r] * diff[r]) : 0.0f;
      }
      if (valid && g == 0) {
        sc[0] += fmaxf(pg1, pg2);
        sc[1] += 0.5f * (nlp - cur.onlp) * (nlp - cur.onlp);
        sc[2] += (fabsf(ratio - 1.0f) > clip) ? 1.0f : 0.0f;
      }
    } else {
      const float v = out[0], ov = cur.ov, R = cur.ret;
      const float dv = v - ov;
      const float vc = ov + fminf(fmaxf(dv, -clip), clip);
      const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
      const float g_clamp = (dv >= -clip && dv <= clip) ? 1.0f : 0.0f;
      const float dvf = (l1 > l2) ? (v - R) : ((l1 < l2) ? (vc - R) * g_clamp : 0.5f * (v - R) + 0.5f * (vc - R) * g_clamp);
      dz3[0] = valid ? a.inv_n * a.vf_coef * dvf : 0.0f;     // (every lane group of the sample holds it; group 0 accounts for it)
      if (valid && g == 0) sc[0] += 0.5f * fmaxf(l1, l2);
    }
    if (KIND == 0 || g == 0) gb3 += dz3;
#if IRRL_MLP_EXP == 1
    if (more) cur = nxt;
    continue;
#endif

    // ---- backward: head ----
    f32x4 d[4];
    if (KIND == 0) {
#pragma unroll
      for (int nt = 0; nt < 4; nt++) *(f32x4 *)&TA[c * LD + 16 * nt + 4 * g] = h2[nt];
      *(f32x4 *)&TB[c * LD + 4 * g] = dz3;
      MU_WAVE_SYNC();
#pragma unroll
      for (int kt = 0; kt < 4; kt++) d[kt] = zero4;
#if IRRL_MLP_EXP == 3
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int kt = 0; kt < 4; kt++) d[kt] = MU_MFMA(wt3[kt][r], dz3[r], d[kt]);
#endif
#if IRRL_MLP_EXP != 2
#pragma unroll
      for (int st = 0; st < 4; st++) {
        const float b = TB[(4 * st + g) * LD + c];
#pragma unroll
        for (int kt = 0; kt < 4; kt++) gw3[kt] = MU_MFMA(TA[(4 * st + g) * LD + 16 * kt + c], b, gw3[kt]);
      }
#endif
#if IRRL_MLP_EXP != 3
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int kt = 0; kt < 4; kt++) d[kt] = MU_MFMA(wt3[kt][r], dz3[r], d[kt]);
#endif
    } else {
      // one output: d h2 = dv w3, d w3 += h2 dv per lane (summed over the sample lanes at the end)
#pragma unroll
      for (int kt = 0; kt < 4; kt++) {
        d[kt] = dz3[0] * *(const f32x4 *)wa3[kt];
        gw3[kt] += dz3[0] * h2[kt];
      }
    }
    f32x4 dz2[4];
#pragma unroll
    for (int kt = 0; kt < 4; kt++) {
      dz2[kt] = d[kt] * (1.0f - h2[kt] * h2[kt]);
      gb2[kt] += dz2[kt];
    }
    // ---- layer 2 ----
    MU_WAVE_SYNC();
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
      *(f32x4 *)&TA[c * LD + 16 * nt + 4 * g] = h1[nt];
      *(f32x4 *)&TB[c * LD + 16 * nt + 4 * g] = dz2[nt];
    }
    MU_WAVE_SYNC();
#pragma unroll
    for (int kt = 0; kt < 4; kt++) d[kt] = zero4;
#if IRRL_MLP_EXP == 3
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++)
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int kt = 0; kt < 4; kt++) d[kt] = MU_MFMA(wt2[kt][n2][r], dz2[n2][r], d[kt]);
#endif
#if IRRL_MLP_EXP != 2
#pragma unroll
    for (int st = 0; st < 4; st++) {
      float b[4];
#pragma unroll
      for (int nt = 0; nt < 4; nt++) b[nt] = TB[(4 * st + g) * LD + 16 * nt + c];
#pragma unroll
      for (int kt = 0; kt < 4; kt++) {
        const float av = TA[(4 * st + g) * LD + 16 * kt + c];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) gw2[kt][nt] = MU_MFMA(av, b[nt], gw2[kt][nt]);
      }
    }
#endif
#if IRRL_MLP_EXP != 3
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++)
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int kt = 0; kt < 4; kt++) d[kt] = MU_MFMA(wt2[kt][n2][r], dz2[n2][r], d[kt]);
#endif
    f32x4 dz1[4];
#pragma unroll
    for (int kt = 0; kt < 4; kt++) {
      dz1[kt] = d[kt] * (1.0f - h1[kt] * h1[kt]);
      gb1[kt] += dz1[kt];
    }
    // ---- layer 1 ----
    MU_WAVE_SYNC();
#pragma unroll
    for (int nt = 0; nt < 4; nt++) *(f32x4 *)&TB[c * LD + 16 * nt + 4 * g] = dz1[nt];
    MU_WAVE_SYNC();
#if IRRL_MLP_EXP != 2
#pragma unroll
    for (int st = 0; st < 4; st++) {
      float b[4];
#pragma unroll
      for (int nt = 0; nt < 4; nt++) b[nt] = TB[(4 * st + g) * LD + 16 * nt + c];
#pragma unroll
      for (int kt = 0; kt < 3; kt++) {
        const float av = X[(4 * st + g) * XLD + 16 * kt + c];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) gw1[kt][nt] = MU_MFMA(av, b[nt], gw1[kt][nt]);
      }
    }
#endif
    MU_WAVE_SYNC();
    if (more) cur = nxt;
  }

  // ---- reduction: bias / logstd / scalar sums over the 16 sample lanes, then the four waves in order through LDS ----
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
      for (int nt = 0; nt < 4; nt++) { gb1[nt][r] += __shfl_xor(gb1[nt][r], off, 64); gb2[nt][r] += __shfl_xor(gb2[nt][r], off, 64); }
      gb3[r] += __shfl_xor(gb3[r], off, 64);
      if (KIND == 1) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) gw3[nt][r] += __shfl_xor(gw3[nt][r], off, 64);
      }
      gls[r] += __shfl_xor(gls[r], off, 64);
      sc[r] += __shfl_xor(sc[r], off, 64);
    }
  }
  __syncthreads();                 // every wave is done with its tiles: the space is reused
  float *red = lds;
  for (int w = 0; w < 4; w++) {
    if (wv == w) {
      const bool first = w == 0;
      auto put = [&](int i, float v) { red[i] = first ? v : red[i] + v; };
      if (c == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
          for (int nt = 0; nt < 4; nt++) { put(IRRL_MLP_P_DB1 + 16 * nt + 4 * g + r, gb1[nt][r]); put(IRRL_MLP_P_DB2 + 16 * nt + 4 * g + r, gb2[nt][r]); }
          put(IRRL_MLP_P_DB3 + 4 * g + r, gb3[r]);
          put(IRRL_MLP_P_DLS + 4 * g + r, gls[r]);
          if (g == 0) put(r, sc[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
#pragma unroll
          for (int kt = 0; kt < 4; kt++) {
            put(IRRL_MLP_P_DW2 + (16 * kt + 4 * g + r) * H + 16 * nt + c, gw2[kt][nt][r]);
            if (kt < 3) put(IRRL_MLP_P_DW1 + (16 * kt + 4 * g + r) * H + 16 * nt + c, gw1[kt][nt][r]);
          }
          // policy: gw3 is a D tile (row = input 16 nt + 4 g + r, column = action c); value: the lane's input 16 nt + 4 g + r, column 0
          put(IRRL_MLP_P_DW3 + (16 * nt + 4 * g + r) * 16 + c, (KIND == 0 || c == 0) ? gw3[nt][r] : 0.0f);
        }
      }
    }
    __syncthreads();
  }
  float *dst = a.partials + (size_t)blockIdx.x * IRRL_MLP_P;
  for (int i = threadIdx.x; i < IRRL_MLP_P; i += 256) dst[i] = red[i];
}

// ---- moments of the raw advantages of one minibatch: sum (R - V), sum (R - V)^2 over the indexed rows, accumulated in double ------
// (ppo2.py:262-263 normalises the advantages per minibatch; the eager form -- two gathers, two casts, a square, two reductions --
// is ~8 launches per optimizer step.)  Two launches, fixed order: per-workgroup partials, then one workgroup adds them.
__global__ void __launch_bounds__(256)
irrl_adv_moments_kernel(const int64_t *__restrict__ idx, size_t n, const float *__restrict__ ret, const float *__restrict__ val, double *__restrict__ part,
                        size_t stride) {
  __shared__ double red[2][4];
  double s = 0.0, ss = 0.0;
  for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (size_t)gridDim.x * 256) {
    const size_t r = (idx ? (size_t)idx[j] : j) * stride;      // stride 1: plain arrays; IRRL_MLP_REC: word 51 of the packed records (val == NULL)
    const double a = (double)(val ? ret[r] - val[r] : ret[r]);   // the advantage is formed in f32, as the rollout stores it (val == NULL: ret IS the advantage)
    s += a; ss += a * a;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); ss += __shfl_down(ss, off, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; }
  __syncthreads();
  if (threadIdx.x < 2) part[2 * blockIdx.x + threadIdx.x] = ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
}
__global__ void __launch_bounds__(64)
irrl_adv_moments_final_kernel(const double *__restrict__ part, int blocks, size_t n, double *__restrict__ sums, float *__restrict__ stats) {
  double s = 0.0, ss = 0.0;
  for (int b = threadIdx.x; b < blocks; b += 64) { s += part[2 * b]; ss += part[2 * b + 1]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); ss += __shfl_down(ss, off, 64); }
  if (threadIdx.x == 0) {
    sums[0] = s; sums[1] = ss; sums[2] = (double)n;
    if (stats) {     // single-process job: (mean, std) as the loss kernels read them -- the same double arithmetic the host side applies to `sums`
      const double mean = s / (double)n, var = fmax(ss / (double)n - mean * mean, 0.0);
      stats[0] = (float)mean; stats[1] = (float)sqrt(var);
    }
  }
}
