// lstm_kernels.hip -- persistent sequence kernels for the stable-baselines LSTM of the reference's
// CustomLSTMPolicy (run_bp_v5.py:143-176; a2c.utils.lstm: z = x wx + h wh + b, gates i,f,o,g, state [c,h],
// state *= (1 - mask) before every step).  These replace the 750-step eager unroll of the PPO2 update
// (ppo2.py:132-134: the train graph back-propagates through the whole rollout) -- SURVEY 8f-2.
//
// Split of work (gfx950):
//   * everything that is NOT sequential is left as large GEMMs for the library (torch.matmul -> rocBLAS):
//     zx = x wx + b for all T*N rows, and in the backward pass dwx = x^T dz, dwh = h_prev^T dz, dx = dz wx^T.
//   * the recurrence -- z_t = zx_t + (h_{t-1} keep_t) wh, the cell, and in reverse dh_{t-1} = keep_t (dz_t wh^T)
//     -- runs in ONE launch per layer: a workgroup owns 16 envs for all T steps; its HID/16 waves own 16 hidden
//     units each, keep their slice of wh in VGPRs as MFMA B-fragments for the whole sequence, and accumulate
//     with v_mfma_f32_16x16x4_f32 (exact f32: bitwise an fmaf chain, so parity with the eager f32 graph is at
//     rounding level).  Per step and wave: 48 MFMAs + lane-local cell math; h (forward) / partial dh (backward)
//     is exchanged between the waves through a few KB of LDS.
//   * gate columns are permuted to [unit][gate] so that the four gates of a unit sit in one lane (four
//     accumulators, same C/D slot) and every global access of zx / gates / dz is a 16-byte vector.
//
// Layouts (all f32): zx, gates, dz [T, N, HID, 4]; cseq, hseq, dh_in [T, N, HID]; masks [T, N] (1.0 = episode
// ended before step t); state0, state_out [N, 2*HID] = [c | h]; wh_p [HID(k)][HID(unit)][4(gate)].
// N must be a multiple of 16 (the Python side pads).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "policy_step.hpp"   // f32x4, LSTM_DEV, fast_sigmoid / fast_tanh, the rollout step of the LSTM policy
#include "lstm_bf16.hpp"     // the sequence kernels on the bf16 matrix cores with compensated operand splits (round 4)

template <int HID>
__global__ void __launch_bounds__(HID / 16 * 64)
lstm_seq_fwd_kernel(const float *__restrict__ zx, const float *__restrict__ wh_p, const float *__restrict__ masks,
                    const float *__restrict__ state0, float *__restrict__ gates, float *__restrict__ cseq,
                    float *__restrict__ hseq, float *__restrict__ state_out, int T, int N) {
  constexpr int KS = HID / 4;       // k-steps of the 16x16x4 MFMA over the hidden index
  constexpr int LD = HID + 1;       // padded LDS row (bank-conflict-free A-fragment reads)
  __shared__ float hbuf[2][16 * LD];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int col = l & 15, rq = l >> 4;
  const int e0 = blockIdx.x * 16;
  const int u = 16 * w + col;  // hidden unit owned by this lane (C/D column)
  // B fragments: B[k = 4kk + rq][j = col] = wh[k][unit u][gate g]
  float bw[KS][4];
#pragma unroll
  for (int kk = 0; kk < KS; kk++)
#pragma unroll
    for (int g = 0; g < 4; g++) bw[kk][g] = wh_p[((size_t)(4 * kk + rq) * HID + u) * 4 + g];
  float c[4], hlast[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int e = e0 + 4 * rq + j;  // C/D rows of this lane
    c[j] = state0[(size_t)e * 2 * HID + u];
    hlast[j] = state0[(size_t)e * 2 * HID + HID + u];
    hbuf[0][(4 * rq + j) * LD + u] = hlast[j];
  }
  f32x4 zcur[4];
#pragma unroll
  for (int j = 0; j < 4; j++) zcur[j] = *(const f32x4 *)&zx[(((size_t)0 * N + e0 + 4 * rq + j) * HID + u) * 4];
  float mA_cur = masks[e0 + col], mC_cur[4];
#pragma unroll
  for (int j = 0; j < 4; j++) mC_cur[j] = masks[e0 + 4 * rq + j];
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < T; t++) {
    // prefetch the next step's input projection while this step computes
    f32x4 znext[4];
    const int tn = (t + 1 < T) ? t + 1 : t;
#pragma unroll
    for (int j = 0; j < 4; j++) znext[j] = *(const f32x4 *)&zx[(((size_t)tn * N + e0 + 4 * rq + j) * HID + u) * 4];
    const float keepA = 1.0f - mA_cur;  // A rows are envs e0 + (l & 15)
    float keepC[4];
#pragma unroll
    for (int j = 0; j < 4; j++) keepC[j] = 1.0f - mC_cur[j];
    const float mA_next = masks[(size_t)tn * N + e0 + col];
    float mC_next[4];
#pragma unroll
    for (int j = 0; j < 4; j++) mC_next[j] = masks[(size_t)tn * N + e0 + 4 * rq + j];
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; g++) acc[g] = (f32x4){zcur[0][g], zcur[1][g], zcur[2][g], zcur[3][g]};
    const float *hb = hbuf[cur];
#pragma unroll
    for (int kk = 0; kk < KS; kk++) {
      const float a = hb[col * LD + 4 * kk + rq] * keepA;  // A[i = col][k = 4kk + rq] = h_{t-1}[env i][k] * keep
#pragma unroll
      for (int g = 0; g < 4; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[kk][g], acc[g], 0, 0, 0);
    }
    float *hn = hbuf[cur ^ 1];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float ig = fast_sigmoid(acc[0][j]), fg = fast_sigmoid(acc[1][j]), og = fast_sigmoid(acc[2][j]), gg = fast_tanh(acc[3][j]);
      const float cn = fg * (c[j] * keepC[j]) + ig * gg;
      const float hn_ = og * fast_tanh(cn);
      c[j] = cn;
      hlast[j] = hn_;
      const size_t row = (size_t)t * N + e0 + 4 * rq + j;
      *(f32x4 *)&gates[(row * HID + u) * 4] = (f32x4){ig, fg, og, gg};
      cseq[row * HID + u] = cn;
      hseq[row * HID + u] = hn_;
      hn[(4 * rq + j) * LD + u] = hn_;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) { zcur[j] = znext[j]; mC_cur[j] = mC_next[j]; }
    mA_cur = mA_next;
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int e = e0 + 4 * rq + j;
    state_out[(size_t)e * 2 * HID + u] = c[j];
    state_out[(size_t)e * 2 * HID + HID + u] = hlast[j];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Forward sequence kernel with the INPUT PROJECTION fused in: z_t = b + x_t wx + (h_{t-1} keep_t) wh.  Same mapping
// as lstm_seq_fwd_kernel, plus the wave's slice of wx as a second set of resident B fragments; x_t is read straight
// from the layer input [T, N, n_in] (prefetched one step ahead), so the [T, N, 4H] zx tensor -- 2.4 GB written by a
// GEMM and read back here at the training shape -- never exists.  KXS = ceil(n_in / 4) k-steps.
// HELPER adds one wave that computes the input projection b + x_{t+1} wx for ALL gate columns one step ahead (it does not
// depend on h) and hands it over through a double-buffered LDS tile; the recurrence waves then start from that tile and issue
// only the HID/4 x 4 recurrent MFMAs before their gate arithmetic.
// SPLIT (with HELPER): the k range of the input projection is shared out so that the helper's SIMD is not the bottleneck -- two
// forward kernels (actor and critic) share a CU, so both helpers sit on the same SIMD and at 12 column tiles x KXS MFMAs per step
// each they alone set the step time (29-32 % MFMA busy overall).  The helper keeps all k-steps but one 16-element group; every
// recurrence wave adds that group for its own columns (4 k-steps x 4 gates = 16 MFMAs, issued before the previous h is needed).
// Whole 16-element groups of the input row travel as 16-byte vectors: lane (env, rq) holds elements 16 m + 4 rq + j and the
// k-step (m, j) pairs them with the wx rows of the same index (a permutation of the k order inside the group: rounding only).
template <int HID, int KXS, bool HELPER, bool SPLIT = false>
__global__ void __launch_bounds__((HID / 16 + (HELPER ? 1 : 0)) * 64)
lstm_seq_fwd_x_kernel(const float *__restrict__ x, const float *__restrict__ wx_p, const float *__restrict__ b_p,
                      const float *__restrict__ wh_p, const float *__restrict__ masks, const float *__restrict__ state0,
                      float *__restrict__ gates, float *__restrict__ cseq, float *__restrict__ hseq,
                      float *__restrict__ state_out, int T, int N, int n_in) {
  constexpr int KS = HID / 4;
  constexpr int LD = HID + 1;
  constexpr int NW = HID / 16;
  constexpr int LDZX = 4 * HID + 4;
  __shared__ float hbuf[2][16 * LD];
  __shared__ float zxbuf[HELPER ? 2 : 1][HELPER ? 16 * LDZX : 1];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int col = l & 15, rq = l >> 4;
  const int e0 = blockIdx.x * 16;
  const int u = 16 * w + col;
  static_assert(!SPLIT || HELPER, "SPLIT shares the input projection between the helper and the recurrence waves");
  constexpr int NG = SPLIT ? KXS / 4 : 0;      // whole 16-element groups of the input row, fetched as vectors (the host checks n_in >= 16 NG)
  constexpr int GW = SPLIT ? NG - 1 : -1;      // the group the recurrence waves keep
  // input element that k-step kk pairs with lane rq
  auto elem = [&](int kk) { return kk < 4 * NG ? 16 * (kk / 4) + 4 * rq + (kk % 4) : 4 * kk + rq; };
  if (HELPER && w == NW) {
    // ---- helper wave: zx_t[env][c] = b[c] + sum_k x_t[env][k] wx[k][c] for every permuted gate column c, one step ahead ----
    // (with SPLIT: over the k-steps outside group GW)
    float bxh[KXS][4 * NW], bias_h[4 * NW];
#pragma unroll
    for (int ct = 0; ct < 4 * NW; ct++) {
      bias_h[ct] = b_p[16 * ct + col];
#pragma unroll
      for (int kk = 0; kk < KXS; kk++) {
        const int k = elem(kk);
        bxh[kk][ct] = (k < n_in) ? wx_p[(size_t)k * HID * 4 + 16 * ct + col] : 0.0f;
      }
    }
    float xh[KXS], xhn[KXS];
    auto fetch_h = [&](int t, float (&dst)[KXS]) {
      const float *row = x + ((size_t)t * N + e0 + col) * n_in;
#pragma unroll
      for (int m = 0; m < NG; m++) {
        if (m == GW) continue;
        const f32x4u v4 = *(const f32x4u *)&row[16 * m + 4 * rq];
        dst[4 * m + 0] = v4[0]; dst[4 * m + 1] = v4[1]; dst[4 * m + 2] = v4[2]; dst[4 * m + 3] = v4[3];
      }
#pragma unroll
      for (int kk = 4 * NG; kk < KXS; kk++) {
        const int k = 4 * kk + rq;
        dst[kk] = row[k < n_in ? k : n_in - 1];
      }
    };
    fetch_h(0, xh);
#ifdef IRRL_PROFILE_FWD   /* diagnostic build (tools/lstm_fwd_phases.py) */
    unsigned long long hw_wait = 0, hw_work = 0, hts = wall_clock64();
#endif
    // two steps per trip, the x registers used alternately: copying xhn to xh at the end of a step would make the helper wait, in
    // the SAME step, for the load it issued at its start (~1 us of exposed HBM latency per step: it was at work for 2.1 us with 0.8 us
    // of MFMAs); now the load of step t + 1 is first touched one whole step after it was issued
    auto helper_step = [&](int t, float (&xc)[KXS], float (&xn_)[KXS]) {   // iteration t produces zx_t; its barrier pairs with the recurrence waves'
      if (t < T) {
        if (t + 1 < T) fetch_h(t + 1, xn_);
        float *zb = zxbuf[t & 1];
#pragma unroll
        for (int ct = 0; ct < 4 * NW; ct++) {
          f32x4 acc = (f32x4){bias_h[ct], bias_h[ct], bias_h[ct], bias_h[ct]};
#pragma unroll
          for (int kk = 0; kk < KXS; kk++)
            if (kk / 4 != GW || kk >= 4 * NG) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[kk], bxh[kk][ct], acc, 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; j++) zb[(4 * rq + j) * LDZX + 16 * ct + col] = acc[j];
        }
      }
#ifdef IRRL_PROFILE_FWD
      { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = wall_clock64(); hw_work += n_ - hts; hts = n_; __builtin_amdgcn_sched_barrier(0); }
#endif
      __syncthreads();
#ifdef IRRL_PROFILE_FWD
      { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = wall_clock64(); hw_wait += n_ - hts; hts = n_; __builtin_amdgcn_sched_barrier(0); }
#endif
    };
    for (int t = 0; t <= T; t += 2) {
      helper_step(t, xh, xhn);
      if (t + 1 <= T) helper_step(t + 1, xhn, xh);
    }
#ifdef IRRL_PROFILE_FWD
    if (blockIdx.x == gridDim.x / 2 && l == 0) { float *o_ = state_out + (size_t)N * 2 * HID; o_[8] = (float)hw_wait; o_[9] = (float)hw_work; }   // one row behind the states: the profiling caller allocates it
#endif
    return;
  }
  float bw[KS][4], bx[HELPER ? (SPLIT ? 4 : 1) : KXS][4];
#pragma unroll
  for (int kk = 0; kk < KS; kk++)
#pragma unroll
    for (int g = 0; g < 4; g++) bw[kk][g] = wh_p[((size_t)(4 * kk + rq) * HID + u) * 4 + g];
  if (!HELPER) {
#pragma unroll
    for (int kk = 0; kk < KXS; kk++) {
      const int k = 4 * kk + rq;
#pragma unroll
      for (int g = 0; g < 4; g++) bx[kk][g] = (k < n_in) ? wx_p[((size_t)k * HID + u) * 4 + g] : 0.0f;
    }
  }
  if (SPLIT) {   // this wave's share of the input projection: group GW against its own 16 units x 4 gates
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int g = 0; g < 4; g++) bx[j][g] = wx_p[((size_t)(16 * GW + 4 * rq + j) * HID + u) * 4 + g];
  }
  f32x4 xw = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  auto fetch_w = [&](int t) -> f32x4 { return *(const f32x4u *)&x[((size_t)t * N + e0 + col) * n_in + 16 * (GW < 0 ? 0 : GW) + 4 * rq]; };
  if (SPLIT) xw = fetch_w(0);
  const f32x4 bias = *(const f32x4 *)&b_p[u * 4];
  float c[4], hlast[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int e = e0 + 4 * rq + j;
    c[j] = state0[(size_t)e * 2 * HID + u];
    hlast[j] = state0[(size_t)e * 2 * HID + HID + u];
    hbuf[0][(4 * rq + j) * LD + u] = hlast[j];
  }
  // A fragments of x_t: A[i = env col][k = 4kk + rq]; columns >= n_in are clamped (their B rows are zero)
  auto fetch_x = [&](int t, float (&dst)[KXS]) {
    const float *row = x + ((size_t)t * N + e0 + col) * n_in;
#pragma unroll
    for (int kk = 0; kk < KXS; kk++) {
      const int k = 4 * kk + rq;
      dst[kk] = row[k < n_in ? k : n_in - 1];
    }
  };
  // per-step inputs fetched one step ahead; two sets used alternately (no register copy of a load that is still in flight, see the helper)
  struct StepIn { float xa[KXS]; f32x4 xw; float mA, mC[4]; };
  StepIn in0, in1;
  if (!HELPER) fetch_x(0, in0.xa);
  in0.xw = xw;
  in0.mA = masks[e0 + col];
#pragma unroll
  for (int j = 0; j < 4; j++) in0.mC[j] = masks[e0 + 4 * rq + j];
  __syncthreads();
#ifdef IRRL_PROFILE_FWD
  unsigned long long ph_[4] = {0, 0, 0, 0}, pts_ = wall_clock64();
#define IRRL_FW_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = wall_clock64(); ph_[i] += n_ - pts_; pts_ = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define IRRL_FW_STAMP(i) do { } while (0)
#endif
  auto rec_step = [&](int t, int cur, StepIn &ic, StepIn &inx) {
    const int tn = (t + 1 < T) ? t + 1 : t;
    if (!HELPER) fetch_x(tn, inx.xa);
    if (SPLIT) inx.xw = fetch_w(tn);
    const float keepA = 1.0f - ic.mA;
    float keepC[4];
#pragma unroll
    for (int j = 0; j < 4; j++) keepC[j] = 1.0f - ic.mC[j];
    inx.mA = masks[(size_t)tn * N + e0 + col];
#pragma unroll
    for (int j = 0; j < 4; j++) inx.mC[j] = masks[(size_t)tn * N + e0 + 4 * rq + j];
    f32x4 acc[4];
    if (HELPER) {
      // b + x_t wx from the helper wave's tile (published before the barrier that ended the previous step)
      const float *zb = zxbuf[t & 1];
      f32x4 zr[4];
#pragma unroll
      for (int j = 0; j < 4; j++) zr[j] = *(const f32x4 *)&zb[(4 * rq + j) * LDZX + 4 * u];
#pragma unroll
      for (int g = 0; g < 4; g++) acc[g] = (f32x4){zr[0][g], zr[1][g], zr[2][g], zr[3][g]};
      if (SPLIT) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int g = 0; g < 4; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ic.xw[j], bx[j][g], acc[g], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; g++) acc[g] = (f32x4){bias[g], bias[g], bias[g], bias[g]};
      // the input half first: it does not depend on the previous step's h (issued while the other waves still publish it)
#pragma unroll
      for (int kk = 0; kk < KXS; kk++)
#pragma unroll
        for (int g = 0; g < 4; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ic.xa[kk], bx[kk][g], acc[g], 0, 0, 0);
    }
    IRRL_FW_STAMP(0);   // prefetch issue, the helper's tile, this wave's share of the input projection
    const float *hb = hbuf[cur];
    float hv[KS];
#pragma unroll
    for (int kk = 0; kk < KS; kk++) hv[kk] = hb[col * LD + 4 * kk + rq] * keepA;
    __builtin_amdgcn_sched_barrier(0);   // all LDS reads of h in front of the MFMAs (a read behind an MFMA is exposed latency)
#pragma unroll
    for (int kk = 0; kk < KS; kk++) {
#pragma unroll
      for (int g = 0; g < 4; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[kk], bw[kk][g], acc[g], 0, 0, 0);
    }
    IRRL_FW_STAMP(1);   // recurrent MFMAs
    float *hn = hbuf[cur ^ 1];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float ig = fast_sigmoid(acc[0][j]), fg = fast_sigmoid(acc[1][j]), og = fast_sigmoid(acc[2][j]), gg = fast_tanh(acc[3][j]);
      const float cn = fg * (c[j] * keepC[j]) + ig * gg;
      const float hn_ = og * fast_tanh(cn);
      c[j] = cn;
      hlast[j] = hn_;
      const size_t row = (size_t)t * N + e0 + 4 * rq + j;
      *(f32x4 *)&gates[(row * HID + u) * 4] = (f32x4){ig, fg, og, gg};
      cseq[row * HID + u] = cn;
      hseq[row * HID + u] = hn_;
      hn[(4 * rq + j) * LD + u] = hn_;
    }
    IRRL_FW_STAMP(2);   // cell, stores, h to LDS
    __syncthreads();
    IRRL_FW_STAMP(3);   // barrier
  };
  for (int t = 0; t < T; t += 2) {
    rec_step(t, 0, in0, in1);
    if (t + 1 < T) rec_step(t + 1, 1, in1, in0);
  }
#ifdef IRRL_PROFILE_FWD
  if (blockIdx.x == gridDim.x / 2 && w == 0 && l == 0)
    for (int i = 0; i < 4; i++) (state_out + (size_t)N * 2 * HID)[i] = (float)ph_[i];
#endif
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int e = e0 + 4 * rq + j;
    state_out[(size_t)e * 2 * HID + u] = c[j];
    state_out[(size_t)e * 2 * HID + HID + u] = hlast[j];
  }
}

template <int HID>
__global__ void __launch_bounds__(HID / 16 * 64)
lstm_seq_bwd_kernel(const float *__restrict__ gates, const float *__restrict__ cseq, const float *__restrict__ masks,
                    const float *__restrict__ state0, const float *__restrict__ dh_in, const float *__restrict__ wh_p,
                    float *__restrict__ dz, int T, int N) {
  constexpr int NW = HID / 16;   // waves per workgroup
  constexpr int LDZ = 64 + 4;    // padded row of a wave's dz staging tile (16 envs x 64 gate columns)
  constexpr int LDP = HID + 1;   // padded row of the partial dh tiles
  __shared__ float dzbuf[NW][16 * LDZ];
  __shared__ float part[2][NW][16 * LDP];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int col = l & 15, rq = l >> 4;
  const int e0 = blockIdx.x * 16;
  const int u = 16 * w + col;
  // B fragments of dh_prev = dz wh^T restricted to this wave's 64 gate columns (K-split over the waves):
  // B[kc = 4kk + rq][j = col] = wh[hidden 16 nt + col][permuted column 64 w + kc]
  float bT[16][NW];
#pragma unroll
  for (int kk = 0; kk < 16; kk++)
#pragma unroll
    for (int nt = 0; nt < NW; nt++) bT[kk][nt] = wh_p[(size_t)(16 * nt + col) * HID * 4 + 64 * w + 4 * kk + rq];
  float dc[4] = {0.0f, 0.0f, 0.0f, 0.0f}, dhrec[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  int pb = 0;
  // software pipeline over time: the operands of step t-1 are requested while step t computes (a step is only a few
  // thousand cycles, so an un-prefetched global load would be most of it)
  f32x4 g_n[4];
  float ct_n[4], cp_n[4], dh_n[4], mk_n[4];
  auto fetch = [&](int t) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int e = e0 + 4 * rq + j;
      const size_t row = (size_t)t * N + e;
      mk_n[j] = masks[row];
      g_n[j] = *(const f32x4 *)&gates[(row * HID + u) * 4];
      ct_n[j] = cseq[row * HID + u];
      cp_n[j] = (t > 0) ? cseq[(row - N) * HID + u] : state0[(size_t)e * 2 * HID + u];
      dh_n[j] = dh_in[row * HID + u];
    }
  };
  fetch(T - 1);
  for (int t = T - 1; t >= 0; t--) {
    float keepC[4];
    f32x4 dz4[4], g4[4];
    float ct[4], cpv[4], dhv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { keepC[j] = 1.0f - mk_n[j]; g4[j] = g_n[j]; ct[j] = ct_n[j]; cpv[j] = cp_n[j]; dhv[j] = dh_n[j]; }
    if (t > 0) fetch(t - 1);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int e = e0 + 4 * rq + j;
      const size_t row = (size_t)t * N + e;
      const float cprev = cpv[j] * keepC[j];
      const float dh = dhv[j] + dhrec[j];
      const float ig = g4[j][0], fg = g4[j][1], og = g4[j][2], gg = g4[j][3];
      const float tc = fast_tanh(ct[j]);
      const float d_o = dh * tc;
      const float dct = dc[j] + dh * og * (1.0f - tc * tc);
      const float d_i = dct * gg, d_g = dct * ig, d_f = dct * cprev;
      dc[j] = dct * fg * keepC[j];
      dz4[j] = (f32x4){d_i * ig * (1.0f - ig), d_f * fg * (1.0f - fg), d_o * og * (1.0f - og), d_g * (1.0f - gg * gg)};
      *(f32x4 *)&dz[(row * HID + u) * 4] = dz4[j];
      // stage for the A operand: dzbuf[w][env row][local column = 4 * (u - 16 w) + gate]
      *(f32x4 *)&dzbuf[w][(4 * rq + j) * LDZ + 4 * col] = dz4[j];
    }
    // own-wave data only: LDS writes of this wave must land before its reads (no workgroup barrier needed)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x4 acc[NW];
#pragma unroll
    for (int nt = 0; nt < NW; nt++) acc[nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
      const float a = dzbuf[w][col * LDZ + 4 * kk + rq];  // A[i = env col][k = 4kk + rq]
#pragma unroll
      for (int nt = 0; nt < NW; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bT[kk][nt], acc[nt], 0, 0, 0);
    }
    // publish this wave's partial dh_prev [16 envs x HID]; sum the NW partials for the own units
#pragma unroll
    for (int nt = 0; nt < NW; nt++)
#pragma unroll
      for (int j = 0; j < 4; j++) part[pb][w][(4 * rq + j) * LDP + 16 * nt + col] = acc[nt][j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float s = 0.0f;
#pragma unroll
      for (int ww = 0; ww < NW; ww++) s += part[pb][ww][(4 * rq + j) * LDP + u];
      dhrec[j] = s * keepC[j];  // h_{t-1} entered step t multiplied by keep_t
    }
    pb ^= 1;  // double-buffered partials: the next step's writes cannot race with slower readers of this one
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward sequence kernel with EVERYTHING that consumes dz fused in.  Besides the recurrence dh_{t-1} = keep_t (dz_t
// wh^T) of lstm_seq_bwd_kernel, every step also feeds the same LDS-staged dz tile into
//   dx_t   = dz_t wx^T                      (second K-split MFMA product, summed over the waves like dh)
//   dwh   += (h_{t-1} keep_t)^T dz_t,  dwx += x_t^T dz_t      (per-workgroup accumulators held in registers for all T)
//   db    += sum_env dz_t                   (per-lane running sums)
// so dz [T, N, 4H] is never written and the three tall GEMMs + the bias reduction that used to re-read it (4 x 2.4 GB
// per layer and pass at the training shape) disappear.  Each workgroup ends with the weight gradients of its 16 envs;
// they are stored as partials [N/16, ...] and summed by one small deterministic reduction on the host side.
// MX = 3 M-tiles cover an input width of up to 48.
// HELPER adds one wave that does nothing but the dwh accumulation for all 4 HID gate columns (HID/16 x 4 HID/16 tiles,
// 144 MFMAs per step at HID = 48) from the double-buffered dz / h_prev tiles of the step the other waves just finished:
// with HID = 48 that is a fourth wave for the CU's fourth SIMD (the kernel's register budget allows one workgroup per
// CU), and the recurrence waves drop from 192 to 144 MFMAs per step.
#ifndef IRRL_BWD_MIN_WAVES
#define IRRL_BWD_MIN_WAVES 1   /* waves per SIMD the backward kernel is compiled for (2: two workgroups -- both stacks -- share a CU) */
#endif
// HX (with HELPER): input M-tiles of the dwx accumulation that the helper takes over as well (the last HX of the MX = 3), 16 MFMAs
// per tile and step off every recurrence wave -- they are the critical path (their MFMAs and their gate arithmetic do not
// overlap), the helper's SIMD has room: 144 + 48 HX MFMAs per step against 144 - 16 HX (96 - 16 HX without dx) plus ~3 k cycles.
template <int HID, bool NEED_DX, bool HELPER, int HX = 0>
__global__ void __launch_bounds__((HID / 16 + (HELPER ? 1 : 0)) * 64, IRRL_BWD_MIN_WAVES)
lstm_seq_bwd_x_kernel(const float *__restrict__ gates, const float *__restrict__ cseq, const float *__restrict__ hseq,
                      const float *__restrict__ x, const float *__restrict__ masks, const float *__restrict__ state0,
                      const float *__restrict__ dh_in, const float *__restrict__ wh_p, const float *__restrict__ wx_p,
                      float *__restrict__ dx, float *__restrict__ dwx_part, float *__restrict__ dwh_part,
                      float *__restrict__ db_part, int T, int N, int n_in) {
  constexpr int NW = HID / 16;
  // dwh M-tiles (16 hidden rows each) kept by the recurrence waves: all of them without the helper, none with it (giving them
  // one tile back when they do not carry the dx product -- 112 : 96 MFMAs instead of 96 : 144 -- measured slower: the
  // recurrence waves are the critical path)
  constexpr int HM = HELPER ? 0 : NW;
  constexpr int MX = 3;                       // input M-tiles (n_in <= 48)
  static_assert(HX >= 0 && HX <= MX && (HX == 0 || HELPER), "HX: dwx M-tiles accumulated by the helper wave");
  constexpr int MXW = MX - HX;                // dwx M-tiles the recurrence waves accumulate (for their own 64 gate columns)
  constexpr int XPW = (MX + NW - 1) / NW;     // input tiles staged / reduced per wave
  constexpr int LDZ = 64 + 4;
  constexpr int LDH = HID + 1;
  constexpr int LDX = 16 * MX + 1;
  constexpr int LDP = HID + (NEED_DX ? 16 * MX : 0) + 1;
  __shared__ float dzbuf[HELPER ? 2 : 1][NW][16 * LDZ];
  __shared__ float hpbuf[2][16 * LDH];
  __shared__ float xbuf[2][16 * LDX];
  __shared__ float part[2][NW][16 * LDP];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int col = l & 15, rq = l >> 4;
  const int e0 = blockIdx.x * 16;
  const int u = 16 * w + col;
  if (HELPER && w == NW) {
    // ---- helper wave: dwh[k][c] += sum_env hprev[env][k] dz[env][c] for every column c, one step behind the barrier ----
    f32x4 accH[NW][4 * NW], accXh[HX > 0 ? HX : 1][4 * NW];
#pragma unroll
    for (int mt = HM; mt < NW; mt++)
#pragma unroll
      for (int ct = 0; ct < 4 * NW; ct++) accH[mt][ct] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < HX; q++)
#pragma unroll
      for (int ct = 0; ct < 4 * NW; ct++) accXh[q][ct] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    int pbh = 0;
#ifdef IRRL_PROFILE_BWD   /* diagnostic build (tools/lstm_bwd_phases.py): where a step goes, summed over the T steps */
    unsigned long long hw_wait = 0, hw_work = 0, hts = wall_clock64();
#endif
    for (int t = T - 1; t >= 0; t--) {
#ifdef IRRL_PROFILE_BWD
      { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = wall_clock64(); hw_work += n_ - hts; hts = n_; __builtin_amdgcn_sched_barrier(0); }
#endif
      __syncthreads();   // the recurrence waves have published dz_t and h_prev_t in buffers [pbh]
#ifdef IRRL_PROFILE_BWD
      { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = wall_clock64(); hw_wait += n_ - hts; hts = n_; __builtin_amdgcn_sched_barrier(0); }
#endif
      const float *hpp = hpbuf[pbh];
      const float *xbh = xbuf[pbh];
      // all LDS operands of the step first, then the MFMAs back to back: an MFMA behind its own ds_read waits out the LDS latency,
      // and a wave's VALU / LDS work never overlaps its MFMAs (tools/microbench/mfma_rate.hip), so every exposed read is lost time
      float ah[4][NW], axh[4][HX > 0 ? HX : 1], bzv[4][NW][4];
#pragma unroll
      for (int sk = 0; sk < 4; sk++) {
#pragma unroll
        for (int mt = HM; mt < NW; mt++) ah[sk][mt] = hpp[(4 * sk + rq) * LDH + 16 * mt + col];
#pragma unroll
        for (int q = 0; q < HX; q++) axh[sk][q] = xbh[(4 * sk + rq) * LDX + 16 * (MXW + q) + col];
#pragma unroll
        for (int ws = 0; ws < NW; ws++)
#pragma unroll
          for (int nt = 0; nt < 4; nt++) bzv[sk][ws][nt] = dzbuf[HELPER ? pbh : 0][ws][(4 * sk + rq) * LDZ + 16 * nt + col];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int sk = 0; sk < 4; sk++) {
#pragma unroll
        for (int ws = 0; ws < NW; ws++)
#pragma unroll
          for (int nt = 0; nt < 4; nt++) {
            const float bz = bzv[sk][ws][nt];
#pragma unroll
            for (int mt = HM; mt < NW; mt++) accH[mt][4 * ws + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[sk][mt], bz, accH[mt][4 * ws + nt], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < HX; q++) accXh[q][4 * ws + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(axh[sk][q], bz, accXh[q][4 * ws + nt], 0, 0, 0);
          }
      }
      pbh ^= 1;
    }
    const size_t blk = blockIdx.x;
#pragma unroll
    for (int ct = 0; ct < 4 * NW; ct++)
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int mt = HM; mt < NW; mt++) dwh_part[(blk * HID + 16 * mt + 4 * rq + r) * (4 * HID) + 16 * ct + col] = accH[mt][ct][r];
#pragma unroll
    for (int q = 0; q < HX; q++)
#pragma unroll
      for (int ct = 0; ct < 4 * NW; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = 16 * (MXW + q) + 4 * rq + r;
          if (i < n_in) dwx_part[(blk * n_in + i) * (4 * HID) + 16 * ct + col] = accXh[q][ct][r];
        }
#ifdef IRRL_PROFILE_BWD
    if (blockIdx.x == gridDim.x / 2 && l == 0) { float *o_ = db_part + (size_t)gridDim.x * 16 * HID; o_[8] = (float)hw_wait; o_[9] = (float)hw_work; }   // one row behind the partials: the profiling caller allocates it
#endif
    return;
  }
  // B fragments (K = this wave's 64 permuted gate columns): dh_prev tiles over the hidden index, dx tiles over the input
  float bT[16][NW];
#pragma unroll
  for (int kk = 0; kk < 16; kk++)
#pragma unroll
    for (int nt = 0; nt < NW; nt++) bT[kk][nt] = wh_p[(size_t)(16 * nt + col) * HID * 4 + 64 * w + 4 * kk + rq];
  float bX[NEED_DX ? 16 : 1][MX];
  if (NEED_DX) {
#pragma unroll
    for (int kk = 0; kk < 16; kk++)
#pragma unroll
      for (int nx = 0; nx < MX; nx++) {
        const int i = 16 * nx + col;
        bX[kk][nx] = (i < n_in) ? wx_p[(size_t)i * HID * 4 + 64 * w + 4 * kk + rq] : 0.0f;
      }
  }
  f32x4 accWh[NW][4], accWx[MXW > 0 ? MXW : 1][4];
#pragma unroll
  for (int nt = 0; nt < 4; nt++) {
#pragma unroll
    for (int mt = 0; mt < NW; mt++) accWh[mt][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int mx = 0; mx < MXW; mx++) accWx[mx][nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  }
  float dbacc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float dc[4] = {0.0f, 0.0f, 0.0f, 0.0f}, dhrec[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  int pb = 0;
  f32x4 g_n[4];
  float ct_n[4], cp_n[4], dh_n[4], mk_n[4], hp_n[4], xs_n[XPW][4];
  auto fetch = [&](int t) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int e = e0 + 4 * rq + j;
      const size_t row = (size_t)t * N + e;
      mk_n[j] = masks[row];
      g_n[j] = *(const f32x4 *)&gates[(row * HID + u) * 4];
      ct_n[j] = cseq[row * HID + u];
      cp_n[j] = (t > 0) ? cseq[(row - N) * HID + u] : state0[(size_t)e * 2 * HID + u];
      hp_n[j] = (t > 0) ? hseq[(row - N) * HID + u] : state0[(size_t)e * 2 * HID + HID + u];
      dh_n[j] = dh_in[row * HID + u];
#pragma unroll
      for (int q = 0; q < XPW; q++) {
        const int i = 16 * (w + q * NW) + col;
        xs_n[q][j] = (w + q * NW < MX && i < n_in) ? x[row * n_in + i] : 0.0f;
      }
    }
  };
  fetch(T - 1);
#ifdef IRRL_PROFILE_BWD
  unsigned long long ph_[6] = {0, 0, 0, 0, 0, 0}, pts_ = wall_clock64();
#define IRRL_BW_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = wall_clock64(); ph_[i] += n_ - pts_; pts_ = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define IRRL_BW_STAMP(i) do { } while (0)
#endif
  for (int t = T - 1; t >= 0; t--) {
    float keepC[4];
    f32x4 dz4[4], g4[4];
    float ct[4], cpv[4], dhv[4];
    float *hp = hpbuf[pb], *xb = xbuf[pb];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      keepC[j] = 1.0f - mk_n[j]; g4[j] = g_n[j]; ct[j] = ct_n[j]; cpv[j] = cp_n[j]; dhv[j] = dh_n[j];
      hp[(4 * rq + j) * LDH + u] = hp_n[j] * keepC[j];          // h_{t-1} as it entered step t
#pragma unroll
      for (int q = 0; q < XPW; q++)
        if (w + q * NW < MX) xb[(4 * rq + j) * LDX + 16 * (w + q * NW) + col] = xs_n[q][j];
    }
    if (t > 0) fetch(t - 1);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float cprev = cpv[j] * keepC[j];
      const float dh = dhv[j] + dhrec[j];
      const float ig = g4[j][0], fg = g4[j][1], og = g4[j][2], gg = g4[j][3];
      const float tc = fast_tanh(ct[j]);
      const float d_o = dh * tc;
      const float dct = dc[j] + dh * og * (1.0f - tc * tc);
      const float d_i = dct * gg, d_g = dct * ig, d_f = dct * cprev;
      dc[j] = dct * fg * keepC[j];
      dz4[j] = (f32x4){d_i * ig * (1.0f - ig), d_f * fg * (1.0f - fg), d_o * og * (1.0f - og), d_g * (1.0f - gg * gg)};
      *(f32x4 *)&dzbuf[HELPER ? pb : 0][w][(4 * rq + j) * LDZ + 4 * col] = dz4[j];
#pragma unroll
      for (int g = 0; g < 4; g++) dbacc[g] += dz4[j][g];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    IRRL_BW_STAMP(0);   // staging of h_prev / x, prefetch issue, gate arithmetic, dz tile
    f32x4 acc[NW], accx[NEED_DX ? MX : 1];
#pragma unroll
    for (int nt = 0; nt < NW; nt++) acc[nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    if (NEED_DX) {
#pragma unroll
      for (int nx = 0; nx < MX; nx++) accx[nx] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    float av[16];
#pragma unroll
    for (int kk = 0; kk < 16; kk++) av[kk] = dzbuf[HELPER ? pb : 0][w][col * LDZ + 4 * kk + rq];  // A[i = env col][k = 4kk + rq]
    __builtin_amdgcn_sched_barrier(0);   // the 16 reads stay in front of the MFMAs (see the helper)
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
      const float a = av[kk];
#pragma unroll
      for (int nt = 0; nt < NW; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bT[kk][nt], acc[nt], 0, 0, 0);
      if (NEED_DX) {
#pragma unroll
        for (int nx = 0; nx < MX; nx++) accx[nx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bX[kk][nx], accx[nx], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
      for (int nt = 0; nt < NW; nt++) part[pb][w][(4 * rq + j) * LDP + 16 * nt + col] = acc[nt][j];
      if (NEED_DX) {
#pragma unroll
        for (int nx = 0; nx < MX; nx++) part[pb][w][(4 * rq + j) * LDP + HID + 16 * nx + col] = accx[nx][j];
      }
    }
    IRRL_BW_STAMP(1);   // recurrence (+ dx) MFMAs, partials to LDS
    __syncthreads();   // partials of all waves, and this step's h_prev / x tiles, are now visible
    IRRL_BW_STAMP(2);   // barrier
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float sacc = 0.0f;
#pragma unroll
      for (int ww = 0; ww < NW; ww++) sacc += part[pb][ww][(4 * rq + j) * LDP + u];
      dhrec[j] = sacc * keepC[j];
    }
    if (NEED_DX) {
#pragma unroll
      for (int q = 0; q < XPW; q++) {
        const int nx = w + q * NW;
        const int i = 16 * nx + col;
        if (nx < MX && i < n_in) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            float sacc = 0.0f;
#pragma unroll
            for (int ww = 0; ww < NW; ww++) sacc += part[pb][ww][(4 * rq + j) * LDP + HID + i];
            dx[((size_t)t * N + e0 + 4 * rq + j) * n_in + i] = sacc;
          }
        }
      }
    }
    IRRL_BW_STAMP(3);   // partial sums -> dh_prev, dx rows
    // weight-gradient accumulation: D[m][c] += sum_env A[m][env] B[env][c], env = 4s + rq
    float bz[4][4], ah[4][NW], ax[4][MX];
#pragma unroll
    for (int sk = 0; sk < 4; sk++) {
#pragma unroll
      for (int nt = 0; nt < 4; nt++) bz[sk][nt] = dzbuf[HELPER ? pb : 0][w][(4 * sk + rq) * LDZ + 16 * nt + col];
#pragma unroll
      for (int mt = 0; mt < HM; mt++) ah[sk][mt] = hp[(4 * sk + rq) * LDH + 16 * mt + col];
#pragma unroll
      for (int mx = 0; mx < MXW; mx++) ax[sk][mx] = xb[(4 * sk + rq) * LDX + 16 * mx + col];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sk = 0; sk < 4; sk++) {
#pragma unroll
      for (int nt = 0; nt < 4; nt++) {
#pragma unroll
        for (int mt = 0; mt < HM; mt++) accWh[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[sk][mt], bz[sk][nt], accWh[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mx = 0; mx < MXW; mx++) accWx[mx][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[sk][mx], bz[sk][nt], accWx[mx][nt], 0, 0, 0);
      }
    }
    IRRL_BW_STAMP(4);   // weight-gradient MFMAs
    pb ^= 1;
  }
#ifdef IRRL_PROFILE_BWD
  if (blockIdx.x == gridDim.x / 2 && w == 0 && l == 0)
    for (int i = 0; i < 5; i++) (db_part + (size_t)gridDim.x * 16 * HID)[i] = (float)ph_[i];
#endif
  // per-workgroup partial gradients; C/D slot (row 4 rq + r, column col) of tile (m, nt) = (input / hidden index, gate column)
  const size_t blk = blockIdx.x;
#pragma unroll
  for (int nt = 0; nt < 4; nt++) {
    const int c = 64 * w + 16 * nt + col;
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
      for (int mt = 0; mt < HM; mt++) dwh_part[(blk * HID + 16 * mt + 4 * rq + r) * (4 * HID) + c] = accWh[mt][nt][r];
#pragma unroll
      for (int mx = 0; mx < MXW; mx++) {
        const int i = 16 * mx + 4 * rq + r;
        if (i < n_in) dwx_part[(blk * n_in + i) * (4 * HID) + c] = accWx[mx][nt][r];
      }
    }
  }
#pragma unroll
  for (int g = 0; g < 4; g++) db_part[(blk * 4 + rq) * (4 * HID) + u * 4 + g] = dbacc[g];
}

// ---------------------------------------------------------------------------------------------------------------
// One ROLLOUT step of the whole CustomLSTMPolicy in a single launch: the device code lives in policy_step.hpp (shared with the
// fused env + policy kernel of env_kernels.hip); this is the stand-alone kernel, a workgroup of 2 HID/16 waves per 16 envs.
template <int HID, int OBK>
__global__ void __launch_bounds__(2 * (HID / 16) * 64)
lstm_policy_step_kernel(PolicyStepArgs a) {
  __shared__ float hbuf[2][16 * (HID + 1)];   // per stack: the h of the layer just computed, [env][unit]
  __shared__ float terms[16][17];
  __shared__ float head_w[HID * 17];          // pi_w [HID][act] then vf_w [HID]: staged once, read by the head threads
  policy_step_body<HID, OBK, 1, 2 * (HID / 16) * 64>(a, blockIdx.x * 16, hbuf, terms, head_w);
}

// ---------------------------------------------------------------------------------------------------------------
// The same single-launch rollout step for MlpPolicy (policies.py:430-446: separate pi / vf nets of two tanh layers of H
// units).  A workgroup owns 16 envs, each of its four waves four of them (mlp_policy_wave_body, policy_step.hpp: both nets on 4 x 4 x 1 MFMAs).
// a.w[] = pi_w1 [ob][H], pi_b1, pi_w2 [H][H], pi_b2, vf_w1, vf_b1, vf_w2, vf_b2 (plain row-major, no permutation).
template <int H>
__global__ void __launch_bounds__(256)
mlp_policy_step_kernel(PolicyStepArgs a) {
  // four independent waves, four envs each (mlp_policy_wave_body: no workgroup barrier)
  __shared__ __attribute__((aligned(16))) float ws[4][MlpWaveLds<H>::FLOATS];
  const int wv = (int)(threadIdx.x >> 6);
  const int e4 = (int)blockIdx.x * 16 + 4 * wv;
  if (e4 >= a.N) return;
  mlp_policy_wave_body<H, false>(a, e4, ws[wv], nullptr, nullptr, (int)(threadIdx.x & 63u));
}

// ---- PPO2 clipped-surrogate loss, forward AND backward in one pass (ppo2.py:152-175 + DiagGaussian neglogp / entropy) ----
// One lane per sample.  Because every term of the loss is a mean over samples, the gradient of a sample's row does not
// depend on the other rows: the kernel writes d loss / d mean [M, A] and d loss / d vpred [M] directly and leaves per-workgroup
// partial sums of the scalars (policy loss, value loss, approx KL, clip fraction) and of d loss / d logstd [A]
// in `partials` [blocks, 4 + A]; the caller adds them up (deterministic order).  adv_stats = (mean, std) of the raw
// advantages (device scalars: with several ranks they are all-reduced first); the kernel normalises on the fly.
template <int A>
__global__ void __launch_bounds__(256)
irrl_ppo_loss_kernel(size_t M, const float *__restrict__ mean, const float *__restrict__ logstd, const float *__restrict__ vpred,
                     const float *__restrict__ actions, const float *__restrict__ returns, const float *__restrict__ old_values,
                     const float *__restrict__ old_neglogp, const float *__restrict__ adv_stats, float cliprange, float vf_coef,
                     float inv_m, float *__restrict__ d_mean, float *__restrict__ d_vpred, float *__restrict__ partials) {
  __shared__ float red[4][4 + A];
  float sd_inv[A], ls_sum = 0.0f;
#pragma unroll
  for (int a = 0; a < A; a++) { const float ls = logstd[a]; sd_inv[a] = __expf(-ls); ls_sum += ls; }
  const float a_mean = adv_stats[0], a_istd = 1.0f / (adv_stats[1] + 1e-8f);
  float acc[4 + A];
#pragma unroll
  for (int i = 0; i < 4 + A; i++) acc[i] = 0.0f;
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < M; r += (size_t)gridDim.x * blockDim.x) {
    float diff[A], q = 0.0f;
#pragma unroll
    for (int a = 0; a < A; a++) { diff[a] = (actions[r * A + a] - mean[r * A + a]) * sd_inv[a]; q += diff[a] * diff[a]; }
    const float nlp = 0.5f * q + 0.918938533204672742f * (float)A + ls_sum;
    const float onlp = old_neglogp[r];
    const float adv = (returns[r] - old_values[r] - a_mean) * a_istd;
    const float ratio = __expf(onlp - nlp);
    const float rc = fminf(fmaxf(ratio, 1.0f - cliprange), 1.0f + cliprange);
    const float pg1 = -adv * ratio, pg2 = -adv * rc;
    const bool inside = (ratio >= 1.0f - cliprange) && (ratio <= 1.0f + cliprange);
    // torch.maximum: a tie splits the gradient evenly; inside the clip range both branches carry d/d ratio = -adv
    const float dpg_dratio = inside ? -adv : ((pg1 > pg2) ? -adv : ((pg1 == pg2) ? -0.5f * adv : 0.0f));
    const float dl_dnlp = inv_m * dpg_dratio * (-ratio);
    const float v = vpred[r], ov = old_values[r], R = returns[r];
    const float dv = v - ov;
    const float vc = ov + fminf(fmaxf(dv, -cliprange), cliprange);
    const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
    const float g_clamp = (dv >= -cliprange && dv <= cliprange) ? 1.0f : 0.0f;
    const float dvf = (l1 > l2) ? (v - R) : ((l1 < l2) ? (vc - R) * g_clamp : 0.5f * (v - R) + 0.5f * (vc - R) * g_clamp);
    d_vpred[r] = inv_m * vf_coef * dvf;
#pragma unroll
    for (int a = 0; a < A; a++) {
      d_mean[r * A + a] = dl_dnlp * (-diff[a] * sd_inv[a]);
      acc[4 + a] += dl_dnlp * (1.0f - diff[a] * diff[a]);
    }
    acc[0] += fmaxf(pg1, pg2);
    acc[1] += 0.5f * fmaxf(l1, l2);
    acc[2] += 0.5f * (nlp - onlp) * (nlp - onlp);
    acc[3] += (fabsf(ratio - 1.0f) > cliprange) ? 1.0f : 0.0f;
  }
  // wave reduction, then across the four waves
#pragma unroll
  for (int i = 0; i < 4 + A; i++) {
    float x = acc[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    acc[i] = x;
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 4 + A; i++) red[w][i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < 4 + A) partials[(size_t)blockIdx.x * (4 + A) + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}


// ---- policy / value HEADS + the PPO2 loss, forward AND backward, in one pass over the rollout -------------------------------
// The heads are skinny linears over millions of rows (mean = h_pi W_pi + b_pi [M,48]x[48,12], v = h_v w_v + b_v): as library
// GEMMs their forward, the two dx products and the two tall dW reductions cost ~3.7 ms per epoch at 4096 x 750 (the dx GEMM
// alone 2.45 ms: a K = 12 contraction writing 590 MB), although every byte is touched once.  Here one lane owns one sample:
// it reads its two latent rows (float4 loads), forms mean / value with the head weights broadcast from LDS, evaluates the loss
// terms and their row gradients exactly like irrl_ppo_loss_kernel, and writes d loss / d h_pi, d loss / d h_v [M,48] straight
// away.  The weight gradient dW_pi = h_pi^T d_mean is accumulated on the matrix cores (v_mfma_f32_16x16x4_f32, K = rows: the
// wave's 64 d_mean rows go through a 4 KB LDS tile into the B operand, the A operand is re-read from the L1-resident latent
// rows); dw_v, the biases and the scalars are per-lane sums reduced once per workgroup.  Per-workgroup partials [blocks, P],
// P = 4 + A + A + 1 + H + H*A: (pg, vf, kl, clipfrac) | d logstd [A] | d b_pi [A] | d b_v | d w_v [H] | d W_pi [H][A].
template <int A, int H>
__global__ void __launch_bounds__(256, 2)
irrl_ppo_heads_loss_kernel(size_t M, const float *__restrict__ h_pi, const float *__restrict__ h_v, const float *__restrict__ pi_w,
                           const float *__restrict__ pi_b, const float *__restrict__ vf_w, const float *__restrict__ vf_b,
                           const float *__restrict__ logstd, const float *__restrict__ actions, const float *__restrict__ returns,
                           const float *__restrict__ old_values, const float *__restrict__ old_neglogp, const float *__restrict__ adv_stats,
                           float cliprange, float vf_coef, float inv_m, float *__restrict__ d_hpi, float *__restrict__ d_hv,
                           float *__restrict__ mean_out, float *__restrict__ value_out, float *__restrict__ partials) {
  static_assert(A == 12 && H == 48, "heads kernel is instantiated for the 48-unit latents and 12 actions of CustomLSTMPolicy");
  constexpr int P = 4 + A + A + 1 + H + H * A;
  constexpr int NT = H / 16;
  __shared__ f32x4 Wl[H][A / 4];      // pi head weights, [unit][action]
  __shared__ float wv[H];
  __shared__ float dm[4][64][17];     // per wave: rows (d_mean[0..11], d_v, 0, 0, 0) of the current 64-sample tile (+1 pad)
  // per wave: the current 64-row tile of ONE latent matrix (h_v, then h_pi), filled by coalesced 16-byte loads (a tile is 12 KiB of
  // contiguous memory), read back row-per-lane for the head and as the MFMA A operand, overwritten with the row gradients and
  // written out in whole lines.  (Row-per-lane global accesses touch 64 cache lines per instruction and use 16 bytes of each; with
  // 8 waves per CU the 24 KiB a wave works on did not survive in L1 between its visits, and the MFMA operands were read a third time.)
  constexpr int TLD = H + 4;          // 52: 16-byte aligned rows; 13 l + q is a bijection mod 16 for the row-per-lane 16-byte accesses
  __shared__ __attribute__((aligned(16))) float tile[4][64 * TLD];
  static_assert(P <= 64 * TLD, "the reduction scratch reuses the tiles");
  float (*red)[64 * TLD] = tile;      // [wave][P] partial sums, written after the last tile (behind a barrier)
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, col = l & 15, rq = l >> 4;
  for (int i = tid; i < H * A; i += 256) ((float *)Wl)[i] = pi_w[i];
  if (tid < H) wv[tid] = vf_w[tid];
  float sd_inv[A], ls_sum = 0.0f;
#pragma unroll
  for (int a = 0; a < A; a++) { const float ls = logstd[a]; sd_inv[a] = __expf(-ls); ls_sum += ls; }
  const float bv = vf_b[0];
  const float a_mean = adv_stats[0], a_istd = 1.0f / (adv_stats[1] + 1e-8f);
  float sc[4] = {0.0f, 0.0f, 0.0f, 0.0f}, dls[A], dbp[A], dbv = 0.0f;
#pragma unroll
  for (int a = 0; a < A; a++) { dls[a] = 0.0f; dbp[a] = 0.0f; }
  f32x4 dW[NT], dWv[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) { dW[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; dWv[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll
  for (int a = A + 1; a < 17; a++) dm[w][l][a] = 0.0f;   // padding columns stay zero
  __syncthreads();
  const size_t n_tiles = (M + 63) / 64;
  float *tl = tile[w];
  for (size_t tile_i = (size_t)blockIdx.x * 4 + w; tile_i < n_tiles; tile_i += (size_t)gridDim.x * 4) {
    const size_t r0 = tile_i * 64;
    const size_t r = r0 + l;
    const bool ok = r < M;
    const size_t rc = ok ? r : M - 1;
    const float live = ok ? 1.0f : 0.0f;
    // coalesced fill of the wave's tile from matrix `src` (rows r0 .. r0 + 63, clamped at the end of the batch), and the reverse
    auto fill = [&](const float *__restrict__ src) {
#pragma unroll 4
      for (int i = 0; i < H * 64 / 256; i++) {
        const int c = l + 64 * i;                     // 16-byte piece of the tile: row c / (H / 4), columns 4 (c % (H / 4)) ..
        const int row = c / (H / 4), cq = c - row * (H / 4);
        const size_t rr = r0 + row < M ? r0 + row : M - 1;
        *(f32x4 *)&tl[row * TLD + 4 * cq] = *(const f32x4 *)&src[rr * H + 4 * cq];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto drain = [&](float *__restrict__ dst) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 4
      for (int i = 0; i < H * 64 / 256; i++) {
        const int c = l + 64 * i;
        const int row = c / (H / 4), cq = c - row * (H / 4);
        if (r0 + row < M) *(f32x4 *)&dst[(r0 + row) * H + 4 * cq] = *(const f32x4 *)&tl[row * TLD + 4 * cq];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };
    // one of the two weight-gradient products for the wave's 64 rows on the matrix cores: A[i = unit][k = row] from the staged tile,
    // B[k = row][j] = the dm tile (columns 0 .. 11 = d_mean, column 12 = d_v)
    auto grad_mfma = [&](f32x4 (&acc)[NT]) {
#pragma unroll 2
      for (int s4 = 0; s4 < 16; s4++) {
        const float bb = dm[w][4 * s4 + rq][col];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tl[(4 * s4 + rq) * TLD + 16 * t + col], bb, acc[t], 0, 0, 0);
      }
    };
    const float R = returns[rc], ov = old_values[rc], onlp = old_neglogp[rc];
    // ================= value stack: v, d_v, d loss / d h_v = d_v w_v, d w_v += h_v^T d_v =================
    fill(h_v);
    float v = bv;
#pragma unroll
    for (int q = 0; q < H / 4; q++) {
      const f32x4 y = *(const f32x4 *)&tl[l * TLD + 4 * q];
#pragma unroll
      for (int j = 0; j < 4; j++) v = __builtin_fmaf(y[j], wv[4 * q + j], v);
    }
    const float dv = v - ov;
    const float vc = ov + fminf(fmaxf(dv, -cliprange), cliprange);
    const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
    const float g_clamp = (dv >= -cliprange && dv <= cliprange) ? 1.0f : 0.0f;
    const float dvf = (l1 > l2) ? (v - R) : ((l1 < l2) ? (vc - R) * g_clamp : 0.5f * (v - R) + 0.5f * (vc - R) * g_clamp);
    const float d_v = live * inv_m * vf_coef * dvf;
    dm[w][l][A] = d_v;
    sc[1] += live * 0.5f * fmaxf(l1, l2);
    dbv += d_v;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    grad_mfma(dWv);                                   // (its columns other than 12 multiply the previous tile's d_mean: never read)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < H / 4; q++)
      *(f32x4 *)&tl[l * TLD + 4 * q] = (f32x4){d_v * wv[4 * q], d_v * wv[4 * q + 1], d_v * wv[4 * q + 2], d_v * wv[4 * q + 3]};
    drain(d_hv);
    // ================= policy stack: mean, loss terms, d_mean, d loss / d h_pi = d_mean W_pi^T, dW_pi += h_pi^T d_mean =================
    fill(h_pi);
    float mean[A];
#pragma unroll
    for (int a = 0; a < A; a++) mean[a] = pi_b[a];
#pragma unroll 3
    for (int q = 0; q < H / 4; q++) {
      const f32x4 x = *(const f32x4 *)&tl[l * TLD + 4 * q];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int k = 4 * q + j;
#pragma unroll
        for (int a4 = 0; a4 < A / 4; a4++) {
          const f32x4 wk = Wl[k][a4];
          mean[4 * a4] = __builtin_fmaf(x[j], wk[0], mean[4 * a4]); mean[4 * a4 + 1] = __builtin_fmaf(x[j], wk[1], mean[4 * a4 + 1]);
          mean[4 * a4 + 2] = __builtin_fmaf(x[j], wk[2], mean[4 * a4 + 2]); mean[4 * a4 + 3] = __builtin_fmaf(x[j], wk[3], mean[4 * a4 + 3]);
        }
      }
    }
    // ---- loss terms and row gradients (same arithmetic as irrl_ppo_loss_kernel) ----
    float diff[A], q2 = 0.0f;
    {
      const f32x4 *pa = (const f32x4 *)(actions + rc * A);
#pragma unroll
      for (int q = 0; q < A / 4; q++) {
        const f32x4 x = pa[q];
#pragma unroll
        for (int j = 0; j < 4; j++) { const int a = 4 * q + j; diff[a] = (x[j] - mean[a]) * sd_inv[a]; q2 += diff[a] * diff[a]; }
      }
    }
    const float nlp = 0.5f * q2 + 0.918938533204672742f * (float)A + ls_sum;
    const float adv = (R - ov - a_mean) * a_istd;
    const float ratio = __expf(onlp - nlp);
    const float rcl = fminf(fmaxf(ratio, 1.0f - cliprange), 1.0f + cliprange);
    const float pg1 = -adv * ratio, pg2 = -adv * rcl;
    const bool inside = (ratio >= 1.0f - cliprange) && (ratio <= 1.0f + cliprange);
    const float dpg_dratio = inside ? -adv : ((pg1 > pg2) ? -adv : ((pg1 == pg2) ? -0.5f * adv : 0.0f));
    const float dl_dnlp = live * inv_m * dpg_dratio * (-ratio);
    float dmean[A];
#pragma unroll
    for (int a = 0; a < A; a++) {
      dmean[a] = dl_dnlp * (-diff[a] * sd_inv[a]);
      dls[a] += dl_dnlp * (1.0f - diff[a] * diff[a]);
      dbp[a] += dmean[a];
      dm[w][l][a] = dmean[a];
    }
    sc[0] += live * fmaxf(pg1, pg2);
    sc[2] += live * 0.5f * (nlp - onlp) * (nlp - onlp);
    sc[3] += (ok && fabsf(ratio - 1.0f) > cliprange) ? 1.0f : 0.0f;
    if (ok && mean_out) {
      f32x4 *pm = (f32x4 *)(mean_out + r * A);
#pragma unroll
      for (int q = 0; q < A / 4; q++) pm[q] = (f32x4){mean[4 * q], mean[4 * q + 1], mean[4 * q + 2], mean[4 * q + 3]};
      value_out[r] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    grad_mfma(dW);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // d loss / d h_pi = d_mean W_pi^T: this lane's row into the tile, then out in whole lines
#pragma unroll 3
    for (int q = 0; q < H / 4; q++) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int k = 4 * q + j;
        float acc = 0.0f;
#pragma unroll
        for (int a4 = 0; a4 < A / 4; a4++) {
          const f32x4 wk = Wl[k][a4];
          acc = __builtin_fmaf(dmean[4 * a4], wk[0], acc); acc = __builtin_fmaf(dmean[4 * a4 + 1], wk[1], acc);
          acc = __builtin_fmaf(dmean[4 * a4 + 2], wk[2], acc); acc = __builtin_fmaf(dmean[4 * a4 + 3], wk[3], acc);
        }
        o[j] = acc;
      }
      *(f32x4 *)&tl[l * TLD + 4 * q] = o;
    }
    drain(d_hpi);
  }
  __syncthreads();   // every wave is done with its tile: the tiles now hold the reduction scratch
  // ---- workgroup reduction: per-lane sums over the wave, then the four waves through LDS ----
  auto wave_sum = [](float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
  };
#pragma unroll
  for (int i = 0; i < 4; i++) { const float x = wave_sum(sc[i]); if (l == 0) red[w][i] = x; }
#pragma unroll
  for (int a = 0; a < A; a++) {
    const float x = wave_sum(dls[a]), y = wave_sum(dbp[a]);
    if (l == 0) { red[w][4 + a] = x; red[w][4 + A + a] = y; }
  }
  { const float x = wave_sum(dbv); if (l == 0) red[w][4 + 2 * A] = x; }
  // MFMA tiles: lane holds D[i = 4 rq + j][col] of tile t -> unit 16 t + 4 rq + j; column = action (dW_pi), column 12 = d w_v
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int unit = 16 * t + 4 * rq + j;
      if (col < A) red[w][4 + 2 * A + 1 + H + unit * A + col] = dW[t][j];
      if (col == A) red[w][4 + 2 * A + 1 + unit] = dWv[t][j];
    }
  __syncthreads();
  for (int i = tid; i < P; i += 256) partials[(size_t)blockIdx.x * P + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

// ---- sum of per-workgroup partial rows: out[c] = sum_r part[r][c'] (fixed order: deterministic) -------------------------------
// The gradient kernels leave one row of partial sums per workgroup ([256, 9216] for a weight matrix of the LSTM); the library's
// reduction over the OUTER dimension of such a matrix takes 250 us (37 GB/s).  Here a workgroup owns 64 columns: its four waves
// each add a quarter of the rows (coalesced 256-byte reads, eight loads in flight per lane), then the four add up through LDS.
// unit_gate_hid > 0 additionally undoes the LSTM kernels' [unit][gate] column permutation inside each group of 4 * hid
// columns: out column g * hid + u  <-  partial column 4 u + g.
__global__ void __launch_bounds__(256)
irrl_sum_rows_kernel(const float *__restrict__ part, int rows, int cols, int unit_gate_hid, float *__restrict__ out) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  float acc = 0.0f;
  if (col < cols) {
    int src = col;
    if (unit_gate_hid > 0) {
      const int w4 = 4 * unit_gate_hid, base = col / w4 * w4, j = col - base;
      src = base + 4 * (j % unit_gate_hid) + j / unit_gate_hid;
    }
    const int per = (rows + 3) / 4, r0 = rg * per, r1 = min(rows, r0 + per);
    const float *p = part + (size_t)r0 * cols + src;
    int r = r0;
    for (; r + 8 <= r1; r += 8) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = p[(size_t)i * cols];
#pragma unroll
      for (int i = 0; i < 8; i++) acc += v[i];
      p += (size_t)8 * cols;
    }
    for (; r < r1; r++) { acc += *p; p += cols; }
  }
  red[rg][cx] = acc;
  __syncthreads();
  if (rg == 0 && col < cols) out[col] = ((red[0][cx] + red[1][cx]) + red[2][cx]) + red[3][cx];
}

extern "C" {

int irrl_sum_rows(const float *part, int rows, int cols, int unit_gate_hid, float *out, void *hip_stream) {
  if (rows <= 0 || cols <= 0 || (unit_gate_hid > 0 && cols % (4 * unit_gate_hid) != 0)) return 1;
  hipLaunchKernelGGL(irrl_sum_rows_kernel, dim3((cols + 63) / 64), dim3(256), 0, (hipStream_t)hip_stream, part, rows, cols, unit_gate_hid, out);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// returns 0 on success; 1 = unsupported shape, 2 = launch error
int irrl_lstm_seq_forward(int hid, int T, int N, const float *zx, const float *wh_p, const float *masks, const float *state0,
                          float *gates, float *cseq, float *hseq, float *state_out, void *hip_stream) {
  if (N <= 0 || T <= 0 || (N % 16) != 0) return 1;
  hipStream_t s = (hipStream_t)hip_stream;
  if (hid == 48) hipLaunchKernelGGL(lstm_seq_fwd_kernel<48>, dim3(N / 16), dim3(192), 0, s, zx, wh_p, masks, state0, gates, cseq, hseq, state_out, T, N);
  else if (hid == 32) hipLaunchKernelGGL(lstm_seq_fwd_kernel<32>, dim3(N / 16), dim3(128), 0, s, zx, wh_p, masks, state0, gates, cseq, hseq, state_out, T, N);
  else if (hid == 64) hipLaunchKernelGGL(lstm_seq_fwd_kernel<64>, dim3(N / 16), dim3(256), 0, s, zx, wh_p, masks, state0, gates, cseq, hseq, state_out, T, N);
  else return 1;
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// forward with the input projection fused: x [T,N,n_in], wx_p [n_in][hid][4], b_p [hid][4]; n_in <= 48
int irrl_lstm_seq_forward_x(int hid, int T, int N, int n_in, const float *x, const float *wx_p, const float *b_p, const float *wh_p,
                            const float *masks, const float *state0, float *gates, float *cseq, float *hseq, float *state_out,
                            void *hip_stream) {
  if (N <= 0 || T <= 0 || (N % 16) != 0 || n_in <= 0 || n_in > 48) return 1;
  hipStream_t s = (hipStream_t)hip_stream;
  const int kxs = (n_in + 3) / 4;
  // HID = 32 / 48: one extra wave computes the input projection one step ahead (IRRL_LSTM_FWD_HELPER=0: plain kernel, for A/B runs)
  static const bool fhelper = [] { const char *e = getenv("IRRL_LSTM_FWD_HELPER"); return !(e && e[0] == '0'); }();
  // the input projection shared between the helper and the recurrence waves (IRRL_LSTM_FWD_SPLIT=0: helper alone, for A/B runs);
  // needs every vectorised 16-element group inside the row
  static const bool fsplit = [] { const char *e = getenv("IRRL_LSTM_FWD_SPLIT"); return !(e && e[0] == '0'); }();
  const bool split = fsplit && n_in >= 16 * (kxs <= 9 ? 2 : 3);
#define IRRL_FX(H, K) \
  do { \
    if (fhelper && H <= 48 && split) hipLaunchKernelGGL((lstm_seq_fwd_x_kernel<H, K, true, true>), dim3(N / 16), dim3((H / 16 + 1) * 64), 0, s, x, wx_p, b_p, wh_p, masks, state0, gates, cseq, hseq, state_out, T, N, n_in); \
    else if (fhelper && H <= 48) hipLaunchKernelGGL((lstm_seq_fwd_x_kernel<H, K, true>), dim3(N / 16), dim3((H / 16 + 1) * 64), 0, s, x, wx_p, b_p, wh_p, masks, state0, gates, cseq, hseq, state_out, T, N, n_in); \
    else hipLaunchKernelGGL((lstm_seq_fwd_x_kernel<H, K, false>), dim3(N / 16), dim3(H / 16 * 64), 0, s, x, wx_p, b_p, wh_p, masks, state0, gates, cseq, hseq, state_out, T, N, n_in); \
  } while (0)
  if (hid == 48 && kxs <= 9) IRRL_FX(48, 9);
  else if (hid == 48) IRRL_FX(48, 12);
  else if (hid == 32 && kxs <= 9) IRRL_FX(32, 9);
  else if (hid == 32) IRRL_FX(32, 12);
  else if (hid == 64 && kxs <= 9) IRRL_FX(64, 9);
  else if (hid == 64) IRRL_FX(64, 12);
  else return 1;
#undef IRRL_FX
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

int irrl_lstm_seq_backward(int hid, int T, int N, const float *gates, const float *cseq, const float *masks, const float *state0,
                           const float *dh_in, const float *wh_p, float *dz, void *hip_stream) {
  if (N <= 0 || T <= 0 || (N % 16) != 0) return 1;
  hipStream_t s = (hipStream_t)hip_stream;
  if (hid == 48) hipLaunchKernelGGL(lstm_seq_bwd_kernel<48>, dim3(N / 16), dim3(192), 0, s, gates, cseq, masks, state0, dh_in, wh_p, dz, T, N);
  else if (hid == 32) hipLaunchKernelGGL(lstm_seq_bwd_kernel<32>, dim3(N / 16), dim3(128), 0, s, gates, cseq, masks, state0, dh_in, wh_p, dz, T, N);
  else if (hid == 64) hipLaunchKernelGGL(lstm_seq_bwd_kernel<64>, dim3(N / 16), dim3(256), 0, s, gates, cseq, masks, state0, dh_in, wh_p, dz, T, N);
  else return 1;
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// One rollout step of the two-stack LSTM policy (see PolicyStepArgs).  lstm_w: HOST array of 12 device pointers.
// returns 0 on success; 1 = unsupported shape, 2 = launch error
int irrl_lstm_policy_step(int hid, int ob_dim, int act_dim, int N, const float *obs, const uint8_t *dones, const float *states_in,
                          float *states_out, const float *const *lstm_w, const float *pi_w, const float *pi_b, const float *vf_w,
                          const float *vf_b, const float *logstd, const float *noise, int rng_on, unsigned rng_seed, long long rng_step,
                          const long long *rng_base, int env_id_offset, float *action, float *clipped, float *value, float *neglogp, long long row, float *mb_obs,
                          float *mb_actions, float *mb_values, float *mb_neglogp, uint8_t *mb_dones, float *mb_rewards,
                          const float *prev_reward, void *hip_stream) {
  if (N <= 0 || ob_dim <= 0 || act_dim <= 0 || act_dim > 16) return 1;
  const int threads = 2 * (hid / 16) * 64;
  if (16 * act_dim + 16 > threads) return 1;
  PolicyStepArgs a;
  a.obs = obs; a.dones = dones; a.states_in = states_in; a.states_out = states_out;
  for (int i = 0; i < 12; i++) a.w[i] = lstm_w[i];
  a.pi_w = pi_w; a.pi_b = pi_b; a.vf_w = vf_w; a.vf_b = vf_b; a.logstd = logstd; a.noise = noise;
  a.action = action; a.clipped = clipped; a.value = value; a.neglogp = neglogp;
  a.row = row; a.rng_base = rng_base;
  const bool rows = row >= 0;
  a.mb_obs = rows ? mb_obs : nullptr; a.mb_actions = rows ? mb_actions : nullptr; a.mb_values = rows ? mb_values : nullptr;
  a.mb_neglogp = rows ? mb_neglogp : nullptr; a.mb_dones = rows ? mb_dones : nullptr; a.mb_rewards = rows ? mb_rewards : nullptr;
  a.prev_reward = (rows && mb_rewards) ? prev_reward : nullptr;
  if (rows && !(mb_obs && mb_actions && mb_values && mb_neglogp && mb_dones)) return 1;
  a.rng_step = rng_step; a.rng_seed = rng_seed; a.rng_on = rng_on; a.env_id_offset = (unsigned)env_id_offset;
  a.N = N; a.ob_dim = ob_dim; a.act_dim = act_dim;
  hipStream_t s = (hipStream_t)hip_stream;
  const int obk = (ob_dim + 3) / 4;
#define IRRL_PS_LAUNCH(H) \
  do { \
    if (obk == 9) hipLaunchKernelGGL((lstm_policy_step_kernel<H, 9>), dim3((N + 15) / 16), dim3(threads), 0, s, a); \
    else hipLaunchKernelGGL((lstm_policy_step_kernel<H, 0>), dim3((N + 15) / 16), dim3(threads), 0, s, a); \
  } while (0)
  if (hid == 48) IRRL_PS_LAUNCH(48);
  else if (hid == 32) IRRL_PS_LAUNCH(32);
  else if (hid == 64) IRRL_PS_LAUNCH(64);
  else return 1;
#undef IRRL_PS_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// backward with dx / dwx / dwh / db fused in (no dz tensor).  dx [T,N,n_in] or NULL (no input gradient wanted);
// dwx_part [N/16, n_in, 4 hid], dwh_part [N/16, hid, 4 hid], db_part [N/16 * 4, 4 hid] are per-workgroup partials in the
// permuted column order: the caller sums them over the first axis.  n_in <= 48.
int irrl_lstm_seq_backward_x(int hid, int T, int N, int n_in, const float *gates, const float *cseq, const float *hseq, const float *x,
                             const float *masks, const float *state0, const float *dh_in, const float *wh_p, const float *wx_p,
                             float *dx, float *dwx_part, float *dwh_part, float *db_part, void *hip_stream) {
  if (N <= 0 || T <= 0 || (N % 16) != 0 || n_in <= 0 || n_in > 48) return 1;
  hipStream_t s = (hipStream_t)hip_stream;
  // HID = 32 / 48: one extra wave for the dwh accumulation (IRRL_LSTM_BWD_HELPER=0 selects the plain kernel for A/B runs)
  static const bool helper = [] { const char *e = getenv("IRRL_LSTM_BWD_HELPER"); return !(e && e[0] == '0'); }();
  // the helper can also accumulate dwx M-tiles (template HX).  IRRL_LSTM_BWD_SHARE = "<with dx><without dx>" tiles, e.g. 21, 10, 01.
  // default 10 (one tile, in the kernel that also carries dx): PPO update 166.5 -> 163.6 ms; 20: 165.2, 01: 172.3, 11: 170.7, 02: 183.2
  // (profiles/r02_ab_bwd_helper_share.log) -- without dx the helper's SIMD is already the longer one
  static const int share = [] { const char *e = getenv("IRRL_LSTM_BWD_SHARE"); return (e && e[0] && e[1]) ? (e[0] - '0') * 10 + (e[1] - '0') : 10; }();
  const int hx = dx ? share / 10 : share % 10;
#define IRRL_BXH(H, D, X) hipLaunchKernelGGL((lstm_seq_bwd_x_kernel<H, D, true, X>), dim3(N / 16), dim3((H / 16 + 1) * 64), 0, s, gates, cseq, hseq, x, masks, state0, dh_in, wh_p, wx_p, dx, dwx_part, dwh_part, db_part, T, N, n_in)
#define IRRL_BX(H, D) \
  do { \
    if (helper && H <= 48 && hx == 1) IRRL_BXH(H, D, 1); \
    else if (helper && H <= 48 && hx == 2) IRRL_BXH(H, D, 2); \
    else if (helper && H <= 48) hipLaunchKernelGGL((lstm_seq_bwd_x_kernel<H, D, true>), dim3(N / 16), dim3((H / 16 + 1) * 64), 0, s, gates, cseq, hseq, x, masks, state0, dh_in, wh_p, wx_p, dx, dwx_part, dwh_part, db_part, T, N, n_in); \
    else hipLaunchKernelGGL((lstm_seq_bwd_x_kernel<H, D, false>), dim3(N / 16), dim3(H / 16 * 64), 0, s, gates, cseq, hseq, x, masks, state0, dh_in, wh_p, wx_p, dx, dwx_part, dwh_part, db_part, T, N, n_in); \
  } while (0)
  if (hid == 48 && dx) IRRL_BX(48, true);
  else if (hid == 48) IRRL_BX(48, false);
  else if (hid == 32 && dx) IRRL_BX(32, true);
  else if (hid == 32) IRRL_BX(32, false);
  else if (hid == 64 && dx) IRRL_BX(64, true);
  else if (hid == 64) IRRL_BX(64, false);
  else return 1;
#undef IRRL_BX
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// One rollout step of MlpPolicy (two tanh layers of `hid` units per net; hid in {64}).  mlp_w: HOST array of 8 device
// pointers pi_w1 [ob][hid], pi_b1, pi_w2 [hid][hid], pi_b2, vf_w1, vf_b1, vf_w2, vf_b2; everything else as irrl_lstm_policy_step
// (no recurrent state).
int irrl_mlp_policy_step(int hid, int ob_dim, int act_dim, int N, const float *obs, const uint8_t *dones, const float *const *mlp_w,
                         const float *pi_w, const float *pi_b, const float *vf_w, const float *vf_b, const float *logstd, const float *noise,
                         int rng_on, unsigned rng_seed, long long rng_step, const long long *rng_base, int env_id_offset, float *action, float *clipped,
                         float *value, float *neglogp, long long row, float *mb_obs, float *mb_actions, float *mb_values, float *mb_neglogp,
                         uint8_t *mb_dones, float *mb_rewards, const float *prev_reward, void *hip_stream) {
  if (N <= 0 || ob_dim <= 0 || ob_dim > 64 || act_dim <= 0 || act_dim > 15 || hid != 64) return 1;
  PolicyStepArgs a;
  a.obs = obs; a.dones = dones; a.states_in = nullptr; a.states_out = nullptr;
  for (int i = 0; i < 12; i++) a.w[i] = i < 8 ? mlp_w[i] : nullptr;
  a.pi_w = pi_w; a.pi_b = pi_b; a.vf_w = vf_w; a.vf_b = vf_b; a.logstd = logstd; a.noise = noise;
  a.action = action; a.clipped = clipped; a.value = value; a.neglogp = neglogp;
  a.row = row; a.rng_base = rng_base;
  const bool rows = row >= 0;
  a.mb_obs = rows ? mb_obs : nullptr; a.mb_actions = rows ? mb_actions : nullptr; a.mb_values = rows ? mb_values : nullptr;
  a.mb_neglogp = rows ? mb_neglogp : nullptr; a.mb_dones = rows ? mb_dones : nullptr; a.mb_rewards = rows ? mb_rewards : nullptr;
  a.prev_reward = (rows && mb_rewards) ? prev_reward : nullptr;
  if (rows && !(mb_obs && mb_actions && mb_values && mb_neglogp && mb_dones)) return 1;
  a.rng_step = rng_step; a.rng_seed = rng_seed; a.rng_on = rng_on; a.env_id_offset = (unsigned)env_id_offset;
  a.N = N; a.ob_dim = ob_dim; a.act_dim = act_dim;
  hipLaunchKernelGGL((mlp_policy_step_kernel<64>), dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)hip_stream, a);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// ---- the sequence kernels on the bf16 matrix cores with compensated operand splits (csrc/lstm_bf16.hpp) ----
// nsplit 2: three plane products per product (~2^-16 relative), 3: six (~2^-24, the f32 level).  Same tensors as the *_x entry points.
#ifndef IRRL_MAX_DEVICES
#define IRRL_MAX_DEVICES 64
#endif
// "has this device granted the kernels' LDS size": -1 = not asked yet, 0 = yes, otherwise the error.  A function-local static of this type is
// constructed exactly once under the C++11 guarantee; filling a slot later is idempotent (every thread that finds -1 asks the same question and
// stores the same answer).
struct IrrlPerDeviceFlag {
  int v[IRRL_MAX_DEVICES];
  IrrlPerDeviceFlag() { for (int i = 0; i < IRRL_MAX_DEVICES; i++) v[i] = -1; }
};
// returns 0 on success; 1 = unsupported shape, 2 = launch error
static int lstm_bf16_allow_lds(const void *kernel, int bytes) {
  // the tiles exceed the 64 KB a kernel gets without asking (gfx950 has 160 KB per CU)
  return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 0 : 2;
}
int irrl_lstm_seq_forward_bf16(int nsplit, int hid, int T, int N, int n_in, const float *x, const float *wx_p, const float *b_p, const float *wh_p,
                               const float *masks, const float *state0, float *gates, float *cseq, float *hseq, float *state_out, void *hip_stream) {
  if (N <= 0 || T <= 0 || (N % 16) != 0 || n_in <= 0 || n_in > LBF_KX || hid != LBF_HID || (nsplit != 2 && nsplit != 3)) return 1;
  LstmFwdBf16Args a;
  a.x = x; a.wx_p = wx_p; a.b_p = b_p; a.wh_p = wh_p; a.masks = masks; a.state0 = state0;
  a.gates = gates; a.cseq = cseq; a.hseq = hseq; a.state_out = state_out; a.T = T; a.N = N; a.n_in = n_in;
  hipStream_t s = (hipStream_t)hip_stream;
  // what is kept for a backward pass: gates + c (the kernel that loads its gates), c alone (gates NULL: the kernel that recomputes them, two
  // planes), nothing (both NULL: inference -- only hseq and state_out are written)
  if (gates != nullptr && cseq == nullptr) return 1;
  const int store = gates != nullptr ? 2 : (cseq != nullptr ? 1 : 0);
#define IRRL_FF(NS, ST) hipLaunchKernelGGL((lstm_seq_fwd_bf16_kernel<NS, ST>), dim3(N / 16), dim3(256), lstm_fwd_bf16_lds_bytes<NS>(), s, a)
  if (nsplit == 2) { if (store == 2) IRRL_FF(2, 2); else if (store == 1) IRRL_FF(2, 1); else IRRL_FF(2, 0); }
  else { if (store == 2) IRRL_FF(3, 2); else if (store == 1) IRRL_FF(3, 1); else IRRL_FF(3, 0); }
#undef IRRL_FF
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

int irrl_lstm_seq_backward_bf16(int nsplit, int hid, int T, int N, int n_in, const float *gates, const float *cseq, const float *hseq, const float *x,
                                const float *masks, const float *state0, const float *dh_in, const float *wh_p, const float *wx_p, const float *b_p,
                                float *dx, float *dwx_part, float *dwh_part, float *db_part, void *hip_stream) {
  if (N <= 0 || T <= 0 || (N % 16) != 0 || n_in <= 0 || n_in > LBF_KX || hid != LBF_HID || (nsplit != 2 && nsplit != 3)) return 1;
  if (gates == nullptr && (nsplit != 2 || b_p == nullptr)) return 1;      // the recomputing kernel: two planes, needs the bias
  LstmBwdBf16Args a;
  a.gates = gates; a.cseq = cseq; a.hseq = hseq; a.x = x; a.masks = masks; a.state0 = state0; a.dh_in = dh_in; a.wh_p = wh_p; a.wx_p = wx_p; a.b_p = b_p;
  a.dx = dx; a.dwx_part = dwx_part; a.dwh_part = dwh_part; a.db_part = db_part; a.T = T; a.N = N; a.n_in = n_in;
  hipStream_t s = (hipStream_t)hip_stream;
  // the opt-in belongs to the CURRENT device (a process may drive several GPUs): remembered per device ordinal
  static IrrlPerDeviceFlag allowed_on;     // function-local static with a constructor: initialised once, thread-safe (C++11)
  int dev_ = 0;
  if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= IRRL_MAX_DEVICES) return 2;
  int &allowed = allowed_on.v[dev_];
  if (allowed < 0) {
    allowed = lstm_bf16_allow_lds((const void *)lstm_seq_bwd_bf16_kernel<2, true>, lstm_bwd_bf16_lds_bytes<2>()) |
              lstm_bf16_allow_lds((const void *)lstm_seq_bwd_bf16_kernel<2, false>, lstm_bwd_bf16_lds_bytes<2>()) |
              lstm_bf16_allow_lds((const void *)lstm_seq_bwd_bf16_kernel<3, true>, lstm_bwd_bf16_lds_bytes<3>()) |
              lstm_bf16_allow_lds((const void *)lstm_seq_bwd_bf16_kernel<3, false>, lstm_bwd_bf16_lds_bytes<3>()) |
              lstm_bf16_allow_lds((const void *)lstm_seq_bwd_bf16_rc_kernel<true>, lstm_bwd_bf16_rc_lds_bytes<2>()) |
              lstm_bf16_allow_lds((const void *)lstm_seq_bwd_bf16_rc_kernel<false>, lstm_bwd_bf16_rc_lds_bytes<2>());
  }
  if (allowed != 0) return 2;
  if (gates == nullptr) {
    if (dx) hipLaunchKernelGGL((lstm_seq_bwd_bf16_rc_kernel<true>), dim3(N / 16), dim3(256), lstm_bwd_bf16_rc_lds_bytes<2>(), s, a);
    else hipLaunchKernelGGL((lstm_seq_bwd_bf16_rc_kernel<false>), dim3(N / 16), dim3(256), lstm_bwd_bf16_rc_lds_bytes<2>(), s, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
  }
#define IRRL_BB(NS, D) hipLaunchKernelGGL((lstm_seq_bwd_bf16_kernel<NS, D>), dim3(N / 16), dim3(256), lstm_bwd_bf16_lds_bytes<NS>(), s, a)
  if (nsplit == 2 && dx) IRRL_BB(2, true);
  else if (nsplit == 2) IRRL_BB(2, false);
  else if (dx) IRRL_BB(3, true);
  else IRRL_BB(3, false);
#undef IRRL_BB
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // extern "C"
