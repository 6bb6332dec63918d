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

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LSTM_DEV __device__ __forceinline__

LSTM_DEV float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
LSTM_DEV float fast_tanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

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

extern "C" {

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

}  // extern "C"
