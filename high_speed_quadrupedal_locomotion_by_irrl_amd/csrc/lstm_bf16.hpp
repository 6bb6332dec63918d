// lstm_bf16.hpp -- the persistent LSTM sequence kernels of the PPO2 update (run_bp_v5.py:143-176 network, ppo2.py:132-134 full-length
// BPTT) on the bf16 matrix cores with COMPENSATED OPERAND SPLITS, f32 accumulation (round 4).
//
// Why: the exact-f32 kernels of lstm_kernels.hip issue v_mfma_f32_16x16x4_f32, which gfx950 runs at the vector rate -- 13.9 ns per
// 16x16 tile and 4 units of K on a SIMD (3.48 ns per unit of K) -- while v_mfma_f32_16x16x32_bf16 takes 7.2 ns for 32 units of K
// (0.226 ns per unit) and v_mfma_f32_16x16x16_bf16 7.3 ns for 16 (tools/microbench/mfma_bf16_rate.hip, profiles/r04_mfma_bf16_rate.log).
// A float is split into NS bf16 planes x = p0 + p1 (+ p2) (each the round-to-nearest bf16 of what the planes before it left over:
// the residuals are exact in f32), and a product of two split operands is the sum of the plane products whose weight is above
// the target: NS = 2 -> p0 q0 + p0 q1 + p1 q0 (3 MFMAs, error ~2^-16 of the product), NS = 3 -> 6 MFMAs (~2^-24, the f32 level).
// Per unit of K that is 0.68 / 1.36 ns against 3.48 ns.  Accumulation stays f32 inside the MFMA, small planes first.
//
// Mapping (HID = 48, n_in <= 48, the reference's network).  A workgroup owns 16 envs for all T steps.
//   forward  (3 compute waves + the loader wave, see the kernel): wave w owns hidden units 16 w .. 16 w + 15 with their four gates; z_t = b + [h_{t-1} keep_t | x_t] [wh ; wx] is
//            ONE contraction over K = 96 = 3 chunks of 32; the A operand ([env][k] bf16 planes, h written by its owners at the end of
//            step t - 1, x_t staged by all lanes one step ahead) lives in a double-buffered LDS tile, the B fragments (weights,
//            split once) stay in registers for the whole sequence.  36 (NS 2) / 72 (NS 3) MFMAs + the cell per wave and step, one
//            workgroup barrier per step.
//   backward (4 waves): waves 0-2 own 16 units each: gate arithmetic -> dz (4 envs x 4 gates per lane), which goes to LDS as an
//            [env][gate column] tile: read row-wise for the recurrence / dx products (K = the 192 gate columns: 6 chunks of 32; N-split:
//            wave w produces dh_prev for ITS units and dx for input columns 16 w .. 16 w + 15, so no partial sums are exchanged) and
//            TRANSPOSED (ds_read_b64_tr_b16) for the weight gradients (K = the 16 envs of the step: v_mfma_f32_16x16x16_bf16), next to
//            (h_{t-1} keep_t)^T and x_t^T (staged by wave 3).  Of the 72 weight-gradient tiles (dwh 3 x 12, dwx 3 x 12) waves 0-2 accumulate
//            gate-column tiles 2 w, 2 w + 1 against all six M-tiles (12 tiles each), wave 3 -- which has nothing else to compute -- tiles 6 .. 11
//            (36 tiles), for all T steps.
//            Double-buffered tiles, one barrier per step.  Per-workgroup partial gradients as in lstm_seq_bwd_x_kernel.
#pragma once
#include "policy_step.hpp"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));

LSTM_DEV unsigned short bf_bits(float x) { const __bf16 b = (__bf16)x; return __builtin_bit_cast(unsigned short, b); }   // v_cvt_pk_bf16_f32: round to nearest even
LSTM_DEV float bf_val(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
template <int NS>
LSTM_DEV void bf_split(float x, unsigned short (&p)[NS]) {
  float r = x;
#pragma unroll
  for (int i = 0; i < NS; i++) { p[i] = bf_bits(r); r -= bf_val(p[i]); }   // the residual of a round-to-nearest bf16 is exact in f32
}
// plane products kept, smallest weight first: NS 2: (0,1) (1,0) (0,0); NS 3: (1,1) (0,2) (2,0) (0,1) (1,0) (0,0)
template <int NS> struct BfProducts;
template <> struct BfProducts<2> { static constexpr int N = 3; static constexpr int A[3] = {0, 1, 0}; static constexpr int B[3] = {1, 0, 0}; };
template <> struct BfProducts<3> { static constexpr int N = 6; static constexpr int A[6] = {1, 0, 2, 0, 1, 0}; static constexpr int B[6] = {1, 2, 0, 1, 0, 0}; };
#define BF_MFMA32(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a_), __builtin_bit_cast(bf16x8_t, b_), c_, 0, 0, 0)
#define BF_MFMA16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, a_), __builtin_bit_cast(s16x4_t, b_), c_, 0, 0, 0)

// ---- forward ---------------------------------------------------------------------------------------------------------------------
struct LstmFwdBf16Args {
  const float *x, *wx_p, *b_p, *wh_p, *masks, *state0;
  float *gates, *cseq, *hseq, *state_out;
  int T, N, n_in;
};
constexpr int LBF_HID = 48, LBF_KX = 48, LBF_KF = LBF_HID + LBF_KX;   // forward contraction: [h | x], 96 = 3 chunks of 32
constexpr int LBF_FROW = LBF_KF + 8;                                    // padded A-tile row (bf16 elements; 208 bytes: 16-byte aligned)

// STORE: what leaves the kernel per env and step.  2 = gates + c + h (1152 bytes: the backward kernel that LOADS its gates -- three planes);
// 1 = c + h (384 bytes; round 6: the two-plane backward kernel recomputes the gates from h_{t-1} and x_t, lstm_seq_bwd_bf16_rc_kernel);
// 0 = INFERENCE (round 5: the critic pass behind an actor-only rollout, ppo2.Runner._critic_pass): h and the final state only, 192 bytes.
//
// ROUND 5, LAST CHANGE: A FOURTH WAVE THAT DOES ALL THE LOADING, WITH FEW, WIDE LOADS.  Rounds 4-5 measured the symptom -- a pair of forward launches
// "bound by its stores" at 3.5 TB/s although the same store mix alone streams at 5.9 -- and a probe found the cause
// (profiles/r05_ab_lstm_fwd_prefetch_same_box.log): with every store kept and NO vector-memory load in the step loop the kernel runs at 751
// instead of 1216 us; the loads and the stores get in each other's way in the CU's one memory pipeline.  So the two kinds of traffic are split
// over different waves and the loads made few: waves 0-2 compute and STORE (no vector-memory load after the prologue); wave 3 LOADS -- the x rows
// (16-byte loads where the row allows, requested XD steps ahead, split to planes and staged into the A tile a step ahead) and the mask rows
// (published through a small LDS ring) -- and stores nothing.  Same arithmetic, same values, same results bit for bit; the pair 2059 -> ~1600 us.
#ifndef IRRL_LBF_FWD_XD
#define IRRL_LBF_FWD_XD 8
#endif
constexpr int LBF_FWD_XD = IRRL_LBF_FWD_XD;                             // steps of x / mask rows the loader wave has in flight
template <int NS> constexpr int lstm_fwd_bf16_lds_bytes() { return 2 * NS * 16 * LBF_FROW * 2 + 4 * 16 * 4; }      // two A tiles + the mask ring
template <int NS, int STORE = 2>
__global__ void __launch_bounds__(256)
lstm_seq_fwd_bf16_kernel(const LstmFwdBf16Args a) {
  constexpr int HID = LBF_HID, KC = LBF_KF / 32, XD = LBF_FWD_XD;
  using PR = BfProducts<NS>;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds_f[];
  // At(buf, plane, env, k)
  auto At = [&](int buf, int p, int env, int k) -> unsigned short * { return lds_f + (((size_t)(buf * NS + p) * 16 + env) * LBF_FROW + k); };
  float *mring = (float *)(lds_f + 2 * NS * 16 * LBF_FROW);            // [4 slots][16 envs]: slot t & 3 holds the masks of step t
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int col = l & 15, rq = l >> 4;
  const int e0 = blockIdx.x * 16;
  const int T = a.T, N = a.N, n_in = a.n_in;

  if (w == 3) {
    // ---------------- the loader wave: lane (env = l / 4, quarter = l % 4) owns input elements 12 quarter .. 12 quarter + 11 of its env ----------------
    const int xe = l >> 2, xq = l & 3;
    // 16-byte loads where a group of four lies inside the row (rows are only 4-byte aligned for n_in = 35), single words at its ragged end, nothing
    // beyond it: few vector-memory instructions per step, so that XD steps of them fit the wave's 63 outstanding operations
    auto load_x = [&](int t, float (&dst)[12]) {
      const float *row = a.x + ((size_t)t * N + e0 + xe) * n_in;
#pragma unroll
      for (int g4 = 0; g4 < 3; g4++) {
        const int k0 = 12 * xq + 4 * g4;
        if (k0 + 3 < n_in) {
          const f32x4u v = *(const f32x4u *)&row[k0];
          dst[4 * g4] = v[0]; dst[4 * g4 + 1] = v[1]; dst[4 * g4 + 2] = v[2]; dst[4 * g4 + 3] = v[3];
        } else if (k0 < n_in) {
#pragma unroll
          for (int i = 0; i < 4; i++) dst[4 * g4 + i] = row[k0 + i < n_in ? k0 + i : n_in - 1];
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++) dst[4 * g4 + i] = 0.0f;
        }
      }
    };
    auto stage_x = [&](int buf, const float (&src)[12]) {
#pragma unroll
      for (int g4 = 0; g4 < 3; g4++) {
        u16x4_t pk[NS];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int k = 12 * xq + 4 * g4 + i;
          unsigned short pl[NS];
          bf_split<NS>(k < n_in ? src[4 * g4 + i] : 0.0f, pl);
#pragma unroll
          for (int p = 0; p < NS; p++) pk[p][i] = pl[p];
        }
#pragma unroll
        for (int p = 0; p < NS; p++) *(u16x4_t *)At(buf, p, xe, HID + 12 * xq + 4 * g4) = pk[p];
      }
    };
    auto load_m = [&](int t) -> float { const int tc = t < T ? t : T - 1; return a.masks[(size_t)tc * N + e0 + (l & 15)]; };
    float xr[XD][12], mr[XD];
    load_x(0, xr[0]);
    stage_x(0, xr[0]);
#pragma unroll
    for (int d = 0; d < XD; d++) {
      if (d + 1 < T) load_x(d + 1, xr[d]);          // x_{d+1}, staged during step d
      mr[d] = load_m(d + 2);                         // the masks of step d + 2, published during step d
    }
    __syncthreads();
    for (int t = 0; t < T; t += XD) {
#pragma unroll
      for (int d = 0; d < XD; d++) {
        const int tt = t + d;
        if (tt < T) {
          if (tt + 1 < T) stage_x((tt + 1) & 1, xr[d]);
          if (l < 16) mring[((tt + 2) & 3) * 16 + l] = mr[d];
          if (tt + 1 + XD < T) load_x(tt + 1 + XD, xr[d]);
          mr[d] = load_m(tt + 2 + XD);
          __syncthreads();
        }
      }
    }
    return;
  }

  // ---------------- waves 0-2: wave w owns hidden units 16 w .. 16 w + 15 with their four gates ----------------
  const int u = 16 * w + col;
  // B fragments: B[k = 32 kc + 8 rq + i][column = unit u, gate g]; k < 48: wh_p[k][u][g], else wx_p[k - 48][u][g] (0 beyond n_in)
  u16x8_t Bf[KC][4][NS];
#pragma unroll
  for (int kc = 0; kc < KC; kc++)
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int k = 32 * kc + 8 * rq + i;
        float v;
        if (k < HID) v = a.wh_p[((size_t)k * HID + u) * 4 + g];
        else v = (k - HID < n_in) ? a.wx_p[((size_t)(k - HID) * HID + u) * 4 + g] : 0.0f;
        unsigned short pl[NS];
        bf_split<NS>(v, pl);
#pragma unroll
        for (int p = 0; p < NS; p++) Bf[kc][g][p][i] = pl[p];
      }
  const f32x4 bias = *(const f32x4 *)&a.b_p[u * 4];
  auto stage_h = [&](int buf, const float (&h)[4], const float (&keep)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      unsigned short pl[NS];
      bf_split<NS>(h[j] * keep[j], pl);
#pragma unroll
      for (int p = 0; p < NS; p++) *At(buf, p, 4 * rq + j, u) = pl[p];
    }
  };
  float c[4], hlast[4], mk_cur[4], mk_nxt[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int e = e0 + 4 * rq + j;
    c[j] = a.state0[(size_t)e * 2 * HID + u];
    hlast[j] = a.state0[(size_t)e * 2 * HID + HID + u];
    mk_cur[j] = a.masks[e];
    mk_nxt[j] = a.masks[(size_t)(T > 1 ? 1 : 0) * N + e];        // (the last loads of these waves: from here on they only store)
  }
  {
    float keep0[4];
#pragma unroll
    for (int j = 0; j < 4; j++) keep0[j] = 1.0f - mk_cur[j];
    stage_h(0, hlast, keep0);
  }
  __syncthreads();
  for (int t = 0; t < T; t++) {
    const int buf = t & 1;
    u16x8_t av[KC][NS];
#pragma unroll
    for (int kc = 0; kc < KC; kc++)
#pragma unroll
      for (int p = 0; p < NS; p++) av[kc][p] = *(const u16x8_t *)At(buf, p, col, 32 * kc + 8 * rq);
    if (t > 0) {      // the masks of step t + 1: published by the loader wave during step t - 1
      const f32x4 m4 = *(const f32x4 *)&mring[((t + 1) & 3) * 16 + 4 * rq];
      mk_nxt[0] = m4[0]; mk_nxt[1] = m4[1]; mk_nxt[2] = m4[2]; mk_nxt[3] = m4[3];
    }
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; g++) acc[g] = (f32x4){bias[g], bias[g], bias[g], bias[g]};
#pragma unroll
    for (int kc = KC - 1; kc >= 0; kc--)      // the x chunks first: they do not depend on the h the other waves have just published
#pragma unroll
      for (int q = 0; q < PR::N; q++)
#pragma unroll
        for (int g = 0; g < 4; g++) acc[g] = BF_MFMA32(av[kc][PR::A[q]], Bf[kc][g][PR::B[q]], acc[g]);
    float keepn[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float keepC = 1.0f - mk_cur[j];
      const float ig = fast_sigmoid(acc[0][j]), fg = fast_sigmoid(acc[1][j]), og = fast_sigmoid(acc[2][j]), gg = fast_tanh(acc[3][j]);
      const float cn = fg * (c[j] * keepC) + ig * gg;
      const float hn = og * fast_tanh(cn);
      c[j] = cn;
      hlast[j] = hn;
      const size_t row = (size_t)t * N + e0 + 4 * rq + j;
#ifndef IRRL_LBF_AB_NO_GATE_STORES      /* A/B switches of tools/build_variants.py (wrong results): which of the forward kernel's stores cost what */
      if (STORE == 2) *(f32x4 *)&a.gates[(row * HID + u) * 4] = (f32x4){ig, fg, og, gg};
#endif
#ifndef IRRL_LBF_AB_NO_CH_STORES
      if (STORE >= 1) a.cseq[row * HID + u] = cn;
      a.hseq[row * HID + u] = hn;
#endif
      keepn[j] = 1.0f - mk_nxt[j];
    }
    if (t + 1 < T) stage_h(buf ^ 1, hlast, keepn);
#pragma unroll
    for (int j = 0; j < 4; j++) mk_cur[j] = mk_nxt[j];
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int e = e0 + 4 * rq + j;
    a.state_out[(size_t)e * 2 * HID + u] = c[j];
    a.state_out[(size_t)e * 2 * HID + HID + u] = hlast[j];
  }
}

// ---- backward --------------------------------------------------------------------------------------------------------------------
struct LstmBwdBf16Args {
  const float *gates, *cseq, *hseq, *x, *masks, *state0, *dh_in, *wh_p, *wx_p, *b_p;      // gates NULL + b_p: the recomputing kernel (two planes)
  float *dx, *dwx_part, *dwh_part, *db_part;
  int T, N, n_in;
};
constexpr int LBF_GC = 4 * LBF_HID;         // gate columns
constexpr int LBF_RROW = LBF_GC + 8;        // dz tile [env][gate column]: padded row, 400 bytes (16-byte aligned)
constexpr int LBF_CROW = 16 + 4;            // [gate column | unit | input][env]: padded row, 40 bytes (8-byte aligned)
template <int NS> constexpr int lstm_bwd_bf16_lds_elems_per_buf() { return NS * (16 * LBF_RROW + LBF_HID * LBF_CROW + LBF_KX * LBF_CROW); }
template <int NS> constexpr int lbf_total_bytes();
template <int NS> constexpr int lstm_bwd_bf16_lds_bytes() { return 2 * lstm_bwd_bf16_lds_elems_per_buf<NS>() * 2 + lbf_total_bytes<NS>(); }

// the weight-gradient products of one step for NCI gate-column tiles (first one: ci0) against all six M-tiles (0-2: hidden rows of dwh, 3-5: input
// rows of dwx); `bufbase` = the step's LDS buffer.  Per accumulator tile the plane products arrive in the same order whoever owns the tile.
template <int NS, int NCI>
LSTM_DEV void lbf_weight_grads(f32x4 (&accW)[6][NCI], const unsigned short *bufbase, int ci0, int col, int rq) {
  using PR = BfProducts<NS>;
  constexpr int OFF_H = NS * 16 * LBF_RROW, OFF_X = OFF_H + NS * LBF_HID * LBF_CROW;
  u16x4_t bz[NCI][NS], am[6][NS];
  // dz as the B operand (k = the step's envs) comes TRANSPOSED out of the [env][gate column] tile the recurrence reads row-wise
  // (ds_read_b64_tr_b16: lane 4 q + p of a 16-lane group addresses row 4 rq + q, columns 4 p .. 4 p + 3 of the 16-column block; lane i gets
  // column i of the four rows) -- no second copy of dz in a [gate column][env] layout
  const int tr_off = (4 * rq + (col >> 2)) * LBF_RROW + 4 * (col & 3);
#pragma unroll
  for (int p = 0; p < NS; p++) {
#pragma unroll
    for (int ci = 0; ci < NCI; ci++) {
      typedef short lbf_s16x4 __attribute__((ext_vector_type(4)));
      const lbf_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lbf_s16x4 __attribute__((address_space(3))) *)(bufbase + (size_t)p * 16 * LBF_RROW + tr_off + 16 * (ci0 + ci)));
      bz[ci][p] = __builtin_bit_cast(u16x4_t, v);
    }
#pragma unroll
    for (int mt = 0; mt < 3; mt++) {
      am[mt][p] = *(const u16x4_t *)(bufbase + OFF_H + ((size_t)p * LBF_HID + 16 * mt + col) * LBF_CROW + 4 * rq);
      am[3 + mt][p] = *(const u16x4_t *)(bufbase + OFF_X + ((size_t)p * LBF_KX + 16 * mt + col) * LBF_CROW + 4 * rq);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < PR::N; q++)
#pragma unroll
    for (int mt = 0; mt < 6; mt++)
#pragma unroll
      for (int ci = 0; ci < NCI; ci++) accW[mt][ci] = BF_MFMA16(am[mt][PR::A[q]], bz[ci][PR::B[q]], accW[mt][ci]);
}
template <int NCI>
LSTM_DEV void lbf_store_weight_grads(const f32x4 (&accW)[6][NCI], const LstmBwdBf16Args &a, int ci0, int col, int rq) {
  const size_t blk = blockIdx.x;
#pragma unroll
  for (int ci = 0; ci < NCI; ci++) {
    const int cc = 16 * (ci0 + ci) + col;
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
      for (int mt = 0; mt < 3; mt++) {
        a.dwh_part[(blk * LBF_HID + 16 * mt + 4 * rq + r) * LBF_GC + cc] = accW[mt][ci][r];
        const int i = 16 * mt + 4 * rq + r;
        if (i < a.n_in) a.dwx_part[(blk * a.n_in + i) * LBF_GC + cc] = accW[3 + mt][ci][r];
      }
    }
  }
}

// TWO-LEVEL ACCUMULATION OF THE WEIGHT GRADIENTS (round 6; three planes = the f32-level arithmetic only).  A tile's accumulator takes the plane
// products of every step: T x PR::N matrix-core additions into ONE f32 value each, every one rounded at the magnitude the running sum has
// reached -- at T = 750 that rounding walk, not the operand splits, is what bf16x6's weight-gradient error consisted of (4.1e-6 / 5.6e-6 of the
// largest entry on dwx / dwh against 1.7e-6 / 2.6e-6 of the exact-f32 kernels, which add 4 instead of 6 products per step and tile;
// profiles/r05_pytest_gpu.log).  So the register accumulators only ever hold LBF_FLUSH steps: every LBF_FLUSH steps they are added (vector ALU,
// round to nearest) to the tile's running total, which lives in LDS -- 72 tiles x 1 KB behind the operand buffers, each element read and written
// by the one lane that owns it: no synchronisation, one ds_read_b128 + ds_write_b128 per tile and flush -- and cleared.  The products of a block
// meet an accumulator that is LBF_FLUSH / T as large; the T / LBF_FLUSH block sums are ordinary f32 additions.  (Two planes: the 2^-16 of the
// dropped plane products dominates, nothing to gain; that kernel is unchanged and has no register to spare.)
#ifndef IRRL_LBF_FLUSH
#define IRRL_LBF_FLUSH 32
#endif
template <int NS> constexpr int lbf_flush_steps() { return NS == 3 ? IRRL_LBF_FLUSH : 0; }      // 0: one accumulation over all T steps (rounds 4-5)
template <int NS> constexpr int lbf_total_bytes() { return lbf_flush_steps<NS>() > 0 ? 72 * 256 * 4 : 0; }
// acc -> total (and cleared), or, behind the last step, total -> acc (what lbf_store_weight_grads then writes out)
template <int NCI>
LSTM_DEV void lbf_flush_weight_grads(f32x4 (&accW)[6][NCI], float *tot, int ci0, int l, bool last) {
#pragma unroll
  for (int ci = 0; ci < NCI; ci++)
#pragma unroll
    for (int mt = 0; mt < 6; mt++) {
      f32x4 *p = (f32x4 *)(tot + ((size_t)((ci0 + ci) * 6 + mt) * 256 + 4 * l));
      const f32x4 sum = *p + accW[mt][ci];
      if (last) accW[mt][ci] = sum;
      else { *p = sum; accW[mt][ci] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }
    }
}
template <int NCI>
LSTM_DEV void lbf_clear_totals(float *tot, int ci0, int l) {
#pragma unroll
  for (int ci = 0; ci < NCI; ci++)
#pragma unroll
    for (int mt = 0; mt < 6; mt++) *(f32x4 *)(tot + ((size_t)((ci0 + ci) * 6 + mt) * 256 + 4 * l)) = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
}

#ifdef IRRL_LBF_AB_OCC2      /* A/B probe of tools/build_variants.py: at most 256 registers, so that the two stacks' launches can share a CU (two waves per SIMD) */
#define LBF_BWD_BOUNDS __launch_bounds__(256, 2)
#else
#define LBF_BWD_BOUNDS __launch_bounds__(256)
#endif
template <int NS, bool NEED_DX>
__global__ void LBF_BWD_BOUNDS
lstm_seq_bwd_bf16_kernel(const LstmBwdBf16Args a) {
  constexpr int HID = LBF_HID, GC = LBF_GC, KX = LBF_KX, KC = GC / 32;
#ifdef IRRL_LBF_DEPTH      /* A/B switch of tools/build_variants.py */
  constexpr int DEPTH = (NS == 2) ? IRRL_LBF_DEPTH : (NEED_DX ? 1 : 2);
#else
#ifndef IRRL_LBF_DEPTH_X6
#define IRRL_LBF_DEPTH_X6 (NEED_DX ? 1 : 2)
#endif
  constexpr int DEPTH = (NS == 2) ? 3 : IRRL_LBF_DEPTH_X6;    // steps of operand loads in flight (36 registers per step; NS 3 has fewer to spare)
#endif
  using PR = BfProducts<NS>;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds_b[];
  constexpr int PER_BUF = lstm_bwd_bf16_lds_elems_per_buf<NS>();
  constexpr int OFF_H = NS * 16 * LBF_RROW, OFF_X = OFF_H + NS * HID * LBF_CROW;
  auto Zr = [&](int buf, int p, int env, int c) -> unsigned short * { return lds_b + (size_t)buf * PER_BUF + ((size_t)p * 16 + env) * LBF_RROW + c; };
  auto Ht = [&](int buf, int p, int k, int env) -> unsigned short * { return lds_b + (size_t)buf * PER_BUF + OFF_H + ((size_t)p * HID + k) * LBF_CROW + env; };
  auto Xt = [&](int buf, int p, int i, int env) -> unsigned short * { return lds_b + (size_t)buf * PER_BUF + OFF_X + ((size_t)p * KX + i) * LBF_CROW + env; };
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int col = l & 15, rq = l >> 4;
  const int e0 = blockIdx.x * 16;
  const int T = a.T, N = a.N, n_in = a.n_in;
  const bool main_wave = w < 3;
  const int u = 16 * (main_wave ? w : 0) + col;
#ifdef IRRL_LBF_MAIN_TILES      /* A/B switch of tools/build_variants.py: gate-column tiles per main wave (3 = 18 accumulator tiles per wave, the even split) */
  constexpr int MAIN_CI = IRRL_LBF_MAIN_TILES;
#else
  // two planes: 2 tiles per main wave (12 accumulator tiles; shared out evenly the main waves' chain got longer: +1-1.5 % per update, round 4);
  // three planes (six products per tile: wave 3's 216 MFMAs per step were the longest chain): 3 each -- 138.1 -> 134.0 ms per update, same box
  // (round 5, profiles/r05_ab_lstm_bwd_x6_tile_split_same_box.log)
  constexpr int MAIN_CI = (NS == 3) ? 3 : 2;
#endif
  constexpr int HELP_PARTS = (12 - 3 * MAIN_CI) / 3;     // the helper wave's tiles in parts of three
  constexpr int FLUSH = lbf_flush_steps<NS>();           // two-level accumulation of the weight gradients (above); 0 = off
  float *totals = (float *)(lds_b + 2 * (size_t)PER_BUF);   // [72 tiles][256]: only touched when FLUSH > 0 (the launch then asks for the bytes)
  if (FLUSH > 0) {      // every wave clears the tiles it owns; nobody else ever touches them
    if (main_wave) lbf_clear_totals<MAIN_CI>(totals, MAIN_CI * w, l);
    else
#pragma unroll
      for (int hf = 0; hf < HELP_PARTS; hf++) lbf_clear_totals<3>(totals, 3 * MAIN_CI + 3 * hf, l);
  }
  // The 72 weight-gradient tiles (12 gate-column tiles x 6 M-tiles): waves 0-2, which also carry the gate arithmetic and the recurrence, own
  // gate-column tiles 2 w, 2 w + 1 (12 accumulator tiles, 36 MFMAs per step); wave 3, which otherwise only stages x_t^T, owns tiles 6 .. 11 (36
  // accumulator tiles, 108 MFMAs per step).  (Shared out evenly -- 18 tiles each -- the main waves were the step's critical path.)
  if (!main_wave) {
    // ---- wave 3: stages x_t^T for the weight gradients (coalesced row reads, two steps ahead) and takes its share of them ----
    float xr[DEPTH + 1][12];
    auto load_x = [&](int t, float (&dst)[12]) {
      const float *x_t = a.x + (size_t)t * N * n_in;      // uniform row pointer + 32-bit lane index
#pragma unroll
      for (int r = 0; r < 12; r++) {
        const int idx = 64 * r + l, env = idx / KX, i = idx % KX;
        dst[r] = x_t[(unsigned)(e0 + env) * (unsigned)n_in + (unsigned)(i < n_in ? i : n_in - 1)];
      }
    };
    auto stage_x = [&](int buf, const float (&src)[12]) {
#pragma unroll
      for (int r = 0; r < 12; r++) {
        const int idx = 64 * r + l, env = idx / KX, i = idx % KX;
        unsigned short pl[NS];
        bf_split<NS>(i < n_in ? src[r] : 0.0f, pl);
#pragma unroll
        for (int p = 0; p < NS; p++) *Xt(buf, p, i, env) = pl[p];
      }
    };
    f32x4 accH[HELP_PARTS][6][3];      // parts of three gate-column tiles: one part's operands are in flight at a time
#pragma unroll
    for (int hf = 0; hf < HELP_PARTS; hf++)
#pragma unroll
      for (int mt = 0; mt < 6; mt++)
#pragma unroll
        for (int ci = 0; ci < 3; ci++) accH[hf][mt][ci] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int HD = DEPTH + 1;
#pragma unroll
    for (int d = 0; d < HD; d++)
      if (T - 1 - d >= 0) load_x(T - 1 - d, xr[d]);
    // (the flush sits BETWEEN blocks of steps, outside the step loop: a test inside it ends the basic block in which this step's matrix-core
    // products run under the next step's staging -- measured: +7 % per update with the test per step, profiles/r06_ab_lstm_bwd_two_level_accumulation.log)
    constexpr int HBLK = FLUSH > 0 ? (FLUSH / HD > 0 ? FLUSH / HD : 1) * HD : (1 << 30);
    for (int tb = T - 1; tb >= 0; tb -= HBLK) {
      const int t_end = tb - HBLK + 1 > 0 ? tb - HBLK + 1 : 0;
      for (int t = tb; t >= t_end; t -= HD) {
#pragma unroll
        for (int d = 0; d < HD; d++) {
          const int tt = t - d;
          if (tt < 0) break;
          stage_x(tt & 1, xr[d]);
          if (tt - HD >= 0) load_x(tt - HD, xr[d]);
          __syncthreads();
#pragma unroll
          for (int hf = 0; hf < HELP_PARTS; hf++) lbf_weight_grads<NS, 3>(accH[hf], lds_b + (size_t)(tt & 1) * PER_BUF, 3 * MAIN_CI + 3 * hf, col, rq);
        }
      }
      if (FLUSH > 0 && t_end > 0) {
#pragma unroll
        for (int hf = 0; hf < HELP_PARTS; hf++) lbf_flush_weight_grads<3>(accH[hf], totals, 3 * MAIN_CI + 3 * hf, l, false);
      }
    }
#pragma unroll
    for (int hf = 0; hf < HELP_PARTS; hf++) {
      if (FLUSH > 0) lbf_flush_weight_grads<3>(accH[hf], totals, 3 * MAIN_CI + 3 * hf, l, true);
      lbf_store_weight_grads<3>(accH[hf], a, 3 * MAIN_CI + 3 * hf, col, rq);
    }
    return;
  }
  // ---- waves 0-2 ----
  // B fragments over K = gate column c = 32 kc + 8 rq + i: dh_prev[env][hidden k' = 16 w + col] = sum_c dz[env][c] wh[k'][c],
  // dx[env][input i' = 16 w + col] = sum_c dz[env][c] wx[i'][c]  (rows of the permuted weight matrices are contiguous in c)
  u16x8_t Bh[KC][NS], Bx[NEED_DX ? KC : 1][NS];
#pragma unroll
  for (int kc = 0; kc < KC; kc++)
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int cidx = 32 * kc + 8 * rq + i;
      unsigned short pl[NS];
      bf_split<NS>(a.wh_p[(size_t)u * GC + cidx], pl);
#pragma unroll
      for (int p = 0; p < NS; p++) Bh[kc][p][i] = pl[p];
      if (NEED_DX) {
        bf_split<NS>((u < n_in) ? a.wx_p[(size_t)u * GC + cidx] : 0.0f, pl);
#pragma unroll
        for (int p = 0; p < NS; p++) Bx[kc][p][i] = pl[p];
      }
    }
  float dbacc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float dc[4] = {0.0f, 0.0f, 0.0f, 0.0f}, dhrec[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  // The operands of a step are requested DEPTH steps ahead (DEPTH register sets used in turn).  A step of this kernel is ~2 us of
  // arithmetic, less than the latency of its 24 KB of loads under a fully loaded memory system (the rows of consecutive steps are 3 MB
  // apart in each of six arrays): with one step in flight the kernel ran at that latency -- 5.4 us per step, 1.4 TB/s (round 4, first
  // version) -- not at its arithmetic.
  f32x4 accW[6][MAIN_CI];
#pragma unroll
  for (int mt = 0; mt < 6; mt++)
#pragma unroll
    for (int ci = 0; ci < MAIN_CI; ci++) accW[mt][ci] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  struct StepOps { f32x4 g[4]; float ct[4], cp[4], dh[4], mk[4], hp[4]; };
  // Addresses as UNIFORM row pointer (scalar registers, advanced per step by the scalar ALU) + a 32-bit per-lane element index: the loads take the
  // `global_load v, v_offset, s[base]` form.  Written with 64-bit per-lane indices ((size_t) t * N + e) * HID + u, every one of a step's 24 loads
  // carried its own 64-bit multiply-add chain on the vector ALU -- ~80 of the main waves' 438 vector instructions per step, on the waves whose
  // issue rate IS the step (round 6; profiles/r06_ab_lstm_scalar_row_pointers_same_box.log).
  auto fetch = [&](int t, StepOps &o) {
    const size_t trow = (size_t)t * N;
    const float *mk_t = a.masks + trow, *ct_t = a.cseq + trow * HID, *dh_t = a.dh_in + trow * HID;
#ifndef IRRL_LBF_AB_RECOMPUTE_PROBE
    const float *g_t = a.gates + trow * HID * 4;
#endif
    // c_{t-1} / h_{t-1}: the rows of step t - 1, or (t == 0, uniform) the initial state [env][c | h]
    const float *cp_t = t > 0 ? a.cseq + (trow - N) * HID : a.state0;
    const float *hp_t = t > 0 ? a.hseq + (trow - N) * HID : a.state0 + HID;
    const unsigned pstride = t > 0 ? (unsigned)HID : 2u * HID;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const unsigned e = (unsigned)(e0 + 4 * rq + j);
      o.mk[j] = mk_t[e];
#ifdef IRRL_LBF_AB_RECOMPUTE_PROBE      /* A/B probe of tools/build_variants.py (WRONG results): what "recompute the gates instead of loading them" would cost here */
      o.g[j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#else
      o.g[j] = *(const f32x4 *)&g_t[(e * HID + u) * 4u];
#endif
      o.ct[j] = ct_t[e * HID + u];
      o.cp[j] = cp_t[e * pstride + u];
      o.hp[j] = hp_t[e * pstride + u];
      o.dh[j] = dh_t[e * HID + u];
    }
  };
  auto step = [&](int t, StepOps &o) {
    const int buf = t & 1;
    float keepC[4], ct[4], cpv[4], dhv[4], hpv[4];
    f32x4 g4[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { keepC[j] = 1.0f - o.mk[j]; g4[j] = o.g[j]; ct[j] = o.ct[j]; cpv[j] = o.cp[j]; dhv[j] = o.dh[j]; hpv[j] = o.hp[j] * keepC[j]; }
    if (t - DEPTH >= 0) fetch(t - DEPTH, o);
#ifdef IRRL_LBF_AB_RECOMPUTE_PROBE
    {
      // the forward kernel's matrix-core stage for this wave's 16 units x 4 gates (PR::N x 3 chunks x 4 gates MFMAs over K = 96) on whatever lies in
      // the other buffer's tile, then the four activations per (env, unit): the instruction mix of a real recompute, not its values
      u16x8_t ap[3][NS];
#pragma unroll
      for (int kc = 0; kc < 3; kc++)
#pragma unroll
        for (int p = 0; p < NS; p++) ap[kc][p] = *(const u16x8_t *)Zr(buf ^ 1, p, col, 32 * kc + 8 * rq);
      f32x4 zacc[4];
#pragma unroll
      for (int g = 0; g < 4; g++) zacc[g] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int kc = 0; kc < 3; kc++)
#pragma unroll
        for (int q = 0; q < PR::N; q++)
#pragma unroll
          for (int g = 0; g < 4; g++) zacc[g] = BF_MFMA32(ap[kc][PR::A[q]], Bh[(kc + g) % KC][PR::B[q]], zacc[g]);
#pragma unroll
      for (int j = 0; j < 4; j++) g4[j] = (f32x4){fast_sigmoid(zacc[0][j]), fast_sigmoid(zacc[1][j]), fast_sigmoid(zacc[2][j]), fast_tanh(zacc[3][j])};
    }
#endif
    // (h_{t-1} keep_t)^T for the weight gradients: Ht[plane][unit u][env 4 rq .. 4 rq + 3]
    {
      u16x4_t pk[NS];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        unsigned short pl[NS];
        bf_split<NS>(hpv[j], pl);
#pragma unroll
        for (int p = 0; p < NS; p++) pk[p][j] = pl[p];
      }
#pragma unroll
      for (int p = 0; p < NS; p++) *(u16x4_t *)Ht(buf, p, u, 4 * rq) = pk[p];
    }
    // gate arithmetic -> dz (env 4 rq + j, unit u, gates i f o g) into the [env][gate column] tile
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
      const float dz4[4] = {d_i * ig * (1.0f - ig), d_f * fg * (1.0f - fg), d_o * og * (1.0f - og), d_g * (1.0f - gg * gg)};
      u16x4_t zr[NS];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        dbacc[g] += dz4[g];
        unsigned short pl[NS];
        bf_split<NS>(dz4[g], pl);
#pragma unroll
        for (int p = 0; p < NS; p++) zr[p][g] = pl[p];
      }
#pragma unroll
      for (int p = 0; p < NS; p++) *(u16x4_t *)Zr(buf, p, 4 * rq + j, 4 * u) = zr[p];
    }
    __syncthreads();      // dz_t, (h_{t-1} keep_t)^T and x_t^T of every wave are visible
    // recurrence (+ dx): A[env = col][k = gate column]
    u16x8_t av[KC][NS];
#pragma unroll
    for (int kc = 0; kc < KC; kc++)
#pragma unroll
      for (int p = 0; p < NS; p++) av[kc][p] = *(const u16x8_t *)Zr(buf, p, col, 32 * kc + 8 * rq);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}, accx = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < PR::N; q++)
#pragma unroll
      for (int kc = 0; kc < KC; kc++) {
        acc = BF_MFMA32(av[kc][PR::A[q]], Bh[kc][PR::B[q]], acc);
        if (NEED_DX) accx = BF_MFMA32(av[kc][PR::A[q]], Bx[kc][PR::B[q]], accx);
      }
#pragma unroll
    for (int j = 0; j < 4; j++) dhrec[j] = acc[j] * keepC[j];
    if (NEED_DX && u < n_in) {
      float *dx_t = a.dx + (size_t)t * N * n_in;      // uniform row pointer
#pragma unroll
      for (int j = 0; j < 4; j++) dx_t[(unsigned)(e0 + 4 * rq + j) * (unsigned)n_in + u] = accx[j];
    }
    lbf_weight_grads<NS, MAIN_CI>(accW, lds_b + (size_t)buf * PER_BUF, MAIN_CI * w, col, rq);
  };
  StepOps ops[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; d++)
    if (T - 1 - d >= 0) fetch(T - 1 - d, ops[d]);
  // blocks of ~FLUSH steps with the flush between them (outside the step loop, see the helper wave); the bias gradient's running sums get the
  // same two levels
  constexpr int MBLK = FLUSH > 0 ? (FLUSH / DEPTH > 0 ? FLUSH / DEPTH : 1) * DEPTH : (1 << 30);
  float dbtot[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int tb = T - 1; tb >= 0; tb -= MBLK) {
    const int t_end = tb - MBLK + 1 > 0 ? tb - MBLK + 1 : 0;
    for (int t = tb; t >= t_end; t -= DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; d++)
        if (t - d >= 0) step(t - d, ops[d]);
    }
    if (FLUSH > 0 && t_end > 0) {
      lbf_flush_weight_grads<MAIN_CI>(accW, totals, MAIN_CI * w, l, false);
#pragma unroll
      for (int g = 0; g < 4; g++) { dbtot[g] += dbacc[g]; dbacc[g] = 0.0f; }
    }
  }
  if (FLUSH > 0) lbf_flush_weight_grads<MAIN_CI>(accW, totals, MAIN_CI * w, l, true);
  lbf_store_weight_grads<MAIN_CI>(accW, a, MAIN_CI * w, col, rq);
  const size_t blk = blockIdx.x;
#pragma unroll
  for (int g = 0; g < 4; g++) a.db_part[(blk * 4 + rq) * GC + u * 4 + g] = dbtot[g] + dbacc[g];
}

// ---- backward with the gates RECOMPUTED (round 6, verdict r5 item 2; two planes = bf16x3, the learner's default) ---------------------------------
// The forward kernel's gate stores are 2/3 of its bytes (768 of 1152 per env and step) and the backward kernel's gate loads half of its own.
// z_t = b + [h_{t-1} keep_t | x_t] [wh ; wx] depends on nothing the backward recurrence produces, and both of its operands are already on their way
// through this kernel: (h_{t-1} keep_t)^T and x_t^T are staged in LDS for the weight gradients.  So the kernel forms z_t itself -- the forward kernel's
// matrix-core stage, instruction for instruction (same operand planes, same products, same order: the recomputed gates ARE the forward's, bit for
// bit) -- and the forward kernel of a two-plane update stores c and h only.
//   * A operand ([env][k] rows, k = 32 kc + 8 rq ..) straight out of the TRANSPOSED tiles the weight gradients read: two ds_read_b64_tr_b16 per
//     (chunk, plane) turn four k-rows x 16 envs into four k-values of one env per lane -- no second, row-major copy of h / x.  For that the tiles
//     of step t must be complete BEFORE step t's gate arithmetic: h / x are staged ONE STEP AHEAD (step t + 1 stages the tiles of step t), which
//     needs a third buffer for them (the weight gradients of step t + 1 are still reading theirs).  The dz tile stays double-buffered.
//   * B operand: the forward kernel's weight fragments, 96 registers + the bias -- the room the gates' three register sets left (48) plus what the
//     block-structured step loop freed (profiles/r06_ab_lstm_bwd_two_level_accumulation.log).
//   * MEASURED, AND NOT THE DEFAULT.  A probe that issued a recompute's matrix-core and activation mix on stale data (no staging, no tile reads, the
//     weight fragments it needed already in registers) had promised 96.3 -> 88.0 ms per update (79.5 with the gate stores gone and the backward kernel
//     unchanged).  The real kernel -- bit-identical to the loading one in every output, tests/test_gpu_ppo.py -- takes the forward pairs from 3.2 to 2.1 ms
//     per epoch and the four backward launches from 5.7 to ~7.5: update 94.4 -> 106.4 ms (same box).  Placing the recompute behind the previous step's
//     barrier instead of in front of the gate arithmetic changes nothing (106.2 / 106.5): the step is not waiting on that chain, the main waves are
//     issue-bound, and the recompute adds ~1 100 clocks of tile reads, fragment reads and transcendentals per step to them.  With the forward fragments
//     in registers instead (IRRL_LBF_RC_BF_REGS) the kernel sits at 512 registers and the update at 118-120 ms.  lstm_fused.RECOMPUTE_GATES /
//     IRRL_LSTM_RECOMPUTE=1 selects it (profiles/r06_ab_lstm_recompute_same_box.log).
// Three planes keep the loading kernel in any case: 72 more MFMAs per wave and step (probe: 129.3 -> 133.3 ms).
constexpr int LBF_RC_Z = 16 * LBF_RROW;                       // per plane: dz tile [env][gate column]
constexpr int LBF_RC_HX = (LBF_HID + LBF_KX) * LBF_CROW;      // per plane: [k = 0..47: unit | 48..95: input][env]
// One set of weight fragments does not fit the register file beside the others (all three: 512 + spills, 240 register-file moves per step): the
// forward fragments live in LDS -- every lane's own 16 bytes per (chunk, gate, plane), written once, read back by the same lane each step (24
// ds_read_b128): update 106 ms.  IRRL_LBF_RC_BF_REGS (A/B): the forward fragments in registers and the dx fragments in LDS instead (12 reads per step
// in the layer-1 launches, none in the others): 120 ms -- what costs is the register file, not the LDS reads (profiles/r06_ab_lstm_recompute_same_box.log).
#ifndef IRRL_LBF_RC_BF_REGS
constexpr int LBF_RC_BF = 3 * (LBF_KF / 32) * 4 * 64 * 8;     // per plane: [wave][chunk][gate][lane] x 8 bf16
#else
constexpr int LBF_RC_BF = 3 * (LBF_GC / 32) * 64 * 8;         // per plane: the dx fragments, [wave][chunk][lane] x 8 bf16
#endif
template <int NS> constexpr int lstm_bwd_bf16_rc_lds_bytes() { return (2 * NS * LBF_RC_Z + 3 * NS * LBF_RC_HX + NS * LBF_RC_BF) * 2; }

template <int NS, int NCI>
LSTM_DEV void lbf_weight_grads_rc(f32x4 (&accW)[6][NCI], const unsigned short *zbase, const unsigned short *hxbase, int ci0, int col, int rq) {
  using PR = BfProducts<NS>;
  u16x4_t bz[NCI][NS], am[6][NS];
  const int tr_off = (4 * rq + (col >> 2)) * LBF_RROW + 4 * (col & 3);
#pragma unroll
  for (int p = 0; p < NS; p++) {
#pragma unroll
    for (int ci = 0; ci < NCI; ci++) {
      typedef short lbf_s16x4 __attribute__((ext_vector_type(4)));
      const lbf_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lbf_s16x4 __attribute__((address_space(3))) *)(zbase + (size_t)p * LBF_RC_Z + tr_off + 16 * (ci0 + ci)));
      bz[ci][p] = __builtin_bit_cast(u16x4_t, v);
    }
#pragma unroll
    for (int mt = 0; mt < 6; mt++) am[mt][p] = *(const u16x4_t *)(hxbase + (size_t)p * LBF_RC_HX + (16 * mt + col) * LBF_CROW + 4 * rq);      // rows 0-47: h, 48-95: x
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < PR::N; q++)
#pragma unroll
    for (int mt = 0; mt < 6; mt++)
#pragma unroll
      for (int ci = 0; ci < NCI; ci++) accW[mt][ci] = BF_MFMA16(am[mt][PR::A[q]], bz[ci][PR::B[q]], accW[mt][ci]);
}

template <bool NEED_DX>
__global__ void __launch_bounds__(256)
lstm_seq_bwd_bf16_rc_kernel(const LstmBwdBf16Args a) {
  constexpr int NS = 2, HID = LBF_HID, GC = LBF_GC, KX = LBF_KX, KC = GC / 32, KF = LBF_KF / 32, MAIN_CI = 2, HELP_PARTS = 2;
  constexpr int DEPTH = 3;      // steps of operand loads in flight: 20 registers per step (the gates are not among the operands any more)
  using PR = BfProducts<NS>;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds_rc[];
  unsigned short *const zt = lds_rc, *const hx = lds_rc + 2 * NS * LBF_RC_Z, *const bfl = hx + 3 * NS * LBF_RC_HX;
  auto Zr = [&](int buf, int p, int env, int c) -> unsigned short * { return zt + ((size_t)(buf * NS + p) * 16 + env) * LBF_RROW + c; };
  auto HX = [&](int hb, int p, int k, int env) -> unsigned short * { return hx + ((size_t)(hb * NS + p) * (HID + KX) + k) * LBF_CROW + env; };
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int col = l & 15, rq = l >> 4;
  const int e0 = blockIdx.x * 16;
  const int T = a.T, N = a.N, n_in = a.n_in;
  if (w == 3) {
    // ---- wave 3: stages x^T ONE STEP AHEAD (rows 48.. of the h / x tile) and owns gate-column tiles 6 .. 11 of the weight gradients ----
    constexpr int HD = DEPTH + 1;
    float xr[HD][12];
    auto load_x = [&](int t, float (&dst)[12]) {
      const float *x_t = a.x + (size_t)t * N * n_in;      // uniform row pointer + 32-bit lane index
#pragma unroll
      for (int r = 0; r < 12; r++) {
        const int idx = 64 * r + l, env = idx / KX, i = idx % KX;
        dst[r] = x_t[(unsigned)(e0 + env) * (unsigned)n_in + (unsigned)(i < n_in ? i : n_in - 1)];
      }
    };
    auto stage_x = [&](int hb, const float (&src)[12]) {
#pragma unroll
      for (int r = 0; r < 12; r++) {
        const int idx = 64 * r + l, env = idx / KX, i = idx % KX;
        unsigned short pl[NS];
        bf_split<NS>(i < n_in ? src[r] : 0.0f, pl);
#pragma unroll
        for (int p = 0; p < NS; p++) *HX(hb, p, HID + i, env) = pl[p];
      }
    };
    f32x4 accH[HELP_PARTS][6][3];
#pragma unroll
    for (int hf = 0; hf < HELP_PARTS; hf++)
#pragma unroll
      for (int mt = 0; mt < 6; mt++)
#pragma unroll
        for (int ci = 0; ci < 3; ci++) accH[hf][mt][ci] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    {
      float x0[12];
      load_x(T - 1, x0);
#pragma unroll
      for (int d = 0; d < HD; d++)
        if (T - 2 - d >= 0) load_x(T - 2 - d, xr[d]);      // xr[d] holds x of the step BEHIND the one slot d processes
      stage_x((T - 1) % 3, x0);
    }
    __syncthreads();
    int hb = (T - 1) % 3;      // tile of the step being processed; the step behind it goes into (hb + 2) % 3
#pragma unroll 1
    for (int t = T - 1; t >= 0; t -= HD) {
#pragma unroll
      for (int d = 0; d < HD; d++) {
        const int tt = t - d;
        if (tt < 0) break;
        const int hb_next = hb == 0 ? 2 : hb - 1;
        if (tt - 1 >= 0) stage_x(hb_next, xr[d]);
        if (tt - 1 - HD >= 0) load_x(tt - 1 - HD, xr[d]);
        __syncthreads();
#pragma unroll
        for (int hf = 0; hf < HELP_PARTS; hf++)
          lbf_weight_grads_rc<NS, 3>(accH[hf], zt + (size_t)(tt & 1) * NS * LBF_RC_Z, hx + (size_t)hb * NS * LBF_RC_HX, 3 * MAIN_CI + 3 * hf, col, rq);
        hb = hb_next;
      }
    }
#pragma unroll
    for (int hf = 0; hf < HELP_PARTS; hf++) lbf_store_weight_grads<3>(accH[hf], a, 3 * MAIN_CI + 3 * hf, col, rq);
    return;
  }
  // ---- waves 0-2: units 16 w .. 16 w + 15 ----
  const int u = 16 * w + col;
  // recurrence / dx fragments over K = gate column (as in lstm_seq_bwd_bf16_kernel)
#ifndef IRRL_LBF_RC_BF_REGS
  constexpr bool BX_LDS = false;
#else
  constexpr bool BX_LDS = true;
#endif
  auto BxL = [&](int kc, int p) -> u16x8_t * { return (u16x8_t *)(bfl + (((size_t)p * 3 + w) * KC + kc) * 64 * 8 + (size_t)l * 8); };
  u16x8_t Bh[KC][NS], Bx[(NEED_DX && !BX_LDS) ? KC : 1][NS];
#pragma unroll
  for (int kc = 0; kc < KC; kc++) {
    u16x8_t fx[NS];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int cidx = 32 * kc + 8 * rq + i;
      unsigned short pl[NS];
      bf_split<NS>(a.wh_p[(size_t)u * GC + cidx], pl);
#pragma unroll
      for (int p = 0; p < NS; p++) Bh[kc][p][i] = pl[p];
      if (NEED_DX) {
        bf_split<NS>((u < n_in) ? a.wx_p[(size_t)u * GC + cidx] : 0.0f, pl);
#pragma unroll
        for (int p = 0; p < NS; p++) fx[p][i] = pl[p];
      }
    }
    if (NEED_DX) {
#pragma unroll
      for (int p = 0; p < NS; p++) {
        if (BX_LDS) *BxL(kc, p) = fx[p];
        else Bx[kc][p] = fx[p];
      }
    }
  }
  // the forward kernel's fragments over K = [h | x]: B[k = 32 kc + 8 rq + i][unit u, gate g].  They live in LDS, every lane's own 16 bytes per
  // (chunk, gate, plane) -- written once here, read back by the same lane each step (24 ds_read_b128 per lane and step): in registers (96) the
  // kernel sat at 490-512 with 240 register-file moves per step and ran 45-65 % longer than the kernel that loads its gates
  // (profiles/r06_ab_lstm_recompute_same_box.log)
  auto BfL = [&](int kc, int g, int p) -> u16x8_t * { return (u16x8_t *)(bfl + ((((size_t)p * 3 + w) * KF + kc) * 4 + g) * 64 * 8 + (size_t)l * 8); };
  u16x8_t Bf[BX_LDS ? KF : 1][4][NS];
#pragma unroll
  for (int kc = 0; kc < KF; kc++)
#pragma unroll
    for (int g = 0; g < 4; g++) {
      u16x8_t fr[NS];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int k = 32 * kc + 8 * rq + i;
        float v;
        if (k < HID) v = a.wh_p[((size_t)k * HID + u) * 4 + g];
        else v = (k - HID < n_in) ? a.wx_p[((size_t)(k - HID) * HID + u) * 4 + g] : 0.0f;
        unsigned short pl[NS];
        bf_split<NS>(v, pl);
#pragma unroll
        for (int p = 0; p < NS; p++) fr[p][i] = pl[p];
      }
#pragma unroll
      for (int p = 0; p < NS; p++) {
        if (BX_LDS) Bf[kc][g][p] = fr[p];
        else *BfL(kc, g, p) = fr[p];
      }
    }
  const f32x4 bias = *(const f32x4 *)&a.b_p[u * 4];
  float dbacc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float dc[4] = {0.0f, 0.0f, 0.0f, 0.0f}, dhrec[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 accW[6][MAIN_CI];
#pragma unroll
  for (int mt = 0; mt < 6; mt++)
#pragma unroll
    for (int ci = 0; ci < MAIN_CI; ci++) accW[mt][ci] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  struct StepOps { float ct[4], cp[4], dh[4], mk[4], hp[4]; };
  auto fetch = [&](int t, StepOps &o) {      // uniform row pointers + 32-bit lane indices (see lstm_seq_bwd_bf16_kernel)
    const size_t trow = (size_t)t * N;
    const float *mk_t = a.masks + trow, *ct_t = a.cseq + trow * HID, *dh_t = a.dh_in + trow * HID;
    const float *cp_t = t > 0 ? a.cseq + (trow - N) * HID : a.state0;
    const float *hp_t = t > 0 ? a.hseq + (trow - N) * HID : a.state0 + HID;
    const unsigned pstride = t > 0 ? (unsigned)HID : 2u * HID;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const unsigned e = (unsigned)(e0 + 4 * rq + j);
      o.mk[j] = mk_t[e];
      o.ct[j] = ct_t[e * HID + u];
      o.cp[j] = cp_t[e * pstride + u];
      o.hp[j] = hp_t[e * pstride + u];
      o.dh[j] = dh_t[e * HID + u];
    }
  };
  // (h_{t-1} keep_t)^T of the step whose operands are `o`: rows 0 .. 47 of tile hb, [unit u][env 4 rq .. 4 rq + 3]
  auto stage_h = [&](int hb, const StepOps &o) {
    u16x4_t pk[NS];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      unsigned short pl[NS];
      bf_split<NS>(o.hp[j] * (1.0f - o.mk[j]), pl);
#pragma unroll
      for (int p = 0; p < NS; p++) pk[p][j] = pl[p];
    }
#pragma unroll
    for (int p = 0; p < NS; p++) *(u16x4_t *)HX(hb, p, u, 4 * rq) = pk[p];
  };
  // z of one step: the forward kernel's products, same planes, same order (x chunks first, then small plane products first, then the gates), from
  // the COMPLETE h / x tile hb.  It runs BEHIND the barrier of the step before (where the tile becomes complete) -- in the stretch whose matrix-core
  // work the vector ALU otherwise only waits for -- and hands its gates to the next step in 16 registers (g4c): in front of the gate arithmetic, where
  // the first version had it, tile reads, 36 dependent-free MFMAs and 16 transcendental chains were all on the step's serial chain (update 106 ms
  // against 96 for the kernel that loads its gates; profiles/r06_ab_lstm_recompute_same_box.log)
  f32x4 g4c[4];
  auto recompute = [&](const int hb) {
    typedef short lbf_s16x4 __attribute__((ext_vector_type(4)));
    u16x8_t av[KF][NS];
#pragma unroll
    for (int kc = 0; kc < KF; kc++)
#pragma unroll
      for (int p = 0; p < NS; p++) {
        // lane 4 q + p4 of the 16-lane group rq addresses k-row 32 kc + 8 rq + 4 h + q, envs 4 p4 .. 4 p4 + 3; lane i receives env i of the four rows
        const unsigned short *base = HX(hb, p, 32 * kc + 8 * rq + (col >> 2), 4 * (col & 3));
        const lbf_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lbf_s16x4 __attribute__((address_space(3))) *)base);
        const lbf_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lbf_s16x4 __attribute__((address_space(3))) *)(base + 4 * LBF_CROW));
        av[kc][p] = __builtin_bit_cast(u16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));      // (a register pair next to a register pair: no moves)
      }
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; g++) acc[g] = (f32x4){bias[g], bias[g], bias[g], bias[g]};
#pragma unroll
    for (int kc = KF - 1; kc >= 0; kc--) {
      u16x8_t bf[4][NS];
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int p = 0; p < NS; p++) bf[g][p] = BX_LDS ? Bf[kc][g][p] : *BfL(kc, g, p);
#pragma unroll
      for (int q = 0; q < PR::N; q++)
#pragma unroll
        for (int g = 0; g < 4; g++) acc[g] = BF_MFMA32(av[kc][PR::A[q]], bf[g][PR::B[q]], acc[g]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) g4c[j] = (f32x4){fast_sigmoid(acc[0][j]), fast_sigmoid(acc[1][j]), fast_sigmoid(acc[2][j]), fast_tanh(acc[3][j])};
  };
  auto step = [&](int t, StepOps &o, const StepOps &onext, const int hb, const int hb_next) {      // hb = t % 3, hb_next = (t - 1) % 3: fixed per slot (DEPTH == 3 tiles)
    const int buf = t & 1;
    float keepC[4], ct[4], cpv[4], dhv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { keepC[j] = 1.0f - o.mk[j]; ct[j] = o.ct[j]; cpv[j] = o.cp[j]; dhv[j] = o.dh[j]; }
    if (t - DEPTH >= 0) fetch(t - DEPTH, o);
    // the h tile of the NEXT step (its operands arrived at least two steps ago)
    if (t - 1 >= 0) stage_h(hb_next, onext);
    f32x4 g4[4];
#pragma unroll
    for (int j = 0; j < 4; j++) g4[j] = g4c[j];
    // gate arithmetic -> dz (env 4 rq + j, unit u, gates i f o g) into the [env][gate column] tile
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
      const float dz4[4] = {d_i * ig * (1.0f - ig), d_f * fg * (1.0f - fg), d_o * og * (1.0f - og), d_g * (1.0f - gg * gg)};
      u16x4_t zr[NS];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        dbacc[g] += dz4[g];
        unsigned short pl[NS];
        bf_split<NS>(dz4[g], pl);
#pragma unroll
        for (int p = 0; p < NS; p++) zr[p][g] = pl[p];
      }
#pragma unroll
      for (int p = 0; p < NS; p++) *(u16x4_t *)Zr(buf, p, 4 * rq + j, 4 * u) = zr[p];
    }
    __syncthreads();      // dz_t and the h / x tiles of step t - 1 of every wave are visible
    u16x8_t az[KC][NS];
#pragma unroll
    for (int kc = 0; kc < KC; kc++)
#pragma unroll
      for (int p = 0; p < NS; p++) az[kc][p] = *(const u16x8_t *)Zr(buf, p, col, 32 * kc + 8 * rq);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}, accx = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    if (NEED_DX && BX_LDS) {
      // the recurrence first (its result starts the next step's chain), then dx with its fragments out of LDS; same products, same order per accumulator
#pragma unroll
      for (int q = 0; q < PR::N; q++)
#pragma unroll
        for (int kc = 0; kc < KC; kc++) acc = BF_MFMA32(az[kc][PR::A[q]], Bh[kc][PR::B[q]], acc);
      // (plane products in the loading kernel's order per accumulator: q outer, chunks inner -- the fragments of all six chunks are live for that)
      u16x8_t bx[KC][NS];
#pragma unroll
      for (int kc = 0; kc < KC; kc++)
#pragma unroll
        for (int p = 0; p < NS; p++) bx[kc][p] = *BxL(kc, p);
#pragma unroll
      for (int q = 0; q < PR::N; q++)
#pragma unroll
        for (int kc = 0; kc < KC; kc++) accx = BF_MFMA32(az[kc][PR::A[q]], bx[kc][PR::B[q]], accx);
    } else {
#pragma unroll
      for (int q = 0; q < PR::N; q++)
#pragma unroll
        for (int kc = 0; kc < KC; kc++) {
          acc = BF_MFMA32(az[kc][PR::A[q]], Bh[kc][PR::B[q]], acc);
          if (NEED_DX) accx = BF_MFMA32(az[kc][PR::A[q]], Bx[kc][PR::B[q]], accx);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) dhrec[j] = acc[j] * keepC[j];
    if (NEED_DX && u < n_in) {
      float *dx_t = a.dx + (size_t)t * N * n_in;      // uniform row pointer
#pragma unroll
      for (int j = 0; j < 4; j++) dx_t[(unsigned)(e0 + 4 * rq + j) * (unsigned)n_in + u] = accx[j];
    }
    lbf_weight_grads_rc<NS, MAIN_CI>(accW, zt + (size_t)buf * NS * LBF_RC_Z, hx + (size_t)hb * NS * LBF_RC_HX, MAIN_CI * w, col, rq);
    if (t - 1 >= 0) recompute(hb_next);      // the gates of step t - 1 (its tile is complete since this step's barrier)
  };
  StepOps ops[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; d++)
    if (T - 1 - d >= 0) fetch(T - 1 - d, ops[d]);
  static_assert(DEPTH == 3, "slot d of a group always works on h / x tile (hb0 - d) mod 3: the groups advance by as many steps as there are tiles");
  const int hb0 = (T - 1) % 3;
  stage_h(hb0, ops[0]);
  __syncthreads();
  recompute(hb0);
#pragma unroll 1
  for (int t = T - 1; t >= 0; t -= DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
      if (t - d >= 0) step(t - d, ops[d], ops[(d + 1) % DEPTH], (hb0 + 3 - d) % 3, (hb0 + 5 - d) % 3);
  }
  lbf_store_weight_grads<MAIN_CI>(accW, a, MAIN_CI * w, col, rq);
  const size_t blk = blockIdx.x;
#pragma unroll
  for (int g = 0; g < 4; g++) a.db_part[(blk * 4 + rq) * GC + u * 4 + g] = dbacc[g];
}
