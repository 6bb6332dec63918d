// mlp_bf16.hpp -- the MlpPolicy gradient kernels of mlp_update.hpp (archi/policies.py:430-446 under the PPO2 loss, ppo2.py:152-175) on
// the bf16 matrix cores with COMPENSATED OPERAND SPLITS, f32 accumulation: same arguments, same partial-sum rows, same loss arithmetic.
//
// Why a second kernel and not a switch in the first: the f32 kernel is MFMA-issue bound (324 v_mfma_f32_16x16x4_f32 per 16-sample tile =
// 4.5 us at 13.9 ns each, MFMA-busy 0.60); the two-plane split x = p0 + p1 (lstm_bf16.hpp) turns every product into three plane products
// on v_mfma_f32_16x16x32_bf16 / _16x16x16_bf16 (7.2 ns for 32 resp. 16 units of K): 186 MFMAs = 1.4 us per tile.  Applied mechanically
// to the f32 kernel that bought nothing (profiles/r04_lstm_precision_error_and_time.log: the planes of 196 weight registers plus the
// split arithmetic spilled).  This kernel is built around the planes:
//   * the weight planes live in LDS, one lane-linear 1 KB block per (layer, output tile, K chunk, plane), written once per workgroup and
//     fetched with ds_read_b128 ONE STAGE AHEAD of the MFMAs that use them (a layer's 16 blocks = 64 registers in flight instead of the
//     f32 kernel's 196 resident weight registers);
//   * everything is still computed transposed (sample = C/D column), and the C layout of a 16x16 f32 tile -- lane (c, g) holds rows
//     4 g .. 4 g + 3 -- IS the K = 16 B-operand layout, two tiles side by side the K = 32 one: a layer's output is split to bf16 where
//     it sits and becomes the next layer's operand without moving (the weight blocks carry the matching permuted k order);
//   * the weight gradients contract over the SAMPLES: each tensor is written once, as planes, into a per-wave [sample][feature] image
//     and read back TRANSPOSED by ds_read_b64_tr_b16 (gfx950) straight into the A / B operand layout of v_mfma_f32_16x16x16_bf16.
// A workgroup is four independent waves, one per SIMD; a wave walks 16-sample tiles.  883 VALU + 186 MFMA + 147 LDS instructions per tile
// (policy network, PMC); the VALU parts of the activations / deltas are zipped by hand with the MFMAs that do not depend on them (below).
// Gradients within 7e-6 of each tensor's largest entry of float64 autograd (tools/mlp_grad_error.py).  One MFMA shape per accumulator
// chain: K = 16 and K = 32 instructions alternating on one accumulator are miscompiled (tools/microbench/mfma_mixed_chain.hip).
#pragma once
#include "mlp_update.hpp"
#include "lstm_bf16.hpp"

typedef short mb_s16x4 __attribute__((ext_vector_type(4)));

constexpr int MB_RS = 72;                      // row stride of a [16 samples][<= 64 features] image, bf16 elements (144 bytes)
constexpr int MB_IMG = 16 * MB_RS;             // one plane of one image
constexpr int MB_WAVE = 4 * 2 * MB_IMG;        // per wave: X | H1 | H2 | D (the deltas of the layer at hand), two planes each
// weight blocks (elements per plane): W1 chunk 0 (K = 32: observations 0..31) | W1 chunk 1 (K = 32: observations 32..63, 32..34 exist) | W2 | W3 | W3^T | W2^T
constexpr int MB_W1A = 0, MB_W1B = MB_W1A + 4 * 512, MB_W2 = MB_W1B + 4 * 512, MB_W3 = MB_W2 + 8 * 512, MB_W3T = MB_W3 + 2 * 512, MB_W2T = MB_W3T + 4 * 256;
constexpr int MB_WPLANE = MB_W2T + 8 * 512;    // 14 336 elements = 28 KB per plane
constexpr int MB_LDS_ELEMS = 2 * MB_WPLANE + 4 * MB_WAVE;
constexpr int mlp_bf16_lds_bytes() { return MB_LDS_ELEMS * 2 + (2 * IRRL_MLP_H + 16) * 4; }
static_assert(4 * MB_WAVE * 2 >= IRRL_MLP_P * 4, "the block reduction reuses the waves' image space");

// feature index of element j (0..7) of lane group g in K chunk m of a 64-feature operand that sits in C layout (two 16-row tiles side by side)
LSTM_DEV int mb_feat(int j, int g, int m) { return 16 * (2 * m + (j >> 2)) + 4 * g + (j & 3); }

#ifdef IRRL_MB_PROFILE   /* diagnostic build (tools/build_variants.py mbprof=-DIRRL_MB_PROFILE, tools/mlp_bf16_phases.py): where a tile goes, per wave, in 100 MHz ticks; the d logstd slots of the partial rows carry the sums (wrong gradients there) */
#define MB_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = wall_clock64(); ph_[i] += (float)(n_ - pts_); pts_ = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MB_STAMP(i) do { } while (0)
#endif

template <int KIND, bool REC = false>      // REC: the samples come as packed 256-byte records (a.rec, mlp_update.hpp IRRL_MLP_REC) instead of five arrays
__global__ void __launch_bounds__(256)
irrl_mlp_ppo_bf16_kernel(const MlpUpdateArgs a) {
  constexpr int OB = IRRL_MLP_OB, H = IRRL_MLP_H, OUT = KIND == 0 ? 12 : 1;
  using PR = BfProducts<2>;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds_m[];
  unsigned short *wimg = lds_m;                                   // [plane][block]
  float *bias = (float *)(lds_m + MB_LDS_ELEMS);                  // b1 | b2 | b3
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, c = l & 15, g = l >> 4;
  unsigned short *img = lds_m + 2 * MB_WPLANE + wv * MB_WAVE;     // this wave's images
  auto IX = [&](int p) { return img + (0 * 2 + p) * MB_IMG; };
  auto IH1 = [&](int p) { return img + (1 * 2 + p) * MB_IMG; };
  auto IH2 = [&](int p) { return img + (2 * 2 + p) * MB_IMG; };
  auto ID = [&](int p) { return img + (3 * 2 + p) * MB_IMG; };

  // ---- weight planes -> LDS (once per workgroup); block element (lane, j) = the A-operand value of that lane ----
  {
    auto put = [&](int off, float v) {
      unsigned short pl[2];
      bf_split<2>(v, pl);
      wimg[off] = pl[0];
      wimg[MB_WPLANE + off] = pl[1];
    };
    for (int i = threadIdx.x; i < MB_WPLANE; i += 256) {
      float v = 0.0f;
      if (i < MB_W1B) {                     // W1 chunk 0: A[out 16 nt + c'][k = 8 g' + j]
        const int nt = i >> 9, ll = (i >> 3) & 63, j = i & 7, cc = ll & 15, gg = ll >> 4;
        v = a.w1[(8 * gg + j) * H + 16 * nt + cc];
      } else if (i < MB_W2) {               // W1 chunk 1: k = 32 + 8 g' + j (a K = 32 block like the others: chains of MIXED MFMA shapes on one
                                            // accumulator -- 16x16x16 between 16x16x32 -- lost part of the shorter instruction's result)
        const int e = i - MB_W1B, nt = e >> 9, ll = (e >> 3) & 63, j = e & 7, cc = ll & 15, gg = ll >> 4, k = 32 + 8 * gg + j;
        v = (k < OB) ? a.w1[k * H + 16 * nt + cc] : 0.0f;
      } else if (i < MB_W3) {               // W2: A[out 16 n2 + c'][k = mb_feat(j, g', m)], block = 2 n2 + m
        const int e = i - MB_W2, blk = e >> 9, ll = (e >> 3) & 63, j = e & 7, cc = ll & 15, gg = ll >> 4;
        v = a.w2[mb_feat(j, gg, blk & 1) * H + 16 * (blk >> 1) + cc];
      } else if (i < MB_W3T) {              // W3 (policy head): A[output c'][k = mb_feat(j, g', m)], block = m
        const int e = i - MB_W3, m = e >> 9, ll = (e >> 3) & 63, j = e & 7, cc = ll & 15, gg = ll >> 4;
        v = (KIND == 0 && cc < OUT) ? a.w3[mb_feat(j, gg, m) * OUT + cc] : 0.0f;
      } else if (i < MB_W2T) {              // W3^T: A[h2 feature 16 kt + c'][k = output 4 g' + j]
        const int e = i - MB_W3T, kt = e >> 8, ll = (e >> 2) & 63, j = e & 3, cc = ll & 15, gg = ll >> 4;
        v = (KIND == 0 && 4 * gg + j < OUT) ? a.w3[(16 * kt + cc) * OUT + 4 * gg + j] : 0.0f;
      } else {                              // W2^T: A[h1 feature 16 kt + c'][k = mb_feat(j, g', m)], block = 2 kt + m
        const int e = i - MB_W2T, blk = e >> 9, ll = (e >> 3) & 63, j = e & 7, cc = ll & 15, gg = ll >> 4;
        v = a.w2[(16 * (blk >> 1) + cc) * H + mb_feat(j, gg, blk & 1)];
      }
      put(i, v);
    }
  }
  if (threadIdx.x < H) { bias[threadIdx.x] = a.b1[threadIdx.x]; bias[H + threadIdx.x] = a.b2[threadIdx.x]; }
  if (threadIdx.x < 16) bias[2 * H + threadIdx.x] = (threadIdx.x < OUT) ? a.b3[threadIdx.x] : 0.0f;
  for (int i = l; i < 8 * MB_IMG; i += 64) img[i] = 0;           // (columns 35..47 of the observation image stay zero)
  __syncthreads();
  auto W8 = [&](int p, int off) -> u16x8_t { return *(const u16x8_t *)(wimg + p * MB_WPLANE + off + 8 * l); };
  auto W4 = [&](int p, int off) -> u16x4_t { return *(const u16x4_t *)(wimg + p * MB_WPLANE + off + 4 * l); };
  // value head: one column -- lane-local f32 arithmetic, as in the f32 kernel
  float wa3[4][4];
#pragma unroll
  for (int n2 = 0; n2 < 4; n2++)
#pragma unroll
    for (int r = 0; r < 4; r++) wa3[n2][r] = (KIND == 1) ? a.w3[16 * n2 + 4 * g + r] : 0.0f;

  float sd_inv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ls_sum = 0.0f;
  if (KIND == 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) if (4 * g + r < OUT) sd_inv[r] = __expf(-a.logstd[4 * g + r]);
    for (int i = 0; i < OUT; i++) ls_sum += a.logstd[i];
  }
  const float a_mean = a.adv_stats[0], a_istd = 1.0f / (a.adv_stats[1] + 1e-8f);
  const float clip = a.cliprange;

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

  // split of a C-layout tile (rows 4 g .. 4 g + 3 of this lane's sample) into its two planes
  auto split4 = [&](const f32x4 v, u16x4_t (&p)[2]) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      unsigned short pl[2];
      bf_split<2>(v[r], pl);
      p[0][r] = pl[0]; p[1][r] = pl[1];
    }
  };
  auto pair8 = [&](const u16x4_t lo, const u16x4_t hi) -> u16x8_t { return (u16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; };
  // transposed read of feature tile `ft` of a [sample][feature] image: element e = [sample 4 g + e][feature 16 ft + c]
  const int tr_off = (4 * g + (c >> 2)) * MB_RS + 4 * (c & 3);
  auto TR = [&](const unsigned short *image, int ft) -> u16x4_t {
    const mb_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mb_s16x4 __attribute__((address_space(3))) *)(image + tr_off + 16 * ft));
    return __builtin_bit_cast(u16x4_t, v);
  };

  // weight operands in flight: 16 K = 32 blocks (one layer's worth) + the head's transposed blocks; requested one stage ahead
  u16x8_t wr[16];
  u16x4_t w3t[8];
  auto load_w1 = [&]() {
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
      for (int p = 0; p < 2; p++) { wr[4 * nt + p] = W8(p, MB_W1A + 512 * nt); wr[4 * nt + 2 + p] = W8(p, MB_W1B + 512 * nt); }
  };
  auto load_w2 = [&]() {
#pragma unroll
    for (int b = 0; b < 8; b++)
#pragma unroll
      for (int p = 0; p < 2; p++) wr[2 * b + p] = W8(p, MB_W2 + 512 * b);
  };
  auto load_w2t = [&]() {
#pragma unroll
    for (int b = 0; b < 8; b++)
#pragma unroll
      for (int p = 0; p < 2; p++) wr[2 * b + p] = W8(p, MB_W2T + 512 * b);
  };
  auto load_head = [&]() {
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int p = 0; p < 2; p++) wr[2 * m + p] = W8(p, MB_W3 + 512 * m);
#pragma unroll
    for (int kt = 0; kt < 4; kt++)
#pragma unroll
      for (int p = 0; p < 2; p++) w3t[2 * kt + p] = W4(p, MB_W3T + 256 * kt);
  };
  struct TileIn { float x[8], xt[3]; f32x4 act; float ret, ov, onlp; };
  const size_t ntiles = (a.n + 15) / 16, stride = (size_t)gridDim.x * 4;
  size_t tile = (size_t)blockIdx.x * 4 + wv;
  auto row_of = [&](size_t t) -> size_t {
    size_t j = t * 16 + c;
    if (j >= a.n) j = a.n - 1;
    return a.idx ? (size_t)a.idx[j] : j;
  };
  auto load_tile = [&](size_t row, TileIn &in) {
    if (REC) {
      // two 128-byte lines per sample, 16-byte loads: observation words 8 g .. 8 g + 7, the tail 32 .. 34 (+ the zero at 35), the action
      // quad, and (return, old value, old neglogp, advantage) as one vector -- the same values the five arrays hold, so the results are bit-identical
      const float *r = a.rec + row * IRRL_MLP_REC;
      const f32x4 x0 = *(const f32x4 *)(r + 8 * g), x1 = *(const f32x4 *)(r + 8 * g + 4), xt = *(const f32x4 *)(r + 32);
      in.x[0] = x0[0]; in.x[1] = x0[1]; in.x[2] = x0[2]; in.x[3] = x0[3]; in.x[4] = x1[0]; in.x[5] = x1[1]; in.x[6] = x1[2]; in.x[7] = x1[3];
      in.xt[0] = xt[0]; in.xt[1] = xt[1]; in.xt[2] = xt[2];
      if (KIND == 0) in.act = *(const f32x4 *)(r + 36 + 4 * (g < 3 ? g : 2));
      const f32x4 s4 = *(const f32x4 *)(r + 48);
      in.ret = s4[0]; in.ov = s4[1];
      if (KIND == 0) in.onlp = s4[2];
      return;
    }
    const float *xr = a.obs + row * OB;
#pragma unroll
    for (int j = 0; j < 8; j++) in.x[j] = xr[8 * g + j];
#pragma unroll
    for (int j = 0; j < 3; j++) in.xt[j] = xr[32 + j];
    if (KIND == 0) {
      in.act = *(const f32x4 *)(a.actions + row * 12 + 4 * (g < 3 ? g : 2));
      in.onlp = a.old_neglogp[row];
    }
    in.ret = a.returns[row];
    in.ov = a.old_values[row];
  };
  TileIn cur, nxt;
  size_t row_next = 0;
  if (tile < ntiles) {
    load_tile(row_of(tile), cur);
    row_next = row_of(tile + stride < ntiles ? tile + stride : tile);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see the f32 kernel
  load_w1();
  // A wave issues one MFMA per 16 clocks and, in a burst of MFMAs, nothing else; in a burst of VALU work the matrix core idles
  // (PMC of the first version: VALU 35 %, MFMA 26 %, LDS 9 %, waiting 20 % of a tile -- one after the other).  So the two are ZIPPED by
  // hand where independent work exists: an activation / delta element is three small VALU parts (MB_PIN keeps the order), and every
  // part carries one MFMA of (a) the next output tile's chain of the same layer, (b) the weight-gradient products of the layer
  // behind -- those of layer 1 are carried over the loop's back edge (their operands wait in registers) and run under the next tile's
  // first layer.
#define MB_PIN() __builtin_amdgcn_sched_barrier(0)
  auto act_a = [&](float x) -> float { return __expf(-2.0f * x); };
  auto act_b = [&](float e) -> float { return 2.0f * __builtin_amdgcn_rcpf(1.0f + e) - 1.0f; };     // == fast_tanh
  u16x4_t pat[3][2], pbz[4][2];     // operands of the PENDING d W1 products (zero in front of the first tile)
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int p2 = 0; p2 < 2; p2++) { pbz[i][p2] = (u16x4_t){0, 0, 0, 0}; if (i < 3) pat[i][p2] = (u16x4_t){0, 0, 0, 0}; }
  auto dw1_mfma = [&](int i) {      // i = 0 .. 35
    const int nt = i / 9, kt = (i / 3) % 3, q = i % 3;
    gw1[kt][nt] = BF_MFMA16(pat[kt][PR::A[q]], pbz[nt][PR::B[q]], gw1[kt][nt]);
  };
#ifdef IRRL_MB_PROFILE
  float ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long pts_ = wall_clock64();
#endif
  for (; tile < ntiles; tile += stride) {
    const bool more = tile + stride < ntiles;
    // the next tile's rows, requested unconditionally (behind the last tile: a valid row once more): a load under a branch costs the
    // register copies of its join, and those wait for the load right where it was issued
    load_tile(row_next, nxt);
    row_next = row_of(tile + 2 * stride < ntiles ? tile + 2 * stride : tile);
    const bool valid = tile * 16 + c < a.n;

    // ---- observations: planes as the B operand (k = 8 g + j; 32 + 8 g + j) and into the image for dW1 ----
    u16x8_t xp[2], xq[2];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      unsigned short pl[2];
      bf_split<2>(cur.x[j], pl);
      xp[0][j] = pl[0]; xp[1][j] = pl[1];
      if (j < 6) { dw1_mfma(j); MB_PIN(); }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      unsigned short pl[2] = {0, 0};
      if (j < 3) bf_split<2>((g == 0) ? cur.xt[j] : 0.0f, pl);
      xq[0][j] = pl[0]; xq[1][j] = pl[1];
    }
#pragma unroll
    for (int p = 0; p < 2; p++) {
      *(u16x8_t *)(IX(p) + c * MB_RS + 8 * g) = xp[p];
      if (g == 0) *(u16x4_t *)(IX(p) + c * MB_RS + 32) = (u16x4_t){xq[p][0], xq[p][1], xq[p][2], xq[p][3]};
    }
    // ---- layer 1 ----
    f32x4 h1[4], h2[4];
    u16x4_t h1p[4][2], h2p[4][2];
    {
      f32x4 acc[4];
      auto chain1 = [&](int nt, int j) {      // MFMA j (0..5) of output tile nt
        if (j == 0) acc[nt] = *(const f32x4 *)&bias[16 * nt + 4 * g];
        const int q = j >> 1;
        if (j & 1) acc[nt] = BF_MFMA32(wr[4 * nt + PR::A[q]], xp[PR::B[q]], acc[nt]);
        else acc[nt] = BF_MFMA32(wr[4 * nt + 2 + PR::A[q]], xq[PR::B[q]], acc[nt]);
      };
#pragma unroll
      for (int j = 0; j < 6; j++) chain1(0, j);
      MB_PIN();
#pragma unroll
      for (int st = 0; st < 4; st++) {
        if (st == 3) { load_w2(); MB_PIN(); }       // every chain of this layer is out: its weight registers take the next layer's
        float e_[4];
#pragma unroll
        for (int j = 0; j < 12; j++) {
          const int r = j / 3, part = j % 3;
          if (st < 3 && j < 6) chain1(st + 1, j);
          else dw1_mfma(st < 3 ? 6 + 6 * st + (j - 6) : 24 + j);
          if (part == 0) e_[r] = act_a(acc[st][r]);
          else if (part == 1) h1[st][r] = act_b(e_[r]);
          else { unsigned short pl[2]; bf_split<2>(h1[st][r], pl); h1p[st][0][r] = pl[0]; h1p[st][1][r] = pl[1]; }
          MB_PIN();
        }
#pragma unroll
        for (int p = 0; p < 2; p++) *(u16x4_t *)(IH1(p) + c * MB_RS + 16 * st + 4 * g) = h1p[st][p];
      }
    }
    MB_STAMP(0);   // observations + layer 1
    // ---- layer 2 ----
    {
      u16x8_t b[2][2];
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int p = 0; p < 2; p++) b[m][p] = pair8(h1p[2 * m][p], h1p[2 * m + 1][p]);
      f32x4 acc[4];
      auto chain2 = [&](int n2, int j) {
        if (j == 0) acc[n2] = *(const f32x4 *)&bias[H + 16 * n2 + 4 * g];
        const int m = j / 3, q = j % 3;
        acc[n2] = BF_MFMA32(wr[4 * n2 + 2 * m + PR::A[q]], b[m][PR::B[q]], acc[n2]);
      };
#pragma unroll
      for (int j = 0; j < 6; j++) chain2(0, j);
      MB_PIN();
#pragma unroll
      for (int st = 0; st < 4; st++) {
        if (st == 3) { if (KIND == 0) load_head(); else load_w2t(); MB_PIN(); }
        float e_[4];
#pragma unroll
        for (int j = 0; j < 12; j++) {
          const int r = j / 3, part = j % 3;
          if (st < 3 && j < 6) chain2(st + 1, j);
          if (part == 0) e_[r] = act_a(acc[st][r]);
          else if (part == 1) h2[st][r] = act_b(e_[r]);
          else { unsigned short pl[2]; bf_split<2>(h2[st][r], pl); h2p[st][0][r] = pl[0]; h2p[st][1][r] = pl[1]; }
          MB_PIN();
        }
#pragma unroll
        for (int p = 0; p < 2; p++) *(u16x4_t *)(IH2(p) + c * MB_RS + 16 * st + 4 * g) = h2p[st][p];
      }
    }
    MB_STAMP(1);   // layer 2
    // ---- head ----
    f32x4 out;
    if (KIND == 0) {
      out = *(const f32x4 *)&bias[2 * H + 4 * g];
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int q = 0; q < PR::N; q++)
          out = BF_MFMA32(wr[2 * m + PR::A[q]], pair8(h2p[2 * m][PR::B[q]], h2p[2 * m + 1][PR::B[q]]), out);
    } else {
      f32x4 pv = h2[0] * *(const f32x4 *)wa3[0];
#pragma unroll
      for (int n2 = 1; n2 < 4; n2++) pv += h2[n2] * *(const f32x4 *)wa3[n2];
      float v = (pv[0] + pv[1]) + (pv[2] + pv[3]);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      out = zero4;
      out[0] = v + bias[2 * H];
    }

    // ---- loss and d loss / d out (the arithmetic of the f32 kernel, line for line) ----
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
      dz3[0] = valid ? a.inv_n * a.vf_coef * dvf : 0.0f;
      if (valid && g == 0) sc[0] += 0.5f * fmaxf(l1, l2);
    }
    if (KIND == 0 || g == 0) gb3 += dz3;

    MB_STAMP(2);   // head + loss
    // ---- backward: head.  In every layer the chain that feeds the next delta (d h = W dz, operands in registers) goes out FIRST; the
    // weight-gradient products (operands back from the images, transposed) are zipped with the VALU parts of that delta ----
    f32x4 d[4];
    u16x4_t z2[4][2], z1[4][2];
    if (KIND == 0) {
      u16x4_t z3[2];
      split4(dz3, z3);
#pragma unroll
      for (int p = 0; p < 2; p++) *(u16x4_t *)(ID(p) + c * MB_RS + 4 * g) = z3[p];
#pragma unroll
      for (int kt = 0; kt < 4; kt++) {
        d[kt] = zero4;
#pragma unroll
        for (int q = 0; q < PR::N; q++) d[kt] = BF_MFMA16(w3t[2 * kt + PR::A[q]], z3[PR::B[q]], d[kt]);
      }
      MU_WAVE_SYNC();
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; kt++) {
        d[kt] = dz3[0] * *(const f32x4 *)wa3[kt];
        gw3[kt] += dz3[0] * h2[kt];
      }
    }
    {
      u16x4_t bz[2], at[4][2];
      if (KIND == 0) {
        bz[0] = TR(ID(0), 0); bz[1] = TR(ID(1), 0);
#pragma unroll
        for (int kt = 0; kt < 4; kt++) { at[kt][0] = TR(IH2(0), kt); at[kt][1] = TR(IH2(1), kt); }
        MB_PIN();
        load_w2t();
        MB_PIN();
      }
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int kt = e >> 2, r = e & 3;
        float t = 0.0f;
#pragma unroll
        for (int part = 0; part < 3; part++) {
          if (KIND == 0 && part == 0 && e < 12) gw3[e / 3] = BF_MFMA16(at[e / 3][PR::A[e % 3]], bz[PR::B[e % 3]], gw3[e / 3]);
          if (part == 0) t = d[kt][r] * (1.0f - h2[kt][r] * h2[kt][r]);
          else if (part == 1) gb2[kt][r] += t;
          else { unsigned short pl[2]; bf_split<2>(t, pl); z2[kt][0][r] = pl[0]; z2[kt][1][r] = pl[1]; }
          MB_PIN();
        }
      }
    }
    MB_STAMP(3);   // d W3, d h2, d z2
    // ---- layer 2 ----
    MU_WAVE_SYNC();      // the head's reads of the delta image are done
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
      for (int p = 0; p < 2; p++) *(u16x4_t *)(ID(p) + c * MB_RS + 16 * nt + 4 * g) = z2[nt][p];
    {
      u16x8_t b[2][2];
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int p = 0; p < 2; p++) b[m][p] = pair8(z2[2 * m][p], z2[2 * m + 1][p]);
#pragma unroll
      for (int kt = 0; kt < 4; kt++) {
        d[kt] = zero4;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
          for (int q = 0; q < PR::N; q++) d[kt] = BF_MFMA32(wr[4 * kt + 2 * m + PR::A[q]], b[m][PR::B[q]], d[kt]);
      }
    }
    MU_WAVE_SYNC();
    {
      u16x4_t at[4][2], bz[4][2];
#pragma unroll
      for (int kt = 0; kt < 4; kt++) { at[kt][0] = TR(IH1(0), kt); at[kt][1] = TR(IH1(1), kt); bz[kt][0] = TR(ID(0), kt); bz[kt][1] = TR(ID(1), kt); }
      MB_PIN();
      load_w1();          // the next tile's first layer
      MB_PIN();
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int kt = e >> 2, r = e & 3;
        float t = 0.0f;
#pragma unroll
        for (int part = 0; part < 3; part++) {
          const int i = 3 * e + part, nt = i / 12, k2 = (i / 3) % 4, q = i % 3;
          gw2[k2][nt] = BF_MFMA16(at[k2][PR::A[q]], bz[nt][PR::B[q]], gw2[k2][nt]);
          if (part == 0) t = d[kt][r] * (1.0f - h1[kt][r] * h1[kt][r]);
          else if (part == 1) gb1[kt][r] += t;
          else { unsigned short pl[2]; bf_split<2>(t, pl); z1[kt][0][r] = pl[0]; z1[kt][1][r] = pl[1]; }
          MB_PIN();
        }
      }
    }
    MB_STAMP(4);   // d W2, d h1, d z1
    // ---- layer 1: the delta goes into the image, the operands of d W1 come back transposed and WAIT: the products run under the next tile ----
    MU_WAVE_SYNC();
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
      for (int p = 0; p < 2; p++) *(u16x4_t *)(ID(p) + c * MB_RS + 16 * nt + 4 * g) = z1[nt][p];
    MU_WAVE_SYNC();
#pragma unroll
    for (int kt = 0; kt < 3; kt++) { pat[kt][0] = TR(IX(0), kt); pat[kt][1] = TR(IX(1), kt); }
#pragma unroll
    for (int nt = 0; nt < 4; nt++) { pbz[nt][0] = TR(ID(0), nt); pbz[nt][1] = TR(ID(1), nt); }
    MU_WAVE_SYNC();
    cur = nxt;
    (void)more;
    MB_STAMP(5);   // d W1
  }
#pragma unroll
  for (int i = 0; i < 36; i++) dw1_mfma(i);      // the last tile's d W1
#ifdef IRRL_MB_PROFILE
  gls = (f32x4){ph_[0], ph_[1], ph_[2], ph_[3]};
  if (g == 1) gls = (f32x4){ph_[4], ph_[5], 0.0f, 0.0f};
  if (g > 1 || c != 0) gls = zero4;
#endif

  // ---- reduction (the f32 kernel's, same partial-sum row) ----
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
  __syncthreads();
  float *red = (float *)(lds_m + 2 * MB_WPLANE);
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
          put(IRRL_MLP_P_DW3 + (16 * nt + 4 * g + r) * 16 + c, (KIND == 0 || c == 0) ? gw3[nt][r] : 0.0f);
        }
      }
    }
    __syncthreads();
  }
  float *dst = a.partials + (size_t)blockIdx.x * IRRL_MLP_P;
  for (int i = threadIdx.x; i < IRRL_MLP_P; i += 256) dst[i] = red[i];
}
