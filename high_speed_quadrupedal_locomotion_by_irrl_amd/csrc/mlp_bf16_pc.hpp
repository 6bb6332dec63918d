// mlp_bf16_pc.hpp -- the MlpPolicy gradient kernel of mlp_bf16.hpp with TWO WAVES PER SIMD (round 5): the same arithmetic on the same values in
// the same order -- partial rows bit-identical to irrl_mlp_ppo_bf16_kernel's -- split over two wave roles.
//
// Why: mlp_bf16.hpp's wave carries everything of a 16-sample tile -- 883 VALU, 186 MFMA, 147 LDS instructions -- on 401 registers, so one wave
// per SIMD, and one wave does not overlap its own MFMAs, VALU work and LDS latencies to any useful degree (PMC: VALU active 0.37, MFMA-busy 0.26,
// waiting 0.19: one after the other; zipping them by hand bought 3.5 %).  A second wave per SIMD needs <= 256 registers per wave, and the
// weight-gradient accumulators alone are 128.  Here a workgroup is four PAIRS of waves (wave w and wave w + 4 share SIMD w):
//   * the PRODUCER (waves 0-3) walks its tiles as before -- forward, loss, the delta chains -- but forms no weight gradient: it leaves every
//     tensor the weight gradients contract ([sample][feature] bf16 planes of x, h1, h2, dz3, dz2, dz1) in the pair's LDS images;
//   * the CONSUMER (waves 4-7) reads the images of tile i TRANSPOSED (ds_read_b64_tr_b16) into registers -- all of a tile's operands, 80
//     registers -- and runs the tile's 96 weight-gradient MFMAs on its 128 accumulator registers while the producer is already in tile i + 1.
// Two workgroup barriers per tile: A = "the images of tile i are complete", B = "the consumer holds tile i's operands in registers"; the producer
// passes B only after the next tile's first layer, which touches no image, so it does not wait for the consumer's reads.
// LDS: the weight planes (56 KB) + four pairs' images, single-buffered and no wider than their tensors (x: 48 + 8 columns, dz3: 16 + 8; three
// separate delta images instead of one rewritten twice): 4 x 23 KB -- 149 KB of the CU's 160.
#pragma once
#include "mlp_bf16.hpp"

constexpr int PC_RSX = 56, PC_RS3 = 24;                                 // row strides (bf16 elements) of the x and dz3 images; the others: MB_RS
constexpr int PC_OX = 0, PC_OH1 = PC_OX + 16 * PC_RSX, PC_OH2 = PC_OH1 + 16 * MB_RS, PC_OD3 = PC_OH2 + 16 * MB_RS, PC_OD2 = PC_OD3 + 16 * PC_RS3,
              PC_OD1 = PC_OD2 + 16 * MB_RS;
constexpr int PC_PLANE = PC_OD1 + 16 * MB_RS;                           // one plane of one pair's images: 5888 elements
constexpr int PC_PAIR = 2 * PC_PLANE;
constexpr int PC_LDS_ELEMS = 2 * MB_WPLANE + 4 * PC_PAIR;
constexpr int mlp_bf16_pc_lds_bytes() { return PC_LDS_ELEMS * 2 + (2 * IRRL_MLP_H + 16) * 4; }
static_assert(4 * PC_PAIR * 2 >= IRRL_MLP_P * 4, "the block reduction reuses the pairs' image space");
static_assert(mlp_bf16_pc_lds_bytes() <= 160 * 1024, "one workgroup per CU: the CU's whole LDS");

// every LDS operation of this wave has completed, then the workgroup's barrier (no vmcnt wait: the next tile's rows stay in flight)
#ifndef IRRL_PC_AB
#define IRRL_PC_AB 0      /* diagnostics (WRONG results), tools/build_variants.py: 1 no transcendentals, 2 no splits of the activations, 3 weight operands fetched once, 4 no image writes, 5 no barriers, 6 one tile's rows for every tile, 7 no consumer products */
#endif
#if IRRL_PC_AB == 5
#define PC_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define PC_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

template <int KIND, bool REC = false>
__global__ void __launch_bounds__(512)
irrl_mlp_ppo_bf16_pc_kernel(const MlpUpdateArgs a) {
  constexpr int OB = IRRL_MLP_OB, H = IRRL_MLP_H, OUT = KIND == 0 ? 12 : 1;
  using PR = BfProducts<2>;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds_m[];
  unsigned short *wimg = lds_m;                                   // [plane][block], as in mlp_bf16.hpp
  float *bias = (float *)(lds_m + PC_LDS_ELEMS);                  // b1 | b2 | b3
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, c = l & 15, g = l >> 4;
  const int pw = wv & 3;
  const bool producer = wv < 4;
  unsigned short *img = lds_m + 2 * MB_WPLANE + pw * PC_PAIR;     // this pair's images
  auto IM = [&](int off, int p) { return img + p * PC_PLANE + off; };

  // ---- weight planes -> LDS (once per workgroup): mlp_bf16.hpp's blocks ----
  {
    auto put = [&](int off, float v) {
      unsigned short pl[2];
      bf_split<2>(v, pl);
      wimg[off] = pl[0];
      wimg[MB_WPLANE + off] = pl[1];
    };
    for (int i = threadIdx.x; i < MB_WPLANE; i += 512) {
      float v = 0.0f;
      if (i < MB_W1B) {
        const int nt = i >> 9, ll = (i >> 3) & 63, j = i & 7, cc = ll & 15, gg = ll >> 4;
        v = a.w1[(8 * gg + j) * H + 16 * nt + cc];
      } else if (i < MB_W2) {
        const int e = i - MB_W1B, nt = e >> 9, ll = (e >> 3) & 63, j = e & 7, cc = ll & 15, gg = ll >> 4, k = 32 + 8 * gg + j;
        v = (k < OB) ? a.w1[k * H + 16 * nt + cc] : 0.0f;
      } else if (i < MB_W3) {
        const int e = i - MB_W2, blk = e >> 9, ll = (e >> 3) & 63, j = e & 7, cc = ll & 15, gg = ll >> 4;
        v = a.w2[mb_feat(j, gg, blk & 1) * H + 16 * (blk >> 1) + cc];
      } else if (i < MB_W3T) {
        const int e = i - MB_W3, m = e >> 9, ll = (e >> 3) & 63, j = e & 7, cc = ll & 15, gg = ll >> 4;
        v = (KIND == 0 && cc < OUT) ? a.w3[mb_feat(j, gg, m) * OUT + cc] : 0.0f;
      } else if (i < MB_W2T) {
        const int e = i - MB_W3T, kt = e >> 8, ll = (e >> 2) & 63, j = e & 3, cc = ll & 15, gg = ll >> 4;
        v = (KIND == 0 && 4 * gg + j < OUT) ? a.w3[(16 * kt + cc) * OUT + 4 * gg + j] : 0.0f;
      } else {
        const int e = i - MB_W2T, blk = e >> 9, ll = (e >> 3) & 63, j = e & 7, cc = ll & 15, gg = ll >> 4;
        v = a.w2[(16 * (blk >> 1) + cc) * H + mb_feat(j, gg, blk & 1)];
      }
      put(i, v);
    }
  }
  if (threadIdx.x < H) { bias[threadIdx.x] = a.b1[threadIdx.x]; bias[H + threadIdx.x] = a.b2[threadIdx.x]; }
  if (threadIdx.x < 16) bias[2 * H + threadIdx.x] = (threadIdx.x < OUT) ? a.b3[threadIdx.x] : 0.0f;
  for (int i = (wv >> 2) * 64 + l; i < PC_PAIR; i += 128) img[i] = 0;      // (columns 35..47 of the observation image stay zero)
  __syncthreads();

  // a wave's tiles: tile0 + it * stride; every wave of the workgroup runs the barriers of `iters` tiles (the count of the pair with the most)
  const size_t ntiles = (a.n + 15) / 16, stride = (size_t)gridDim.x * 4;
  const size_t base = (size_t)blockIdx.x * 4;
  const size_t iters = ntiles > base ? (ntiles - base - 1) / stride + 1 : 0;
  const size_t tile0 = base + pw;
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
  float *red = (float *)(lds_m + 2 * MB_WPLANE);

  if (!producer) {
    // ================================ CONSUMER: the weight gradients ================================
#ifndef IRRL_PC_NO_PRIO
    __builtin_amdgcn_s_setprio(0);      // the producer on the same SIMD owns the tile's critical path: its instructions go first
#endif
    f32x4 gw1[3][4], gw2[4][4], gw3[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int j = 0; j < 4; j++) { gw2[i][j] = zero4; if (i < 3) gw1[i][j] = zero4; }
      gw3[i] = zero4;
    }
    // transposed read of feature tile `ft` of a [sample][feature] image with row stride RS: element e = [sample 4 g + e][feature 16 ft + c]
    auto TR = [&](const unsigned short *image, int rs, int ft) -> u16x4_t {
      const mb_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mb_s16x4 __attribute__((address_space(3))) *)(image + (4 * g + (c >> 2)) * rs + 4 * (c & 3) + 16 * ft));
      return __builtin_bit_cast(u16x4_t, v);
    };
    u16x4_t xa[3][2], h1a[4][2], h2a[4][2], d3[2], d2[4][2], d1[4][2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
      d3[p] = (u16x4_t){0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 4; i++) { h1a[i][p] = d3[p]; h2a[i][p] = d3[p]; d2[i][p] = d3[p]; d1[i][p] = d3[p]; if (i < 3) xa[i][p] = d3[p]; }
    }
    // The tile's 96 products (84 for the value network) are PACED: issued back to back they own the SIMD's matrix core for ~0.7 us right
    // behind barrier B, exactly where the producer runs the dependent MFMA chains of its second layer (measured: that phase 0.73 -> 1.66 us);
    // in groups of four with a short sleep in between they spread over the producer's tile and fill the matrix core's idle slots instead.
#ifndef IRRL_PC_PACE
#define IRRL_PC_PACE 2      /* s_sleep argument between groups (units of 64 clocks); 0 = back to back */
#endif
    auto pace = [&]() {
      if (IRRL_PC_PACE > 0) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(IRRL_PC_PACE); __builtin_amdgcn_sched_barrier(0); }
    };
    auto products = [&]() {
      // per accumulator the three plane products in mlp_bf16.hpp's order (q = 0, 1, 2); consecutive MFMAs go to different accumulators
#pragma unroll
      for (int q = 0; q < PR::N; q++) {
        if (KIND == 0) {
#pragma unroll
          for (int kt = 0; kt < 4; kt++) gw3[kt] = BF_MFMA16(h2a[kt][PR::A[q]], d3[PR::B[q]], gw3[kt]);
          pace();
        }
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
#pragma unroll
          for (int k2 = 0; k2 < 4; k2++) gw2[k2][nt] = BF_MFMA16(h1a[k2][PR::A[q]], d2[nt][PR::B[q]], gw2[k2][nt]);
          pace();
        }
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
#pragma unroll
          for (int kt = 0; kt < 3; kt++) gw1[kt][nt] = BF_MFMA16(xa[kt][PR::A[q]], d1[nt][PR::B[q]], gw1[kt][nt]);
          pace();
        }
      }
    };
    bool have = false;
    for (size_t it = 0; it < iters; it++) {
      PC_BARRIER();                                    // B: the last tile's operands are in registers (nothing pending in front of the first tile)
      if (have && IRRL_PC_AB != 7) products();
      PC_BARRIER();                                    // A: the producer's images of tile `it` are complete
      have = tile0 + it * stride < ntiles;
      if (have) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
#pragma unroll
          for (int kt = 0; kt < 3; kt++) xa[kt][p] = TR(IM(PC_OX, p), PC_RSX, kt);
#pragma unroll
          for (int kt = 0; kt < 4; kt++) {
            h1a[kt][p] = TR(IM(PC_OH1, p), MB_RS, kt);
            d2[kt][p] = TR(IM(PC_OD2, p), MB_RS, kt);
            d1[kt][p] = TR(IM(PC_OD1, p), MB_RS, kt);
            if (KIND == 0) h2a[kt][p] = TR(IM(PC_OH2, p), MB_RS, kt);
          }
          if (KIND == 0) d3[p] = TR(IM(PC_OD3, p), PC_RS3, 0);
        }
      }
    }
    if (have) products();
    // ---- reduction: the pairs add up in order, the two roles of a pair own disjoint entries of the row ----
    __syncthreads();
    for (int w = 0; w < 4; w++) {
      if (pw == w) {
        const bool first = w == 0;
        auto put = [&](int i, float v) { red[i] = first ? v : red[i] + v; };
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
          for (int nt = 0; nt < 4; nt++) {
#pragma unroll
            for (int kt = 0; kt < 4; kt++) {
              put(IRRL_MLP_P_DW2 + (16 * kt + 4 * g + r) * H + 16 * nt + c, gw2[kt][nt][r]);
              if (kt < 3) put(IRRL_MLP_P_DW1 + (16 * kt + 4 * g + r) * H + 16 * nt + c, gw1[kt][nt][r]);
            }
            if (KIND == 0) put(IRRL_MLP_P_DW3 + (16 * nt + 4 * g + r) * 16 + c, gw3[nt][r]);
          }
        }
      }
      __syncthreads();
    }
  } else {
    // ================================ PRODUCER: forward, loss, delta chains ================================
#ifndef IRRL_PC_NO_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    auto W8 = [&](int p, int off) -> u16x8_t { return *(const u16x8_t *)(wimg + p * MB_WPLANE + off + 8 * l); };
    auto W4 = [&](int p, int off) -> u16x4_t { return *(const u16x4_t *)(wimg + p * MB_WPLANE + off + 4 * l); };
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
    f32x4 gw3[4], gb1[4], gb2[4], gb3, gls, sc;       // gw3: the value head's lane-local gradient (KIND 1)
#pragma unroll
    for (int i = 0; i < 4; i++) { gw3[i] = zero4; gb1[i] = zero4; gb2[i] = zero4; }
    gb3 = zero4; gls = zero4; sc = zero4;
    auto split4 = [&](const f32x4 v, u16x4_t (&p)[2]) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        unsigned short pl[2];
        bf_split<2>(v[r], pl);
        p[0][r] = pl[0]; p[1][r] = pl[1];
      }
    };
    auto pair8 = [&](const u16x4_t lo, const u16x4_t hi) -> u16x8_t { return (u16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; };
    u16x8_t wr[16];
    u16x4_t w3t[8];
    auto load_w1 = [&]() {
#pragma unroll
      for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int p = 0; p < 2; p++) { wr[4 * nt + p] = W8(p, MB_W1A + 512 * nt); wr[4 * nt + 2 + p] = W8(p, MB_W1B + 512 * nt); }
    };
    auto load_w2 = [&]() {
      if (IRRL_PC_AB == 3) return;
#pragma unroll
      for (int b = 0; b < 8; b++)
#pragma unroll
        for (int p = 0; p < 2; p++) wr[2 * b + p] = W8(p, MB_W2 + 512 * b);
    };
    auto load_w2t = [&]() {
      if (IRRL_PC_AB == 3) return;
#pragma unroll
      for (int b = 0; b < 8; b++)
#pragma unroll
        for (int p = 0; p < 2; p++) wr[2 * b + p] = W8(p, MB_W2T + 512 * b);
    };
    auto load_head = [&]() {
      if (IRRL_PC_AB == 3) return;
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
    auto row_of = [&](size_t t) -> size_t {
      size_t j = t * 16 + c;
      if (j >= a.n) j = a.n - 1;
      return a.idx ? (size_t)a.idx[j] : j;
    };
    auto load_tile = [&](size_t row, TileIn &in) {
      if (REC) {
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
    size_t tile = tile0;
    if (tile < ntiles) {
      load_tile(row_of(tile), cur);
      row_next = row_of(tile + stride < ntiles ? tile + stride : tile);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see the f32 kernel
    load_w1();
#if IRRL_PC_AB == 1
    auto act_a = [&](float x) -> float { return -2.0f * x; };
    auto act_b = [&](float e) -> float { return 2.0f * (1.0f + e) - 1.0f; };
#else
    auto act_a = [&](float x) -> float { return __expf(-2.0f * x); };
    auto act_b = [&](float e) -> float { return 2.0f * __builtin_amdgcn_rcpf(1.0f + e) - 1.0f; };     // == fast_tanh
#endif
#ifdef IRRL_MB_PROFILE   /* tools/mlp_bf16_phases.py: where a producer's tile goes, in 100 MHz ticks (the d logstd slots carry the sums) */
    float ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pts_ = wall_clock64();
#endif
    for (size_t it = 0; it < iters; it++, tile += stride) {
      if (tile >= ntiles) {      // this pair has run out of tiles: the other pairs' barriers
        PC_BARRIER();
        PC_BARRIER();
        continue;
      }
      if (IRRL_PC_AB != 6) {
        load_tile(row_next, nxt);
        row_next = row_of(tile + 2 * stride < ntiles ? tile + 2 * stride : tile);
      }
      const bool valid = tile * 16 + c < a.n;

      // ---- observations: planes as the B operand (k = 8 g + j; 32 + 8 g + j) ----
      u16x8_t xp[2], xq[2];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        unsigned short pl[2];
        bf_split<2>(cur.x[j], pl);
        xp[0][j] = pl[0]; xp[1][j] = pl[1];
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        unsigned short pl[2] = {0, 0};
        if (j < 3) bf_split<2>((g == 0) ? cur.xt[j] : 0.0f, pl);
        xq[0][j] = pl[0]; xq[1][j] = pl[1];
      }
      // ---- layer 1 (no image is touched before barrier B) ----
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
          if (st == 3) { load_w2(); MB_PIN(); }
          float e_[4];
#pragma unroll
          for (int j = 0; j < 12; j++) {
            const int r = j / 3, part = j % 3;
            if (st < 3 && j < 6) chain1(st + 1, j);
            if (part == 0) e_[r] = act_a(acc[st][r]);
            else if (part == 1) h1[st][r] = act_b(e_[r]);
            else {
#if IRRL_PC_AB == 2
              h1p[st][0][r] = (unsigned short)(__builtin_bit_cast(unsigned, h1[st][r]) >> 16); h1p[st][1][r] = 0;
#else
              unsigned short pl[2]; bf_split<2>(h1[st][r], pl); h1p[st][0][r] = pl[0]; h1p[st][1][r] = pl[1];
#endif
            }
            MB_PIN();
          }
        }
      }
      MB_STAMP(0);   // observations + layer 1
      PC_BARRIER();       // B: the consumer holds the last tile's operands in registers -- the images are free
      MB_STAMP(1);   // waiting at B
#pragma unroll
      for (int p = 0; p < 2; p++) {
        if (IRRL_PC_AB != 4) *(u16x8_t *)(IM(PC_OX, p) + c * PC_RSX + 8 * g) = xp[p];
        if (IRRL_PC_AB != 4 && g == 0) *(u16x4_t *)(IM(PC_OX, p) + c * PC_RSX + 32) = (u16x4_t){xq[p][0], xq[p][1], xq[p][2], xq[p][3]};
#pragma unroll
        for (int st = 0; st < 4; st++) if (IRRL_PC_AB != 4) *(u16x4_t *)(IM(PC_OH1, p) + c * MB_RS + 16 * st + 4 * g) = h1p[st][p];
      }
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
            else {
#if IRRL_PC_AB == 2
              h2p[st][0][r] = (unsigned short)(__builtin_bit_cast(unsigned, h2[st][r]) >> 16); h2p[st][1][r] = 0;
#else
              unsigned short pl[2]; bf_split<2>(h2[st][r], pl); h2p[st][0][r] = pl[0]; h2p[st][1][r] = pl[1];
#endif
            }
            MB_PIN();
          }
          if (KIND == 0) {
#pragma unroll
            for (int p = 0; p < 2; p++) if (IRRL_PC_AB != 4) *(u16x4_t *)(IM(PC_OH2, p) + c * MB_RS + 16 * st + 4 * g) = h2p[st][p];
          }
        }
      }
      MB_STAMP(2);   // x, h1 images + layer 2
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

      MB_STAMP(3);   // head + loss
      // ---- backward: head ----
      f32x4 d[4];
      u16x4_t z2[4][2], z1[4][2];
      if (KIND == 0) {
        u16x4_t z3[2];
        split4(dz3, z3);
#pragma unroll
        for (int p = 0; p < 2; p++) if (IRRL_PC_AB != 4) *(u16x4_t *)(IM(PC_OD3, p) + c * PC_RS3 + 4 * g) = z3[p];
#pragma unroll
        for (int kt = 0; kt < 4; kt++) {
          d[kt] = zero4;
#pragma unroll
          for (int q = 0; q < PR::N; q++) d[kt] = BF_MFMA16(w3t[2 * kt + PR::A[q]], z3[PR::B[q]], d[kt]);
        }
        MB_PIN();
        load_w2t();
        MB_PIN();
      } else {
#pragma unroll
        for (int kt = 0; kt < 4; kt++) {
          d[kt] = dz3[0] * *(const f32x4 *)wa3[kt];
          gw3[kt] += dz3[0] * h2[kt];
        }
      }
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int kt = e >> 2, r = e & 3;
        const float t = d[kt][r] * (1.0f - h2[kt][r] * h2[kt][r]);
        gb2[kt][r] += t;
        unsigned short pl[2];
        bf_split<2>(t, pl);
        z2[kt][0][r] = pl[0]; z2[kt][1][r] = pl[1];
      }
      MB_STAMP(4);   // d h2, d z2
      // ---- layer 2 ----
#pragma unroll
      for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int p = 0; p < 2; p++) if (IRRL_PC_AB != 4) *(u16x4_t *)(IM(PC_OD2, p) + c * MB_RS + 16 * nt + 4 * g) = z2[nt][p];
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
      MB_PIN();
      if (IRRL_PC_AB != 3) load_w1();          // the next tile's first layer
      MB_PIN();
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int kt = e >> 2, r = e & 3;
        const float t = d[kt][r] * (1.0f - h1[kt][r] * h1[kt][r]);
        gb1[kt][r] += t;
        unsigned short pl[2];
        bf_split<2>(t, pl);
        z1[kt][0][r] = pl[0]; z1[kt][1][r] = pl[1];
      }
#pragma unroll
      for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int p = 0; p < 2; p++) if (IRRL_PC_AB != 4) *(u16x4_t *)(IM(PC_OD1, p) + c * MB_RS + 16 * nt + 4 * g) = z1[nt][p];
      MB_STAMP(5);   // d h1, d z1
      PC_BARRIER();       // A: this tile's images are complete
      MB_STAMP(6);   // waiting at A
      if (IRRL_PC_AB != 6) cur = nxt;
      MB_STAMP(7);   // the next tile's rows arriving
    }
#ifdef IRRL_MB_PROFILE
    gls = (f32x4){ph_[0], ph_[1], ph_[2], ph_[3]};
    if (g == 1) gls = (f32x4){ph_[4], ph_[5], ph_[6], ph_[7]};
    if (g > 1 || c != 0) gls = zero4;
#endif
    // ---- reduction (mlp_bf16.hpp's, same partial-sum row) ----
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
    for (int w = 0; w < 4; w++) {
      if (pw == w) {
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
        if (KIND == 1) {
#pragma unroll
          for (int r = 0; r < 4; r++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++) put(IRRL_MLP_P_DW3 + (16 * nt + 4 * g + r) * 16 + c, (c == 0) ? gw3[nt][r] : 0.0f);
        }
      }
      __syncthreads();
    }
  }
  float *dst = a.partials + (size_t)blockIdx.x * IRRL_MLP_P;
  for (int i = threadIdx.x; i < IRRL_MLP_P; i += 512) dst[i] = red[i];
}
