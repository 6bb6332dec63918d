// hbm_rate.hip -- what HBM rates does an MI355X give to the ACCESS SHAPES of the LSTM sequence kernels (csrc/lstm_bf16.hpp)?
// Those kernels run one workgroup per 16 envs (256 workgroups at 4096 envs) that walks the T steps of a [T][N][C] array: per step a
// workgroup touches one contiguous 16 x C x 4 byte piece, consecutive steps are N x C x 4 bytes apart.  The forward kernel stores
// (4.1 GB per launch at 3.4 TB/s), the backward kernel loads with its requests three steps ahead (5.8 GB at 3.6 TB/s).  This probe
// separates the chip's ceiling for that shape from what the kernels reach:
//   fill / read / copy with a plain grid-stride loop over the whole chip (many workgroups)            -- the ceiling
//   the walking shape: 256 (or 512, 1024) workgroups, each its own 16-env column, stores resp. loads  -- the shape's ceiling
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/hbm_rate.hip -o tools/microbench/hbm_rate && ./hbm_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) fill_k(f32x4 *p, size_t n4, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = (f32x4){v, v, v, v};
}
__global__ void __launch_bounds__(256) read_k(const f32x4 *p, size_t n4, float *out) {
  f32x4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) s += p[i];
  if (s[0] + s[1] + s[2] + s[3] == 1.2345f) out[0] = 1.0f;
}
__global__ void __launch_bounds__(256) copy_k(const f32x4 *p, f32x4 *q, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) q[i] = p[i];
}
// the walking shape.  C4 = float4 per env row (72: gates 48 + c 12 + h 12 as the forward kernel stores them, in one array here);
// a workgroup of 192 lanes owns EPW envs; per step it moves EPW * C4 float4, lanes contiguous.
template <int MODE, int DEPTH>   // MODE 0 store, 1 load (DEPTH steps in flight), 2 load + store (different arrays)
__global__ void __launch_bounds__(192) walk_k(f32x4 *p, f32x4 *q, int T, int N, int C4, int EPW, float *out) {
  const size_t row4 = (size_t)N * C4;
  const size_t base = (size_t)blockIdx.x * EPW * C4;
  const int per = EPW * C4;
  f32x4 s = {0, 0, 0, 0};
  if (MODE == 0) {
    for (int t = 0; t < T; t++)
      for (int i = threadIdx.x; i < per; i += 192) p[t * row4 + base + i] = (f32x4){(float)t, 1.0f, 2.0f, 3.0f};
  } else {
    // per <= 6 * 192 for the shapes used: up to 6 float4 per lane and step, DEPTH steps requested ahead
    f32x4 ring[DEPTH][6];
    auto issue = [&](int t, f32x4 (&r)[6]) {
#pragma unroll
      for (int j = 0; j < 6; j++) { const int i = threadIdx.x + 192 * j; if (i < per) r[j] = __builtin_nontemporal_load(&p[t * row4 + base + i]); }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) if (d < T) issue(d, ring[d]);
    for (int t0 = 0; t0 < T; t0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; d++) {
        const int t = t0 + d;
        if (t >= T) break;
#pragma unroll
        for (int j = 0; j < 6; j++) { const int i = threadIdx.x + 192 * j; if (i < per) { s += ring[d][j]; if (MODE == 2) q[t * row4 + base + i] = ring[d][j]; } }
        if (t + DEPTH < T) issue(t + DEPTH, ring[d]);
      }
    }
    if (s[0] + s[1] + s[2] + s[3] == 1.2345f) out[0] = 1.0f;
  }
}

// the forward LSTM kernel's STORE MIX per step and workgroup (16 envs, 48 units): gates [env][unit][4] as float4 (lane (col, rq) of wave w: unit
// 16 w + col, envs 4 rq + j: 256-byte pieces), c and h [env][unit] as dwords (64-byte pieces), three separate arrays -- or (PACKED) one array of
// [env][unit][6]-float records (24 bytes: gates, c, h together), or (WIDE) c and h through float4 stores of 256-byte pieces
template <int KIND>   // 0: as the kernel; 1: gates only; 2: c/h only; 3: packed records; 4: c/h as float4
__global__ void __launch_bounds__(192) lstm_store_mix_k(float *gates, float *cs, float *hs, int T, int N) {
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, col = l & 15, rq = l >> 4, u = 16 * w + col, e0 = blockIdx.x * 16;
  for (int t = 0; t < T; t++) {
    const float v = (float)(t + tid);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const size_t row = (size_t)t * N + e0 + 4 * rq + j;
      if (KIND == 0 || KIND == 1) *(f32x4 *)&gates[(row * 48 + u) * 4] = (f32x4){v, v + 1, v + 2, v + 3};
      if (KIND == 0 || KIND == 2) { cs[row * 48 + u] = v; hs[row * 48 + u] = v + 1; }
      if (KIND == 3) { float *r = gates + (row * 48 + u) * 6; *(float2 *)r = make_float2(v, v + 1); *(float2 *)(r + 2) = make_float2(v + 2, v + 3); *(float2 *)(r + 4) = make_float2(v, v + 1); }
    }
    if (KIND == 4) {      // the workgroup's 16 x 48 block of c and of h is contiguous (3 KB each): 192 lanes x float4 per array
      const size_t base = ((size_t)t * N + e0) * 48;
      *(f32x4 *)&cs[base + 4 * tid] = (f32x4){v, v, v, v};
      *(f32x4 *)&hs[base + 4 * tid] = (f32x4){v, v, v, v};
#pragma unroll
      for (int j = 0; j < 4; j++) { const size_t row = (size_t)t * N + e0 + 4 * rq + j; *(f32x4 *)&gates[(row * 48 + u) * 4] = (f32x4){v, v + 1, v + 2, v + 3}; }
    }
  }
}

static hipEvent_t e0, e1;
template <class F> static double timed(F f, int reps = 3) {
  double best = 1e30;
  for (int r = 0; r < reps; r++) {
    (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int T = 750, N = 4096, C4 = 72;                 // 750 x 4096 x 288 floats = 3.54 GB: one layer's saved activations
  const size_t n4 = (size_t)T * N * C4, bytes = n4 * 16;
  f32x4 *p, *q; float *out;
  if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&q, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(p, 0, bytes); (void)hipMemset(q, 0, bytes);
  printf("array: %d x %d x %d floats = %.2f GB\n", T, N, 4 * C4, bytes / 1e9);
  for (int wg : {1024, 4096, 16384}) {
    double a = timed([&] { hipLaunchKernelGGL(fill_k, dim3(wg), dim3(256), 0, 0, p, n4, 1.0f); });
    double b = timed([&] { hipLaunchKernelGGL(read_k, dim3(wg), dim3(256), 0, 0, p, n4, out); });
    double c = timed([&] { hipLaunchKernelGGL(copy_k, dim3(wg), dim3(256), 0, 0, p, q, n4); });
    printf("grid-stride, %5d workgroups: fill %6.3f ms = %5.2f TB/s   read %6.3f ms = %5.2f TB/s   copy %6.3f ms = %5.2f TB/s (read + write)\n", wg, a, bytes / a / 1e9, b,
           bytes / b / 1e9, c, 2.0 * bytes / c / 1e9);
  }
  for (int epw : {16, 8, 4}) {
    const int wgs = N / epw;
    double a = timed([&] { hipLaunchKernelGGL((walk_k<0, 1>), dim3(wgs), dim3(192), 0, 0, p, q, T, N, C4, epw, out); });
    double b1 = timed([&] { hipLaunchKernelGGL((walk_k<1, 1>), dim3(wgs), dim3(192), 0, 0, p, q, T, N, C4, epw, out); });
    double b3 = timed([&] { hipLaunchKernelGGL((walk_k<1, 3>), dim3(wgs), dim3(192), 0, 0, p, q, T, N, C4, epw, out); });
    double b6 = timed([&] { hipLaunchKernelGGL((walk_k<1, 6>), dim3(wgs), dim3(192), 0, 0, p, q, T, N, C4, epw, out); });
    double c3 = timed([&] { hipLaunchKernelGGL((walk_k<2, 3>), dim3(wgs), dim3(192), 0, 0, p, q, T, N, C4, epw, out); });
    printf("walking, %4d workgroups x %2d envs: store %6.3f ms = %5.2f TB/s   load depth 1 / 3 / 6: %5.2f / %5.2f / %5.2f TB/s   load + store (depth 3) %5.2f TB/s\n", wgs, epw, a,
           bytes / a / 1e9, bytes / b1 / 1e9, bytes / b3 / 1e9, bytes / b6 / 1e9, 2.0 * bytes / c3 / 1e9);
  }
  {
    float *g = (float *)p, *cc = (float *)q, *hh = (float *)q + (size_t)T * N * 48;      // gates 2.36 GB in p (as packed records: 3.54 GB), c | h 0.59 GB each in q
    const double gb = (double)T * N * 48 * 4;
    const char *names[5] = {"gates float4 + c, h dwords (the kernel)", "gates only", "c, h only", "packed [unit][6] records", "gates float4 + c, h as float4 blocks"};
    const double bytes_k[5] = {6 * gb, 4 * gb, 2 * gb, 6 * gb, 6 * gb};
    double ms;
    for (int k = 0; k < 5; k++) {
      if (k == 0) ms = timed([&] { hipLaunchKernelGGL((lstm_store_mix_k<0>), dim3(N / 16), dim3(192), 0, 0, g, cc, hh, T, N); });
      if (k == 1) ms = timed([&] { hipLaunchKernelGGL((lstm_store_mix_k<1>), dim3(N / 16), dim3(192), 0, 0, g, cc, hh, T, N); });
      if (k == 2) ms = timed([&] { hipLaunchKernelGGL((lstm_store_mix_k<2>), dim3(N / 16), dim3(192), 0, 0, g, cc, hh, T, N); });
      if (k == 3) ms = timed([&] { hipLaunchKernelGGL((lstm_store_mix_k<3>), dim3(N / 16), dim3(192), 0, 0, g, cc, hh, T, N); });
      if (k == 4) ms = timed([&] { hipLaunchKernelGGL((lstm_store_mix_k<4>), dim3(N / 16), dim3(192), 0, 0, g, cc, hh, T, N); });
      printf("LSTM forward store mix, %-44s %6.3f ms = %5.2f TB/s\n", names[k], ms, bytes_k[k] / ms / 1e9);
    }
  }
  return 0;
}
