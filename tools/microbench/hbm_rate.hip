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
  return 0;
}
