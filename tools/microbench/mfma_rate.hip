// mfma_rate.hip -- what does a v_mfma_f32_16x16x4_f32 cost a SIMD on MI355X when the whole chip does nothing else?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_rate.hip -o tools/microbench/mfma_rate && ./mfma_rate
// 256 / 512 workgroups of 256 threads (one / two waves per SIMD on every CU), NACC independent accumulators per wave, ITER x NACC
// MFMAs per wave; a second kernel interleaves 4 independent v_fma_f32 per MFMA (does a wave's VALU work run under its MFMAs?).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ITER 20000

template <int NACC, int VALU_PER_MFMA>
__global__ void __launch_bounds__(256) k(float *out, float a, float b) {
  f32x4 acc[NACC];
  float v[4] = {a, b, a + 1.0f, b + 1.0f};
  for (int i = 0; i < NACC; i++) acc[i] = (f32x4){a, b, a, b};
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) {
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < VALU_PER_MFMA; j++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[j & 3]) : "v"(a), "v"(b));
    }
  }
  float s = v[0] + v[1] + v[2] + v[3];
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// two kinds of wave in one workgroup of 512 threads (two waves per SIMD): waves 0-3 issue only MFMAs, waves 4-7 only v_fma_f32
// (VALU_PER_MFMA of them for every MFMA of a sibling) -- does the SIMD overlap a pure-MFMA wave with a pure-VALU wave?
template <int NACC, int VALU_PER_MFMA>
__global__ void __launch_bounds__(512) k_split(float *out, float a, float b) {
  f32x4 acc[NACC];
  float v[4] = {a, b, a + 1.0f, b + 1.0f};
  for (int i = 0; i < NACC; i++) acc[i] = (f32x4){a, b, a, b};
  const bool mfma_wave = (threadIdx.x >> 6) < 4;
  for (int it = 0; it < ITER; it++) {
    if (mfma_wave) {
#pragma unroll
      for (int i = 0; i < NACC; i++) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    } else {
#pragma unroll
      for (int i = 0; i < NACC * VALU_PER_MFMA; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 3]) : "v"(a), "v"(b));
    }
  }
  float s = v[0] + v[1] + v[2] + v[3];
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NACC, int VPM>
static void run_split(const char *name, float *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_split<NACC, VPM>), dim3(256), dim3(512), 0, 0, out, 1.0f, 1e-3f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s  256 workgroups: %7.2f ms, %6.2f ns per MFMA per SIMD (MFMA wave + VALU wave side by side)\n", name, ms, ms * 1e6 / ((double)ITER * NACC));
}

template <int NACC, int VPM>
static void run(const char *name, int blocks, float *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, VPM>), dim3(blocks), dim3(256), 0, 0, out, 1.0f, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)ITER * NACC * (blocks / 256);           // MFMAs each SIMD executed
  const double tflops = 2048.0 * ITER * NACC * 4.0 * blocks / (ms * 1e-3) / 1e12;
  printf("%-44s %4d workgroups: %7.2f ms, %6.2f ns per MFMA per SIMD, %6.1f TFLOP/s f32 matrix\n", name, blocks, ms, ms * 1e6 / per_simd, tflops);
}

int main() {
  float *out; (void)hipMalloc(&out, 512 * 256 * 4);
  run<8, 0>("8 independent accumulators", 256, out);
  run<8, 0>("8 independent accumulators", 512, out);
  run<2, 0>("2 independent accumulators", 256, out);
  run<1, 0>("1 accumulator (dependent chain)", 256, out);
  run<8, 4>("8 accumulators + 4 v_fma_f32 per MFMA", 256, out);
  run<8, 7>("8 accumulators + 7 v_fma_f32 per MFMA", 256, out);
  run<8, 12>("8 accumulators + 12 v_fma_f32 per MFMA", 256, out);
  run<8, 7>("8 accumulators + 7 v_fma_f32 per MFMA", 512, out);
  run_split<8, 4>("MFMA-only wave next to a wave with 4 v_fma per MFMA", out);
  run_split<8, 7>("MFMA-only wave next to a wave with 7 v_fma per MFMA", out);
  run_split<8, 12>("MFMA-only wave next to a wave with 12 v_fma per MFMA", out);
  return 0;
}
