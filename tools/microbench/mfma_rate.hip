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
  float *out; hipMalloc(&out, 512 * 256 * 4);
  run<8, 0>("8 independent accumulators", 256, out);
  run<8, 0>("8 independent accumulators", 512, out);
  run<2, 0>("2 independent accumulators", 256, out);
  run<1, 0>("1 accumulator (dependent chain)", 256, out);
  run<8, 4>("8 accumulators + 4 v_fma_f32 per MFMA", 256, out);
  run<8, 7>("8 accumulators + 7 v_fma_f32 per MFMA", 256, out);
  run<8, 12>("8 accumulators + 12 v_fma_f32 per MFMA", 256, out);
  run<8, 7>("8 accumulators + 7 v_fma_f32 per MFMA", 512, out);
  return 0;
}
