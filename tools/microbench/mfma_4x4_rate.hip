// mfma_4x4_rate.hip -- what does a v_mfma_f32_4x4x1_16b_f32 cost a SIMD on MI355X (the wave-level policy steps of csrc/policy_step.hpp run on it)?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_4x4_rate.hip -o tools/microbench/mfma_4x4_rate && ./mfma_4x4_rate
// NACC independent accumulators per wave (the dependent distance of a chain), one / two waves per SIMD; with and without an LDS operand read per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ITER 20000

template <int NACC, int LDSR>
__global__ void __launch_bounds__(256) k(float *out, float a, float b) {
  __shared__ float w[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) w[i] = b + (float)i * 1e-6f;
  __syncthreads();
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (f32x4){a, b, a, b};
  const int l = threadIdx.x & 63;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) {
      float bb = b;
      if (LDSR) bb = w[((it * NACC + i) * 64 + l) & 4095];
      acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, bb, acc[i], 0, 0, 0);
    }
  }
  float s = 0.0f;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int LDSR>
static void run(const char *name, int blocks, float *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LDSR>), dim3(blocks), dim3(256), 0, 0, out, 1.0f, 1e-3f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)ITER * NACC * (blocks / 256);
  printf("%-52s %4d workgroups: %7.2f ms, %6.2f ns per MFMA per SIMD\n", name, blocks, ms, ms * 1e6 / per_simd);
}

int main() {
  float *out; (void)hipMalloc(&out, 512 * 256 * sizeof(float));
  run<1, 0>("4x4x1, 1 chain (dependent back to back)", 256, out);
  run<2, 0>("4x4x1, 2 chains", 256, out);
  run<3, 0>("4x4x1, 3 chains", 256, out);
  run<4, 0>("4x4x1, 4 chains", 256, out);
  run<8, 0>("4x4x1, 8 chains", 256, out);
  run<3, 0>("4x4x1, 3 chains, two waves per SIMD", 512, out);
  run<3, 1>("4x4x1, 3 chains + one ds_read_b32 per MFMA", 256, out);
  run<3, 1>("4x4x1, 3 chains + ds_read_b32, two waves per SIMD", 512, out);
  run<8, 1>("4x4x1, 8 chains + ds_read_b32", 256, out);
  return 0;
}
