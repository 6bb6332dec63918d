// mfma_bf16_rate.hip -- cost of the bf16 MFMA forms a compensated-split LSTM kernel would use on MI355X, one wave per SIMD, beside
// the exact-f32 form it replaces, and whether independent VALU work of the SAME wave runs under them (it does not under the f32 form:
// tools/microbench/mfma_rate.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_bf16_rate.hip -o tools/microbench/mfma_bf16_rate && ./mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define ITER 20000

// KIND 0: v_mfma_f32_16x16x4_f32, 1: v_mfma_f32_16x16x16_bf16 (K = 16), 2: v_mfma_f32_16x16x32_bf16 (K = 32)
template <int KIND, int NACC, int VPM>
__global__ void __launch_bounds__(256) k(float *out, float a, float b) {
  f32x4 acc[NACC];
  float v[4] = {a, b, a + 1.0f, b + 1.0f};
  for (int i = 0; i < NACC; i++) acc[i] = (f32x4){a, b, a, b};
  bf16x4 a4 = {1, 2, 3, 4}, b4 = {4, 3, 2, 1};
  bf16x8 a8, b8;
  for (int i = 0; i < 8; i++) { a8[i] = (__bf16)(a + i); b8[i] = (__bf16)(b - i); }
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) {
      if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (KIND == 1) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a4), "v"(b4));
      if (KIND == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a8), "v"(b8));
#pragma unroll
      for (int j = 0; j < VPM; j++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[j & 3]) : "v"(a), "v"(b));
    }
  }
  float s = v[0] + v[1] + v[2] + v[3];
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int NACC, int VPM>
static void run(const char *name, float *out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NACC, VPM>), dim3(256), dim3(256), 0, 0, out, 1.0f, 1e-3f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const int K = KIND == 0 ? 4 : (KIND == 1 ? 16 : 32);
  printf("%-52s %7.2f ms, %6.2f ns per MFMA per SIMD, %6.3f ns per unit of K (16x16 tile)\n", name, ms, ms * 1e6 / ((double)ITER * NACC), ms * 1e6 / ((double)ITER * NACC * K));
}

int main() {
  float *out;
  (void)hipMalloc(&out, 256 * 256 * sizeof(float));
  run<0, 8, 0>("f32 16x16x4, 8 accumulators", out);
  run<1, 8, 0>("bf16 16x16x16, 8 accumulators", out);
  run<1, 1, 0>("bf16 16x16x16, 1 accumulator (dependent chain)", out);
  run<2, 8, 0>("bf16 16x16x32, 8 accumulators", out);
  run<2, 1, 0>("bf16 16x16x32, 1 accumulator (dependent chain)", out);
  run<1, 8, 1>("bf16 16x16x16 + 1 v_fma_f32 each", out);
  run<1, 8, 2>("bf16 16x16x16 + 2 v_fma_f32 each", out);
  run<1, 8, 4>("bf16 16x16x16 + 4 v_fma_f32 each", out);
  run<2, 8, 1>("bf16 16x16x32 + 1 v_fma_f32 each", out);
  run<2, 8, 2>("bf16 16x16x32 + 2 v_fma_f32 each", out);
  run<2, 8, 4>("bf16 16x16x32 + 4 v_fma_f32 each", out);
  run<2, 8, 8>("bf16 16x16x32 + 8 v_fma_f32 each", out);
  run<0, 8, 4>("f32 16x16x4 + 4 v_fma_f32 each", out);
  return 0;
}
