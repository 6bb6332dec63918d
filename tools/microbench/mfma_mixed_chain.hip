// mfma_mixed_chain.hip -- does a chain of MIXED bf16 MFMA shapes on ONE accumulator give the sum of its products?  (csrc/mlp_bf16.hpp's first version
// accumulated a K = 16 v_mfma_f32_16x16x16_bf16 and a K = 32 v_mfma_f32_16x16x32_bf16 into the same registers in turn and lost 4-86 % of the K = 16
// contributions, no fault.)  Exact small-integer data: every product and sum is exact in f32, so any difference from the reference is an error.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_mixed_chain.hip -o tools/microbench/mfma_mixed_chain && ./mfma_mixed_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
__device__ inline unsigned short bfb(float x) { const __bf16 b = (__bf16)x; return __builtin_bit_cast(unsigned short, b); }
#define M32(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), c_, 0, 0, 0)
#define M16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a_), __builtin_bit_cast(s16x4, b_), c_, 0, 0, 0)

// out[0]: mixed chain on one accumulator (16, 32, 16, 32, 16, 32); out[1]: the same six products, the K = 16 ones and the K = 32 ones on their own accumulators, added at the end
__global__ void k(float *out, int reps) {
  const int l = threadIdx.x, c = l & 15, g = l >> 4;
  u16x8 a8[3], b8[3];
  u16x4 a4[3], b4[3];
  for (int q = 0; q < 3; q++) {
    for (int j = 0; j < 8; j++) { a8[q][j] = bfb((float)((c + 2 * g + j + q) % 5 - 2)); b8[q][j] = bfb((float)((3 * c + g + 2 * j + q) % 7 - 3)); }
    for (int j = 0; j < 4; j++) { a4[q][j] = bfb((float)((2 * c + g + j + q) % 5 - 2)); b4[q][j] = bfb((float)((c + 3 * g + j + 2 * q) % 7 - 3)); }
  }
  f32x4 mixed = {1.0f, 2.0f, 3.0f, 4.0f}, s16 = {0, 0, 0, 0}, s32 = {1.0f, 2.0f, 3.0f, 4.0f};
  for (int r = 0; r < reps; r++)
#pragma unroll
    for (int q = 0; q < 3; q++) {
      mixed = M16(a4[q], b4[q], mixed);
      mixed = M32(a8[q], b8[q], mixed);
    }
  for (int r = 0; r < reps; r++)
#pragma unroll
    for (int q = 0; q < 3; q++) { s16 = M16(a4[q], b4[q], s16); s32 = M32(a8[q], b8[q], s32); }
  const f32x4 sep = s16 + s32;
  for (int i = 0; i < 4; i++) { out[(0 * 64 + l) * 4 + i] = mixed[i]; out[(1 * 64 + l) * 4 + i] = sep[i]; }
}

int main() {
  float *d; (void)hipMalloc(&d, 2 * 64 * 4 * sizeof(float));
  static float h[2 * 64 * 4];
  for (int reps : {1, 4}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, reps);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // reference on the host from the documented operand layouts
    int bad_mixed = 0, bad_sep = 0; double worst = 0;
    for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) {
      const int col = l & 15, row = 4 * (l >> 4) + i;
      double ref = (double)(i + 1);
      for (int q = 0; q < 3; q++) {
        double s = 0;
        for (int g = 0; g < 4; g++) {
          for (int j = 0; j < 8; j++) s += (double)((row + 2 * g + j + q) % 5 - 2) * (double)((3 * col + g + 2 * j + q) % 7 - 3);
          for (int j = 0; j < 4; j++) s += (double)((2 * row + g + j + q) % 5 - 2) * (double)((col + 3 * g + j + 2 * q) % 7 - 3);
        }
        ref += reps * s;
      }
      const double em = fabs(h[(0 * 64 + l) * 4 + i] - ref), es = fabs(h[(1 * 64 + l) * 4 + i] - ref);
      if (em > 0) bad_mixed++;
      if (es > 0) bad_sep++;
      if (em > worst) worst = em;
    }
    printf("%d repetition(s) of (K16, K32) x 3: mixed chain on one accumulator: %d of 256 elements wrong (worst |error| %.0f); separate accumulators: %d wrong\n", reps, bad_mixed, worst, bad_sep);
  }
  return 0;
}
