// How do the matrix cores ROUND?  One wave, D = A B + C with operands chosen so that the exact answer is known to the bit, for the three
// instructions the learner's kernels use: v_mfma_f32_16x16x16_bf16 (weight gradients of the LSTM update), v_mfma_f32_16x16x32_bf16
// (recurrences), v_mfma_f32_16x16x4_f32 (the exact-f32 kernels).  Every element of A is `a`, of B is `b` (so D[i][j] = K a b + C for all i, j), C = c.
//   case 1  K a b = 0.75 ulp(c)              round-to-nearest: c + ulp   truncation: c
//   case 2  K a b = 0.50 ulp(c) (a tie)      round-to-nearest-even: c    (c has an even mantissa)
//   case 3  K a b = 1.50 ulp(c)              round-to-nearest-even: c + 2 ulp ... (tie to even), truncation: c + ulp
//   case 4  every product = ulp(c) / K: the products only reach one ulp TOGETHER: exact sum before the one rounding: c + ulp; products aligned
//           to c and cut one by one: c
//   case 6  c = -1, K a b = 2^-26 (a quarter of the spacing below 1): nearest: -1, toward zero: -1 + 2^-24
//   case 5  c = 0, K products of 2^-140 (subnormal sum): flushed or kept
//   case 7  long accumulation: 4096 dependent MFMAs adding K a b = 0.3 ulp(1) to c = 1 -- the drift of a sequential f32 accumulator
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/mfma_rounding.hip -o tools/microbench/mfma_rounding
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

__device__ short bf16_bits(float v) { unsigned u = __float_as_uint(v); return (short)(u >> 16); }   // operands are exact in bf16 by construction

template <int KIND> __device__ f32x4 mfma(float a, float b, f32x4 c) {
  if (KIND == 0) {            // 16x16x16 bf16: 4 bf16 per lane and operand
    s16x4 av, bv;
    for (int i = 0; i < 4; i++) { av[i] = bf16_bits(a); bv[i] = bf16_bits(b); }
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, c, 0, 0, 0);
  } else if (KIND == 1) {     // 16x16x32 bf16: 8 bf16 per lane and operand
    s16x8 av, bv;
    for (int i = 0; i < 8; i++) { av[i] = bf16_bits(a); bv[i] = bf16_bits(b); }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
  } else {                    // 16x16x4 f32
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
}
template <int KIND> __global__ void probe(const float *abc, int n_cases, int chain, float *out) {
  for (int k = 0; k < n_cases; k++) {
    const float a = abc[3 * k], b = abc[3 * k + 1], c0 = abc[3 * k + 2];
    f32x4 c = {c0, c0, c0, c0};
    c = mfma<KIND>(a, b, c);
    if (threadIdx.x == 0) out[k] = c[0];
  }
  // the long chain
  const float a = abc[3 * n_cases], b = abc[3 * n_cases + 1], c0 = abc[3 * n_cases + 2];
  f32x4 c = {c0, c0, c0, c0};
  for (int i = 0; i < chain; i++) c = mfma<KIND>(a, b, c);
  if (threadIdx.x == 0) out[n_cases] = c[0];
}
static unsigned bits(float v) { unsigned u; memcpy(&u, &v, 4); return u; }
int main() {
  const char *names[3] = {"v_mfma_f32_16x16x16_bf16", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x4_f32"};
  const int Ks[3] = {16, 32, 4};
  const int chain = 4096;
  for (int kind = 0; kind < 3; kind++) {
    const int K = Ks[kind];
    const float ulp = ldexpf(1.0f, -23), c = 1.0f;
    // K a b with a = 1: b = target / K (a power of two times 1, 1.5 or 0.75: exact in bf16)
    float cases[7][3] = {{1.0f, 0.75f * ulp / K, c}, {1.0f, 0.5f * ulp / K, c}, {1.0f, 1.5f * ulp / K, c}, {1.0f, ulp / K, c},
                         {ldexpf(1.0f, -70), ldexpf(1.0f, -70), 0.0f}, {1.0f, 0.125f * ulp / K, -c},
                         {1.0f, 0.3125f * ulp / K, c}};     // last row = the chain's operands (0.3125 = 5/16: exact)
    float *d_abc, *d_out, out[8];
    hipMalloc(&d_abc, sizeof(cases)); hipMalloc(&d_out, sizeof(out));
    hipMemcpy(d_abc, cases, sizeof(cases), hipMemcpyHostToDevice);
    if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, d_abc, 6, chain, d_out);
    if (kind == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, d_abc, 6, chain, d_out);
    if (kind == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, d_abc, 6, chain, d_out);
    hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost);
    printf("%s (K = %d)\n", names[kind], K);
    printf("  1  c + 0.75 ulp            -> %.9g (0x%08x)   nearest: %.9g   truncation: %.9g\n", out[0], bits(out[0]), c + ulp, c);
    printf("  2  c + 0.50 ulp (tie)      -> %.9g (0x%08x)   nearest-even: %.9g\n", out[1], bits(out[1]), c);
    printf("  3  c + 1.50 ulp (tie)      -> %.9g (0x%08x)   nearest-even: %.9g   truncation: %.9g\n", out[2], bits(out[2]), c + 2 * ulp, c + ulp);
    printf("  4  K products of ulp / K   -> %.9g (0x%08x)   exact sum first: %.9g   cut one by one: %.9g\n", out[3], bits(out[3]), c + ulp, c);
    printf("  5  K products of 2^-140    -> %.9g (0x%08x)   kept: %.9g   flushed: 0\n", out[4], bits(out[4]), K * ldexp(1.0, -140));
    printf("  6  -1 + 2^-26              -> %.9g (0x%08x)   nearest: %.9g   toward zero: %.9g\n", out[5], bits(out[5]), -c, -c + ulp / 2);
    const double exact = 1.0 + chain * 0.3125 * (double)ulp;
    printf("  7  %d dependent MFMAs each adding 0.3125 ulp(1) to c = 1: %.9g   exact %.9g   (sequential nearest: stays 1; stochastic / wider accumulator: grows)\n",
           chain, out[6], exact);
    hipFree(d_abc); hipFree(d_out);
  }
  return 0;
}
