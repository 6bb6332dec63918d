// valu_issue.hip -- how fast ONE wave (and two waves) per SIMD issue plain, packed and DPP f32 VALU instructions on gfx950.
// Decides which levers of the env-step kernel (one resident wave per SIMD, VALU issue bound) are worth pulling:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_issue.hip -o tools/microbench/valu_issue && ./valu_issue
// Every kernel runs ITER x 64 instructions of one kind per wave on 16 independent accumulators (or 1 for the dependent chain).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
#define ITER 4096

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ void __launch_bounds__(512) k(float *out, float x, float y, unsigned long long mask) {
  float a[16];
  f2 p[16];
  for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 1e-3f + i; p[i] = (f2){a[i], a[i] + 1.0f}; }
  f2 x2 = {x, x}, y2 = {y, y};
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (KIND == 0) {        // v_fma_f32, 16 independent chains
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        REP16(X)
#undef X
      } else if (KIND == 1) {  // v_pk_fma_f32, 16 independent chains (2 flops x 2 per lane)
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(x2), "v"(y2));
        REP16(X)
#undef X
      } else if (KIND == 2) {  // dependent chain of v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(x), "v"(y));
        REP16(X)
#undef X
      } else if (KIND == 3) {  // v_mov_b32_dpp (quad_perm) + v_fma_f32 consuming it
#define X(i) { float t; asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(a[(i + 8) & 15])); \
               asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(t), "v"(y)); }
        REP16(X)
#undef X
      } else if (KIND == 4) {  // v_fmac_f32_dpp: the move folded into the FMA
#define X(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 8) & 15]), "v"(y));
        REP16(X)
#undef X
      } else if (KIND == 5) {  // v_add_f32_dpp (what the compiler does fold)
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 8) & 15]), "v"(y));
        REP16(X)
#undef X
      } else if (KIND == 6) {  // v_pk_mul_f32
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(x2));
        REP16(X)
#undef X
      } else if (KIND == 7) {  // v_pk_add_f32
#define X(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(x2));
        REP16(X)
#undef X
      } else if (KIND == 8) {  // v_cndmask_b32 (selects are ~20 % of the env kernel's VALU stream)
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x));
        REP16(X)
#undef X
      } else if (KIND == 9) {  // v_rcp_f32 (quarter rate?)
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        REP16(X)
#undef X
      } else if (KIND == 10) {  // row_ror DPP add (between legs)
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %2 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 8) & 15]), "v"(y));
        REP16(X)
#undef X
      } else if (KIND == 12) {  // v_cndmask_b32 e64 with an SGPR-pair mask
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "s"(mask));
        REP16(X)
#undef X
      } else if (KIND == 13) {
#define X(i) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[i]) : "v"(a[(i + 8) & 15]));
        REP16(X)
#undef X
      } else if (KIND == 14) {
#define X(i) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
        REP16(X)
#undef X
      } else if (KIND == 15) {
#define X(i) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
        REP16(X)
#undef X
      } else if (KIND == 16) {
#define X(i) asm volatile("s_nop 0");
        REP16(X)
#undef X
      } else if (KIND == 17) {
#define X(i) asm volatile("s_nop 1");
        REP16(X)
#undef X
      } else if (KIND == 18) {
#define X(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
        REP16(X)
#undef X
      } else if (KIND == 19) {
#define X(i) asm volatile("v_mul_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[(i + 8) & 15]), "v"(x));
        REP16(X)
#undef X
      } else if (KIND == 20) {  // compare + select pair
#define X(i) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %0\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
        REP16(X)
#undef X
      } else if (KIND == 21) {
#define X(i) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
        REP16(X)
#undef X
      } else if (KIND == 22) {
#define X(i) asm volatile("v_rsq_f32_e32 %0, %0" : "+v"(a[i]));
        REP16(X)
#undef X
      } else if (KIND == 23) {  // two interleaved dependent chains
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i & 1]) : "v"(x), "v"(y));
        REP16(X)
#undef X
      } else if (KIND == 24) {  // four interleaved dependent chains
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i & 3]) : "v"(x), "v"(y));
        REP16(X)
#undef X
      } else if (KIND == 25) {  // cndmask e32 with vcc, sources distinct from the destination
#define X(i) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(x), "v"(y));
        REP16(X)
#undef X
      } else if (KIND == 26) {  // v_sub_f32 + v_mul (plain VOP2 mix)
#define X(i) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
        REP16(X)
#undef X
      } else if (KIND == 11) {  // pk_fma with op_sel broadcasting the low half of src1 (scalar x vector form)
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(x2), "v"(y2));
        REP16(X)
#undef X
      }
    }
  }
  float s = 0.0f;
  for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static double run(const char *name, int waves_per_simd, double flops_per_inst, float *d_out) {
  // 256 CUs x 4 SIMDs: one block per CU of 4 (or 8) waves puts 1 (or 2) waves on every SIMD
  dim3 grid(256), block(64 * 4 * waves_per_simd);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d_out, 1.0001f, 1e-7f, 0x5555aaaa3333ccccull);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d_out, 1.0001f, 1e-7f, 0x5555aaaa3333ccccull);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.0f;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)ITER * 64.0;  // per wave
  const double ns_per_inst = ms * 1e6 / insts;
  printf("%-44s waves/SIMD %d : %7.3f ns per wave-instruction  (%6.2f cycles @2.4GHz)  %7.1f TFLOP/s\n", name, waves_per_simd, ns_per_inst, ns_per_inst * 2.4,
         flops_per_inst * 64.0 * insts * 1024.0 * waves_per_simd / (ms * 1e-3) / 1e12);
  return ns_per_inst;
}

// DPP read-after-write hazard: does the hardware interlock when the DPP source was written by the previous VALU instruction?
__global__ void hazard(float *out, const float *in) {
  float v = in[threadIdx.x], w = in[threadIdx.x + 64];
  float r0, r1, r2;
  asm volatile("v_add_f32_e32 %0, %1, %1\n v_add_f32_dpp %2, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(r0), "+v"(v), "=&v"(r1) : );
  asm volatile("v_add_f32_e32 %0, %1, %1\n s_nop 1\n v_add_f32_dpp %2, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(r0), "+v"(w), "=&v"(r2) : );
  out[threadIdx.x] = r1;
  out[threadIdx.x + 64] = r2;
}
int main() {
  {
    float h_in[128], h_out[128], *d_in, *d_o;
    for (int i = 0; i < 64; i++) { h_in[i] = (float)i; h_in[64 + i] = (float)i; }
    hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_o, sizeof(h_out));
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(hazard, dim3(1), dim3(64), 0, 0, d_o, d_in);
    hipMemcpy(h_out, d_o, sizeof(h_out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) if (h_out[i] != h_out[64 + i]) bad++;
    printf("DPP RAW hazard without s_nop: %d of 64 lanes differ from the s_nop version (0 = hardware interlocks or no hazard)\n", bad);
  }
  float *d_out;
  hipMalloc(&d_out, 256 * 512 * sizeof(float));
  for (int w = 1; w <= 2; w++) {
    run<0>("v_fma_f32 x16 independent", w, 2, d_out);
    run<1>("v_pk_fma_f32 x16 independent", w, 4, d_out);
    run<11>("v_pk_fma_f32 op_sel_hi (scalar bcast)", w, 4, d_out);
    run<2>("v_fma_f32 dependent chain", w, 2, d_out);
    run<3>("v_mov_b32_dpp + v_fma_f32 (2 insts)", w, 1, d_out);
    run<4>("v_fmac_f32_dpp", w, 2, d_out);
    run<5>("v_add_f32_dpp quad_perm", w, 1, d_out);
    run<10>("v_add_f32_dpp row_ror:4", w, 1, d_out);
    run<6>("v_pk_mul_f32", w, 2, d_out);
    run<7>("v_pk_add_f32", w, 2, d_out);
    run<8>("v_cndmask_b32", w, 0, d_out);
    run<9>("v_rcp_f32", w, 1, d_out);
    run<12>("v_cndmask_b32_e64 sgpr mask", w, 0, d_out);
    run<25>("v_cndmask_b32_e32 vcc, dst != src", w, 0, d_out);
    run<20>("v_cmp_lt_f32 + v_cndmask (2 insts)", w, 0, d_out);
    run<13>("v_mov_b32_e32", w, 0, d_out);
    run<14>("v_mul_f32_e32", w, 1, d_out);
    run<15>("v_add_f32_e32", w, 1, d_out);
    run<26>("v_sub_f32_e32", w, 1, d_out);
    run<21>("v_max_f32_e32", w, 1, d_out);
    run<18>("v_fmac_f32_e32 (VOP2)", w, 2, d_out);
    run<19>("v_mul_f32_dpp", w, 1, d_out);
    run<22>("v_rsq_f32", w, 1, d_out);
    run<16>("s_nop 0", w, 0, d_out);
    run<17>("s_nop 1", w, 0, d_out);
    run<23>("v_fma_f32 2 interleaved chains", w, 2, d_out);
    run<24>("v_fma_f32 4 interleaved chains", w, 2, d_out);
  }
  return 0;
}
