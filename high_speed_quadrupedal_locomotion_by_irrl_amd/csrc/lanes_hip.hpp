// lanes_hip.hpp -- lane primitives for gfx950 (CDNA4), wave64.
//
// Mapping used by every env kernel: ONE DPP QUAD PER ROBOT, ONE LEG PER LANE.
//   lane = threadIdx.x & 63, leg = lane & 3 (FR, FL, HR, HL), env = blockIdx.x * 16 + (lane >> 2).
// The four legs of a robot only couple through the floating base, so every cross-lane exchange of
// the dynamics is a reduction or a broadcast inside a quad -- exactly what DPP quad_perm does in the
// VALU operand path (no LDS, no ds_bpermute).  A wave carries 16 robots.
//
// env_core.hpp is written against the names defined here (vf/vi/vu/vm + helpers) so that the very
// same algorithm source can also be instantiated by the host-side lane emulation under
// tests/host_emulation (a debugging aid for the no-GPU build container; never part of the product).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define IRRL_DEV __device__ __forceinline__
// value made opaque to the optimizer (no instruction): what comes out is "some register", not the expression that produced it
#define IRRL_OPAQUE(x) asm volatile("" : "+v"(x))

// KERNEL ARGUMENTS READ WHERE THEY ARE USED (round 6).  EnvParams (92 words) + EnvState (26 pointers) + a kernel's own pointers are more than the
// 102 SGPRs.  Read as by-value arguments, every field is an invariant load that the optimizer hoists to the kernel's entry, and what does not
// fit is parked in VGPR lanes: a v_writelane up front and a v_readlane -- a VALU issue slot of the one resident wave -- at every use (step
// kernel: 119 spilled SGPRs, 217 v_readlane; multi-step kernel: 217 / 605).  irrl_kernarg<T>(off) names the argument in the kernarg segment
// itself (constant address space: s_load_dword[xN] where the field is used), irrl_refresh() makes that pointer opaque once per step of the
// multi-step kernels so that the loads of step k stay inside step k.  Same values, same arithmetic: results are bit-identical.
#define IRRL_CONST_AS __attribute__((address_space(4)))
template <class T> IRRL_DEV const T &irrl_kernarg(unsigned off) {
  const IRRL_CONST_AS char *p = (const IRRL_CONST_AS char *)__builtin_amdgcn_kernarg_segment_ptr() + off;
  return *(const T *)(const IRRL_CONST_AS T *)p;
}
template <class T> IRRL_DEV const T &irrl_refresh(const T &r) {
  const IRRL_CONST_AS T *p = (const IRRL_CONST_AS T *)&r;
  asm volatile("" : "+s"(p));
  return *(const T *)p;
}

typedef float vf;
typedef int32_t vi;
typedef uint32_t vu;
typedef bool vm;

// a PAIR of f32 per lane in an even-aligned register pair: +, -, * and a * b + c on pairs are ONE v_pk_{add,mul,fma}_f32 each
// (2.55 ns against 2.28 ns for a scalar FMA with one resident wave per SIMD, i.e. 1.3 ns per useful FMA), and op_sel broadcasts
// either half of an operand for free (the compiler emits it for the .xx / .yy swizzles and for pk2(s, s)).  Used where the
// algebra IS 2-vectors from the start -- the tangential plane of the published per-contact solve -- not as a vectorizer target.
typedef float vf2 __attribute__((ext_vector_type(2)));

#define IRRL_SUB0_ONLY_BEGIN {
#define IRRL_SUB0_ONLY_END }

namespace lanes {

IRRL_DEV vf2 pk2(vf a, vf b) { vf2 r; r.x = a; r.y = b; return r; }
IRRL_DEV vf pk_lo(vf2 a) { return a.x; }
IRRL_DEV vf pk_hi(vf2 a) { return a.y; }
IRRL_DEV vf pk_hsum(vf2 a) { return a.x + a.y; }
IRRL_DEV vf2 pk_sel(vm m, vf2 a, vf2 b) { vf2 r; r.x = m ? a.x : b.x; r.y = m ? a.y : b.y; return r; }

IRRL_DEV vi leg_id() { return (vi)(threadIdx.x & 3u); }

// ---- DPP quad primitives ----
template <int CTRL>
IRRL_DEV float dpp_quad(float x) {
  // old = 0 with bound_ctrl:1 (every source lane of a quad_perm is valid, so `old` is never used): this form lets the
  // DPP-combine pass fold the move into the consuming VALU op (v_add_f32_dpp, v_mul_f32_dpp, v_fmac_f32_dpp ...)
  int xi = __builtin_bit_cast(int, x);
  int r = __builtin_amdgcn_update_dpp(0, xi, CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(float, r);
}
template <int CTRL>
IRRL_DEV int dpp_quad_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }

// layout-neutral names used by env_core.hpp: legs_sum = sum over the robot's four legs (result in every lane),
// legs_bcast<K> = value held by leg K.  In this layout the four legs are the four lanes of a DPP quad.
// (quad_perm [1,0,3,2] then [2,3,0,1])
IRRL_DEV vf legs_sum(vf x) {
  x += dpp_quad<0xB1>(x);
  x += dpp_quad<0x4E>(x);
  return x;
}
IRRL_DEV vi legs_sum_i(vi x) {
  x += dpp_quad_i<0xB1>(x);
  x += dpp_quad_i<0x4E>(x);
  return x;
}
// value of lane K of the quad in every lane (quad_perm [K,K,K,K]); K is a compile-time constant
template <int K>
IRRL_DEV vf legs_bcast(vf x) { return dpp_quad<K * 0x55>(x); }
template <int K>
IRRL_DEV vi legs_bcast_i(vi x) { return dpp_quad_i<K * 0x55>(x); }
template <int K>
IRRL_DEV vu legs_bcast_u(vu x) { return (vu)dpp_quad_i<K * 0x55>((vi)x); }

// ---- masks / selects ----
IRRL_DEV vf vsel(vm m, vf a, vf b) { return m ? a : b; }
IRRL_DEV vi vsel_i(vm m, vi a, vi b) { return m ? a : b; }
IRRL_DEV vu vsel_u(vm m, vu a, vu b) { return m ? a : b; }
IRRL_DEV bool wave_any(vm m) { return __ballot(m) != 0ull; }
// wave-uniform maximum of a small non-negative per-lane integer (0..4)
IRRL_DEV int wave_max_small(vi x) {
  int r = 0;
  if (__ballot(x >= 1) != 0ull) r = 1;
  if (__ballot(x >= 2) != 0ull) r = 2;
  if (__ballot(x >= 3) != 0ull) r = 3;
  if (__ballot(x >= 4) != 0ull) r = 4;
  return r;
}

// ---- math (full-precision OCML forms: parity with the f64 oracle is judged at fp32 tolerance) ----
// v_sqrt_f32 / v_rcp_f32 / v_rsq_f32: one instruction each, ~1 ulp -- far inside the fp32 parity tolerance;
// the IEEE-exact expansions of sqrtf() and 1.0f/x cost ~10 VALU instructions apiece in a kernel that is bound
// by single-wave issue rate.
IRRL_DEV vf v_sqrt(vf x) { return __builtin_amdgcn_sqrtf(x); }
IRRL_DEV vf v_rcp(vf x) { return __builtin_amdgcn_rcpf(x); }
IRRL_DEV vf v_rsqrt(vf x) { return __builtin_amdgcn_rsqf(x); }
IRRL_DEV vf v_floor(vf x) { return __builtin_floorf(x); }
IRRL_DEV vi f2i(vf x) { return (vi)x; }
IRRL_DEV vf v_sin(vf x) { return sinf(x); }
IRRL_DEV vf v_cos(vf x) { return cosf(x); }
IRRL_DEV void v_sincos(vf x, vf &s, vf &c) { sincosf(x, &s, &c); }
IRRL_DEV vf v_asin(vf x) { return asinf(x); }
IRRL_DEV vf v_acos(vf x) { return acosf(x); }
IRRL_DEV vf v_exp(vf x) { return expf(x); }
IRRL_DEV vf v_log(vf x) { return logf(x); }
IRRL_DEV vf v_fmod(vf x, vf y) { return fmodf(x, y); }
// exp for the reward terms (arguments <= 0, results compared at 2e-4): v_exp_f32 on x log2(e), ~1e-6 relative
IRRL_DEV vf v_exp_fast(vf x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
// fmod(x, y) for 0 <= x < 2^10 y, y > 0 with inv_y ~ 1 / y: EXACT like fmodf (x - q y is representable, so the fma does not
// round; the guessed quotient is corrected by at most one), without fmodf's generic loop
IRRL_DEV vf v_fmod_pos(vf x, vf y, vf inv_y) {
  vf q = __builtin_floorf(x * inv_y);
  vf r = __builtin_fmaf(-q, y, x);
  r = (r < 0.0f) ? r + y : r;
  r = (r >= y) ? r - y : r;
  return r;
}
IRRL_DEV vf v_abs(vf x) { return fabsf(x); }
IRRL_DEV vf v_min(vf a, vf b) { return a < b ? a : b; }   // fmin(a,b) for non-NaN operands
IRRL_DEV vf v_max(vf a, vf b) { return a > b ? a : b; }
IRRL_DEV vu to_u(vi x) { return (vu)x; }
IRRL_DEV vf u2f(vu x) { return (float)x; }
IRRL_DEV vf i2f(vi x) { return (float)x; }
IRRL_DEV vu mulhi_u32(vu a, vu b) { return __umulhi(a, b); }

// ---- memory ----
IRRL_DEV vf ld(const float *p, vi idx) { return p[idx]; }
IRRL_DEV vi ld_i(const int32_t *p, vi idx) { return p[idx]; }
IRRL_DEV vu ld_u(const uint32_t *p, vi idx) { return p[idx]; }
IRRL_DEV void st(float *p, vi idx, vf v) { p[idx] = v; }
IRRL_DEV void st_i(int32_t *p, vi idx, vi v) { p[idx] = v; }
IRRL_DEV void st_u(uint32_t *p, vi idx, vu v) { p[idx] = v; }
IRRL_DEV void st_u8(uint8_t *p, vi idx, vi v) { p[idx] = (uint8_t)v; }
// a block executed by the lanes of mask m only (ONE exec-mask region for many stores); stm*: plain stores inside such a block
#define IRRL_MASKED_BEGIN(m) if (m) {
#define IRRL_MASKED_END }
IRRL_DEV void stm(float *p, vi idx, vf v) { p[idx] = v; }
IRRL_DEV void stm_i(int32_t *p, vi idx, vi v) { p[idx] = v; }
IRRL_DEV void stm_u(uint32_t *p, vi idx, vu v) { p[idx] = v; }
IRRL_DEV void stm_u8(uint8_t *p, vi idx, vi v) { p[idx] = (uint8_t)v; }
// store only from the lanes whose mask is set
IRRL_DEV void st_if(vm m, float *p, vi idx, vf v) { if (m) p[idx] = v; }
IRRL_DEV void st_i_if(vm m, int32_t *p, vi idx, vi v) { if (m) p[idx] = v; }
IRRL_DEV void st_u_if(vm m, uint32_t *p, vi idx, vu v) { if (m) p[idx] = v; }
IRRL_DEV void st_u8_if(vm m, uint8_t *p, vi idx, vi v) { if (m) p[idx] = (uint8_t)v; }

}  // namespace lanes
