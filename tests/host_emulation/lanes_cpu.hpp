// lanes_cpu.hpp -- host-side emulation of ONE DPP quad (4 lanes in lockstep) so that the kernel source
// env_core.hpp can be executed and debugged in the GPU-less build container.
//
// TEST-ONLY: compiled into tests/host_emulation/_build/libirrl_emu.so by the test-suite; it is not part
// of the product library, is not reachable from the C-ABI (include/irrl_env.h) and is never used for a
// result the product reports.  A "wave" here is a single quad, so wave_any == quad_any.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#define IRRL_DEV inline
#define IRRL_OPAQUE(x) do { } while (0)
// on the GPU this region runs on sub-lane 0 of every leg only; the emulation lets every lane compute (same values)
#define IRRL_SUB0_ONLY_BEGIN {
#define IRRL_SUB0_ONLY_END }
#define IRRL_MASKED_BEGIN(m) { lanes::mask_scope irrl_scope_(m);
#define IRRL_MASKED_END }

// W lanes emulate ONE robot: W = 4 (one DPP quad per robot, one leg per lane) or W = 16 (one DPP row per robot: the quad
// (i >> 2) is the leg, the lane inside the quad (i & 3) a sub-lane that splits the leg's work)
#ifndef IRRL_EMU_W
#define IRRL_EMU_W 4
#endif
static constexpr int W = IRRL_EMU_W;
#if IRRL_EMU_W == 16
#define IRRL_L16 1
#endif

struct vm { bool v[W]; vm() {} vm(bool b) { for (int i = 0; i < W; i++) v[i] = b; } };
struct vf { float v[W]; vf() {} vf(float s) { for (int i = 0; i < W; i++) v[i] = s; } };
struct vi { int32_t v[W]; vi() {} vi(int32_t s) { for (int i = 0; i < W; i++) v[i] = s; } };
struct vu { uint32_t v[W]; vu() {} vu(uint32_t s) { for (int i = 0; i < W; i++) v[i] = s; } };

#define LANEWISE_BIN(T, R, OP)                                                                         \
  inline R operator OP(T a, T b) { R r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] OP b.v[i]; return r; }
LANEWISE_BIN(vf, vf, +) LANEWISE_BIN(vf, vf, -) LANEWISE_BIN(vf, vf, *) LANEWISE_BIN(vf, vf, /)
LANEWISE_BIN(vf, vm, <) LANEWISE_BIN(vf, vm, >) LANEWISE_BIN(vf, vm, <=) LANEWISE_BIN(vf, vm, >=)
LANEWISE_BIN(vi, vi, +) LANEWISE_BIN(vi, vi, -) LANEWISE_BIN(vi, vi, *) LANEWISE_BIN(vi, vi, &)
LANEWISE_BIN(vi, vm, ==) LANEWISE_BIN(vi, vm, !=) LANEWISE_BIN(vi, vm, <) LANEWISE_BIN(vi, vm, >=)
LANEWISE_BIN(vu, vu, +) LANEWISE_BIN(vu, vu, *) LANEWISE_BIN(vu, vu, ^)
LANEWISE_BIN(vm, vm, &) LANEWISE_BIN(vm, vm, |)
// scalar on either side
#define SCALAR_BIN(T, S, R, OP)                          \
  inline R operator OP(T a, S b) { return a OP T(b); }   \
  inline R operator OP(S a, T b) { return T(a) OP b; }
SCALAR_BIN(vf, float, vf, +) SCALAR_BIN(vf, float, vf, -) SCALAR_BIN(vf, float, vf, *) SCALAR_BIN(vf, float, vf, /)
SCALAR_BIN(vf, float, vm, <) SCALAR_BIN(vf, float, vm, >) SCALAR_BIN(vf, float, vm, <=) SCALAR_BIN(vf, float, vm, >=)
SCALAR_BIN(vi, int, vi, +) SCALAR_BIN(vi, int, vi, -) SCALAR_BIN(vi, int, vi, *) SCALAR_BIN(vi, int, vi, &)
SCALAR_BIN(vi, int, vm, ==) SCALAR_BIN(vi, int, vm, !=) SCALAR_BIN(vi, int, vm, <) SCALAR_BIN(vi, int, vm, >=)
SCALAR_BIN(vu, uint32_t, vu, +) SCALAR_BIN(vu, uint32_t, vu, *) SCALAR_BIN(vu, uint32_t, vu, ^)
inline vf operator-(vf a) { vf r; for (int i = 0; i < W; i++) r.v[i] = -a.v[i]; return r; }
inline vm operator!(vm a) { vm r; for (int i = 0; i < W; i++) r.v[i] = !a.v[i]; return r; }
inline vu operator>>(vu a, int s) { vu r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] >> s; return r; }
inline vf &operator+=(vf &a, vf b) { a = a + b; return a; }
inline vf &operator-=(vf &a, vf b) { a = a - b; return a; }

// the GPU's packed f32 pair (v_pk_*_f32): two lane vectors here
struct vf2 { vf x, y; };
inline vf2 operator+(vf2 a, vf2 b) { return vf2{a.x + b.x, a.y + b.y}; }
inline vf2 operator-(vf2 a, vf2 b) { return vf2{a.x - b.x, a.y - b.y}; }
inline vf2 operator*(vf2 a, vf2 b) { return vf2{a.x * b.x, a.y * b.y}; }
inline vf2 operator*(vf a, vf2 b) { return vf2{a * b.x, a * b.y}; }
inline vf2 operator*(vf2 a, vf b) { return vf2{a.x * b, a.y * b}; }
inline vf2 operator*(float a, vf2 b) { return vf2{a * b.x, a * b.y}; }
inline vf2 operator-(vf2 a) { return vf2{-a.x, -a.y}; }

namespace lanes {
inline vf2 pk2(vf a, vf b) { return vf2{a, b}; }
inline vf pk_lo(vf2 a) { return a.x; }
inline vf pk_hi(vf2 a) { return a.y; }
inline vf pk_hsum(vf2 a) { return a.x + a.y; }
#if IRRL_EMU_W == 4
inline vi leg_id() { vi r; for (int i = 0; i < W; i++) r.v[i] = i; return r; }
inline vf legs_sum(vf x) {
  // same association as the two DPP steps: (x_i + x_{i^1}) + (x_{i^2} + x_{i^3})
  vf a, r;
  for (int i = 0; i < 4; i++) a.v[i] = x.v[i] + x.v[i ^ 1];
  for (int i = 0; i < 4; i++) r.v[i] = a.v[i] + a.v[i ^ 2];
  return r;
}
inline vi legs_sum_i(vi x) { int s = x.v[0] + x.v[1] + x.v[2] + x.v[3]; return vi(s); }
template <int K> inline vf legs_bcast(vf x) { return vf(x.v[K]); }
template <int K> inline vi legs_bcast_i(vi x) { return vi(x.v[K]); }
template <int K> inline vu legs_bcast_u(vu x) { return vu(x.v[K]); }
#else
inline vi leg_id() { vi r; for (int i = 0; i < W; i++) r.v[i] = (i >> 2) & 3; return r; }
inline vi sub_id() { vi r; for (int i = 0; i < W; i++) r.v[i] = i & 3; return r; }
// row_ror:4 then row_ror:8 -- lane i adds lane (i + 4) mod 16, then (i + 8) mod 16
inline vf legs_sum(vf x) {
  vf a, r;
  for (int i = 0; i < 16; i++) a.v[i] = x.v[i] + x.v[(i + 4) & 15];
  for (int i = 0; i < 16; i++) r.v[i] = a.v[i] + a.v[(i + 8) & 15];
  return r;
}
inline vi legs_sum_i(vi x) {
  vi a, r;
  for (int i = 0; i < 16; i++) a.v[i] = x.v[i] + x.v[(i + 4) & 15];
  for (int i = 0; i < 16; i++) r.v[i] = a.v[i] + a.v[(i + 8) & 15];
  return r;
}
template <int K> inline vf legs_bcast(vf x) { return vf(x.v[4 * K]); }      // row_newbcast:4K (sub-lane 0 of leg K)
template <int K> inline vi legs_bcast_i(vi x) { return vi(x.v[4 * K]); }
template <int K> inline vu legs_bcast_u(vu x) { return vu(x.v[4 * K]); }
// same sub-lane of the leg D quads away (the rotation direction is immaterial to the algorithms built on it)
template <int N> inline vf row_bcast(vf x) { return vf(x.v[N]); }            // row_newbcast:N
template <int D> inline vf legs_rot(vf x) { vf r; for (int i = 0; i < 16; i++) r.v[i] = x.v[(i + 4 * D) & 15]; return r; }
inline vf sub_sum(vf x) {
  vf a, r;
  for (int i = 0; i < 16; i++) a.v[i] = x.v[i] + x.v[i ^ 1];
  for (int i = 0; i < 16; i++) r.v[i] = a.v[i] + a.v[i ^ 2];
  return r;
}
template <int K> inline vf sub_bcast(vf x) { vf r; for (int i = 0; i < 16; i++) r.v[i] = x.v[(i & ~3) | K]; return r; }
template <int K> inline vi sub_bcast_i(vi x) { vi r; for (int i = 0; i < 16; i++) r.v[i] = x.v[(i & ~3) | K]; return r; }
template <int K> inline vu sub_bcast_u(vu x) { vu r; for (int i = 0; i < 16; i++) r.v[i] = x.v[(i & ~3) | K]; return r; }
template <int K> inline vf sub_bcast_fma(vf x, vf y, vf acc) { return acc + sub_bcast<K>(x) * y; }
template <int D> inline vf legs_rot_fma(vf x, vf y, vf acc) { return acc + legs_rot<D>(x) * y; }
// inclusive suffix sum over the sub-lanes (sub-lane s gets x_s + .. + x_3); REQUIRES x_3 == 0 (quad_perm [1,2,3,3], [2,3,3,3])
inline vf sub_suffix_sum(vf x) {
  vf a, r;
  const int p1[4] = {1, 2, 3, 3}, p2[4] = {2, 3, 3, 3};
  for (int i = 0; i < 16; i++) a.v[i] = x.v[i] + x.v[(i & ~3) | p1[i & 3]];
  for (int i = 0; i < 16; i++) r.v[i] = a.v[i] + a.v[(i & ~3) | p2[i & 3]];
  return r;
}
// inclusive prefix sum over the sub-lanes (sub-lane s gets x_0 + .. + x_s)
inline vf sub_prefix_sum(vf x) {
  vf a, r;
  for (int i = 0; i < 16; i++) a.v[i] = x.v[i] + (((i & 3) >= 1) ? x.v[i - 1] : 0.0f);
  for (int i = 0; i < 16; i++) r.v[i] = a.v[i] + (((i & 3) >= 2) ? a.v[i - 2] : 0.0f);
  return r;
}
#endif
inline vf vsel(vm m, vf a, vf b) { vf r; for (int i = 0; i < W; i++) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
inline vf2 pk_sel(vm m, vf2 a, vf2 b) { return vf2{vsel(m, a.x, b.x), vsel(m, a.y, b.y)}; }
inline vi vsel_i(vm m, vi a, vi b) { vi r; for (int i = 0; i < W; i++) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
inline vu vsel_u(vm m, vu a, vu b) { vu r; for (int i = 0; i < W; i++) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
inline bool wave_any(vm m) { bool r = false; for (int i = 0; i < W; i++) r = r || m.v[i]; return r; }
inline int wave_max_small(vi x) { int r = 0; for (int i = 0; i < W; i++) r = x.v[i] > r ? x.v[i] : r; return r; }
#define LANEWISE_FN(name, expr) inline vf name(vf x) { vf r; for (int i = 0; i < W; i++) { float a = x.v[i]; r.v[i] = (expr); } return r; }
LANEWISE_FN(v_sqrt, std::sqrt(a)) LANEWISE_FN(v_rcp, 1.0f / a) LANEWISE_FN(v_sin, std::sin(a)) LANEWISE_FN(v_cos, std::cos(a))
LANEWISE_FN(v_asin, std::asin(a)) LANEWISE_FN(v_acos, std::acos(a)) LANEWISE_FN(v_exp, std::exp(a)) LANEWISE_FN(v_log, std::log(a))
LANEWISE_FN(v_abs, std::fabs(a)) LANEWISE_FN(v_rsqrt, 1.0f / std::sqrt(a)) LANEWISE_FN(v_floor, std::floor(a))
inline vi f2i(vf x) { vi r; for (int i = 0; i < W; i++) r.v[i] = (int32_t)x.v[i]; return r; }
inline void v_sincos(vf x, vf &s, vf &c) { s = v_sin(x); c = v_cos(x); }
inline vf v_fmod(vf x, vf y) { vf r; for (int i = 0; i < W; i++) r.v[i] = std::fmod(x.v[i], y.v[i]); return r; }
inline vf v_exp_fast(vf x) { vf r; for (int i = 0; i < W; i++) r.v[i] = std::exp2(x.v[i] * 1.4426950408889634f); return r; }
inline vf v_fmod_pos(vf x, vf y, vf inv_y) {
  vf r;
  for (int i = 0; i < W; i++) {
    float q = std::floor(x.v[i] * inv_y.v[i]);
    float t = std::fma(-q, y.v[i], x.v[i]);
    if (t < 0.0f) t += y.v[i];
    if (t >= y.v[i]) t -= y.v[i];
    r.v[i] = t;
  }
  return r;
}
inline vf v_min(vf a, vf b) { return vsel(a < b, a, b); }
inline vf v_max(vf a, vf b) { return vsel(a > b, a, b); }
inline vu to_u(vi x) { vu r; for (int i = 0; i < W; i++) r.v[i] = (uint32_t)x.v[i]; return r; }
inline vf u2f(vu x) { vf r; for (int i = 0; i < W; i++) r.v[i] = (float)x.v[i]; return r; }
inline vf i2f(vi x) { vf r; for (int i = 0; i < W; i++) r.v[i] = (float)x.v[i]; return r; }
inline vu mulhi_u32(vu a, vu b) { vu r; for (int i = 0; i < W; i++) r.v[i] = (uint32_t)(((uint64_t)a.v[i] * b.v[i]) >> 32); return r; }
inline vu mulhi_u32(vu a, uint32_t b) { return mulhi_u32(a, vu(b)); }
inline vf ld(const float *p, vi idx) { vf r; for (int i = 0; i < W; i++) r.v[i] = p[idx.v[i]]; return r; }
inline vi ld_i(const int32_t *p, vi idx) { vi r; for (int i = 0; i < W; i++) r.v[i] = p[idx.v[i]]; return r; }
inline vu ld_u(const uint32_t *p, vi idx) { vu r; for (int i = 0; i < W; i++) r.v[i] = p[idx.v[i]]; return r; }
// masked blocks: the emulation keeps a current lane mask, stm* store under it
// (internal linkage: the 4- and 16-lane emulation libraries live in one test process and must not share this object)
static vm &cur_mask() { static thread_local vm m(true); return m; }
struct mask_scope { vm saved; explicit mask_scope(vm m) : saved(cur_mask()) { cur_mask() = saved & m; } ~mask_scope() { cur_mask() = saved; } };
inline void stm(float *p, vi idx, vf v) { for (int i = 0; i < W; i++) if (cur_mask().v[i]) p[idx.v[i]] = v.v[i]; }
inline void stm_i(int32_t *p, vi idx, vi v) { for (int i = 0; i < W; i++) if (cur_mask().v[i]) p[idx.v[i]] = v.v[i]; }
inline void stm_u(uint32_t *p, vi idx, vu v) { for (int i = 0; i < W; i++) if (cur_mask().v[i]) p[idx.v[i]] = v.v[i]; }
inline void stm_u8(uint8_t *p, vi idx, vi v) { for (int i = 0; i < W; i++) if (cur_mask().v[i]) p[idx.v[i]] = (uint8_t)v.v[i]; }
inline void st_if(vm m, float *p, vi idx, vf v) { for (int i = 0; i < W; i++) if (m.v[i]) p[idx.v[i]] = v.v[i]; }
inline void st_i_if(vm m, int32_t *p, vi idx, vi v) { for (int i = 0; i < W; i++) if (m.v[i]) p[idx.v[i]] = v.v[i]; }
inline void st_u_if(vm m, uint32_t *p, vi idx, vu v) { for (int i = 0; i < W; i++) if (m.v[i]) p[idx.v[i]] = v.v[i]; }
inline void st_u8_if(vm m, uint8_t *p, vi idx, vi v) { for (int i = 0; i < W; i++) if (m.v[i]) p[idx.v[i]] = (uint8_t)v.v[i]; }
}  // namespace lanes
