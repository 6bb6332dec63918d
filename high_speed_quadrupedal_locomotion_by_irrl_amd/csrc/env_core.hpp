// env_core.hpp -- the env.step() hot path of the BlackPanther quadruped task as lane-generic code.
//
// Included AFTER a lane-primitive header (lanes_hip.hpp on the GPU) that defines vf/vi/vu/vm and the
// `lanes::` helpers.  Mapping: one DPP quad per robot, one leg per lane (lanes_hip.hpp).  All control
// flow on per-lane conditions is written as selects; the only branches are wave-uniform
// (`wave_any(...)`, kernel parameters), so a wave of 16 robots runs in lockstep without divergence.
//
// What the reference computes here (file:line under
// /root/reference/IRRL/FlexibleRobotRaisimGym/flex_gym/env/, ENV = env/BlackPanther_V55/Environment.hpp,
// VEC = VectorizedEnvironment.hpp):
//   step prologue ENV:700-708, substep loop ENV:758-774 (PD law, torque_clamp ENV:1273-1312, RaiSim
//   integrate ENV:768), updateObservation ENV:956-1004, contact_information_update ENV:1199-1243,
//   DeepMimicRewardUpdate ENV:1444-1548, command_obs_update ENV:1010-1109, gait_generator_manual
//   ENV:1756-1890, inverse_kinematics ENV:1687-1751, contact_obs_update ENV:1116-1194,
//   isTerminalState ENV:1553-1578, reset ENV:547-635, observe ENV:1248-1268, perAgentStep VEC:352-372,
//   constructor randomisation ENV:435-477.
//
// Dynamics formulation (RaiSim is closed source; this is the build's own):
//   u~ = [R^T v, R^T w, qd];  M_B(q) a + b(q,u~) = tau + sum J^T f,  a = physical accelerations in
//   base-frame components.  M_B has the arrow structure  [[A, B_1..B_4], [B_l^T, C_l]]  (legs couple
//   only through the base), so each lane factors its own 3x3 block C_l, the quad reduces the 6x6 Schur
//   complement S = A - sum_l B_l C_l^-1 B_l^T (21 numbers), every lane Cholesky-factors S redundantly,
//   and all solves with M are a 6x6 triangular solve plus lane-local 3x3 work.  The Delassus operator of
//   the toe contacts is G_ll' = Y_l^T Y_l' + delta_ll' E_l with Y_l = L^-1 K_l^T (6x3, lane-local), so
//   the Gauss-Seidel contact sweep only exchanges one 6-vector z = sum_l Y_l lambda_l inside the quad.
//
// The file may be included more than once with different IRRL_CORE_NS / IRRL_CRUTIAL (env_kernels.hip does: the namespace
// `irrl` decides the meteorite paths by the run-time flag, `irrl_plain` has them compiled out for pools without Crutial).
#include "env_params.h"

#ifndef IRRL_CORE_NS
#define IRRL_CORE_NS irrl
#endif
#ifndef IRRL_CRUTIAL
#define IRRL_CRUTIAL(P) ((P).crutial != 0)   /* Crutial: True -- the meteorite (ENV:273-284, 731-740, 815-861) */
#endif

namespace IRRL_CORE_NS {
using namespace lanes;

// ---------------------------------------------------------------------------------------------
// small vector helpers
// ---------------------------------------------------------------------------------------------
// RULE, the template parameter of the step's device functions: bit 0 = the per-contact rule (EnvParams::contact_rule: 1 = the published one),
// bit 1 = THE SHIPPED SOLVER SETTINGS AS COMPILE-TIME CONSTANTS (ContactSolver bit 1 = simultaneous sweeps, ContactExit = 1, a ContactTolerance
// above zero, ContactIterations = 6, eight substeps per control step: what every shipped configuration runs).  The sweep loop then carries neither the other solver's loop nor the per-sweep tests of two run-time flags: same
// arithmetic, bit-identical results, multi-step kernel 30.0 -> 29.3 -> 29.0 -> 28.7 us per step, one launch per step 40.5 -> 40.1 -> 39.1 -> 38.5 us
// (profiles/r06_ab_default_solver_compile_time_same_box.log).
// The launcher takes the RULE = 3 kernels when the pool's settings are those and the RULE = 1 / 0 kernels otherwise (irrl_env_abi.hip).
#define IRRL_SOLVER_FIXED(RULE) (((RULE) & 2) != 0)
#define IRRL_RULE_SHIPPED 3
// bit 2 = FLAT GROUND (Terrain: False) as a compile-time constant: the toe's substep no longer asks for the height field.  The step kernel and the
// multi-step kernel exist in this form too (RULE = 7; the rollout kernels keep the run-time test): one launch per step 39.5 -> 38.8 us, multi-step
// kernel -0.15 us per step in the probe (profiles/r06_ab_default_solver_compile_time_same_box.log); (x - 0) * 1 is x: bit-identical.
#define IRRL_FLAT_GROUND(RULE) (((RULE) & 4) != 0)
#define IRRL_RULE_SHIPPED_FLAT 7
#define IRRL_UNLIKELY(x) __builtin_expect(!!(x), 0)   /* rare wave-uniform branches: laid out off the hot path */
struct v3 { vf x, y, z; };
struct sym3 { vf xx, xy, xz, yy, yz, zz; };

IRRL_DEV v3 mk3(vf x, vf y, vf z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
IRRL_DEV v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
IRRL_DEV v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
IRRL_DEV v3 operator*(vf s, v3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
IRRL_DEV vf dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
IRRL_DEV v3 cross(v3 a, v3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
IRRL_DEV v3 mul(sym3 A, v3 v) {
  return mk3(A.xx * v.x + A.xy * v.y + A.xz * v.z, A.xy * v.x + A.yy * v.y + A.yz * v.z,
             A.xz * v.x + A.yz * v.y + A.zz * v.z);
}
IRRL_DEV sym3 operator+(sym3 a, sym3 b) {
  sym3 r; r.xx = a.xx + b.xx; r.xy = a.xy + b.xy; r.xz = a.xz + b.xz; r.yy = a.yy + b.yy; r.yz = a.yz + b.yz; r.zz = a.zz + b.zz;
  return r;
}
// I = Ix ex ex^T + Iy ey ey^T + Iz ez ez^T + Iyz (ey ez^T + ez ey^T): body inertia rotated into the base frame
IRRL_DEV sym3 rot_inertia(v3 ex, v3 ey, v3 ez, vf Ix, vf Iy, vf Iz, vf Iyz) {
  sym3 r;
  r.xx = Ix * ex.x * ex.x + Iy * ey.x * ey.x + Iz * ez.x * ez.x + Iyz * (2.0f * ey.x * ez.x);
  r.xy = Ix * ex.x * ex.y + Iy * ey.x * ey.y + Iz * ez.x * ez.y + Iyz * (ey.x * ez.y + ez.x * ey.y);
  r.xz = Ix * ex.x * ex.z + Iy * ey.x * ey.z + Iz * ez.x * ez.z + Iyz * (ey.x * ez.z + ez.x * ey.z);
  r.yy = Ix * ex.y * ex.y + Iy * ey.y * ey.y + Iz * ez.y * ez.y + Iyz * (2.0f * ey.y * ez.y);
  r.yz = Ix * ex.y * ex.z + Iy * ey.y * ey.z + Iz * ez.y * ez.z + Iyz * (ey.y * ez.z + ez.y * ey.z);
  r.zz = Ix * ex.z * ex.z + Iy * ey.z * ey.z + Iz * ez.z * ez.z + Iyz * (2.0f * ey.z * ez.z);
  return r;
}
// same with ey.x == 0 (hip/knee frames share the abad y axis (0, c0, s0))
IRRL_DEV sym3 rot_inertia_y0(v3 ex, v3 ey, v3 ez, vf Ix, vf Iy, vf Iz, vf Iyz) {
  sym3 r;
  r.xx = Ix * ex.x * ex.x + Iz * ez.x * ez.x;
  r.xy = Ix * ex.x * ex.y + Iz * ez.x * ez.y + Iyz * (ez.x * ey.y);
  r.xz = Ix * ex.x * ex.z + Iz * ez.x * ez.z + Iyz * (ez.x * ey.z);
  r.yy = Ix * ex.y * ex.y + Iy * ey.y * ey.y + Iz * ez.y * ez.y + Iyz * (2.0f * ey.y * ez.y);
  r.yz = Ix * ex.y * ex.z + Iy * ey.y * ey.z + Iz * ez.y * ez.z + Iyz * (ey.y * ez.z + ez.y * ey.z);
  r.zz = Ix * ex.z * ex.z + Iy * ey.z * ey.z + Iz * ez.z * ez.z + Iyz * (2.0f * ey.z * ez.z);
  return r;
}
// ... and no product of inertia (shank)
IRRL_DEV sym3 rot_inertia_y0d(v3 ex, v3 ey, v3 ez, vf Ix, vf Iy, vf Iz) {
  sym3 r;
  r.xx = Ix * ex.x * ex.x + Iz * ez.x * ez.x;
  r.xy = Ix * ex.x * ex.y + Iz * ez.x * ez.y;
  r.xz = Ix * ex.x * ex.z + Iz * ez.x * ez.z;
  r.yy = Ix * ex.y * ex.y + Iy * ey.y * ey.y + Iz * ez.y * ez.y;
  r.yz = Ix * ex.y * ex.z + Iy * ey.y * ey.z + Iz * ez.y * ez.z;
  r.zz = Ix * ex.z * ex.z + Iy * ey.z * ey.z + Iz * ez.z * ez.z;
  return r;
}
// inertia about the base origin: I_B + m (|c|^2 1 - c c^T)
IRRL_DEV sym3 shift_to_origin(sym3 I, vf m, v3 c) {
  vf cc = dot(c, c);
  sym3 r;
  r.xx = I.xx + m * (cc - c.x * c.x); r.xy = I.xy - m * c.x * c.y; r.xz = I.xz - m * c.x * c.z;
  r.yy = I.yy + m * (cc - c.y * c.y); r.yz = I.yz - m * c.y * c.z; r.zz = I.zz + m * (cc - c.z * c.z);
  return r;
}
IRRL_DEV v3 legs_sum3(v3 a) { return mk3(legs_sum(a.x), legs_sum(a.y), legs_sum(a.z)); }

// rotation matrix body->world from quaternion (w,x,y,z); rows r0,r1,r2
struct rot3 { v3 r0, r1, r2; };
IRRL_DEV rot3 quat_to_rot(vf w, vf x, vf y, vf z) {
  rot3 R;
  R.r0 = mk3(1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y - w * z), 2.0f * (x * z + w * y));
  R.r1 = mk3(2.0f * (x * y + w * z), 1.0f - 2.0f * (x * x + z * z), 2.0f * (y * z - w * x));
  R.r2 = mk3(2.0f * (x * z - w * y), 2.0f * (y * z + w * x), 1.0f - 2.0f * (x * x + y * y));
  return R;
}
IRRL_DEV v3 rot_mul(rot3 R, v3 v) { return mk3(dot(R.r0, v), dot(R.r1, v), dot(R.r2, v)); }          // R v
IRRL_DEV v3 rot_tmul(rot3 R, v3 v) {                                                                  // R^T v
  return mk3(R.r0.x * v.x + R.r1.x * v.y + R.r2.x * v.z, R.r0.y * v.x + R.r1.y * v.y + R.r2.y * v.z,
             R.r0.z * v.x + R.r1.z * v.y + R.r2.z * v.z);
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG: counter = (env, episode, step, purpose), key = (seed, "IRR1")
// ---------------------------------------------------------------------------------------------
struct rng4 { vf u0, u1, u2, u3; };
// section markers for instruction-count breakdowns of the ISA (tools/isa_sections.py builds with -DIRRL_MARKS)
#if defined(IRRL_MARKS) && defined(__HIP_DEVICE_COMPILE__)
#define IRRL_MARK(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; IRRL_MARK " name); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define IRRL_MARK(name) do { } while (0)
#endif
IRRL_DEV rng4 philox_u01(vu seed, vu env, vu episode, vu step, vu purpose) {
  vu c0 = env, c1 = episode, c2 = step, c3 = purpose;
  vu k0 = seed, k1 = 0x49525231u;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    vu hi0 = mulhi_u32(c0, 0xD2511F53u), lo0 = c0 * 0xD2511F53u;
    vu hi1 = mulhi_u32(c2, 0xCD9E8D57u), lo1 = c2 * 0xCD9E8D57u;
    vu n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 = k0 + 0x9E3779B9u; k1 = k1 + 0xBB67AE85u;
  }
  const float s = 1.0f / 16777216.0f;
  rng4 o;
  o.u0 = u2f(c0 >> 8) * s; o.u1 = u2f(c1 >> 8) * s; o.u2 = u2f(c2 >> 8) * s; o.u3 = u2f(c3 >> 8) * s;
  return o;
}
// 16 consecutive uniforms (purposes p .. p+3): lane l of the quad generates block l, DPP broadcasts
// make all 16 visible to every lane.  out[i] == slot (i&3) of purpose p + (i>>2).
IRRL_DEV void legs_rng16(vu seed, vu env, vu episode, vu step, vu purpose, vf out[16]) {
  rng4 r = philox_u01(seed, env, episode, step, purpose + to_u(leg_id()));
  out[0] = legs_bcast<0>(r.u0); out[1] = legs_bcast<0>(r.u1); out[2] = legs_bcast<0>(r.u2); out[3] = legs_bcast<0>(r.u3);
  out[4] = legs_bcast<1>(r.u0); out[5] = legs_bcast<1>(r.u1); out[6] = legs_bcast<1>(r.u2); out[7] = legs_bcast<1>(r.u3);
  out[8] = legs_bcast<2>(r.u0); out[9] = legs_bcast<2>(r.u1); out[10] = legs_bcast<2>(r.u2); out[11] = legs_bcast<2>(r.u3);
  out[12] = legs_bcast<3>(r.u0); out[13] = legs_bcast<3>(r.u1); out[14] = legs_bcast<3>(r.u2); out[15] = legs_bcast<3>(r.u3);
}
// the same 16 values from a draw the lanes already hold (r = philox_u01(..., purpose + leg) of each leg's lane)
IRRL_DEV void legs_gather16(const rng4 &r, vf out[16]) {
  out[0] = legs_bcast<0>(r.u0); out[1] = legs_bcast<0>(r.u1); out[2] = legs_bcast<0>(r.u2); out[3] = legs_bcast<0>(r.u3);
  out[4] = legs_bcast<1>(r.u0); out[5] = legs_bcast<1>(r.u1); out[6] = legs_bcast<1>(r.u2); out[7] = legs_bcast<1>(r.u3);
  out[8] = legs_bcast<2>(r.u0); out[9] = legs_bcast<2>(r.u1); out[10] = legs_bcast<2>(r.u2); out[11] = legs_bcast<2>(r.u3);
  out[12] = legs_bcast<3>(r.u0); out[13] = legs_bcast<3>(r.u1); out[14] = legs_bcast<3>(r.u2); out[15] = legs_bcast<3>(r.u3);
}
// One step's noise draws made up front.  A step of a noisy configuration consumes the action noise (1 draw per robot, or 4 with
// independent factors) and the three observation-noise vectors (3 draws each, legs 0-2): with 16 lanes per robot that is ONE
// Philox evaluation per lane -- sub-lane 0 of leg l draws OBS_JOINT + l, sub-lane 1 OBS_JVEL + l, sub-lane 2 OBS_NORMAL + l,
// sub-lane 3 the action noise -- instead of four in a row (1 in every lane + 3 in the epilogue).  Same (purpose, slot) addresses,
// same numbers.
struct StepNoise { rng4 joint, jvel, normal; };
// pick element (3*leg + k) of a 12-vector that every lane holds
IRRL_DEV vf pick_leg(const vf v[12], vi leg, int k) {
  return vsel(leg == 0, v[k], vsel(leg == 1, v[3 + k], vsel(leg == 2, v[6 + k], v[9 + k])));
}
IRRL_DEV vf pick4(vf a, vf b, vf c, vf d, vi leg) { return vsel(leg == 0, a, vsel(leg == 1, b, vsel(leg == 2, c, d))); }

// ---------------------------------------------------------------------------------------------
// robot model constants (black_panther.urdf:18-165; numbers in SURVEY 8a-M)
// ---------------------------------------------------------------------------------------------
#define IRRL_PI_REF 3.1415926f          /* ENV:45 */
#define IRRL_TOE_RADIUS 0.0275f         /* URDF:148 */
#define IRRL_TOE_Z (-0.19f)             /* URDF:163 */
#define IRRL_GRAV 9.81f
#define IRRL_L_HIP 0.085f               /* ENV:1949-1952 */
#define IRRL_L_THIGH 0.209f
#define IRRL_L_CALF 0.2175f

// per-lane (per-leg) model + the shared base body
struct LegModel {
  vf mA, mT, mS;          // abad, thigh, shank(+toe) masses
  v3 comA, comT, comS;    // COMs in the body frames
  vf m0; v3 com0;         // base body
  vf mu, rest, rest_thr;  // contact material
  vf dz;                  // shank-joint z offset (ENV:472-476)
  vf sf, sy;              // +1 front / -1 hind ; -1 right / +1 left
};
// shank + toe merged through the fixed joint (URDF:104-165): inertia about the merged COM (z axis only moves)
#define IRRL_S_M1 0.064f
#define IRRL_S_Z1 (-0.0865f)
#define IRRL_S_M2 0.05f
#define IRRL_S_Z2 (-0.19f)

IRRL_DEV void model_signs(LegModel &m, vi leg) {
  m.sf = vsel(leg < 2, 1.0f, -1.0f);
  m.sy = vsel((leg & 1) == 0, -1.0f, 1.0f);
}
IRRL_DEV void model_nominal(LegModel &m, vi leg) {
  model_signs(m, leg);
  m.m0 = 3.72f; m.com0 = mk3(0.0f, 0.0f, -0.003f);
  m.mA = 0.54f; m.comA = mk3(m.sf * 0.058f, m.sy * 0.00485f, 0.0f);
  m.mT = 0.636f; m.comT = mk3(0.0f, -m.sy * 0.019f, -0.01865f);
  const float mt = IRRL_S_M1 + IRRL_S_M2;
  const float zc = (IRRL_S_M1 * IRRL_S_Z1 + IRRL_S_M2 * IRRL_S_Z2) / mt;
  m.mS = mt; m.comS = mk3(0.0f, 0.0f, zc);
  m.mu = 0.6f; m.rest = 0.2f; m.rest_thr = 0.01f;  // ENV:433
  m.dz = 0.0f;
}
// ENV:435-477 with the counter RNG (purposes DR_*); identical draw addresses in the oracle
IRRL_DEV void model_randomize(LegModel &m, vi leg, vu seed, vu env, vu episode) {
  // No contraction in here: which multiply-add pairs of these expressions fuse must not depend on the kernel the function is inlined into --
  // with RandomizePerEpisode the in-step reset runs it inside the step kernel, the multi-step persistent kernels and the rollout kernels,
  // which promise bit-identical pools (round 4: the persistent launch's robots left the step kernel's by an ulp of a link mass after
  // their first reset).  Runs once per episode: the cost is nothing.
#pragma clang fp contract(off)
  model_nominal(m, leg);
  vf u[16];
  rng4 r = philox_u01(seed, env, episode, 0u, IRRL_P_DR_MATERIAL);
  m.mu = r.u0 * 0.6f + 0.4f; m.rest = r.u1 * 0.3f; m.rest_thr = r.u2 * 2.0f;
  legs_rng16(seed, env, episode, 0u, IRRL_P_DR_MASS, u);
  vf f[13];
#pragma unroll
  for (int i = 0; i < 13; i++) f[i] = (u[i] - 0.5f) / 0.5f * 0.15f + 1.0f;
  m.m0 = m.m0 * f[0];
  m.mA = m.mA * pick4(f[1], f[4], f[7], f[10], leg);
  m.mT = m.mT * pick4(f[2], f[5], f[8], f[11], leg);
  m.mS = m.mS * pick4(f[3], f[6], f[9], f[12], leg);
  vf c[48];
  legs_rng16(seed, env, episode, 0u, IRRL_P_DR_COM, c);
  legs_rng16(seed, env, episode, 0u, IRRL_P_DR_COM + 4u, c + 16);
  legs_rng16(seed, env, episode, 0u, IRRL_P_DR_COM + 8u, c + 32);
#pragma unroll
  for (int i = 0; i < 39; i++) c[i] = (2.0f * c[i] - 1.0f) * 0.02f;
  m.com0 = m.com0 + mk3(c[0], c[1], c[2]);
  // body index of leg l: abad 1+3l, thigh 2+3l, shank 3+3l -> element 3*body + axis
  m.comA = m.comA + mk3(pick4(c[3], c[12], c[21], c[30], leg), pick4(c[4], c[13], c[22], c[31], leg), pick4(c[5], c[14], c[23], c[32], leg));
  m.comT = m.comT + mk3(pick4(c[6], c[15], c[24], c[33], leg), pick4(c[7], c[16], c[25], c[34], leg), pick4(c[8], c[17], c[26], c[35], leg));
  m.comS = m.comS + mk3(pick4(c[9], c[18], c[27], c[36], leg), pick4(c[10], c[19], c[28], c[37], leg), pick4(c[11], c[20], c[29], c[38], leg));
  rng4 t = philox_u01(seed, env, episode, 0u, IRRL_P_DR_THIGH);
  m.dz = (t.u0 - 0.5f) / 0.5f * 0.01f;
}

// ---------------------------------------------------------------------------------------------
// curve helpers / IK / torque clamp (ENV:86-156, 1273-1312, 1687-1751)
// ---------------------------------------------------------------------------------------------
IRRL_DEV vf bezier_w(vf s) { return s * s * s + 3.0f * (s * s * (1.0f - s)); }
IRRL_DEV vf gauss_bump(vf x, vf width, vf height) {
  return height * v_exp(-(x - width / 2.0f) * (x - width / 2.0f) / (2.0f * (width / 6.0f) * (width / 6.0f)));
}
// the same bump for width 1 (the only width the gait generator uses, ENV:1835): exponent -18 (x - 1/2)^2
IRRL_DEV vf gauss_bump_w1(vf x, vf height) { return height * v_exp_fast(-18.0f * ((x - 0.5f) * (x - 0.5f))); }
IRRL_DEV void sincos_fast(vf x, vf &s, vf &c);
// phase >= 0: fmod(phase, 1) = phase - floor(phase) exactly; ONE sine of the branch's argument (Cody-Waite, 1e-7 absolute)
IRRL_DEV vf smooth_raw(vf phase, vf slope, vf lam, vf inv_lam, vf inv_one_minus_lam) {
  vf f = phase - v_floor(phase);
  vm first = f < lam;
  vf arg = vsel(first, f * inv_lam, (f - lam) * inv_one_minus_lam) * (2.0f * IRRL_PI_REF);
  vf sn, cs;
  sincos_fast(arg, sn, cs);
  return vsel(first, sn, -sn) * slope + 0.5f;
}
IRRL_DEV vf smooth_function(vf phase, vf slope, vf lam, vf inv_lam, vf inv_one_minus_lam) {
  vf t = smooth_raw(phase, slope, lam, inv_lam, inv_one_minus_lam);
  return vsel(t > 1.0f, 1.0f, vsel(t < 0.0f, 0.0f, t));
}
IRRL_DEV vf smooth_function2(vf phase, vf slope, vf lam, vf inv_lam, vf inv_one_minus_lam) {
  vf t = smooth_raw(phase, slope, lam, inv_lam, inv_one_minus_lam);
  return vsel(t > 1.0f, 0.0f, vsel(t < 0.0f, 1.0f, 1.0f - t));
}
// ENV:1687-1751.  th0/th1/th2 hold the previous ("stale") values on entry; `valid*` report which
// slots were overwritten so the caller can reproduce the shared temp[3] chaining across legs.
IRRL_DEV void inverse_kinematics(vf x, vf y, vf z, vf max_len, vm is_right, vf &th0, vf &th1, vf &th2, vm &ok0, vm &ok1, vm &ok2) {
  const float l_hip = IRRL_L_HIP, l_thigh = IRRL_L_THIGH, l_calf = IRRL_L_CALF;
  vf ll2 = x * x + y * y + z * z;
  vf ll = v_sqrt(ll2);
  vm too_long = ll > max_len;
  vf sc = (max_len - 1e-5f) * v_rsqrt(ll2);
  x = vsel(too_long, x * sc, x); y = vsel(too_long, y * sc, y); z = vsel(too_long, z * sc, z);
  vf root = v_sqrt(y * y * (z * z + y * y - l_hip * l_hip));
  vf iden = v_rcp(z * z + y * y);
  vf temp = vsel(is_right, -z * l_hip - root, z * l_hip + root) * iden;
  ok0 = v_abs(temp) <= 1.0f;
  th0 = vsel(ok0, v_asin(temp), th0);
  vf lr = v_sqrt(x * x + y * y + z * z - l_hip * l_hip);
  lr = vsel(lr > (l_thigh + l_calf), (l_thigh + l_calf - 1e-4f), lr);
  vf t2 = (l_thigh * l_thigh + l_calf * l_calf - lr * lr) * (0.5f / (l_thigh * l_calf)) + 1e-5f;
  ok2 = v_abs(t2) <= 1.0f;
  th2 = vsel(ok2, -(IRRL_PI_REF - v_acos(t2)), th2);
  vf ilr = v_rcp(lr);
  vf a1 = x * ilr;
  vf a2 = (lr * lr + l_thigh * l_thigh - l_calf * l_calf) * ((0.5f / l_thigh) * ilr) - 1e-5f;
  ok1 = (v_abs(a1) <= 1.0f) & (v_abs(a2) <= 1.0f);
  th1 = vsel(ok1, v_acos(a2) - v_asin(a1), th1);
}
// ENV:1273-1312 for one joint; knee (k == 2) carries the 1.55f ratio
IRRL_DEV vf torque_clamp1(vf tau, vf qd, int k, const EnvParams &P) {
  const float ratio = (k == 2) ? 1.55f : 1.0f;
  vf w = qd * ratio;
  vf up = vsel(w > P.w_crit, P.tau_max - (w - P.w_crit) * P.clamp_r, P.tau_max) * ratio;
  vf low = vsel(w < -P.w_crit, (-P.w_max - w) * P.clamp_inv_den * -P.tau_max, -P.tau_max) * ratio;
  return v_max(v_min(tau, up), low);
}

// ---------------------------------------------------------------------------------------------
// lane context: everything one leg-lane keeps in registers across the step
// ---------------------------------------------------------------------------------------------
struct EnvLane {
  // leg-local
  vf q[3], qd[3];
  vf ptl[3], tql[3], tq[3];
  vf jr[3], jrl[3], jdr[3], eer[3];
  vf lamw[3];
  vi in_contact; vf contact;
  vu ccount;              // toe-substeps in the contact list (diagnostic counter, EnvState::contact_count)
#ifdef IRRL_PROFILE_WAVES
  int prof_ranksteps, prof_flags;   // wave-uniform: Gauss-Seidel rank steps executed, bit 0 reset path, bit 1 box path, bits 8.. sweeps
#endif
  // per-env (replicated in the 4 lanes of the quad)
  v3 pos; vf qw, qx, qy, qz; v3 vw, ww;
  vf cmd[3], cmdf[3];
  vf t0; vi frame; vu episode; vf up_height;
  vf ob_cmd[3], ob_phase[2], ob_post[3], ob_omega[3];  // env-level part of the raw observation
  vf ob_q[3], ob_qd[3];                                // leg-level part (joint angles / rates + noise)
  vf obl_env[11], obl_q[3], obl_qd[3];                 // obDouble_last_ (only meaningful with ObsFilter)
  LegModel m;
  // Crutial: True -- the meteorite (replicated in the robot's lanes; untouched otherwise)
  v3 sp, sv; vf srad, smass; vi sdyn;
  // scratch carried from the dynamics to the epilogue
  v3 bodyLinVel, bodyAngVel;
};

// sin/cos for joint angles: Cody-Waite reduction by pi/2 (three-part constant, exact products for |k| < 2^8)
// and the Cephes single-precision minimax polynomials on [-pi/4, pi/4]; ~1e-7 absolute error, branch-free,
// no large-argument slow path (joint angles stay within a few radians).
IRRL_DEV void sincos_fast(vf x, vf &s, vf &c) {
  vf kf = v_floor(x * 0.63661977236758134f + 0.5f);
  vf r = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
  vf z = r * r;
  vf sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  vf cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  vi q = f2i(kf) & 3;
  vm swap = (q & 1) != 0;
  vf ss = vsel(swap, cp, sp), cc = vsel(swap, sp, cp);
  s = vsel((q & 2) != 0, -ss, ss);
  c = vsel(((q + 1) & 2) != 0, -cc, cc);
}

// leg kinematics in the base frame
struct LegKin {
  v3 ay, az;            // abad frame axes (ax = e_x)
  v3 tx, tz;            // thigh frame axes (ty = ay)
  v3 sx, sz;            // shank frame axes (sy = ay)
  v3 h;                 // hip / knee joint axis = -ay
  v3 pA, pT, pS, ptoe;  // joint origins and toe frame origin
};
IRRL_DEV LegKin leg_fk_trig(const LegModel &m, vf s0, vf c0, vf s1, vf c1, vf s12, vf c12) {
  LegKin k;
  k.ay = mk3(0.0f, c0, s0); k.az = mk3(0.0f, -s0, c0);
  k.tx = mk3(c1, -s0 * s1, c0 * s1); k.tz = mk3(-s1, -s0 * c1, c0 * c1);
  k.sx = mk3(c12, -s0 * s12, c0 * s12); k.sz = mk3(-s12, -s0 * c12, c0 * c12);
  k.h = mk3(0.0f, -c0, -s0);
  k.pA = mk3(m.sf * 0.212f, m.sy * 0.051f, 0.0f);
  k.pT = k.pA + (m.sy * 0.085f) * k.ay;
  k.pS = k.pT + (-0.201f + m.dz) * k.tz;
  k.ptoe = k.pS + IRRL_TOE_Z * k.sz;
  return k;
}
IRRL_DEV LegKin leg_fk(const LegModel &m, vf q0, vf q1, vf q2) {
  vf s0, c0, s1, c1, s12, c12;
  sincos_fast(q0, s0, c0); sincos_fast(q1, s1, c1); sincos_fast(q1 + q2, s12, c12);
  return leg_fk_trig(m, s0, c0, s1, c1, s12, c12);
}

// Everything the velocity update needs from the factorised dynamics.
struct LegDyn {
  vf L6[21];      // Cholesky factor of the 6x6 base Schur complement, lower-tri row-major, diagonal stored INVERTED
  sym3 Ci;        // C_l^-1
  vf X[6][3];     // B_l C_l^-1  (so D_l = X^T)
  vf bias_b[6];   // base rows of b (quad-reduced)
  vf bias_l[3];   // leg rows of b
};
#define L6I(i, j) ((i) * ((i) + 1) / 2 + (j))

// forward substitution L y = r (diagonal already inverted)
IRRL_DEV void l6_fwd(const vf L[21], vf r[6]) {
#pragma unroll
  for (int i = 0; i < 6; i++) {
    vf v = r[i];
#pragma unroll
    for (int j = 0; j < i; j++) v -= L[L6I(i, j)] * r[j];
    r[i] = v * L[L6I(i, i)];
  }
}
// back substitution L^T x = y
IRRL_DEV void l6_bwd(const vf L[21], vf r[6]) {
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    vf v = r[i];
#pragma unroll
    for (int j = i + 1; j < 6; j++) v -= L[L6I(j, i)] * r[j];
    r[i] = v * L[L6I(i, i)];
  }
}

// CRBA + RNEA + Schur factorisation for the whole robot, leg-parallel.
// wB: base angular velocity (base comps), a0: fictitious base acceleration = 9.81 * R^T e_z.
IRRL_DEV void leg_dynamics(const LegModel &m, const LegKin &k, const vf qd[3], v3 wB, v3 a0, LegDyn &D) {
  const v3 ex = mk3(1.0f, 0.0f, 0.0f);
  // ---- body COMs and inertias in the base frame ----
  v3 rcA = mk3(m.comA.x, m.comA.y * k.ay.y + m.comA.z * k.az.y, m.comA.y * k.ay.z + m.comA.z * k.az.z);  // ay.x = az.x = 0
  v3 rcT = m.comT.x * k.tx + m.comT.y * k.ay + m.comT.z * k.tz;
  v3 rcS = m.comS.x * k.sx + m.comS.y * k.ay + m.comS.z * k.sz;
  v3 cA = k.pA + rcA, cT = k.pT + rcT, cS = k.pS + rcS;
  sym3 IA;  // URDF:62, diagonal in the abad frame whose x axis is the base x axis
  IA.xx = 0.000391f; IA.xy = 0.0f; IA.xz = 0.0f;
  IA.yy = 0.000739f * k.ay.y * k.ay.y + 0.000488f * k.az.y * k.az.y;
  IA.yz = 0.000739f * k.ay.y * k.ay.z + 0.000488f * k.az.y * k.az.z;
  IA.zz = 0.000739f * k.ay.z * k.ay.z + 0.000488f * k.az.z * k.az.z;
  sym3 IT = rot_inertia_y0(k.tx, k.ay, k.tz, 0.001724f, 0.001907f, 0.000468f, -m.sy * 0.000228f);  // URDF:90
  const float zc = (IRRL_S_M1 * IRRL_S_Z1 + IRRL_S_M2 * IRRL_S_Z2) / (IRRL_S_M1 + IRRL_S_M2);
  const float d1 = IRRL_S_Z1 - zc, d2 = IRRL_S_Z2 - zc;
  const float ISx = 0.000716f + IRRL_S_M1 * d1 * d1 + 0.000025f + IRRL_S_M2 * d2 * d2;           // URDF:116,153
  const float ISy = 0.000721f + IRRL_S_M1 * d1 * d1 + 0.000025f + IRRL_S_M2 * d2 * d2;
  const float ISz = 0.000012f + 0.000025f;
  sym3 IS = rot_inertia_y0d(k.sx, k.ay, k.sz, ISx, ISy, ISz);

  // ---- CRBA: composite (mass, first moment, inertia about the base origin) up the chain ----
  vf mcS = m.mS, mcT = m.mT + m.mS, mcA = m.mA + mcT;
  v3 hS = m.mS * cS, hT = m.mT * cT + hS, hA = m.mA * cA + hT;
  sym3 IoS = shift_to_origin(IS, m.mS, cS);
  sym3 IoT = shift_to_origin(IT, m.mT, cT) + IoS;
  sym3 IoA = shift_to_origin(IA, m.mA, cA) + IoT;
  // columns: P_j = s_j x (h_j - m_j p_j),  L_j = Io_j s_j - h_j x (s_j x p_j)
  v3 dA = hA - mcA * k.pA;
  v3 PA = mk3(0.0f, -dA.z, dA.y);                               // e_x x dA
  v3 exp_ = mk3(0.0f, -k.pA.z, k.pA.y);                         // e_x x pA
  v3 LA = mk3(IoA.xx, IoA.xy, IoA.xz) - cross(hA, exp_);        // Io e_x - h x (e_x x pA)
  v3 PT = cross(k.h, hT - mcT * k.pT), LT = mul(IoT, k.h) - cross(hT, cross(k.h, k.pT));
  v3 PS = cross(k.h, hS - mcS * k.pS), LS = mul(IoS, k.h) - cross(hS, cross(k.h, k.pS));
  vf B[6][3];
  B[0][0] = PA.x; B[1][0] = PA.y; B[2][0] = PA.z; B[3][0] = LA.x; B[4][0] = LA.y; B[5][0] = LA.z;
  B[0][1] = PT.x; B[1][1] = PT.y; B[2][1] = PT.z; B[3][1] = LT.x; B[4][1] = LT.y; B[5][1] = LT.z;
  B[0][2] = PS.x; B[1][2] = PS.y; B[2][2] = PS.z; B[3][2] = LS.x; B[4][2] = LS.y; B[5][2] = LS.z;
  // C_l (3x3 sym): M[k,j] = s_k . (L_j - p_k x P_j), rotor inertia on the diagonal (URDF:56,84,110)
  vf C00 = (LA.x - (k.pA.y * PA.z - k.pA.z * PA.y)) + 0.003708f;
  vf C01 = LT.x - (k.pA.y * PT.z - k.pA.z * PT.y);
  vf C02 = LS.x - (k.pA.y * PS.z - k.pA.z * PS.y);
  vf C11 = dot(k.h, LT - cross(k.pT, PT)) + 0.003708f;
  vf C12 = dot(k.h, LS - cross(k.pT, PS));
  vf C22 = dot(k.h, LS - cross(k.pS, PS)) + 0.008966f;
  // inverse of the symmetric 3x3 by cofactors
  {
    vf a = C11 * C22 - C12 * C12, b = C02 * C12 - C01 * C22, c = C01 * C12 - C02 * C11;
    vf idet = v_rcp(C00 * a + C01 * b + C02 * c);
    D.Ci.xx = a * idet; D.Ci.xy = b * idet; D.Ci.xz = c * idet;
    D.Ci.yy = (C00 * C22 - C02 * C02) * idet; D.Ci.yz = (C01 * C02 - C00 * C12) * idet; D.Ci.zz = (C00 * C11 - C01 * C01) * idet;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    D.X[i][0] = B[i][0] * D.Ci.xx + B[i][1] * D.Ci.xy + B[i][2] * D.Ci.xz;
    D.X[i][1] = B[i][0] * D.Ci.xy + B[i][1] * D.Ci.yy + B[i][2] * D.Ci.yz;
    D.X[i][2] = B[i][0] * D.Ci.xz + B[i][1] * D.Ci.yz + B[i][2] * D.Ci.zz;
  }
  // ---- base block from the quad-reduced whole-robot composite ----
  vf mtot = m.m0 + legs_sum(mcA);
  v3 htot = m.m0 * m.com0 + legs_sum3(hA);
  sym3 I0;  // base body inertia about its COM is diagonal in the base frame (URDF:21)
  I0.xx = 0.016269f; I0.xy = 0.0f; I0.xz = 0.0f; I0.yy = 0.050813f; I0.yz = 0.0f; I0.zz = 0.060989f;
  sym3 Io0 = shift_to_origin(I0, m.m0, m.com0);
  sym3 Iot;
  Iot.xx = Io0.xx + legs_sum(IoA.xx); Iot.xy = Io0.xy + legs_sum(IoA.xy); Iot.xz = Io0.xz + legs_sum(IoA.xz);
  Iot.yy = Io0.yy + legs_sum(IoA.yy); Iot.yz = Io0.yz + legs_sum(IoA.yz); Iot.zz = Io0.zz + legs_sum(IoA.zz);
  // Schur complement S = A - sum_l X_l B_l^T  (lower triangle, 21 entries)
  vf S[21];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) S[L6I(i, j)] = legs_sum(D.X[i][0] * B[j][0] + D.X[i][1] * B[j][1] + D.X[i][2] * B[j][2]);
  // A = [[m 1, -[h]x], [[h]x, Io]] ; lower triangle: rows 3-5 x cols 0-2 hold [h]x
  vf A[21];
#pragma unroll
  for (int i = 0; i < 21; i++) A[i] = 0.0f;
  A[L6I(0, 0)] = mtot; A[L6I(1, 1)] = mtot; A[L6I(2, 2)] = mtot;
  A[L6I(3, 1)] = -htot.z; A[L6I(3, 2)] = htot.y;
  A[L6I(4, 0)] = htot.z; A[L6I(4, 2)] = -htot.x;
  A[L6I(5, 0)] = -htot.y; A[L6I(5, 1)] = htot.x;
  A[L6I(3, 3)] = Iot.xx; A[L6I(4, 3)] = Iot.xy; A[L6I(4, 4)] = Iot.yy; A[L6I(5, 3)] = Iot.xz; A[L6I(5, 4)] = Iot.yz; A[L6I(5, 5)] = Iot.zz;
#pragma unroll
  for (int i = 0; i < 21; i++) S[i] = A[i] - S[i];
  // Cholesky S = L L^T, diagonal stored inverted
#pragma unroll
  for (int j = 0; j < 6; j++) {
    vf d = S[L6I(j, j)];
#pragma unroll
    for (int c = 0; c < j; c++) d -= D.L6[L6I(j, c)] * D.L6[L6I(j, c)];
    vf inv = v_rsqrt(d);
    D.L6[L6I(j, j)] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      vf v = S[L6I(i, j)];
#pragma unroll
      for (int c = 0; c < j; c++) v -= D.L6[L6I(i, c)] * D.L6[L6I(j, c)];
      D.L6[L6I(i, j)] = v * inv;
    }
  }

  // ---- RNEA bias (classical Newton-Euler, gravity folded into a0) ----
  v3 sq0 = mk3(qd[0], 0.0f, 0.0f);
  v3 wA = wB + sq0;
  v3 alA = mk3(0.0f, wB.z * qd[0], -wB.y * qd[0]);  // wB x (qd0 e_x)
  v3 aA = a0 + cross(wB, cross(wB, k.pA));
  v3 sq1 = qd[1] * k.h;
  v3 dT = k.pT - k.pA;
  v3 wT = wA + sq1;
  v3 alT = alA + cross(wA, sq1);
  v3 aT = aA + cross(alA, dT) + cross(wA, cross(wA, dT));
  v3 sq2 = qd[2] * k.h;
  v3 dS = k.pS - k.pT;
  v3 wS = wT + sq2;
  v3 alS = alT + cross(wT, sq2);
  v3 aS = aT + cross(alT, dS) + cross(wT, cross(wT, dS));
  v3 fA = m.mA * (aA + cross(alA, rcA) + cross(wA, cross(wA, rcA)));
  v3 fT = m.mT * (aT + cross(alT, rcT) + cross(wT, cross(wT, rcT)));
  v3 fS = m.mS * (aS + cross(alS, rcS) + cross(wS, cross(wS, rcS)));
  v3 nA = mul(IA, alA) + cross(wA, mul(IA, wA)) + cross(rcA, fA);
  v3 nT = mul(IT, alT) + cross(wT, mul(IT, wT)) + cross(rcT, fT);
  v3 nS = mul(IS, alS) + cross(wS, mul(IS, wS)) + cross(rcS, fS);
  D.bias_l[2] = dot(k.h, nS);
  v3 NT = nT + nS + cross(dS, fS);
  v3 FT = fT + fS;
  D.bias_l[1] = dot(k.h, NT);
  v3 NA = nA + NT + cross(dT, FT);
  v3 FA = fA + FT;
  D.bias_l[0] = NA.x;
  v3 Nleg = NA + cross(k.pA, FA);
  v3 f0 = m.m0 * (a0 + cross(wB, cross(wB, m.com0)));
  v3 n0 = cross(wB, mul(I0, wB)) + cross(m.com0, f0);
  v3 Fb = f0 + legs_sum3(FA);
  v3 Nb = n0 + legs_sum3(Nleg);
  D.bias_b[0] = Fb.x; D.bias_b[1] = Fb.y; D.bias_b[2] = Fb.z; D.bias_b[3] = Nb.x; D.bias_b[4] = Nb.y; D.bias_b[5] = Nb.z;
}

// x = M_B^-1 r for r = (rb: shared base rows, rl: this leg's rows).  Results: xb (replicated), xl.
IRRL_DEV void solve_M(const LegDyn &D, const vf rb[6], const vf rl[3], vf xb[6], vf xl[3]) {
#pragma unroll
  for (int i = 0; i < 6; i++) xb[i] = rb[i] - legs_sum(D.X[i][0] * rl[0] + D.X[i][1] * rl[1] + D.X[i][2] * rl[2]);
  l6_fwd(D.L6, xb);
  l6_bwd(D.L6, xb);
  vf c0 = D.Ci.xx * rl[0] + D.Ci.xy * rl[1] + D.Ci.xz * rl[2];
  vf c1 = D.Ci.xy * rl[0] + D.Ci.yy * rl[1] + D.Ci.yz * rl[2];
  vf c2 = D.Ci.xz * rl[0] + D.Ci.yz * rl[1] + D.Ci.zz * rl[2];
#pragma unroll
  for (int i = 0; i < 6; i++) { c0 -= D.X[i][0] * xb[i]; c1 -= D.X[i][1] * xb[i]; c2 -= D.X[i][2] * xb[i]; }
  xl[0] = c0; xl[1] = c1; xl[2] = c2;
}

// Per-contact constants of the block solve: G^-1 (cofactors), G n and n.G n.
struct ContactBlock { sym3 G, Gi; v3 Gn; vf nGn; };
IRRL_DEV ContactBlock make_contact_block(sym3 G, v3 n) {
  ContactBlock B;
  B.G = G;
  vf a = G.yy * G.zz - G.yz * G.yz, b = G.xz * G.yz - G.xy * G.zz, c = G.xy * G.yz - G.xz * G.yy;
  vf idet = v_rcp(G.xx * a + G.xy * b + G.xz * c);
  B.Gi.xx = a * idet; B.Gi.xy = b * idet; B.Gi.xz = c * idet;
  B.Gi.yy = (G.xx * G.zz - G.xz * G.xz) * idet; B.Gi.yz = (G.xy * G.xz - G.xx * G.yz) * idet; B.Gi.zz = (G.xx * G.yy - G.xy * G.xy) * idet;
  B.Gn = mul(G, n);
  B.nGn = dot(n, B.Gn);
  return B;
}
// one-contact solve (same decision order as the oracle's solve_contact): separating -> 0; sticking solution that pushes
// and lies inside the cone -> keep; else slide: friction mu lam_n along the sticking impulse's tangential direction with
// the normal velocity condition kept exact (normal impulse capped at 5 x the frictionless one near the jamming corner).
IRRL_DEV v3 solve_contact(const ContactBlock &B, v3 c, v3 n, vf vstar, vf mu, vm relevant) {
  vf cn = dot(c, n) - vstar;
  v3 rhs = mk3(vstar * n.x - c.x, vstar * n.y - c.y, vstar * n.z - c.z);
  v3 l = mul(B.Gi, rhs);
  vf ln = dot(l, n);
  v3 lt = l - ln * n;
  vf lt2 = dot(lt, lt);
  vm sticking = (lt2 <= mu * mu * ln * ln) & (ln > 0.0f);   // admissible sticking impulse: pushes, inside the cone
  vm sep = cn >= 0.0f;
#ifdef IRRL_GS_FASTPATH
  // (measured slower on gfx950, 62 vs 57 us per step at 4096 envs: the wave-uniform branch breaks the interleaving of
  // independent dependency chains and most waves hold at least one sliding / lifting foot; kept for reference)
  vm plain = (!sep) & sticking & (ln > 0.0f);
  if (!wave_any(relevant & !plain)) return l;
#endif
  vf mcn = -cn;
  v3 w = n + (mu * v_rsqrt(v_max(lt2, 1e-30f))) * lt;
  vf nGw = dot(B.Gn, w);  // n.G w == (G n).w, G symmetric
  // jamming corner n.G w -> 0: normal impulse capped at 5 x the frictionless one (continuous and bounded)
  v3 slide = (mcn * v_rcp(v_max(nGw, 0.2f * B.nGn))) * w;
  v3 r;
  r.x = vsel(sep, 0.0f, vsel(sticking, l.x, slide.x));
  r.y = vsel(sep, 0.0f, vsel(sticking, l.y, slide.y));
  r.z = vsel(sep, 0.0f, vsel(sticking, l.z, slide.z));
  return r;
}

// ---------------------------------------------------------------------------------------------
// THE PUBLISHED PER-CONTACT RULE (ContactSolver bit 0; EnvParams::contact_rule): RaiSim -- world_->integrate(), ENV:768 -- resolves
// contacts with the per-contact iteration of Hwangbo, Lee & Hutter (RA-L 2018): each single-contact problem solved EXACTLY under
// Signorini, Coulomb and maximum dissipation --
//   opening  (c.n >= v*): lam = 0;   sticking (lam_s = -G^-1 (c - v* n) pushes, inside the cone): lam = lam_s;
//   slipping: the point of {cone boundary} x {v_n+ = v*} that minimises the post-impact kinetic energy
//             h(lam) = 1/2 lam^T G lam + lam^T (c - v* n).
// The paper bisects on the polar angle of that conic; here the same point comes from a 2x2 trust-region problem with ONE scalar
// unknown:  contact frame (t1, t2, n); the normal condition eliminates lam_n = alpha + beta . x (alpha = -(c.n - v*) / G_nn,
// beta = -G_tn / G_nn, x = tangential impulse), so h = 1/2 x^T A x + b^T x with A = G_tt - G_tn G_tn^T / G_nn, b = c_t + alpha G_tn,
// free minimiser x* = -A^-1 b = the sticking impulse.  The cone section |x| <= mu (alpha + beta . x) is, in xi = x / alpha, the
// FIXED ellipse (xi - xi_c)^T P (xi - xi_c) <= 1, P = (1 - mu^2 beta beta^T) s / mu^2, xi_c = mu^2 beta / s, s = 1 - mu^2 |beta|^2
// (focus at the origin, eccentricity mu |beta|).  The A-metric projection of xi* onto it: (A + gamma P)(xi - xi_c) = A (xi* - xi_c),
// gamma >= 0 from (xi - xi_c)^T P (xi - xi_c) = 1, i.e. p(gamma) / sqrt(r(gamma)) = sigma with p = det(A + gamma P), r a quadratic,
// sigma = |xi* - xi_c| -- concave and increasing, so Newton is monotone after its first step from any start; the start is the root
// of the isotropic problem along the far point's direction d, gamma_0 = a (sigma sqrt(rho) - 1) / rho (a = d^T A d, rho = d^T P d),
// and IRRL_MD_NEWTON = 2 steps from there are converged to 2e-9 (f64) on the robot's own sliding problems.
// Caps (not reached on the shipped configurations, mu |beta| <= 0.86 there): eccentricity above sqrt(1 - SMIN) (the conic turns
// into a parabola / hyperbola at the jamming corner) -> beta shortened in the cone section; sigma > SIGMAX (a barely pressing
// contact sliding fast, impulse ~1e-4 of a pressing one) -> far point pulled in; mu < MUMIN -> frictionless, lam = alpha n.
// Everything that depends only on (G, n, mu) is per-substep (ContactBlockMD); the CPU restatement the parity tests compare with
// runs the same steps with the same constants.
// ---------------------------------------------------------------------------------------------
#define IRRL_MD_NEWTON 2
#define IRRL_MD_SMIN 0.04f
#define IRRL_MD_SIGMAX 1.0e4f
#define IRRL_MD_MUMIN 1.0e-6f
// The tangential plane's algebra is written on PAIRS (vf2: one v_pk_*_f32 per pair operation on the GPU): column pairs of the 2x2
// matrices, so that M v = M_col1 v.x + M_col2 v.y is two packed instructions with the halves of v broadcast by op_sel.
struct ContactBlockMD {
  vf2 Tx, Ty, Tz;          // (t1.k, t2.k), k = x, y, z: the frame's tangents, packed by component
  vf2 g, be, zc;           // G_tn, beta = -G_tn / G_nn, centre of the cone section's ellipse
  vf2 AC1, AC2;            // columns of A = G_tt - G_tn G_tn^T / G_nn (symmetric)
  vf2 NI1, NI2;            // columns of -A^-1
  vf2 PC1, PC2, QC1, QC2;  // columns of P and of adj(P)
  vf mu2, ignn, detA, idetA2, detP, detP2, mix;
};
IRRL_DEV ContactBlockMD make_contact_block_md(sym3 G, v3 n, vf mu_in) {
  ContactBlockMD B;
  const vf mu = v_max(mu_in, IRRL_MD_MUMIN);
  B.mu2 = mu * mu;
  // branch-free orthonormal frame around n (Duff et al. 2017)
  const vf sg = vsel(n.z >= 0.0f, 1.0f, -1.0f);
  const vf fa = -v_rcp(sg + n.z), fb = n.x * n.y * fa;
  B.Tx = pk2(1.0f + sg * n.x * n.x * fa, fb);
  B.Ty = pk2(sg * fb, sg + n.y * n.y * fa);
  B.Tz = pk2(-sg * n.x, -n.y);
  // G (t1 | t2) by rows, G n
  const vf2 GTx = G.xx * B.Tx + G.xy * B.Ty + G.xz * B.Tz, GTy = G.xy * B.Tx + G.yy * B.Ty + G.yz * B.Tz, GTz = G.xz * B.Tx + G.yz * B.Ty + G.zz * B.Tz;
  const v3 Gn = mul(G, n);
  const vf2 a1 = pk_lo(B.Tx) * GTx + pk_lo(B.Ty) * GTy + pk_lo(B.Tz) * GTz;     // (t1.G t1, t1.G t2)
  const vf2 a2 = pk_hi(B.Tx) * GTx + pk_hi(B.Ty) * GTy + pk_hi(B.Tz) * GTz;     // (t2.G t1, t2.G t2)
  B.g = Gn.x * B.Tx + Gn.y * B.Ty + Gn.z * B.Tz;
  B.ignn = v_rcp(dot(n, Gn));
  const vf2 gs = B.ignn * B.g;
  B.AC1 = a1 - pk_lo(gs) * B.g;
  B.AC2 = a2 - pk_hi(gs) * B.g;
  B.be = -gs;
  const vf A11 = pk_lo(B.AC1), A12 = pk_hi(B.AC1), A22 = pk_hi(B.AC2);
  B.detA = A11 * A22 - A12 * A12;
  const vf idetA = v_rcp(B.detA);
  B.idetA2 = idetA * idetA;
  B.NI1 = (-idetA) * pk2(A22, -A12);
  B.NI2 = (-idetA) * pk2(-A12, A11);
  const vf mb2 = B.mu2 * pk_hsum(B.be * B.be);
  vf shrink = mu;
  if (IRRL_UNLIKELY(wave_any(mb2 > 1.0f - IRRL_MD_SMIN)))      // the jamming corner: not reached on the shipped configurations
    shrink = vsel(mb2 <= 1.0f - IRRL_MD_SMIN, mu, mu * v_sqrt((1.0f - IRRL_MD_SMIN) * v_rcp(v_max(mb2, 1e-30f))));
  const vf2 e = shrink * B.be;                                   // mu beta (capped)
  const vf e1 = pk_lo(e), e2 = pk_hi(e);
  const vf s = 1.0f - pk_hsum(e * e), is = v_rcp(s), ims = s * v_rcp(B.mu2);
  const vf P11 = (1.0f - e1 * e1) * ims, P12 = -e1 * e2 * ims, P22 = (1.0f - e2 * e2) * ims;
  B.PC1 = pk2(P11, P12); B.PC2 = pk2(P12, P22);
  B.QC1 = pk2(P22, -P12); B.QC2 = pk2(-P12, P11);
  B.detP = P11 * P22 - P12 * P12;
  B.detP2 = 2.0f * B.detP;
  B.mix = A22 * P11 - 2.0f * A12 * P12 + A11 * P22;
  B.zc = (mu * is) * e;
  return B;
}
IRRL_DEV v3 solve_contact_md(const ContactBlockMD &B, v3 c, v3 n, vf vstar, vf mu_in, vm relevant) {
  const vf cn = dot(c, n) - vstar;
  const vm sep = cn >= 0.0f;
  const vf alpha = -cn * B.ignn;
  const vf2 b = c.x * B.Tx + c.y * B.Ty + c.z * B.Tz + alpha * B.g;
  const vf2 x = pk_lo(b) * B.NI1 + pk_hi(b) * B.NI2;              // the sticking impulse's tangential part
  const vf ln = alpha + pk_hsum(B.be * x);
  const vm sticking = (pk_hsum(x * x) <= B.mu2 * ln * ln) & (ln > 0.0f);
  const vm frictionless = mu_in <= IRRL_MD_MUMIN;
  const vf zero = 0.0f;
  vf2 X = pk_sel(frictionless, pk2(zero, zero), x);
  // slipping (wave-uniform skip when no contact whose result is kept slides -- feet standing still, the others in flight: 2.2 us
  // of the step at 4096 envs, same box A/B)
  if (wave_any(relevant & !sep & !sticking & !frictionless)) {
    const vf ia = v_rcp(v_max(alpha, 1e-30f));
    vf2 d = ia * x - B.zc;
    const vf s2 = pk_hsum(d * d);
    const vf isg = v_rsqrt(v_max(s2, 1e-30f));
    const vf sig = v_min(s2 * isg, IRRL_MD_SIGMAX);
    d = isg * d;
    const vf2 w = pk_lo(d) * B.AC1 + pk_hi(d) * B.AC2;            // A d
    const vf2 u = B.detA * d;                                      // adj(A) A d
    const vf2 q = pk_lo(w) * B.QC1 + pk_hi(w) * B.QC2;            // adj(P) A d
    const vf2 Pu = pk_lo(u) * B.PC1 + pk_hi(u) * B.PC2, Pq = pk_lo(q) * B.PC1 + pk_hi(q) * B.PC2;
    const vf c0 = pk_hsum(u * Pu), c1 = 2.0f * pk_hsum(q * Pu), c2 = pk_hsum(q * Pq), c22 = 2.0f * c2;
    const vf rho = c0 * B.idetA2;
    vf gam = v_max(pk_hsum(w * d) * (sig * v_sqrt(rho) - 1.0f) * v_rcp(rho), 0.0f);
#pragma unroll
    for (int it = 0; it < IRRL_MD_NEWTON; it++) {
      const vf p = (B.detP * gam + B.mix) * gam + B.detA, r = (c2 * gam + c1) * gam + c0;
      const vf dp = B.detP2 * gam + B.mix, dr = c22 * gam + c1;
      gam += r * (sig * v_sqrt(r) - p) * v_rcp(dp * r - 0.5f * p * dr);
    }
    const vf kk = sig * v_rcp((B.detP * gam + B.mix) * gam + B.detA);
    const vf2 Xs = alpha * (kk * (u + gam * q) + B.zc);
    X = pk_sel(!sticking & !frictionless, Xs, X);
  }
  const vf lnn = vsel(sep, 0.0f, alpha + pk_hsum(B.be * X));   // normal velocity condition exact
  const vf X1 = vsel(sep, 0.0f, pk_lo(X)), X2 = vsel(sep, 0.0f, pk_hi(X));
  return mk3(X1 * pk_lo(B.Tx) + X2 * pk_hi(B.Tx) + lnn * n.x, X1 * pk_lo(B.Ty) + X2 * pk_hi(B.Ty) + lnn * n.y, X1 * pk_lo(B.Tz) + X2 * pk_hi(B.Tz) + lnn * n.z);
}
// |G^-1 dc|^2: the size of a contact's (linear, sticking) answer to a change dc of its contact-point velocity -- what the NEXT sweep
// would change at this contact if the other contacts' last change dlambda moved its velocity by dc (EnvParams::contact_exit).
// Published rule: G^-1 through the contact frame's Schur complement that the block already holds (alpha' = -dc.n / G_nn,
// x' = -A^-1 (dc_t + alpha' G_tn), lam_n' = alpha' + beta . x'); first rule: the explicit inverse.
IRRL_DEV vf answer_norm2_md(const ContactBlockMD &B, v3 dc, v3 n) {
  const vf alpha = -dot(dc, n) * B.ignn;
  const vf2 b = dc.x * B.Tx + dc.y * B.Ty + dc.z * B.Tz + alpha * B.g;
  const vf2 x = pk_lo(b) * B.NI1 + pk_hi(b) * B.NI2;
  const vf ln = alpha + pk_hsum(B.be * x);
  return pk_hsum(x * x) + ln * ln;
}
IRRL_DEV vf answer_norm2(const ContactBlock &B, v3 dc) {
  const v3 l = mul(B.Gi, dc);
  return dot(l, l);
}

// a contact that is solved once (trunk-box corners, meteorite): block + solve in one go, by the pool's rule (RULE: compile time --
// the step kernel is instantiated once per rule, so neither rule's live values weigh on the other's register allocation)
template <int RULE>
IRRL_DEV v3 solve_contact_once(sym3 G, v3 c, v3 n, vf vstar, vf mu, vm relevant) {
  if (RULE) return solve_contact_md(make_contact_block_md(G, n, mu), c, n, vstar, mu, relevant);
  return solve_contact(make_contact_block(G, n), c, n, vstar, mu, relevant);
}

// height and unit normal of the ground below world point (x, y): bilinear cell of the shared height field
IRRL_DEV void terrain_sample(const EnvParams &P, vf x, vf y, vf &h, v3 &n) {
  // cell coordinate (x - x0) / dx = x / dx + (whole cells ix0 + a fraction): formed as x / dx + fraction, the whole cells added to the
  // INTEGER index.  Formed literally, x - x0 is ~250 m and carries 1.5e-5 m of f32 rounding -- 1.5e-4 of a cell, 100x the rounding of
  // a toe's height on flat ground (round 4: the full-size terrain parity run showed it as a 20x larger position error than on flat
  // ground); this way the rounding is that of |x| / dx <~ 150 cells: 1e-6 m.
  vf fx = x * P.hf_inv_dx + P.hf_fx, fy = y * P.hf_inv_dy + P.hf_fy;
  fx = v_min(v_max(fx, P.hf_xlo), P.hf_xhi);     // the table's edge: 0 <= cell coordinate <= n - 1.001
  fy = v_min(v_max(fy, P.hf_ylo), P.hf_yhi);
  vf fi = v_floor(fx), fj = v_floor(fy);
  vi idx = (f2i(fi) + P.hf_ix0) * P.hf_ny + (f2i(fj) + P.hf_iy0);
  vf tx = fx - fi, ty = fy - fj;
  vf h00 = ld(P.height, idx), h01 = ld(P.height, idx + 1), h10 = ld(P.height, idx + P.hf_ny), h11 = ld(P.height, idx + P.hf_ny + 1);
  vf a = h00 + ty * (h01 - h00), b = h10 + ty * (h11 - h10);
  h = a + tx * (b - a);
  vf dhdx = (b - a) * P.hf_inv_dx;
  vf dhdy = ((h01 - h00) + tx * ((h11 - h10) - (h01 - h00))) * P.hf_inv_dy;
  vf inv = v_rsqrt(dhdx * dhdx + dhdy * dhdy + 1.0f);
  n = mk3(-dhdx * inv, -dhdy * inv, inv);
}


// ---------------------------------------------------------------------------------------------
// Trunk collision box (URDF:26: 0.3 x 0.2 x 0.1 centred on the base origin; collision body "body/0", ENV:242): its eight
// corners are point contacts against the same ground and material as the toes.  Corner b: x = +-0.15 (bit 2 set: -),
// y = +-0.1 (bit 1 set: -), z = -+0.05 (bit 0 set: +) -- the oracle's numbering.  A corner only reaches the ground while
// the robot is falling over (the episode ends at z < 0.15 m or 60 degrees of tilt) or on rough terrain, so everything
// here sits behind wave-uniform tests and is written for clarity, not speed.  Each corner is OWNED by one lane of the
// robot (16 lanes: leg quad (b >> 1), sub-lane (b & 1); 4 lanes: lane (b >> 1), slot (b & 1)).  The toes iterate first; the
// corners follow in ONE pass of sequential impulses (0..7, cold start): each touching corner sees the velocity produced so far,
// solves its own contact exactly and is applied at once -- nothing is re-iterated (what a later corner does to an earlier one
// and to the toes is left to the next 0.25 ms substep).  Corners act on the base only: Y_c = L^-1 Jb_c^T.
// ---------------------------------------------------------------------------------------------
#ifdef IRRL_L16
#define IRRL_NCPL 1
IRRL_DEV vf robot_sum(vf x) { return legs_sum(sub_sum(x)); }
template <int LANE> IRRL_DEV vf robot_bcast(vf x) { return row_bcast<LANE>(x); }   // lane LANE of the robot's 16
IRRL_DEV vi box_corner_id(int) { return leg_id() * 2 + (sub_id() & 1); }
IRRL_DEV vm box_corner_owner() { return sub_id() < 2; }
#else
#define IRRL_NCPL 2
IRRL_DEV vf robot_sum(vf x) { return legs_sum(x); }
template <int LANE> IRRL_DEV vf robot_bcast(vf x) { return legs_bcast<LANE>(x); }
IRRL_DEV vi box_corner_id(int j) { return leg_id() * 2 + j; }
IRRL_DEV vm box_corner_owner() { return vm(true); }
#endif
#define IRRL_BOX_HX 0.15f
#define IRRL_BOX_HY 0.1f
#define IRRL_BOX_HZ 0.05f

struct BoxContacts {
  vi id[IRRL_NCPL];
  vm own[IRRL_NCPL];          // slot j of this lane holds an ACTIVE corner that this lane owns
  vf Y[IRRL_NCPL][3][6];      // rows of K L^-T, K = [1 | -[x]x]
  sym3 G[IRRL_NCPL];          // the corner's own Delassus block
  v3 n[IRRL_NCPL], cfree[IRRL_NCPL];
  vf vstar[IRRL_NCPL];
  vf zc[6];                   // sum over the robot's corners of Y^T lambda (same value in all its lanes)
};
// cheap wave-level pre-test: can any corner of this robot be at or below the highest ground?
IRRL_DEV vm box_near_ground(const EnvParams &P, vf pos_z, v3 r2) {
  vf ext = IRRL_BOX_HX * v_abs(r2.x) + IRRL_BOX_HY * v_abs(r2.y) + IRRL_BOX_HZ * v_abs(r2.z);
  return pos_z - ext <= P.hf_max;
}
// detection + per-corner operators.  ub: free base velocity (after dt M^-1 (tau - b)), vB / wB: base velocity before the substep.
IRRL_DEV bool box_setup(const EnvParams &P, const EnvLane &L, const rot3 &R, const vf L6[21], const vf ub[6], v3 vB, v3 wB, BoxContacts &B) {
  bool any = false;
  const vm owner = box_corner_owner();
#pragma unroll
  for (int i = 0; i < 6; i++) B.zc[i] = 0.0f;
#pragma unroll
  for (int j = 0; j < IRRL_NCPL; j++) {
    const vi b = box_corner_id(j);
    B.id[j] = b;
    v3 x = mk3(vsel((b & 4) != 0, -IRRL_BOX_HX, IRRL_BOX_HX), vsel((b & 2) != 0, -IRRL_BOX_HY, IRRL_BOX_HY), vsel((b & 1) != 0, IRRL_BOX_HZ, -IRRL_BOX_HZ));
    v3 cw = rot_mul(R, x);
    vf hgt = 0.0f;
    v3 nw = mk3(0.0f, 0.0f, 1.0f);
    if (P.terrain) terrain_sample(P, L.pos.x + cw.x, L.pos.y + cw.y, hgt, nw);
    vf gap = (L.pos.z + cw.z - hgt) * nw.z;
    B.own[j] = owner & (gap <= 0.0f);
    B.n[j] = rot_tmul(R, nw);
    if (!wave_any(B.own[j])) {
      B.own[j] = vm(false);
      B.vstar[j] = 0.0f; B.cfree[j] = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int i = 0; i < 6; i++) B.Y[j][r][i] = 0.0f;
      sym3 I; I.xx = 1.0f; I.xy = 0.0f; I.xz = 0.0f; I.yy = 1.0f; I.yz = 0.0f; I.zz = 1.0f;
      B.G[j] = I;
      continue;
    }
    any = true;
    // rows of [1 | -[x]x], then Y row = L^-1 (row)^T
    const vf z0 = 0.0f, o1 = 1.0f;
    vf K[3][6] = {{o1, z0, z0, z0, x.z, -x.y}, {z0, o1, z0, -x.z, z0, x.x}, {z0, z0, o1, x.y, -x.x, z0}};
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int i = 0; i < 6; i++) B.Y[j][r][i] = K[r][i];
      l6_fwd(L6, B.Y[j][r]);
    }
    sym3 G;
    vf g[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = r; c < 3; c++) {
        vf acc = B.Y[j][r][0] * B.Y[j][c][0];
#pragma unroll
        for (int i = 1; i < 6; i++) acc += B.Y[j][r][i] * B.Y[j][c][i];
        g[r][c] = acc;
      }
    G.xx = g[0][0]; G.xy = g[0][1]; G.xz = g[0][2]; G.yy = g[1][1]; G.yz = g[1][2]; G.zz = g[2][2];
    B.G[j] = G;
    v3 vpre = vB + cross(wB, x);
    B.cfree[j] = mk3(ub[0], ub[1], ub[2]) + cross(mk3(ub[3], ub[4], ub[5]), x);
    vf vn = dot(vpre, B.n[j]);
    B.vstar[j] = vsel(vn < -L.m.rest_thr, -L.m.rest * vn, 0.0f);
  }
  return any;
}
// Gauss-Seidel step of corner B (compile-time: its owner lane is a DPP broadcast source).  zt: sum over the toes of
// Y^T lambda (replicated).  Adds |dlambda|^2 to dd in the owner lane; updates B.lam and the replicated B.zc.
// One corner of the pass (compile-time CB: its owner lane is a DPP broadcast source): the corner sees the velocity the toes and
// the corners before it have produced, solves its single-contact problem exactly and is applied at once (B.zc advances).
template <int RULE, int CB>
IRRL_DEV void box_corner_step(BoxContacts &B, vf mu) {
  constexpr int j = (IRRL_NCPL == 2) ? (CB & 1) : 0;
  const vm mine = B.own[j] & (B.id[j] == CB);
  if (!wave_any(mine)) return;
  v3 c = B.cfree[j];
#pragma unroll
  for (int i = 0; i < 6; i++) { c.x += B.Y[j][0][i] * B.zc[i]; c.y += B.Y[j][1][i] * B.zc[i]; c.z += B.Y[j][2][i] * B.zc[i]; }
  v3 ln = solve_contact_once<RULE>(B.G[j], c, B.n[j], B.vstar[j], mu, mine);
  v3 dl = mk3(vsel(mine, ln.x, 0.0f), vsel(mine, ln.y, 0.0f), vsel(mine, ln.z, 0.0f));
  // zc += Y^T dl of the owner lane, broadcast to the robot's lanes
#ifdef IRRL_L16
  constexpr int OWNER = (CB >> 1) * 4 + (CB & 1);
#else
  constexpr int OWNER = CB >> 1;
#endif
#pragma unroll
  for (int i = 0; i < 6; i++) B.zc[i] += robot_bcast<OWNER>(B.Y[j][0][i] * dl.x + B.Y[j][1][i] * dl.y + B.Y[j][2][i] * dl.z);
}
// THE TRUNK-BOX PASS: one pass of sequential impulses over the touching corners (0..7, cold start) behind the toe iteration.
// ub: base twist with the toe impulses already applied.  -> dxb: the change of the base twist the corner impulses cause (the
// caller adds it to ub and takes D dxb off the joint rates); false when no corner of the wave's robots touches.
template <int RULE>
IRRL_DEV bool box_pass(const EnvParams &P, const EnvLane &L, const rot3 &R, const vf L6[21], const vf ub[6], v3 vB, v3 wB, vf dxb[6]) {
  BoxContacts B;
  if (!box_setup(P, L, R, L6, ub, vB, wB, B)) return false;
  box_corner_step<RULE, 0>(B, L.m.mu); box_corner_step<RULE, 1>(B, L.m.mu); box_corner_step<RULE, 2>(B, L.m.mu); box_corner_step<RULE, 3>(B, L.m.mu);
  box_corner_step<RULE, 4>(B, L.m.mu); box_corner_step<RULE, 5>(B, L.m.mu); box_corner_step<RULE, 6>(B, L.m.mu); box_corner_step<RULE, 7>(B, L.m.mu);
#pragma unroll
  for (int i = 0; i < 6; i++) dxb[i] = B.zc[i];
  l6_bwd(L6, dxb);
  return true;
}

// ---------------------------------------------------------------------------------------------
// The meteorite of Crutial: True (Environment.hpp:273-284, 731-740, 815-861).  The reference creates CubeNum "steel" spheres
// at the same point with the same radius, mass and velocity (cube_place_radius = 0, ENV:1976): one sphere of CubeNum times
// the mass here.  Every lane of a robot carries the same copy and does the same arithmetic (no lane specialisation: the path
// runs for ~0.2 s out of every 1 s of a Crutial pool and not at all otherwise).
//   meteoriteAttack(true)  ENV:815-842: parked (STATIC) at gc_ + (0.05, 0, 1.0), radius (t/5 + 1) cube_len, mass t/5 + 0.2 each
//   meteoriteAttack(false) ENV:845-858: a parked sphere becomes DYNAMIC with velocity (gv_[0], gv_[1], -5)
// ---------------------------------------------------------------------------------------------
#define IRRL_CUBE_LEN 0.08f   // ENV:1974
IRRL_DEV void sphere_place(const EnvParams &P, EnvLane &L, v3 base, vf t, vm m) {
  L.sp.x = vsel(m, base.x + 0.05f, L.sp.x); L.sp.y = vsel(m, base.y, L.sp.y); L.sp.z = vsel(m, base.z + 1.0f, L.sp.z);
  L.sv.x = vsel(m, 0.0f, L.sv.x); L.sv.y = vsel(m, 0.0f, L.sv.y); L.sv.z = vsel(m, 0.0f, L.sv.z);
  L.srad = vsel(m, (t * 0.2f + 1.0f) * IRRL_CUBE_LEN, L.srad);
  L.smass = vsel(m, (t * 0.2f + 0.2f) * P.cube_num, L.smass);
  L.sdyn = vsel_i(m, 0, L.sdyn);
}
IRRL_DEV void sphere_release(EnvLane &L, vm m) {
  m = m & (L.sdyn == 0);
  L.sv.x = vsel(m, L.vw.x, L.sv.x); L.sv.y = vsel(m, L.vw.y, L.sv.y); L.sv.z = vsel(m, -5.0f, L.sv.z);
  L.sdyn = vsel_i(m, 1, L.sdyn);
}
// One pass of sequential impulses behind the toes and the corners, released spheres only:
//   sphere - trunk box ("steel"-"steel": mu 0, e 0.95, threshold 0.001, ENV:244): closest point q of the box (URDF:26) to the
//     centre, normal from the sphere into the box; Delassus block = K M^-1 K^T of q (base only, K = [1 | -[q]x]) + 1 / m_s;
//   sphere - ground (this env's default material pair), block 1 / m_s;
// then the sphere integrates (semi-implicit Euler).  ub: base twist so far; -> dxb: what the sphere does to it (false: nothing).
template <int RULE>
IRRL_DEV bool sphere_pass(const EnvParams &P, EnvLane &L, const rot3 &R, const vf L6[21], const vf ub[6], v3 vB, v3 wB, vf dxb[6]) {
  const float dt = P.sim_dt;
  const vm dyn = L.sdyn != 0;
  const v3 sv_pre = L.sv;
  const vf ms_inv = v_rcp(v_max(L.smass, 1e-6f));
  L.sv.z = vsel(dyn, L.sv.z - 9.81f * dt, L.sv.z);
  v3 cB = rot_tmul(R, L.sp - L.pos);
  v3 qB = mk3(v_min(v_max(cB.x, -IRRL_BOX_HX), IRRL_BOX_HX), v_min(v_max(cB.y, -IRRL_BOX_HY), IRRL_BOX_HY), v_min(v_max(cB.z, -IRRL_BOX_HZ), IRRL_BOX_HZ));
  v3 dd = cB - qB;
  vf dist2 = dot(dd, dd);
  const vm hit = dyn & (dist2 < L.srad * L.srad);
  const bool any = wave_any(hit);
  if (any) {
    vf inv = v_rsqrt(v_max(dist2, 1e-18f));
    vm outside = dist2 > 1e-18f;   // centre inside the box: the sphere leaves through the top face
    v3 n = mk3(vsel(outside, -dd.x * inv, 0.0f), vsel(outside, -dd.y * inv, 0.0f), vsel(outside, -dd.z * inv, -1.0f));
    const vf z0 = 0.0f, o1 = 1.0f;
    vf Y[3][6] = {{o1, z0, z0, z0, qB.z, -qB.y}, {z0, o1, z0, -qB.z, z0, qB.x}, {z0, z0, o1, qB.y, -qB.x, z0}};
#pragma unroll
    for (int r = 0; r < 3; r++) l6_fwd(L6, Y[r]);
    vf g[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = r; c < 3; c++) {
        vf acc = Y[r][0] * Y[c][0];
#pragma unroll
        for (int i = 1; i < 6; i++) acc += Y[r][i] * Y[c][i];
        g[r][c] = acc;
      }
    sym3 G;
    G.xx = g[0][0] + ms_inv; G.xy = g[0][1]; G.xz = g[0][2]; G.yy = g[1][1] + ms_inv; G.yz = g[1][2]; G.zz = g[2][2] + ms_inv;
    v3 svB = rot_tmul(R, L.sv), svB_pre = rot_tmul(R, sv_pre);
    v3 c = mk3(ub[0], ub[1], ub[2]) + cross(mk3(ub[3], ub[4], ub[5]), qB) - svB;
    v3 pre = vB + cross(wB, qB) - svB_pre;
    vf vn = dot(pre, n);
    vf vs = vsel(vn < -0.001f, -0.95f * vn, 0.0f);
    v3 lam = solve_contact_once<RULE>(G, c, n, vs, 0.0f, hit);
    lam = mk3(vsel(hit, lam.x, 0.0f), vsel(hit, lam.y, 0.0f), vsel(hit, lam.z, 0.0f));
#pragma unroll
    for (int i = 0; i < 6; i++) dxb[i] = Y[0][i] * lam.x + Y[1][i] * lam.y + Y[2][i] * lam.z;
    l6_bwd(L6, dxb);
    v3 sw = rot_mul(R, svB - ms_inv * lam);
    L.sv.x = vsel(hit, sw.x, L.sv.x); L.sv.y = vsel(hit, sw.y, L.sv.y); L.sv.z = vsel(hit, sw.z, L.sv.z);
  }
  {
    vf hgt = 0.0f;
    v3 nw = mk3(0.0f, 0.0f, 1.0f);
    if (P.terrain) terrain_sample(P, L.sp.x, L.sp.y, hgt, nw);
    const vm gnd = dyn & ((L.sp.z - hgt) * nw.z - L.srad <= 0.0f);
    if (wave_any(gnd)) {
      sym3 G; G.xx = ms_inv; G.xy = 0.0f; G.xz = 0.0f; G.yy = ms_inv; G.yz = 0.0f; G.zz = ms_inv;
      vf vn = dot(sv_pre, nw);
      vf vs = vsel(vn < -L.m.rest_thr, -L.m.rest * vn, 0.0f);
      v3 lam = solve_contact_once<RULE>(G, L.sv, nw, vs, L.m.mu, gnd);
      L.sv.x = vsel(gnd, L.sv.x + ms_inv * lam.x, L.sv.x); L.sv.y = vsel(gnd, L.sv.y + ms_inv * lam.y, L.sv.y); L.sv.z = vsel(gnd, L.sv.z + ms_inv * lam.z, L.sv.z);
    }
  }
  L.sp.x = vsel(dyn, L.sp.x + dt * L.sv.x, L.sp.x); L.sp.y = vsel(dyn, L.sp.y + dt * L.sv.y, L.sp.y); L.sp.z = vsel(dyn, L.sp.z + dt * L.sv.z, L.sp.z);
  return any;
}

// ---------------------------------------------------------------------------------------------
// one physics substep (ENV:761-768): PD + clamp, then the build's integrate()
// ---------------------------------------------------------------------------------------------

#ifdef IRRL_L16
// ---------------------------------------------------------------------------------------------
// 16 lanes per robot: the quad (lane >> 2) is the leg, the sub-lane s = lane & 3 owns body / joint s of that leg
// (0 abad, 1 thigh, 2 shank+toe, 3 spare with zero mass).  Per-body work (inertia rotation, Newton-Euler forces),
// CRBA columns, rows of the contact operators and the joint integration are split over the sub-lanes; composite
// inertias / subtree forces are SUFFIX SUMS over the quad (2 DPP adds), sums over legs are row rotations.
// Values tagged (R) are replicated in the quad, (D) differ per sub-lane.
// ---------------------------------------------------------------------------------------------
template <int RULE>
IRRL_DEV void physics_substep(const EnvParams &P, EnvLane &L, const vf pT[3]) {
  const float dt = P.sim_dt;
  const vi sub = sub_id();
  const vm is0 = sub == 0, is1 = sub == 1, is3 = sub == 3, ge1 = sub >= 1, ge2 = sub >= 2;
#define PICK3(a, b, c) vsel(is0, (a), vsel(is1, (b), (c)))   /* sub-lanes 2 and 3 take c */
  IRRL_MARK("pd");
  // (D) PD law for joint `sub`, 1 % blend with the normalised torque_last, speed-dependent clamp (ENV:762-765, 1273-1312)
  vf q_s = PICK3(L.q[0], L.q[1], L.q[2]), qd_s = PICK3(L.qd[0], L.qd[1], L.qd[2]);
  vf tau_s;
  {
    vf pT_s = PICK3(pT[0], pT[1], pT[2]), tql_s = PICK3(L.tql[0], L.tql[1], L.tql[2]);
    vf kp_s = PICK3(vf(P.kp[0]), vf(P.kp[1]), vf(P.kp[2])), kd_s = PICK3(vf(P.kd[0]), vf(P.kd[1]), vf(P.kd[2]));
    vf t = (pT_s - q_s) * kp_s - qd_s * kd_s;
    t = 0.99f * t + (1.0f - 0.99f) * tql_s;
    vf ratio = vsel(ge2, 1.55f, 1.0f);
    vf w = qd_s * ratio;
    vf up = vsel(w > P.w_crit, P.tau_max - (w - P.w_crit) * P.clamp_r, P.tau_max) * ratio;
    vf low = vsel(w < -P.w_crit, (-P.w_max - w) * P.clamp_inv_den * -P.tau_max, -P.tau_max) * ratio;
    tau_s = v_max(v_min(t, up), low);
    L.tq[0] = sub_bcast<0>(tau_s); L.tq[1] = sub_bcast<1>(tau_s); L.tq[2] = sub_bcast<2>(tau_s);
  }
  IRRL_MARK("fk");
  // (R) base frame quantities
  rot3 R = quat_to_rot(L.qw, L.qx, L.qy, L.qz);
  v3 vB = rot_tmul(R, L.vw), wB = rot_tmul(R, L.ww);
  v3 a0 = IRRL_GRAV * R.r2;
  // FK: each sub-lane evaluates one sincos (q0 | q1 | q1 + q2), the quad shares them
  LegKin k;
  {
    vf ang = PICK3(L.q[0], L.q[1], L.q[1] + L.q[2]), sn, cs;
    sincos_fast(ang, sn, cs);
    k = leg_fk_trig(L.m, sub_bcast<0>(sn), sub_bcast<0>(cs), sub_bcast<1>(sn), sub_bcast<1>(cs), sub_bcast<2>(sn), sub_bcast<2>(cs));
  }
  v3 nB = R.r2;
  vf hgt = 0.0f, nwz = 1.0f;
  if (!IRRL_FLAT_GROUND(RULE) && P.terrain) {
    v3 cw = rot_mul(R, k.ptoe);
    v3 nw;
    terrain_sample(P, L.pos.x + cw.x, L.pos.y + cw.y, hgt, nw);
    nB = rot_tmul(R, nw);
    nwz = nw.z;
  }
  IRRL_MARK("body");
  // (D) the sub-lane's own body: frame, joint axis / origin, inertial parameters (sub-lane 3 carries zero mass)
  const v3 ex = mk3(PICK3(vf(1.0f), k.tx.x, k.sx.x), PICK3(vf(0.0f), k.tx.y, k.sx.y), PICK3(vf(0.0f), k.tx.z, k.sx.z));
  const v3 ez = mk3(PICK3(vf(0.0f), k.tz.x, k.sz.x), PICK3(k.az.y, k.tz.y, k.sz.y), PICK3(k.az.z, k.tz.z, k.sz.z));
  const v3 ey = k.ay;
  const v3 p_s = mk3(PICK3(k.pA.x, k.pT.x, k.pS.x), PICK3(k.pA.y, k.pT.y, k.pS.y), PICK3(k.pA.z, k.pT.z, k.pS.z));
  const v3 ax = mk3(vsel(is0, 1.0f, 0.0f), vsel(is0, 0.0f, k.h.y), vsel(is0, 0.0f, k.h.z));
  const float zc = (IRRL_S_M1 * IRRL_S_Z1 + IRRL_S_M2 * IRRL_S_Z2) / (IRRL_S_M1 + IRRL_S_M2);
  const float d1 = IRRL_S_Z1 - zc, d2 = IRRL_S_Z2 - zc;
  const float ISx = 0.000716f + IRRL_S_M1 * d1 * d1 + 0.000025f + IRRL_S_M2 * d2 * d2;
  const float ISy = 0.000721f + IRRL_S_M1 * d1 * d1 + 0.000025f + IRRL_S_M2 * d2 * d2;
  const float ISz = 0.000012f + 0.000025f;
  const vf m_s = vsel(is3, 0.0f, PICK3(L.m.mA, L.m.mT, L.m.mS));
  const vf live = vsel(is3, 0.0f, 1.0f);
  const v3 com_s = mk3(PICK3(L.m.comA.x, L.m.comT.x, L.m.comS.x), PICK3(L.m.comA.y, L.m.comT.y, L.m.comS.y), PICK3(L.m.comA.z, L.m.comT.z, L.m.comS.z));
  const vf Ix = live * PICK3(vf(0.000391f), vf(0.001724f), vf(ISx)), Iy = live * PICK3(vf(0.000739f), vf(0.001907f), vf(ISy));
  const vf Iz = live * PICK3(vf(0.000488f), vf(0.000468f), vf(ISz)), Iyz = vsel(is1, -L.m.sy * 0.000228f, 0.0f);
  const vf rotor = PICK3(vf(0.003708f), vf(0.003708f), vf(0.008966f));
  v3 rc = com_s.x * ex + com_s.y * ey + com_s.z * ez;
  v3 c = p_s + rc;
  sym3 IB = rot_inertia_y0(ex, ey, ez, Ix, Iy, Iz, Iyz);
  IRRL_MARK("composite");
  // composite (mass, first moment, inertia about the base origin) of the subtree hanging on joint `sub`
  sym3 Io = shift_to_origin(IB, m_s, c);
  vf mc = sub_suffix_sum(m_s);
  v3 hc = mk3(sub_suffix_sum(m_s * c.x), sub_suffix_sum(m_s * c.y), sub_suffix_sum(m_s * c.z));
  sym3 Ioc;
  Ioc.xx = sub_suffix_sum(Io.xx); Ioc.xy = sub_suffix_sum(Io.xy); Ioc.xz = sub_suffix_sum(Io.xz);
  Ioc.yy = sub_suffix_sum(Io.yy); Ioc.yz = sub_suffix_sum(Io.yz); Ioc.zz = sub_suffix_sum(Io.zz);
  IRRL_MARK("crba_ci");
  // CRBA column of joint `sub` (D) and its entries of C_l
  v3 Pc = cross(ax, hc - mc * p_s);
  v3 Lc = mul(Ioc, ax) - cross(hc, cross(ax, p_s));
  vf Bs[6] = {Pc.x, Pc.y, Pc.z, Lc.x, Lc.y, Lc.z};
  vf ck0 = (Lc.x - (k.pA.y * Pc.z - k.pA.z * Pc.y)) + vsel(is0, rotor, 0.0f);        // e_x . (L - pA x P)
  vf ck1 = dot(k.h, Lc - cross(k.pT, Pc)) + vsel(is1, rotor, 0.0f);
  vf ck2 = dot(k.h, Lc - cross(k.pS, Pc)) + vsel(sub == 2, rotor, 0.0f);
  LegDyn D;
  {
    vf C00 = sub_bcast<0>(ck0), C01 = sub_bcast<1>(ck0), C02 = sub_bcast<2>(ck0), C11 = sub_bcast<1>(ck1), C12 = sub_bcast<2>(ck1), C22 = sub_bcast<2>(ck2);
    vf a = C11 * C22 - C12 * C12, b = C02 * C12 - C01 * C22, cc = C01 * C12 - C02 * C11;
    vf idet = v_rcp(C00 * a + C01 * b + C02 * cc);
    D.Ci.xx = a * idet; D.Ci.xy = b * idet; D.Ci.xz = cc * idet;
    D.Ci.yy = (C00 * C22 - C02 * C02) * idet; D.Ci.yz = (C01 * C02 - C00 * C12) * idet; D.Ci.zz = (C00 * C11 - C01 * C01) * idet;
  }
  // column `sub` of C^-1 (zero for the spare sub-lane) and of X = B C^-1
  const vf ci0 = live * PICK3(D.Ci.xx, D.Ci.xy, D.Ci.xz), ci1 = live * PICK3(D.Ci.xy, D.Ci.yy, D.Ci.yz), ci2 = live * PICK3(D.Ci.xz, D.Ci.yz, D.Ci.zz);
  vf Xs[6];
#pragma unroll
  for (int i = 0; i < 6; i++) Xs[i] = sub_bcast_fma<2>(Bs[i], ci2, sub_bcast_fma<1>(Bs[i], ci1, sub_bcast<0>(Bs[i]) * ci0));
  IRRL_MARK("schur");
  // whole-robot composite -> base block A; Schur complement S = A - sum_legs sum_sub X[:,s] B[:,s]^T
  vf mtot = L.m.m0 + sub_bcast<0>(legs_sum(mc));
  v3 htot = L.m.m0 * L.m.com0 + mk3(sub_bcast<0>(legs_sum(hc.x)), sub_bcast<0>(legs_sum(hc.y)), sub_bcast<0>(legs_sum(hc.z)));
  sym3 I0;
  I0.xx = 0.016269f; I0.xy = 0.0f; I0.xz = 0.0f; I0.yy = 0.050813f; I0.yz = 0.0f; I0.zz = 0.060989f;
  sym3 Io0 = shift_to_origin(I0, L.m.m0, L.m.com0);
  sym3 Iot;
  Iot.xx = Io0.xx + sub_bcast<0>(legs_sum(Ioc.xx)); Iot.xy = Io0.xy + sub_bcast<0>(legs_sum(Ioc.xy)); Iot.xz = Io0.xz + sub_bcast<0>(legs_sum(Ioc.xz));
  Iot.yy = Io0.yy + sub_bcast<0>(legs_sum(Ioc.yy)); Iot.yz = Io0.yz + sub_bcast<0>(legs_sum(Ioc.yz)); Iot.zz = Io0.zz + sub_bcast<0>(legs_sum(Ioc.zz));
  vf S[21];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) S[L6I(i, j)] = legs_sum(sub_sum(Xs[i] * Bs[j]));
  {
    vf A[21];
#pragma unroll
    for (int i = 0; i < 21; i++) A[i] = 0.0f;
    A[L6I(0, 0)] = mtot; A[L6I(1, 1)] = mtot; A[L6I(2, 2)] = mtot;
    A[L6I(3, 1)] = -htot.z; A[L6I(3, 2)] = htot.y;
    A[L6I(4, 0)] = htot.z; A[L6I(4, 2)] = -htot.x;
    A[L6I(5, 0)] = -htot.y; A[L6I(5, 1)] = htot.x;
    A[L6I(3, 3)] = Iot.xx; A[L6I(4, 3)] = Iot.xy; A[L6I(4, 4)] = Iot.yy; A[L6I(5, 3)] = Iot.xz; A[L6I(5, 4)] = Iot.yz; A[L6I(5, 5)] = Iot.zz;
#pragma unroll
    for (int i = 0; i < 21; i++) S[i] = A[i] - S[i];
  }
  IRRL_MARK("chol");
#pragma unroll
  for (int j = 0; j < 6; j++) {
    vf d = S[L6I(j, j)];
#pragma unroll
    for (int cidx = 0; cidx < j; cidx++) d -= D.L6[L6I(j, cidx)] * D.L6[L6I(j, cidx)];
    vf inv = v_rsqrt(d);
    D.L6[L6I(j, j)] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      vf v = S[L6I(i, j)];
#pragma unroll
      for (int cidx = 0; cidx < j; cidx++) v -= D.L6[L6I(i, cidx)] * D.L6[L6I(j, cidx)];
      D.L6[L6I(i, j)] = v * inv;
    }
  }
  IRRL_MARK("rnea");
#ifdef IRRL_RNEA_SERIAL
  // RNEA: kinematic recursion down the chain, advanced only as far as the own body (sub-lane s stops after body s)
  v3 sq0 = mk3(L.qd[0], 0.0f, 0.0f);
  v3 w = wB + sq0;
  v3 al = mk3(0.0f, wB.z * L.qd[0], -wB.y * L.qd[0]);
  v3 a = a0 + cross(wB, cross(wB, k.pA));
  {
    v3 sq1 = L.qd[1] * k.h, dT = k.pT - k.pA;
    v3 w1 = w + sq1, al1 = al + cross(w, sq1), a1 = a + cross(al, dT) + cross(w, cross(w, dT));
    v3 sq2 = L.qd[2] * k.h, dS = k.pS - k.pT;
    v3 w2 = w1 + sq2, al2 = al1 + cross(w1, sq2), a2 = a1 + cross(al1, dS) + cross(w1, cross(w1, dS));
    w = mk3(vsel(ge2, w2.x, vsel(ge1, w1.x, w.x)), vsel(ge2, w2.y, vsel(ge1, w1.y, w.y)), vsel(ge2, w2.z, vsel(ge1, w1.z, w.z)));
    al = mk3(vsel(ge2, al2.x, vsel(ge1, al1.x, al.x)), vsel(ge2, al2.y, vsel(ge1, al1.y, al.y)), vsel(ge2, al2.z, vsel(ge1, al1.z, al.z)));
    a = mk3(vsel(ge2, a2.x, vsel(ge1, a1.x, a.x)), vsel(ge2, a2.y, vsel(ge1, a1.y, a.y)), vsel(ge2, a2.z, vsel(ge1, a1.z, a.z)));
  }
#else
  // RNEA: kinematic recursion down the chain as PREFIX sums over the sub-lanes.  Joint s contributes
  //   w += s_s qd_s,   al += w_parent x (s_s qd_s),   a += al_parent x d_s + w_parent x (w_parent x d_s)
  // with d_s the joint origin relative to the parent's; every lane forms its own joint's terms from its parent's
  // (w, al) = own prefix minus own term, then two DPP adds per component accumulate them down the chain.
  v3 sq = qd_s * ax;
  sq = mk3(vsel(is3, 0.0f, sq.x), vsel(is3, 0.0f, sq.y), vsel(is3, 0.0f, sq.z));   // the spare lane repeats the shank: no joint of its own
  v3 w = wB + mk3(sub_prefix_sum(sq.x), sub_prefix_sum(sq.y), sub_prefix_sum(sq.z));
  v3 wp = w - sq;                                              // parent's angular velocity
  v3 dal = cross(wp, sq);
  v3 al = mk3(sub_prefix_sum(dal.x), sub_prefix_sum(dal.y), sub_prefix_sum(dal.z));
  v3 alp = al - dal;                                           // parent's angular acceleration
  v3 dj = mk3(vsel(is3, 0.0f, PICK3(k.pA.x, k.pT.x - k.pA.x, k.pS.x - k.pT.x)), vsel(is3, 0.0f, PICK3(k.pA.y, k.pT.y - k.pA.y, k.pS.y - k.pT.y)),
              vsel(is3, 0.0f, PICK3(k.pA.z, k.pT.z - k.pA.z, k.pS.z - k.pT.z)));
  v3 da = cross(alp, dj) + cross(wp, cross(wp, dj));
  v3 a = a0 + mk3(sub_prefix_sum(da.x), sub_prefix_sum(da.y), sub_prefix_sum(da.z));
#endif
  v3 f = m_s * (a + cross(al, rc) + cross(w, cross(w, rc)));
  v3 n0 = mul(IB, al) + cross(w, mul(IB, w)) + cross(c, f);   // moment about the BASE origin: n + (p_s + rc) x f
  v3 F = mk3(sub_suffix_sum(f.x), sub_suffix_sum(f.y), sub_suffix_sum(f.z));
  v3 N0 = mk3(sub_suffix_sum(n0.x), sub_suffix_sum(n0.y), sub_suffix_sum(n0.z));
  vf b_s = dot(ax, N0 - cross(p_s, F));                         // joint bias = axis . moment about the joint origin
  {
    v3 fb = L.m.m0 * (a0 + cross(wB, cross(wB, L.m.com0)));
    v3 nb = cross(wB, mul(I0, wB)) + cross(L.m.com0, fb);
    D.bias_b[0] = fb.x + sub_bcast<0>(legs_sum(F.x)); D.bias_b[1] = fb.y + sub_bcast<0>(legs_sum(F.y)); D.bias_b[2] = fb.z + sub_bcast<0>(legs_sum(F.z));
    D.bias_b[3] = nb.x + sub_bcast<0>(legs_sum(N0.x)); D.bias_b[4] = nb.y + sub_bcast<0>(legs_sum(N0.y)); D.bias_b[5] = nb.z + sub_bcast<0>(legs_sum(N0.z));
  }
  IRRL_MARK("free");
  // free velocity u_free = u + dt M^-1 (tau - damping qd - b)
  vf rl_s = live * (tau_s - 0.01f * qd_s - b_s);
  vf xb[6];
#pragma unroll
  for (int i = 0; i < 6; i++) xb[i] = -D.bias_b[i] - legs_sum(sub_sum(Xs[i] * rl_s));
  l6_fwd(D.L6, xb);
  l6_bwd(D.L6, xb);
  vf xl_s = sub_bcast_fma<2>(rl_s, ci2, sub_bcast_fma<1>(rl_s, ci1, ci0 * sub_bcast<0>(rl_s)));   // row `sub` of C^-1 (symmetric)
#pragma unroll
  for (int i = 0; i < 6; i++) xl_s -= Xs[i] * xb[i];
  vf ub[6];
  ub[0] = vB.x + dt * xb[0]; ub[1] = vB.y + dt * xb[1]; ub[2] = vB.z + dt * xb[2];
  ub[3] = wB.x + dt * xb[3]; ub[4] = wB.y + dt * xb[4]; ub[5] = wB.z + dt * xb[5];
  vf ul_s = qd_s + dt * xl_s;

  IRRL_MARK("contact_setup");
  // ---- contact: toe sphere against the ground ----
  vf gap = (L.pos.z + dot(R.r2, k.ptoe) - hgt) * nwz - IRRL_TOE_RADIUS;
  vm active = gap <= 0.0f;
#ifdef IRRL_NO_BOX   /* A/B switch of tools/build_variants.py: the build without the trunk-box collider */
  const bool box_near = false;
#else
  const bool box_near = wave_any(box_near_ground(P, L.pos.z, R.r2));
#endif
  if (wave_any(active)) {
    v3 x = k.ptoe - IRRL_TOE_RADIUS * nB;
    // (D) column `sub` of the leg Jacobian, then all of it (R)
    v3 jc = live * cross(ax, x - p_s);
    vf Jl[3][3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
      Jl[0][kk] = (kk == 0) ? sub_bcast<0>(jc.x) : ((kk == 1) ? sub_bcast<1>(jc.x) : sub_bcast<2>(jc.x));
      Jl[1][kk] = (kk == 0) ? sub_bcast<0>(jc.y) : ((kk == 1) ? sub_bcast<1>(jc.y) : sub_bcast<2>(jc.y));
      Jl[2][kk] = (kk == 0) ? sub_bcast<0>(jc.z) : ((kk == 1) ? sub_bcast<1>(jc.z) : sub_bcast<2>(jc.z));
    }
    // row r = sub of every contact operator (sub-lanes 2 and 3 both take row 2; the spare lane's results are never read)
    const vf jl0 = PICK3(Jl[0][0], Jl[1][0], Jl[2][0]), jl1 = PICK3(Jl[0][1], Jl[1][1], Jl[2][1]), jl2 = PICK3(Jl[0][2], Jl[1][2], Jl[2][2]);
    // base columns [1 | -[x]x], row r
    vf jb[6];
    jb[0] = vsel(is0, 1.0f, 0.0f); jb[1] = vsel(is1, 1.0f, 0.0f); jb[2] = vsel(ge2, 1.0f, 0.0f);
    jb[3] = PICK3(vf(0.0f), -x.z, x.y); jb[4] = PICK3(x.z, vf(0.0f), -x.x); jb[5] = PICK3(-x.y, x.x, vf(0.0f));
    // K row = Jb row - Jl row . D, Y row = L^-1 K row
    vf Yr[6];
    const vf njl0 = -jl0, njl1 = -jl1, njl2 = -jl2;
#pragma unroll
    for (int i = 0; i < 6; i++) Yr[i] = sub_bcast_fma<2>(Xs[i], njl2, sub_bcast_fma<1>(Xs[i], njl1, sub_bcast_fma<0>(Xs[i], njl0, jb[i])));
    l6_fwd(D.L6, Yr);
    // JC row = Jl row . C^-1
    vf jc0 = jl0 * D.Ci.xx + jl1 * D.Ci.xy + jl2 * D.Ci.xz, jc1 = jl0 * D.Ci.xy + jl1 * D.Ci.yy + jl2 * D.Ci.yz, jc2 = jl0 * D.Ci.xz + jl1 * D.Ci.yz + jl2 * D.Ci.zz;
    // all three Y rows in every sub-lane
    vf Ya[3][6];
#pragma unroll
    for (int i = 0; i < 6; i++) { Ya[0][i] = sub_bcast<0>(Yr[i]); Ya[1][i] = sub_bcast<1>(Yr[i]); Ya[2][i] = sub_bcast<2>(Yr[i]); }
    // own Delassus block, row r: Y_r . Y_c + JC_r . Jl_c
    vf gr[3];
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
      vf acc = jc0 * Jl[cc][0] + jc1 * Jl[cc][1] + jc2 * Jl[cc][2];
#pragma unroll
      for (int i = 0; i < 6; i++) acc += Yr[i] * Ya[cc][i];
      gr[cc] = acc;
    }
    sym3 G;
    G.xx = sub_bcast<0>(gr[0]); G.xy = sub_bcast<0>(gr[1]); G.xz = sub_bcast<0>(gr[2]);
    G.yy = sub_bcast<1>(gr[1]); G.yz = sub_bcast<1>(gr[2]); G.zz = sub_bcast<2>(gr[2]);
    // per-substep constants of the single-contact solve, by the pool's rule (compile time; the other block does not exist)
    ContactBlock CB; ContactBlockMD CM;
    if (RULE) CM = make_contact_block_md(G, nB, L.m.mu); else CB = make_contact_block(G, nB);
    // contact-point velocity rows: before the step (restitution) and free
    vf ul0 = sub_bcast<0>(ul_s), ul1 = sub_bcast<1>(ul_s), ul2 = sub_bcast<2>(ul_s);
    vf vpre_r = jl0 * L.qd[0] + jl1 * L.qd[1] + jl2 * L.qd[2] + jb[0] * vB.x + jb[1] * vB.y + jb[2] * vB.z + jb[3] * wB.x + jb[4] * wB.y + jb[5] * wB.z;
    vf cfree_r = sub_bcast_fma<2>(ul_s, jl2, sub_bcast_fma<1>(ul_s, jl1, jl0 * ul0));
#pragma unroll
    for (int i = 0; i < 6; i++) cfree_r += jb[i] * ub[i];
    const vf nB_r = live * PICK3(nB.x, nB.y, nB.z);
    vf vn = sub_sum(vpre_r * nB_r);
    vf vstar = vsel(vn < -L.m.rest_thr, -L.m.rest * vn, 0.0f);
    // warm start (world -> base components); zero unless the foot was already in the contact list
    v3 lam = rot_tmul(R, mk3(L.lamw[0], L.lamw[1], L.lamw[2]));
    vm warm = active & (L.in_contact != 0);
    lam.x = vsel(warm, lam.x, 0.0f); lam.y = vsel(warm, lam.y, 0.0f); lam.z = vsel(warm, lam.z, 0.0f);
    // partner blocks, row r: G_{l,p} = Y_l Y_p^T for the three other legs, reached by row rotations.  (The alternative -- the
    // 6-vector z = sum_legs Y_l^T lam_l formed once per sweep, which is what the 4-lane layout does -- costs four DEPENDENT DPP
    // adds per component in this layout's sweep loop: 40.8 us per step against 37.5 with the explicit blocks, same box.)
    vf gx1[3], gx2[3], gx3[3];
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
      vf a1 = Yr[0] * legs_rot<1>(Ya[cc][0]), a2 = Yr[0] * legs_rot<2>(Ya[cc][0]), a3 = Yr[0] * legs_rot<3>(Ya[cc][0]);
#pragma unroll
      for (int i = 1; i < 6; i++) { a1 = legs_rot_fma<1>(Ya[cc][i], Yr[i], a1); a2 = legs_rot_fma<2>(Ya[cc][i], Yr[i], a2); a3 = legs_rot_fma<3>(Ya[cc][i], Yr[i], a3); }
      gx1[cc] = a1; gx2[cc] = a2; gx3[cc] = a3;
    }
  IRRL_MARK("gs");
    // rank of this contact among the robot's active contacts (leg order FR,FL,HR,HL)
    vi rank = 0;
    int nrank = 1;   // (this block only runs when some toe of the wave touches)
    if (!IRRL_SOLVER_FIXED(RULE) && P.contact_jacobi == 0) {
      vi leg = leg_id();
      vi act_i = vsel_i(active, 1, 0);
      vi a0i = legs_bcast_i<0>(act_i), a1i = legs_bcast_i<1>(act_i), a2i = legs_bcast_i<2>(act_i), a3i = legs_bcast_i<3>(act_i);
      rank = vsel_i(leg == 0, 0, vsel_i(leg == 1, a0i, vsel_i(leg == 2, a0i + a1i, a0i + a1i + a2i)));
      nrank = wave_max_small(a0i + a1i + a2i + a3i);
    }
    // ContactSolver bit 1 (default): the contacts of a robot update SIMULTANEOUSLY from the sweep's starting iterate -- one solve
    // per sweep instead of one per contact.  The toes couple only through the heavy base (off-diagonal Delassus blocks are a
    // fraction of the diagonal ones), so this converges almost as fast as Gauss-Seidel (measured on the oracle: 2.4 vs 2.2
    // sweeps per substep, p99 5 vs 4, same fixed point), and the step no longer lasts as long as the wave whose robots
    // happen to have the most feet on the ground (tools/wave_spread.py).
    const bool jacobi = IRRL_SOLVER_FIXED(RULE) || P.contact_jacobi != 0;
    const float tol2 = P.contact_tol * P.contact_tol;
    // row r of the partner legs' coupling sum_p G_lp v_p for a 3-vector v every leg holds (lam or its last change): three independent
    // accumulation chains (one per partner leg), then two adds -- the single resident wave issues a DEPENDENT VALU instruction every
    // ~3.7 ns against 2.3 ns for independent ones, and this chain heads every sweep's critical path
#define IRRL_COUPLING_ROW(V, INIT)                                                                                              \
    (legs_rot_fma<1>((V).z, gx1[2], legs_rot_fma<1>((V).y, gx1[1], legs_rot_fma<1>((V).x, gx1[0], (INIT)))) +                  \
     (legs_rot_fma<2>((V).z, gx2[2], legs_rot_fma<2>((V).y, gx2[1], legs_rot<2>((V).x) * gx2[0])) +                             \
      legs_rot_fma<3>((V).z, gx3[2], legs_rot_fma<3>((V).y, gx3[1], legs_rot<3>((V).x) * gx3[0]))))
    if (jacobi) {
      // ONE loop for both exit rules (EnvParams::contact_exit).  The contact-point velocity is carried from sweep to sweep: the
      // coupling of the impulses' last change dl is what the next sweep's solve needs, and -- answered linearly by each contact
      // (answer_norm2*) -- it is also the PREDICTION of how far that sweep would move the impulses: with contact_exit the loop is
      // left before a sweep that would change them by less than the tolerance, instead of after a sweep that did (the
      // confirming sweep: one solve of ~130-200 VALU instructions per substep, against ~25 for the prediction).
      const bool predicted = IRRL_SOLVER_FIXED(RULE) || P.contact_exit != 0;
      const vf cvr0 = IRRL_COUPLING_ROW(lam, cfree_r);
      v3 cv = mk3(sub_bcast<0>(cvr0), sub_bcast<1>(cvr0), sub_bcast<2>(cvr0));
      const int sweep_cap = IRRL_SOLVER_FIXED(RULE) ? IRRL_SHIPPED_SWEEP_CAP : P.contact_iters;
// (unrolled by two -- the back end emits the same code for 2, 3 and 6 with the cap a constant --: the common one- and two-sweep substeps run through
      // fewer taken branches; multi-step kernel 28.4 -> 27.9 us per step, one launch per step 38.3 -> 37.5 us, same box; the 4-lane layout's loop too: -1 ... -2 %)
_Pragma("unroll 2")
      for (int it = 0; it < sweep_cap; it++) {
#ifdef IRRL_PROFILE_WAVES
        L.prof_ranksteps += 1; L.prof_flags += 256;
#endif
        v3 ln = RULE ? solve_contact_md(CM, cv, nB, vstar, L.m.mu, active) : solve_contact(CB, cv, nB, vstar, L.m.mu, active);
        v3 dl = mk3(vsel(active, ln.x - lam.x, 0.0f), vsel(active, ln.y - lam.y, 0.0f), vsel(active, ln.z - lam.z, 0.0f));
        lam = lam + dl;
        if (it + 1 >= sweep_cap) break;
        vf l2 = 0.0f;
        if (IRRL_SOLVER_FIXED(RULE) || tol2 > 0.0f) {
          l2 = legs_sum(vsel(active, dot(lam, lam), 0.0f));
          if (!predicted) {
            vm unconverged = legs_sum(dot(dl, dl)) > tol2 * l2 + 1e-20f;
            if (!wave_any(unconverged)) break;
          }
        }
        const vf dcr = IRRL_COUPLING_ROW(dl, vf(0.0f));
        const v3 dc = mk3(sub_bcast<0>(dcr), sub_bcast<1>(dcr), sub_bcast<2>(dcr));
        // (asking only when the last change was below 100 tolerances -- a cheap necessary condition in front of the ~25 instructions of
        // the prediction -- was measured and dropped: 18.0 instead of 16.1 sweeps per step, the wave 0.75 us slower, same box)
        if (predicted && (IRRL_SOLVER_FIXED(RULE) || tol2 > 0.0f)) {
          const vf p2 = RULE ? answer_norm2_md(CM, dc, nB) : answer_norm2(CB, dc);
          vm unconverged = legs_sum(vsel(active, p2, 0.0f)) > tol2 * l2 + 1e-20f;
          if (!wave_any(unconverged)) break;
        }
        cv = cv + dc;
      }
    } else {
    for (int it = 0; it < P.contact_iters; it++) {
      vf d2 = 0.0f;
#ifdef IRRL_PROFILE_WAVES
      L.prof_ranksteps += nrank; L.prof_flags += 256;
#endif
      for (int rk = 0; rk < nrank; rk++) {
        const vf cvr = IRRL_COUPLING_ROW(lam, cfree_r);
        v3 cv = mk3(sub_bcast<0>(cvr), sub_bcast<1>(cvr), sub_bcast<2>(cvr));
        vm commit = active & (rank == rk);
        v3 ln = RULE ? solve_contact_md(CM, cv, nB, vstar, L.m.mu, commit) : solve_contact(CB, cv, nB, vstar, L.m.mu, commit);
        v3 dl = mk3(vsel(commit, ln.x - lam.x, 0.0f), vsel(commit, ln.y - lam.y, 0.0f), vsel(commit, ln.z - lam.z, 0.0f));
        lam = lam + dl;
        d2 += dot(dl, dl);
      }
      if (tol2 > 0.0f) {
        vf l2 = legs_sum(vsel(active, dot(lam, lam), 0.0f));
        vm unconverged = legs_sum(d2) > tol2 * l2 + 1e-20f;
        if (!wave_any(unconverged)) break;
      }
    }
    }
#undef IRRL_COUPLING_ROW
  IRRL_MARK("contact_apply");
    lam.x = vsel(active, lam.x, 0.0f); lam.y = vsel(active, lam.y, 0.0f); lam.z = vsel(active, lam.z, 0.0f);
    // z = sum_legs sum_rows Y_r lam_r drives the base; the leg gets C^-1 Jl^T lam - D xb
    const vf lam_r = live * PICK3(lam.x, lam.y, lam.z);
    vf xbc[6];
#pragma unroll
    for (int i = 0; i < 6; i++) xbc[i] = legs_sum(sub_sum(Yr[i] * lam_r));
    l6_bwd(D.L6, xbc);
#pragma unroll
    for (int i = 0; i < 6; i++) ub[i] += xbc[i];
    {
      vf v0 = sub_sum(jc0 * lam_r), v1 = sub_sum(jc1 * lam_r), v2 = sub_sum(jc2 * lam_r);   // columns of JC^T lam
      vf v = PICK3(v0, v1, v2);
#pragma unroll
      for (int i = 0; i < 6; i++) v -= Xs[i] * xbc[i];
      ul_s += v;
    }
    v3 lw = rot_mul(R, lam);
    L.lamw[0] = lw.x; L.lamw[1] = lw.y; L.lamw[2] = lw.z;
  } else {
    L.lamw[0] = 0.0f; L.lamw[1] = 0.0f; L.lamw[2] = 0.0f;
  }
  // trunk-box corners (rare: robots falling over, rough terrain): one pass of sequential impulses behind the toe iteration
  if (IRRL_UNLIKELY(box_near)) {
    vf dxb[6];
    if (box_pass<RULE>(P, L, R, D.L6, ub, vB, wB, dxb)) {
#ifdef IRRL_PROFILE_WAVES
      L.prof_flags |= 2;
#endif
#pragma unroll
      for (int i = 0; i < 6; i++) { ub[i] += dxb[i]; ul_s -= Xs[i] * dxb[i]; }
    }
  }
  if (IRRL_UNLIKELY(IRRL_CRUTIAL(P)) && wave_any(L.sdyn != 0)) {
    vf dxb[6];
    if (sphere_pass<RULE>(P, L, R, D.L6, ub, vB, wB, dxb)) {
#pragma unroll
      for (int i = 0; i < 6; i++) { ub[i] += dxb[i]; ul_s -= Xs[i] * dxb[i]; }
    }
  }
  IRRL_MARK("integrate");
  L.in_contact = vsel_i(active, 1, 0);
  L.ccount = L.ccount + to_u(L.in_contact);
  // back to world-frame gv, then positions (semi-implicit Euler); joint `sub` integrates in its own lane
  L.vw = rot_mul(R, mk3(ub[0], ub[1], ub[2]));
  L.ww = rot_mul(R, mk3(ub[3], ub[4], ub[5]));
  {
    vf qn = q_s + dt * ul_s;
    L.q[0] = sub_bcast<0>(qn); L.q[1] = sub_bcast<1>(qn); L.q[2] = sub_bcast<2>(qn);
    L.qd[0] = sub_bcast<0>(ul_s); L.qd[1] = sub_bcast<1>(ul_s); L.qd[2] = sub_bcast<2>(ul_s);
  }
  L.pos = L.pos + dt * L.vw;
  {
    vf hx = 0.5f * dt * L.ww.x, hy = 0.5f * dt * L.ww.y, hz = 0.5f * dt * L.ww.z;
    vf w1 = L.qw - hx * L.qx - hy * L.qy - hz * L.qz;
    vf x1 = L.qx + hx * L.qw + hy * L.qz - hz * L.qy;
    vf y1 = L.qy - hx * L.qz + hy * L.qw + hz * L.qx;
    vf z1 = L.qz + hx * L.qy - hy * L.qx + hz * L.qw;
    vf inv = v_rsqrt(w1 * w1 + x1 * x1 + y1 * y1 + z1 * z1);
    L.qw = w1 * inv; L.qx = x1 * inv; L.qy = y1 * inv; L.qz = z1 * inv;
  }
  IRRL_MARK("end");
#undef PICK3
}
#else
template <int RULE>
IRRL_DEV void physics_substep(const EnvParams &P, EnvLane &L, const vf pT[3]) {
  const float dt = P.sim_dt;
  // PD law, 1 % blend with the normalised torque_last, speed-dependent clamp
  vf tau[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    vf t = (pT[k] - L.q[k]) * P.kp[k] - L.qd[k] * P.kd[k];
    t = 0.99f * t + (1.0f - 0.99f) * L.tql[k];
    tau[k] = torque_clamp1(t, L.qd[k], k, P);
    L.tq[k] = tau[k];
  }
  rot3 R = quat_to_rot(L.qw, L.qx, L.qy, L.qz);
  v3 vB = rot_tmul(R, L.vw), wB = rot_tmul(R, L.ww);
  v3 a0 = IRRL_GRAV * R.r2;              // gravity folded into the base acceleration: g R^T e_z
  LegKin k = leg_fk(L.m, L.q[0], L.q[1], L.q[2]);
  // ground under this lane's toe (issued early: the four table loads overlap the dynamics below)
  v3 nB = R.r2;                          // contact normal in base components; R^T e_z on the plane
  vf hgt = 0.0f;
  vf nwz = 1.0f;
  if (!IRRL_FLAT_GROUND(RULE) && P.terrain) {
    v3 cw = rot_mul(R, k.ptoe);
    v3 nw;
    terrain_sample(P, L.pos.x + cw.x, L.pos.y + cw.y, hgt, nw);
    nB = rot_tmul(R, nw);
    nwz = nw.z;
  }
  LegDyn D;
  leg_dynamics(L.m, k, L.qd, wB, a0, D);
  // free velocity u_free = u + dt M^-1 (tau - damping qd - b)
  vf rb[6], rl[3], xb[6], xl[3];
#pragma unroll
  for (int i = 0; i < 6; i++) rb[i] = -D.bias_b[i];
#pragma unroll
  for (int j = 0; j < 3; j++) rl[j] = tau[j] - 0.01f * L.qd[j] - D.bias_l[j];  // joint damping URDF:56
  solve_M(D, rb, rl, xb, xl);
  vf ub[6], ul[3];
  ub[0] = vB.x + dt * xb[0]; ub[1] = vB.y + dt * xb[1]; ub[2] = vB.z + dt * xb[2];
  ub[3] = wB.x + dt * xb[3]; ub[4] = wB.y + dt * xb[4]; ub[5] = wB.z + dt * xb[5];
#pragma unroll
  for (int j = 0; j < 3; j++) ul[j] = L.qd[j] + dt * xl[j];

  // ---- contact: toe sphere against the plane z = 0 ----
  // sphere against the locally planar ground: centre-to-tangent-plane distance minus the radius
  vf gap = (L.pos.z + dot(R.r2, k.ptoe) - hgt) * nwz - IRRL_TOE_RADIUS;
  vm active = gap <= 0.0f;
#ifdef IRRL_NO_BOX   /* A/B switch of tools/build_variants.py: the build without the trunk-box collider */
  const bool box_near = false;
#else
  const bool box_near = wave_any(box_near_ground(P, L.pos.z, R.r2));
#endif
  if (wave_any(active)) {
    v3 x = k.ptoe - IRRL_TOE_RADIUS * nB;
    // leg columns of the contact Jacobian
    v3 jA = cross(mk3(1.0f, 0.0f, 0.0f), x - k.pA), jT = cross(k.h, x - k.pT), jS = cross(k.h, x - k.pS);
    vf Jl[3][3] = {{jA.x, jT.x, jS.x}, {jA.y, jT.y, jS.y}, {jA.z, jT.z, jS.z}};
    // base columns [1 | -[x]x]
    vf Jb[3][6] = {{1.0f, 0.0f, 0.0f, 0.0f, x.z, -x.y}, {0.0f, 1.0f, 0.0f, -x.z, 0.0f, x.x}, {0.0f, 0.0f, 1.0f, x.y, -x.x, 0.0f}};
    // K = Jb - Jl D, Y = L^-1 K^T
    vf Y[3][6];
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int i = 0; i < 6; i++) Y[r][i] = Jb[r][i] - (Jl[r][0] * D.X[i][0] + Jl[r][1] * D.X[i][1] + Jl[r][2] * D.X[i][2]);
      l6_fwd(D.L6, Y[r]);
    }
    // JC = Jl C^-1, E = JC Jl^T, G = Y Y^T + E
    vf JC[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      JC[r][0] = Jl[r][0] * D.Ci.xx + Jl[r][1] * D.Ci.xy + Jl[r][2] * D.Ci.xz;
      JC[r][1] = Jl[r][0] * D.Ci.xy + Jl[r][1] * D.Ci.yy + Jl[r][2] * D.Ci.yz;
      JC[r][2] = Jl[r][0] * D.Ci.xz + Jl[r][1] * D.Ci.yz + Jl[r][2] * D.Ci.zz;
    }
    vf Gm[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = r; c < 3; c++) {
        vf acc = JC[r][0] * Jl[c][0] + JC[r][1] * Jl[c][1] + JC[r][2] * Jl[c][2];
#pragma unroll
        for (int i = 0; i < 6; i++) acc += Y[r][i] * Y[c][i];
        Gm[r][c] = acc;
      }
    sym3 G; G.xx = Gm[0][0]; G.xy = Gm[0][1]; G.xz = Gm[0][2]; G.yy = Gm[1][1]; G.yz = Gm[1][2]; G.zz = Gm[2][2];
    ContactBlock CB; ContactBlockMD CM;
    if (RULE) CM = make_contact_block_md(G, nB, L.m.mu); else CB = make_contact_block(G, nB);
    // contact-point velocities: before the step (restitution) and free
    vf vpre[3], cfree[3];
    const vf upre[6] = {vB.x, vB.y, vB.z, wB.x, wB.y, wB.z};
#pragma unroll
    for (int r = 0; r < 3; r++) {
      vf a1 = Jl[r][0] * L.qd[0] + Jl[r][1] * L.qd[1] + Jl[r][2] * L.qd[2];
      vf a2 = Jl[r][0] * ul[0] + Jl[r][1] * ul[1] + Jl[r][2] * ul[2];
#pragma unroll
      for (int i = 0; i < 6; i++) { a1 += Jb[r][i] * upre[i]; a2 += Jb[r][i] * ub[i]; }
      vpre[r] = a1; cfree[r] = a2;
    }
    vf vn = vpre[0] * nB.x + vpre[1] * nB.y + vpre[2] * nB.z;
    vf vstar = vsel(vn < -L.m.rest_thr, -L.m.rest * vn, 0.0f);
    // warm start (world -> base components); zero unless the foot was already in the contact list
    v3 lam = rot_tmul(R, mk3(L.lamw[0], L.lamw[1], L.lamw[2]));
    vm warm = active & (L.in_contact != 0);
    lam.x = vsel(warm, lam.x, 0.0f); lam.y = vsel(warm, lam.y, 0.0f); lam.z = vsel(warm, lam.z, 0.0f);
    vi leg = leg_id();
    // rank of this contact among the robot's active contacts (leg order FR,FL,HR,HL)
    vi act_i = vsel_i(active, 1, 0);
    vi a0i = legs_bcast_i<0>(act_i), a1i = legs_bcast_i<1>(act_i), a2i = legs_bcast_i<2>(act_i);
    vi rank = vsel_i(leg == 0, 0, vsel_i(leg == 1, a0i, vsel_i(leg == 2, a0i + a1i, a0i + a1i + a2i)));
    int nrank = wave_max_small(legs_sum_i(act_i));
    const bool jacobi = IRRL_SOLVER_FIXED(RULE) || P.contact_jacobi != 0;   // ContactSolver bit 1: simultaneous updates (see the 16-lane instantiation)
    const float tol2 = P.contact_tol * P.contact_tol;
    // velocity at this contact from the OTHER legs' 3-vectors v (impulses, or their last change): Y_l (z - Y_l^T v_l), z = sum_legs
    // Y_l^T v_l (G_{l,p} = Y_l Y_p^T).  One 6-vector reduction per solve instead of round 1's explicit partner blocks (4 x 9 x 6 FMAs
    // + 72 broadcasts per substep): 90.2 us against 99.4 us per step at 32768 envs, same box.
    auto coupling = [&](const v3 &v, v3 acc) {
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const vf w = Y[0][i] * v.x + Y[1][i] * v.y + Y[2][i] * v.z;
        const vf o = legs_sum(w) - w;
        acc.x += Y[0][i] * o; acc.y += Y[1][i] * o; acc.z += Y[2][i] * o;
      }
      return acc;
    };
    if (jacobi) {
      // one loop for both exit rules, contact-point velocity carried from sweep to sweep (see the 16-lane instantiation)
      const bool predicted = IRRL_SOLVER_FIXED(RULE) || P.contact_exit != 0;
      v3 cv = coupling(lam, mk3(cfree[0], cfree[1], cfree[2]));
      const int sweep_cap = IRRL_SOLVER_FIXED(RULE) ? IRRL_SHIPPED_SWEEP_CAP : P.contact_iters;
      if (nrank > 0)
_Pragma("unroll 2")
      for (int it = 0; it < sweep_cap; it++) {
        v3 ln = RULE ? solve_contact_md(CM, cv, nB, vstar, L.m.mu, active) : solve_contact(CB, cv, nB, vstar, L.m.mu, active);
        v3 dl = mk3(vsel(active, ln.x - lam.x, 0.0f), vsel(active, ln.y - lam.y, 0.0f), vsel(active, ln.z - lam.z, 0.0f));
        lam = lam + dl;
        if (it + 1 >= sweep_cap) break;
        vf l2 = 0.0f;
        if (IRRL_SOLVER_FIXED(RULE) || tol2 > 0.0f) {
          l2 = legs_sum(vsel(active, dot(lam, lam), 0.0f));
          if (!predicted) {
            vm unconverged = legs_sum(dot(dl, dl)) > tol2 * l2 + 1e-20f;
            if (!wave_any(unconverged)) break;
          }
        }
        const v3 dc = coupling(dl, mk3(0.0f, 0.0f, 0.0f));
        if (predicted && (IRRL_SOLVER_FIXED(RULE) || tol2 > 0.0f)) {
          const vf p2 = RULE ? answer_norm2_md(CM, dc, nB) : answer_norm2(CB, dc);
          vm unconverged = legs_sum(vsel(active, p2, 0.0f)) > tol2 * l2 + 1e-20f;
          if (!wave_any(unconverged)) break;
        }
        cv = cv + dc;
      }
    } else {
    for (int it = 0; it < P.contact_iters; it++) {
      vf d2 = 0.0f;
      for (int rk = 0; rk < nrank; rk++) {
        v3 cv = coupling(lam, mk3(cfree[0], cfree[1], cfree[2]));
        vm commit = active & (rank == rk);
        v3 ln = RULE ? solve_contact_md(CM, cv, nB, vstar, L.m.mu, commit) : solve_contact(CB, cv, nB, vstar, L.m.mu, commit);
        v3 dl = mk3(vsel(commit, ln.x - lam.x, 0.0f), vsel(commit, ln.y - lam.y, 0.0f), vsel(commit, ln.z - lam.z, 0.0f));
        lam = lam + dl;
        d2 += dot(dl, dl);
      }
      // build-defined early exit: every robot of the wave has sum |dlam|^2 <= tol^2 sum |lam|^2
      if (tol2 > 0.0f) {
        vf l2 = legs_sum(vsel(active, dot(lam, lam), 0.0f));
        vm unconverged = legs_sum(d2) > tol2 * l2 + 1e-20f;
        if (!wave_any(unconverged)) break;
      }
    }
    }
    // z = sum_l Y_l^T lam_l (6-vector shared by the quad) drives the base velocity update
    lam.x = vsel(active, lam.x, 0.0f); lam.y = vsel(active, lam.y, 0.0f); lam.z = vsel(active, lam.z, 0.0f);
    vf z[6];
#pragma unroll
    for (int i = 0; i < 6; i++) z[i] = legs_sum(Y[0][i] * lam.x + Y[1][i] * lam.y + Y[2][i] * lam.z);

    // velocity update: base part L^-T z, leg part C^-1 Jl^T lam - D xb
    vf xbc[6];
#pragma unroll
    for (int i = 0; i < 6; i++) xbc[i] = z[i];
    l6_bwd(D.L6, xbc);
#pragma unroll
    for (int i = 0; i < 6; i++) ub[i] += xbc[i];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      vf v = JC[0][j] * lam.x + JC[1][j] * lam.y + JC[2][j] * lam.z;
#pragma unroll
      for (int i = 0; i < 6; i++) v -= D.X[i][j] * xbc[i];
      ul[j] += v;
    }
    v3 lw = rot_mul(R, lam);
    L.lamw[0] = lw.x; L.lamw[1] = lw.y; L.lamw[2] = lw.z;
  } else {
    L.lamw[0] = 0.0f; L.lamw[1] = 0.0f; L.lamw[2] = 0.0f;
  }
  // trunk-box corners (rare: robots falling over, rough terrain): one pass of sequential impulses behind the toe iteration
  if (IRRL_UNLIKELY(box_near)) {
    vf dxb[6];
    if (box_pass<RULE>(P, L, R, D.L6, ub, vB, wB, dxb)) {
#pragma unroll
      for (int i = 0; i < 6; i++) {
        ub[i] += dxb[i];
#pragma unroll
        for (int j = 0; j < 3; j++) ul[j] -= D.X[i][j] * dxb[i];
      }
    }
  }
  if (IRRL_UNLIKELY(IRRL_CRUTIAL(P)) && wave_any(L.sdyn != 0)) {
    vf dxb[6];
    if (sphere_pass<RULE>(P, L, R, D.L6, ub, vB, wB, dxb)) {
#pragma unroll
      for (int i = 0; i < 6; i++) {
        ub[i] += dxb[i];
#pragma unroll
        for (int j = 0; j < 3; j++) ul[j] -= D.X[i][j] * dxb[i];
      }
    }
  }
  L.in_contact = vsel_i(active, 1, 0);
  L.ccount = L.ccount + to_u(L.in_contact);
  // back to world-frame gv, then positions (semi-implicit Euler)
  L.vw = rot_mul(R, mk3(ub[0], ub[1], ub[2]));
  L.ww = rot_mul(R, mk3(ub[3], ub[4], ub[5]));
#pragma unroll
  for (int j = 0; j < 3; j++) { L.qd[j] = ul[j]; L.q[j] += dt * ul[j]; }
  L.pos = L.pos + dt * L.vw;
  {
    vf hx = 0.5f * dt * L.ww.x, hy = 0.5f * dt * L.ww.y, hz = 0.5f * dt * L.ww.z;
    vf w1 = L.qw - hx * L.qx - hy * L.qy - hz * L.qz;
    vf x1 = L.qx + hx * L.qw + hy * L.qz - hz * L.qy;
    vf y1 = L.qy - hx * L.qz + hy * L.qw + hz * L.qx;
    vf z1 = L.qz + hx * L.qy - hy * L.qx + hz * L.qw;
    vf inv = v_rsqrt(w1 * w1 + x1 * x1 + y1 * y1 + z1 * z1);
    L.qw = w1 * inv; L.qx = x1 * inv; L.qy = y1 * inv; L.qz = z1 * inv;
  }
}

#endif  // IRRL_L16

// ---------------------------------------------------------------------------------------------
// task logic
// ---------------------------------------------------------------------------------------------
IRRL_DEV vf env_time(const EnvParams &P, const EnvLane &L) { return L.t0 + i2f(L.frame) * P.control_dt; }

// ENV:1756-1890 for this lane's leg.  The reference keeps ONE temp[3] across the legs (and across the
// two passes of the first call), so a failed asin/acos slot inherits the previous leg's angle: the
// chain is replayed with three DPP hand-offs.
IRRL_DEV void gait_leg_pass(const EnvParams &P, const EnvLane &L, vf t_eval, vf gait_step, vf side_step, vf rot_step, vf up_height,
                            vf prev[3] /* temp[] entering leg 0 */, vf th_out[3], v3 &toe_out) {
  vi leg = leg_id();
  vf phase_l = pick4(P.phase[0], P.phase[1], P.phase[2], P.phase[3], leg);
  vf rp = v_fmod_pos(t_eval + phase_l * P.period, P.period, P.inv_period) * P.inv_period;
  vf anti = vsel(leg < 2, 1.0f, -1.0f);
  vf hx = gait_step / 2.0f, hy = side_step / 2.0f + anti * rot_step / 2.0f, hy2 = -side_step / 2.0f + -anti * rot_step / 2.0f;
  // stance: bezier from (+hx, hy) to (-hx, hy2); swing: from (-hx, hy2) to (+hx, hy) with a gaussian lift
  vm stance = rp < P.lam;
  vf s = vsel(stance, rp * P.inv_lam, (rp - P.lam) * P.inv_one_minus_lam);
  vf bw = bezier_w(s);
  vf p0x = vsel(stance, hx, -hx), pfx = vsel(stance, -hx, hx);
  vf p0y = vsel(stance, hy, hy2), pfy = vsel(stance, hy2, hy);
  v3 toe;
  toe.x = p0x + bw * (pfx - p0x);
  toe.y = p0y + bw * (pfy - p0y);
  vf zst = -P.stand_height + bw * (-P.stand_height - -P.stand_height);
  toe.z = vsel(stance, zst, -P.stand_height + gauss_bump_w1(s, up_height));
  vf toff = pick4(-IRRL_L_HIP + P.lean_front, IRRL_L_HIP - P.lean_front, -IRRL_L_HIP + P.lean_hind, IRRL_L_HIP - P.lean_hind, leg);
  vf th0 = 0.0f, th1 = 0.0f, th2 = 0.0f;
  vm ok0, ok1, ok2;
  inverse_kinematics(toe.x, toe.y + toff, toe.z, P.max_len, (leg & 1) == 0, th0, th1, th2, ok0, ok1, ok2);
  // chain the stale values: leg 0 falls back to prev[], leg i to leg i-1's final value
  vf f0 = vsel(ok0, th0, prev[0]), f1 = vsel(ok1, th1, prev[1]), f2 = vsel(ok2, th2, prev[2]);
  {
    vf p0 = legs_bcast<0>(f0), p1 = legs_bcast<0>(f1), p2 = legs_bcast<0>(f2);
    vm me = leg == 1;
    f0 = vsel(me & !ok0, p0, f0); f1 = vsel(me & !ok1, p1, f1); f2 = vsel(me & !ok2, p2, f2);
  }
  {
    vf p0 = legs_bcast<1>(f0), p1 = legs_bcast<1>(f1), p2 = legs_bcast<1>(f2);
    vm me = leg == 2;
    f0 = vsel(me & !ok0, p0, f0); f1 = vsel(me & !ok1, p1, f1); f2 = vsel(me & !ok2, p2, f2);
  }
  {
    vf p0 = legs_bcast<2>(f0), p1 = legs_bcast<2>(f1), p2 = legs_bcast<2>(f2);
    vm me = leg == 3;
    f0 = vsel(me & !ok0, p0, f0); f1 = vsel(me & !ok1, p1, f1); f2 = vsel(me & !ok2, p2, f2);
  }
  th_out[0] = f0; th_out[1] = f1; th_out[2] = f2;
  // temp[] leaving leg 3 (input of the next pass)
  prev[0] = legs_bcast<3>(f0); prev[1] = legs_bcast<3>(f1); prev[2] = legs_bcast<3>(f2);
  toe_out = toe;
}
IRRL_DEV void gait_generator_manual(const EnvParams &P, EnvLane &L, bool is_first) {
  vf t = env_time(P, L);
  vf gait_step = L.cmdf[0] * P.lam * P.period;
  if (P.wildcat) gait_step = -gait_step;
  vf side_step = L.cmdf[1] * P.lam * P.period;
  vf rot_step = L.cmdf[2] * P.period * 0.4f;
  if (P.height_variable) {  // ENV:1779-1792
    vf ratio = v_abs(L.cmdf[0]) / P.Vx;
    if (P.Vy > 0.0f) ratio = v_max(ratio, v_abs(L.cmdf[1]) / P.Vy);
    if (P.Omega > 0.0f) ratio = v_max(ratio, v_abs(L.cmdf[2] / P.Omega));
    L.up_height = vsel(ratio > 0.1f, P.up_height_max, ratio * P.up_height_max);
  }
  vf prev[3] = {0.0f, 0.0f, 0.0f};
  vf th[3];
  v3 toe;
  if (is_first) {
    gait_leg_pass(P, L, t - P.control_dt, gait_step, side_step, rot_step, L.up_height, prev, th, toe);
    L.jrl[0] = th[0]; L.jrl[1] = -th[1]; L.jrl[2] = -th[2];
  }
  gait_leg_pass(P, L, t, gait_step, side_step, rot_step, L.up_height, prev, th, toe);
  L.jr[0] = th[0]; L.jr[1] = -th[1]; L.jr[2] = -th[2];
#pragma unroll
  for (int j = 0; j < 3; j++) { L.jdr[j] = (L.jr[j] - L.jrl[j]) * P.inv_control_dt; L.jrl[j] = L.jr[j]; }
  // EndEffectorRef = toe + hip offset (ENV:331-334)
  L.eer[0] = toe.x + L.m.sf * 0.19f; L.eer[1] = toe.y + L.m.sy * 0.058f; L.eer[2] = toe.z + 0.0f;
}

// ENV:1010-1109, ManualTraj branch
IRRL_DEV vi ref_row_base(const EnvParams &P, const EnvLane &L) {
  vf fr = v_min(v_max(i2f(L.frame), 0.0f), (float)(P.ref_rows - 1));   // row index clamped to the table (exact in f32)
  return f2i(fr) * 30;
}
IRRL_DEV void command_obs_update(const EnvParams &P, EnvLane &L, vu env, bool flag_reset) {
  if (P.manual) return;
  if (P.ref_traj) {
    // ENV:1100-1107 + gait_generator() ENV:1667-1671: command and joint reference straight from row frame_idx of the table
    const vi b = ref_row_base(P, L), leg3 = leg_id() * 3;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      L.ob_cmd[i] = ld(P.ref, b + (27 + i));
      L.cmdf[i] = L.ob_cmd[i];
      L.jr[i] = ld(P.ref, b + leg3 + i);
      L.jdr[i] = ld(P.ref, b + leg3 + (12 + i));
    }
    return;
  }
  rng4 r = philox_u01(P.seed, env, L.episode, to_u(L.frame), flag_reset ? IRRL_P_RESET_CMD : IRRL_P_CMD);
  vm resample = r.u0 < P.cmd_resample_p;
  if (flag_reset) resample = resample | vm(true);
  vf t = r.u1, v = r.u2;
  vm bx = (0.2f < t) & (t <= 0.7f);
  vm by = (!bx) & (0.7f < t) & (t <= 0.85f);
  vm bw = (!bx) & (!by);
  L.cmd[0] = vsel(resample & bx, v * P.Vx + (1.0f - v) * 0.0f, L.cmd[0]);
  L.cmd[1] = vsel(resample & by, v * P.Vy + (1.0f - v) * -P.Vy, L.cmd[1]);
  L.cmd[2] = vsel(resample & bw, v * P.Omega + (1.0f - v) * -P.Omega, L.cmd[2]);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    L.cmdf[i] = flag_reset ? L.cmd[i] : (L.cmdf[i] * 0.995f + L.cmd[i] * (1.0f - 0.995f));
    L.ob_cmd[i] = L.cmdf[i];
  }
  gait_generator_manual(P, L, flag_reset);
}

// ENV:1116-1194
IRRL_DEV void contact_obs_update(const EnvParams &P, EnvLane &L) {
  if (!P.time_based_contact) {
    L.contact = vsel(L.in_contact != 0, 1.0f, 0.0f);
  } else {
    vf phase_l = pick4(P.phase[0], P.phase[1], P.phase[2], P.phase[3], leg_id());
    vf rp = v_fmod_pos(env_time(P, L) + phase_l * P.period, P.period, P.inv_period) * P.inv_period;
    L.contact = vsel(rp < P.lam, 1.0f, 0.0f);
  }
}

// ENV:956-1004
IRRL_DEV void update_observation(const EnvParams &P, EnvLane &L, vu env, const StepNoise *pre = nullptr) {
  vi leg = leg_id();
  vf t = env_time(P, L);
  L.ob_cmd[0] = 0.0f; L.ob_cmd[1] = 0.0f; L.ob_cmd[2] = 0.0f;  // ENV:960 zeroes the buffer
  if (P.ref_traj) {   // ENV:972: phase columns 25-26 of the table
    const vi b = ref_row_base(P, L);
    L.ob_phase[0] = ld(P.ref, b + 25);
    L.ob_phase[1] = ld(P.ref, b + 26);
  } else {
    sincos_fast(P.two_pi_over_period * t, L.ob_phase[0], L.ob_phase[1]);   // t stays below a few seconds: |x| < 2^8 pi/2
  }
  vf nj[3] = {0.0f, 0.0f, 0.0f}, nv[3] = {0.0f, 0.0f, 0.0f}, nn[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (P.obs_noise != 0.0f) {
    vf u[16];
    if (pre) legs_gather16(pre->joint, u); else legs_rng16(P.seed, env, L.episode, to_u(L.frame), IRRL_P_OBS_JOINT, u);
#pragma unroll
    for (int k = 0; k < 3; k++) nj[k] = (2.0f * pick_leg(u, leg, k) - 1.0f) * 0.002f * P.obs_noise;
    if (pre) legs_gather16(pre->jvel, u); else legs_rng16(P.seed, env, L.episode, to_u(L.frame), IRRL_P_OBS_JVEL, u);
#pragma unroll
    for (int k = 0; k < 3; k++) nv[k] = (2.0f * pick_leg(u, leg, k) - 1.0f) * 0.8f * P.obs_noise;
    if (pre) legs_gather16(pre->normal, u); else legs_rng16(P.seed, env, L.episode, to_u(L.frame), IRRL_P_OBS_NORMAL, u);
    // Box-Muller: normal k from uniforms (2k, 2k+1); lane l evaluates normals l and 4 + (l & 1)
    vf ua = pick4(u[0], u[2], u[4], u[6], leg), ub = pick4(u[1], u[3], u[5], u[7], leg);
    vf uc = vsel((leg & 1) == 0, u[8], u[10]), ud = vsel((leg & 1) == 0, u[9], u[11]);
    vf na = v_sqrt(-2.0f * v_log(1.0f - ua)) * v_cos(6.283185307179586f * ub);
    vf nb = v_sqrt(-2.0f * v_log(1.0f - uc)) * v_cos(6.283185307179586f * ud);
    nn[0] = legs_bcast<0>(na); nn[1] = legs_bcast<1>(na); nn[2] = legs_bcast<2>(na); nn[3] = legs_bcast<3>(na);
    nn[4] = legs_bcast<0>(nb); nn[5] = legs_bcast<1>(nb);
  }
#pragma unroll
  for (int k = 0; k < 3; k++) { L.ob_q[k] = nj[k] + L.q[k]; L.ob_qd[k] = nv[k] + L.qd[k]; }
  rot3 R = quat_to_rot(L.qw, L.qx, L.qy, L.qz);
  L.ob_post[0] = R.r2.x + (nn[0] * 0.02f) * P.obs_noise;
  L.ob_post[1] = R.r2.y + (nn[1] * 0.02f) * P.obs_noise;
  L.ob_post[2] = R.r2.z + (nn[2] * 0.02f) * P.obs_noise;
  L.bodyLinVel = rot_tmul(R, L.vw);
  L.bodyAngVel = rot_tmul(R, L.ww);
  L.ob_omega[0] = L.bodyAngVel.x + P.obs_noise * (nn[3] * 0.5f);
  L.ob_omega[1] = L.bodyAngVel.y + P.obs_noise * (nn[4] * 0.5f);
  L.ob_omega[2] = L.bodyAngVel.z + P.obs_noise * (nn[5] * 0.5f);
}

// toe frame position (base comps) and world-frame linear speed (ENV:1224-1231)
IRRL_DEV void toe_state(const EnvLane &L, v3 &xB, vf &speed) {
  LegKin k = leg_fk(L.m, L.q[0], L.q[1], L.q[2]);
  xB = k.ptoe;
  v3 v = L.bodyLinVel + cross(L.bodyAngVel, k.ptoe);
  v = v + L.qd[0] * cross(mk3(1.0f, 0.0f, 0.0f), k.ptoe - k.pA) + L.qd[1] * cross(k.h, k.ptoe - k.pT) + L.qd[2] * cross(k.h, k.ptoe - k.pS);
  speed = v_sqrt(dot(v, v));  // |R v| = |v|
}

// ENV:1199-1243 + 1444-1548.  Returns the reward; fills extra[6] (ENV:942-950 order fixed by this build).
IRRL_DEV vf reward_update(const EnvParams &P, EnvLane &L, vf extra[6]) {
  vi leg = leg_id();
  v3 xB; vf vel_norm;
  toe_state(L, xB, vel_norm);
  vf force_norm = vsel(L.in_contact != 0, v_sqrt(L.lamw[0] * L.lamw[0] + L.lamw[1] * L.lamw[1] + L.lamw[2] * L.lamw[2]) * P.inv_control_dt, 0.0f);
  // per-leg partial sums, summed FR,FL,HR,HL by the quad reduction
  vf d0 = xB.x - L.eer[0], d1 = xB.y - L.eer[1], d2 = xB.z - L.eer[2];
  vf ee2 = legs_sum(d0 * d0 + d1 * d1 + d2 * d2);
  vf j2 = 0.0f, jd2 = 0.0f, tn2 = 0.0f, td2 = 0.0f, tn[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    vf a = L.jr[k] - L.q[k]; j2 += a * a;
    vf b = L.jdr[k] - L.qd[k]; jd2 += b * b;
    tn[k] = L.tq[k] * ((k == 2) ? (1.0f / 27.0f) : (1.0f / 18.0f));  // ENV:354
    tn2 += tn[k] * tn[k];
    vf c = tn[k] - L.tql[k]; td2 += c * c;
  }
  j2 = legs_sum(j2); jd2 = legs_sum(jd2); tn2 = legs_sum(tn2); td2 = legs_sum(td2);
#pragma unroll
  for (int k = 0; k < 3; k++) L.tql[k] = tn[k];  // ENV:1515 keeps the NORMALISED torque
  vf EE = P.c_ee * v_exp_fast(-40.0f * ee2);
  vf dz = L.pos.z - P.stand_height;
  vf BC = P.c_pos * v_exp_fast(-80.0f * (dz * dz));
  vf BA = P.c_att * v_exp_fast(-80.0f * (L.ob_post[0] * L.ob_post[0] + L.ob_post[1] * L.ob_post[1]));
  vf JR = P.c_joint * 0.25f * v_exp_fast(-2.0f * j2);
  vf JD = P.c_joint * 0.75f * v_exp_fast(-P.control_dt * jd2);
  vf lx = P.wildcat ? -L.cmdf[0] : L.cmdf[0];
  vf e0 = L.bodyLinVel.x - lx, e1 = L.bodyLinVel.y - L.cmdf[1], e2 = L.bodyLinVel.z;
  vf g0 = L.bodyAngVel.x, g1 = L.bodyAngVel.y, g2 = L.bodyAngVel.z - L.cmdf[2];
  vf VR = P.c_vel / 2.0f * v_exp_fast(-2.0f * (e0 * e0 + e1 * e1 + e2 * e2)) + P.c_vel / 2.0f * v_exp_fast(-2.0f * (g0 * g0 + g1 * g1 + g2 * g2));
  vf TR = P.c_torque / 2.0f * v_exp_fast(-0.1f * tn2) + P.c_torque / 2.0f * v_exp_fast((-0.1f * P.inv_control_dt) * td2);
  vf phase_l = pick4(P.phase[0], P.phase[1], P.phase[2], P.phase[3], leg);
  vf rp = v_fmod_pos(env_time(P, L) + phase_l * P.period, P.period, P.inv_period) * P.inv_period;
  // smooth_function and smooth_function2 share the raw curve: s1 = clamp(t, 0, 1), s2 = 1 - s1
  vf sraw = smooth_raw(rp, 2.0f, P.lam, P.inv_lam, P.inv_one_minus_lam);
  vf s1 = vsel(sraw > 1.0f, 1.0f, vsel(sraw < 0.0f, 0.0f, sraw));
  vf s2 = vsel(sraw > 1.0f, 0.0f, vsel(sraw < 0.0f, 1.0f, 1.0f - sraw));
  vf fn = force_norm * (1.0f / 12.5f);
  vf cr = 4.0f * vel_norm * vel_norm * s1 + 2.0f * fn * fn * s2;
  vf CR = P.c_contact * v_exp_fast(-2.0f * legs_sum(cr));
  extra[0] = EE; extra[1] = BC; extra[2] = L.pos.z; extra[3] = BA; extra[4] = JR; extra[5] = VR;
  return EE + BC + JR + JD + VR + BA + TR + CR;  // ENV:1546-1547 order
}

// ENV:547-635 for every lane (callers mask the result) WITHOUT its last three statements -- contact_obs_update, command_obs_update
// (false) and the frame increment (ENV:627-629): a control step ends with exactly the same three (ENV:784-785), so the step kernel
// runs them ONCE for the whole wave after it has merged the freshly reset robots (reset_lane_tail).
IRRL_DEV void reset_lane_head(const EnvParams &P, EnvLane &L, vu env) {
  vi leg = leg_id();
  L.episode = L.episode + 1u;
  L.frame = 0;
  if (P.randomize_per_episode && P.stochastic) model_randomize(L.m, leg, P.seed, env, L.episode);
  rng4 rt = philox_u01(P.seed, env, L.episode, 0u, IRRL_P_RESET_TIME);
  L.t0 = P.manual ? 0.0f : rt.u0;
  // ENV:608-612: the meteorite is parked above gc_, which still holds the base position of the state BEFORE this reset (the new
  // state is set further down, ENV:617-623), sized by the new episode's start time
  if (IRRL_CRUTIAL(P)) sphere_place(P, L, L.pos, L.t0, vm(true));
  if (P.ref_traj) {
    // ENV:538-539, 571: random start frame in the first half of the table, frame_len + 10 rows before its end
    const int span = P.ref_rows / 2 - (int)(P.max_time / P.control_dt) - 10;
    vf u = rt.u1;
    vf shaped = vsel((0.0f < u) & (u < 0.5f), u * 4.0f / 3.0f, (u * 2.0f + 1.0f) / 3.0f);   // sampling_reshape ENV:71-81
    L.frame = span > 0 ? f2i(v_floor((float)span * shaped)) : 0;
  }
#pragma unroll
  for (int i = 0; i < 3; i++) { L.cmdf[i] = 0.0f; L.tql[i] = 0.0f; }
  command_obs_update(P, L, env, true);
  contact_obs_update(P, L);
  float nominal[3] = {0.0f, -0.78f, 1.57f};
  L.pos.z = 0.35f; L.qw = 1.0f; L.qx = 0.0f; L.qy = 0.0f; L.qz = 0.0f;
  L.vw = mk3(0.0f, 0.0f, 0.0f); L.ww = mk3(0.0f, 0.0f, 0.0f);
  if (P.manual) {
    L.pos.x = 0.0f; L.pos.y = 0.0f;
    L.q[0] = L.m.sy * P.abad; L.q[1] = nominal[1]; L.q[2] = nominal[2];
    L.qd[0] = 0.0f; L.qd[1] = 0.0f; L.qd[2] = 0.0f;
  } else {
    vf nj[3], nv[3];
    if (P.shared_noise) {
      rng4 r = philox_u01(P.seed, env, L.episode, 0u, IRRL_P_RESET_JOINT);
#pragma unroll
      for (int k = 0; k < 3; k++) { nj[k] = 2.0f * r.u0 - 1.0f; nv[k] = 2.0f * r.u1 - 1.0f; }
    } else {
      vf ua[16], ub[16];
      legs_rng16(P.seed, env, L.episode, 0u, IRRL_P_RESET_JOINT_IND, ua);
      legs_rng16(P.seed, env, L.episode, 0u, IRRL_P_RESET_JOINT_IND + 3u, ub);
      // 24 consecutive uniforms: joints 0..11 then rates 0..11; the second call starts at element 12
#pragma unroll
      for (int k = 0; k < 3; k++) { nj[k] = 2.0f * pick_leg(ua, leg, k) - 1.0f; nv[k] = 2.0f * pick_leg(ub, leg, k) - 1.0f; }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      L.q[k] = L.jr[k] * (nj[k] * 0.3f) + L.jr[k];
      L.qd[k] = L.jdr[k] * (nv[k] * 0.3f) + L.jdr[k];
    }
    rng4 rb = philox_u01(P.seed, env, L.episode, 0u, IRRL_P_RESET_BASE);
    vf vx = L.cmdf[0] * ((2.0f * rb.u0 - 1.0f) * 0.2f + 1.0f);
    L.vw.x = P.wildcat ? -vx : vx;
    L.vw.y = L.cmdf[1] * ((2.0f * rb.u1 - 1.0f) * 0.2f + 1.0f);
    L.ww.z = L.cmdf[2] * ((2.0f * rb.u2 - 1.0f) * 0.2f + 1.0f);
    rng4 rx = philox_u01(P.seed, env, L.episode, 0u, IRRL_P_RESET_XY);
    L.pos.x = rx.u0 * 5.0f + (1.0f - rx.u0) * -5.0f;
    L.pos.y = rx.u1 * 5.0f + (1.0f - rx.u1) * -5.0f;
  }
  update_observation(P, L, env);
  // obDouble_last_ = obDouble_ (ENV:626)
  L.obl_env[0] = L.ob_cmd[0]; L.obl_env[1] = L.ob_cmd[1]; L.obl_env[2] = L.ob_cmd[2];
  L.obl_env[3] = L.ob_phase[0]; L.obl_env[4] = L.ob_phase[1];
#pragma unroll
  for (int k = 0; k < 3; k++) { L.obl_env[5 + k] = L.ob_post[k]; L.obl_env[8 + k] = L.ob_omega[k]; L.obl_q[k] = L.ob_q[k]; L.obl_qd[k] = L.ob_qd[k]; }
}
IRRL_DEV void reset_lane_tail(const EnvParams &P, EnvLane &L, vu env) {
  contact_obs_update(P, L);
  command_obs_update(P, L, env, false);
  L.frame = L.frame + 1;
}
IRRL_DEV void reset_lane(const EnvParams &P, EnvLane &L, vu env) {
  reset_lane_head(P, L, env);
  reset_lane_tail(P, L, env);
}

// field-wise select of two lane contexts (done ? a : b)
IRRL_DEV void select_lane(vm m, const EnvLane &a, EnvLane &b) {
#pragma unroll
  for (int k = 0; k < 3; k++) {
    b.q[k] = vsel(m, a.q[k], b.q[k]); b.qd[k] = vsel(m, a.qd[k], b.qd[k]);
    b.tql[k] = vsel(m, a.tql[k], b.tql[k]);
    b.jr[k] = vsel(m, a.jr[k], b.jr[k]); b.jrl[k] = vsel(m, a.jrl[k], b.jrl[k]); b.jdr[k] = vsel(m, a.jdr[k], b.jdr[k]);
    b.eer[k] = vsel(m, a.eer[k], b.eer[k]);
    b.cmd[k] = vsel(m, a.cmd[k], b.cmd[k]); b.cmdf[k] = vsel(m, a.cmdf[k], b.cmdf[k]);
    b.ob_cmd[k] = vsel(m, a.ob_cmd[k], b.ob_cmd[k]); b.ob_post[k] = vsel(m, a.ob_post[k], b.ob_post[k]);
    b.ob_omega[k] = vsel(m, a.ob_omega[k], b.ob_omega[k]); b.ob_q[k] = vsel(m, a.ob_q[k], b.ob_q[k]); b.ob_qd[k] = vsel(m, a.ob_qd[k], b.ob_qd[k]);
    b.obl_q[k] = vsel(m, a.obl_q[k], b.obl_q[k]); b.obl_qd[k] = vsel(m, a.obl_qd[k], b.obl_qd[k]);
  }
#pragma unroll
  for (int k = 0; k < 11; k++) b.obl_env[k] = vsel(m, a.obl_env[k], b.obl_env[k]);
  b.ob_phase[0] = vsel(m, a.ob_phase[0], b.ob_phase[0]); b.ob_phase[1] = vsel(m, a.ob_phase[1], b.ob_phase[1]);
  b.contact = vsel(m, a.contact, b.contact);
  b.pos.x = vsel(m, a.pos.x, b.pos.x); b.pos.y = vsel(m, a.pos.y, b.pos.y); b.pos.z = vsel(m, a.pos.z, b.pos.z);
  b.qw = vsel(m, a.qw, b.qw); b.qx = vsel(m, a.qx, b.qx); b.qy = vsel(m, a.qy, b.qy); b.qz = vsel(m, a.qz, b.qz);
  b.vw.x = vsel(m, a.vw.x, b.vw.x); b.vw.y = vsel(m, a.vw.y, b.vw.y); b.vw.z = vsel(m, a.vw.z, b.vw.z);
  b.ww.x = vsel(m, a.ww.x, b.ww.x); b.ww.y = vsel(m, a.ww.y, b.ww.y); b.ww.z = vsel(m, a.ww.z, b.ww.z);
  b.t0 = vsel(m, a.t0, b.t0); b.frame = vsel_i(m, a.frame, b.frame); b.episode = vsel_u(m, a.episode, b.episode);
  b.up_height = vsel(m, a.up_height, b.up_height);
  b.sp.x = vsel(m, a.sp.x, b.sp.x); b.sp.y = vsel(m, a.sp.y, b.sp.y); b.sp.z = vsel(m, a.sp.z, b.sp.z);
  b.sv.x = vsel(m, a.sv.x, b.sv.x); b.sv.y = vsel(m, a.sv.y, b.sv.y); b.sv.z = vsel(m, a.sv.z, b.sv.z);
  b.srad = vsel(m, a.srad, b.srad); b.smass = vsel(m, a.smass, b.smass); b.sdyn = vsel_i(m, a.sdyn, b.sdyn);
  b.bodyLinVel.x = vsel(m, a.bodyLinVel.x, b.bodyLinVel.x); b.bodyLinVel.y = vsel(m, a.bodyLinVel.y, b.bodyLinVel.y); b.bodyLinVel.z = vsel(m, a.bodyLinVel.z, b.bodyLinVel.z);
  b.bodyAngVel.x = vsel(m, a.bodyAngVel.x, b.bodyAngVel.x); b.bodyAngVel.y = vsel(m, a.bodyAngVel.y, b.bodyAngVel.y); b.bodyAngVel.z = vsel(m, a.bodyAngVel.z, b.bodyAngVel.z);
  // model (only changes with RandomizePerEpisode)
  b.m.mA = vsel(m, a.m.mA, b.m.mA); b.m.mT = vsel(m, a.m.mT, b.m.mT); b.m.mS = vsel(m, a.m.mS, b.m.mS); b.m.m0 = vsel(m, a.m.m0, b.m.m0);
  b.m.comA.x = vsel(m, a.m.comA.x, b.m.comA.x); b.m.comA.y = vsel(m, a.m.comA.y, b.m.comA.y); b.m.comA.z = vsel(m, a.m.comA.z, b.m.comA.z);
  b.m.comT.x = vsel(m, a.m.comT.x, b.m.comT.x); b.m.comT.y = vsel(m, a.m.comT.y, b.m.comT.y); b.m.comT.z = vsel(m, a.m.comT.z, b.m.comT.z);
  b.m.comS.x = vsel(m, a.m.comS.x, b.m.comS.x); b.m.comS.y = vsel(m, a.m.comS.y, b.m.comS.y); b.m.comS.z = vsel(m, a.m.comS.z, b.m.comS.z);
  b.m.com0.x = vsel(m, a.m.com0.x, b.m.com0.x); b.m.com0.y = vsel(m, a.m.com0.y, b.m.com0.y); b.m.com0.z = vsel(m, a.m.com0.z, b.m.com0.z);
  b.m.mu = vsel(m, a.m.mu, b.m.mu); b.m.rest = vsel(m, a.m.rest, b.m.rest); b.m.rest_thr = vsel(m, a.m.rest_thr, b.m.rest_thr); b.m.dz = vsel(m, a.m.dz, b.m.dz);
}

// ---------------------------------------------------------------------------------------------
// state pool <-> lane context
// ---------------------------------------------------------------------------------------------
// for_step: the step overwrites the applied torque, the contact flags and the whole raw observation before it reads them
// (physics_substep / contact_obs_update / update_observation), so their 51 words per robot are not fetched
IRRL_DEV void load_lane(const EnvParams &P, const EnvState &S, vi env, vi leg, EnvLane &L, bool for_step = false) {
  vi j12 = env * 12 + leg * 3, gcb = env * 19, gvb = env * 18;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    L.q[k] = ld(S.gc, gcb + 7 + leg * 3 + k); L.qd[k] = ld(S.gv, gvb + 6 + leg * 3 + k);
    L.ptl[k] = ld(S.ptarget_last, j12 + k); L.tql[k] = ld(S.torque_last, j12 + k); L.tq[k] = for_step ? vf(0.0f) : ld(S.torque, j12 + k);
    L.jr[k] = ld(S.joint_ref, j12 + k); L.jrl[k] = ld(S.joint_ref_last, j12 + k); L.jdr[k] = ld(S.joint_dot_ref, j12 + k);
    L.eer[k] = ld(S.ee_ref, j12 + k); L.lamw[k] = ld(S.lam_w, j12 + k);
    L.cmd[k] = ld(S.command, env * 3 + k); L.cmdf[k] = ld(S.command_filtered, env * 3 + k);
  }
  L.in_contact = ld_i(S.in_contact, env * 4 + leg); L.contact = for_step ? vf(0.0f) : ld(S.contact, env * 4 + leg);
  L.ccount = 0u;
  if (S.contact_count) L.ccount = ld_u(S.contact_count, env * 4 + leg);
  L.pos = mk3(ld(S.gc, gcb), ld(S.gc, gcb + 1), ld(S.gc, gcb + 2));
  L.qw = ld(S.gc, gcb + 3); L.qx = ld(S.gc, gcb + 4); L.qy = ld(S.gc, gcb + 5); L.qz = ld(S.gc, gcb + 6);
  L.vw = mk3(ld(S.gv, gvb), ld(S.gv, gvb + 1), ld(S.gv, gvb + 2));
  L.ww = mk3(ld(S.gv, gvb + 3), ld(S.gv, gvb + 4), ld(S.gv, gvb + 5));
  L.t0 = ld(S.t0, env); L.frame = ld_i(S.frame_idx, env); L.episode = ld_u(S.episode, env); L.up_height = ld(S.up_height, env);
  model_signs(L.m, leg);
  L.m.mu = ld(S.material, env * 3); L.m.rest = ld(S.material, env * 3 + 1); L.m.rest_thr = ld(S.material, env * 3 + 2);
  L.m.m0 = ld(S.mass, env * 13); L.m.mA = ld(S.mass, env * 13 + 1 + leg * 3); L.m.mT = ld(S.mass, env * 13 + 2 + leg * 3); L.m.mS = ld(S.mass, env * 13 + 3 + leg * 3);
  vi cb = env * 39;
  L.m.com0 = mk3(ld(S.com, cb), ld(S.com, cb + 1), ld(S.com, cb + 2));
  L.m.comA = mk3(ld(S.com, cb + 3 + leg * 9), ld(S.com, cb + 4 + leg * 9), ld(S.com, cb + 5 + leg * 9));
  L.m.comT = mk3(ld(S.com, cb + 6 + leg * 9), ld(S.com, cb + 7 + leg * 9), ld(S.com, cb + 8 + leg * 9));
  L.m.comS = mk3(ld(S.com, cb + 9 + leg * 9), ld(S.com, cb + 10 + leg * 9), ld(S.com, cb + 11 + leg * 9));
  L.m.dz = ld(S.thigh_dz, env);
  if (IRRL_CRUTIAL(P)) {
    vi sb = env * 9;
    L.sp = mk3(ld(S.sphere, sb), ld(S.sphere, sb + 1), ld(S.sphere, sb + 2));
    L.sv = mk3(ld(S.sphere, sb + 3), ld(S.sphere, sb + 4), ld(S.sphere, sb + 5));
    L.srad = ld(S.sphere, sb + 6); L.smass = ld(S.sphere, sb + 7); L.sdyn = f2i(ld(S.sphere, sb + 8));
  } else {
    L.sp = mk3(0.0f, 0.0f, 0.0f); L.sv = mk3(0.0f, 0.0f, 0.0f); L.srad = 0.0f; L.smass = 0.0f; L.sdyn = 0;
  }
  // raw observation (needed by observe()/isTerminalState() between steps, and by the ObsFilter history)
  vi ob = env * 35;
  if (for_step) {
#pragma unroll
    for (int k = 0; k < 3; k++) { L.ob_cmd[k] = 0.0f; L.ob_post[k] = 0.0f; L.ob_omega[k] = 0.0f; L.ob_q[k] = 0.0f; L.ob_qd[k] = 0.0f; }
    L.ob_phase[0] = 0.0f; L.ob_phase[1] = 0.0f;
  } else {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      L.ob_cmd[k] = ld(S.ob, ob + k); L.ob_post[k] = ld(S.ob, ob + 29 + k); L.ob_omega[k] = ld(S.ob, ob + 32 + k);
      L.ob_q[k] = ld(S.ob, ob + 5 + leg * 3 + k); L.ob_qd[k] = ld(S.ob, ob + 17 + leg * 3 + k);
    }
    L.ob_phase[0] = ld(S.ob, ob + 3); L.ob_phase[1] = ld(S.ob, ob + 4);
  }
  // obDouble_last_ is only read by the observation filter (ENV:1251-1256): leave it in HBM otherwise
  if (P.obs_filter) {
#pragma unroll
    for (int k = 0; k < 3; k++) { L.obl_q[k] = ld(S.ob_last, ob + 5 + leg * 3 + k); L.obl_qd[k] = ld(S.ob_last, ob + 17 + leg * 3 + k); }
#pragma unroll
    for (int k = 0; k < 5; k++) L.obl_env[k] = ld(S.ob_last, ob + k);
#pragma unroll
    for (int k = 0; k < 6; k++) L.obl_env[5 + k] = ld(S.ob_last, ob + 29 + k);
  } else {
#pragma unroll
    for (int k = 0; k < 3; k++) { L.obl_q[k] = 0.0f; L.obl_qd[k] = 0.0f; }
#pragma unroll
    for (int k = 0; k < 11; k++) L.obl_env[k] = 0.0f;
  }
  L.bodyLinVel = mk3(0.0f, 0.0f, 0.0f); L.bodyAngVel = mk3(0.0f, 0.0f, 0.0f);
}

IRRL_DEV void store_lane(const EnvParams &P, const EnvState &S, vi env, vi leg, vm valid, const EnvLane &L, bool store_model) {
  vi j12 = env * 12 + leg * 3, gcb = env * 19, gvb = env * 18, ob = env * 35;
  vm lead = valid & (leg == 0);
  // two exec-mask regions (per-leg words, per-env words) instead of one per store
  IRRL_MASKED_BEGIN(valid)
#pragma unroll
  for (int k = 0; k < 3; k++) {
    stm(S.gc, gcb + 7 + leg * 3 + k, L.q[k]); stm(S.gv, gvb + 6 + leg * 3 + k, L.qd[k]);
    stm(S.ptarget_last, j12 + k, L.ptl[k]); stm(S.torque_last, j12 + k, L.tql[k]); stm(S.torque, j12 + k, L.tq[k]);
    stm(S.joint_ref, j12 + k, L.jr[k]); stm(S.joint_ref_last, j12 + k, L.jrl[k]); stm(S.joint_dot_ref, j12 + k, L.jdr[k]);
    stm(S.ee_ref, j12 + k, L.eer[k]); stm(S.lam_w, j12 + k, L.lamw[k]);
    stm(S.ob, ob + 5 + leg * 3 + k, L.ob_q[k]); stm(S.ob, ob + 17 + leg * 3 + k, L.ob_qd[k]);
    if (P.obs_filter) { stm(S.ob_last, ob + 5 + leg * 3 + k, L.obl_q[k]); stm(S.ob_last, ob + 17 + leg * 3 + k, L.obl_qd[k]); }
  }
  stm_i(S.in_contact, env * 4 + leg, L.in_contact); stm(S.contact, env * 4 + leg, L.contact);
  if (S.contact_count) stm_u(S.contact_count, env * 4 + leg, L.ccount);
  if (store_model) {
    stm(S.mass, env * 13 + 1 + leg * 3, L.m.mA); stm(S.mass, env * 13 + 2 + leg * 3, L.m.mT); stm(S.mass, env * 13 + 3 + leg * 3, L.m.mS);
    vi cb = env * 39;
    stm(S.com, cb + 3 + leg * 9, L.m.comA.x); stm(S.com, cb + 4 + leg * 9, L.m.comA.y); stm(S.com, cb + 5 + leg * 9, L.m.comA.z);
    stm(S.com, cb + 6 + leg * 9, L.m.comT.x); stm(S.com, cb + 7 + leg * 9, L.m.comT.y); stm(S.com, cb + 8 + leg * 9, L.m.comT.z);
    stm(S.com, cb + 9 + leg * 9, L.m.comS.x); stm(S.com, cb + 10 + leg * 9, L.m.comS.y); stm(S.com, cb + 11 + leg * 9, L.m.comS.z);
  }
  IRRL_MASKED_END
  IRRL_MASKED_BEGIN(lead)
#pragma unroll
  for (int k = 0; k < 3; k++) {
    stm(S.command, env * 3 + k, L.cmd[k]); stm(S.command_filtered, env * 3 + k, L.cmdf[k]);
    stm(S.ob, ob + k, L.ob_cmd[k]); stm(S.ob, ob + 29 + k, L.ob_post[k]); stm(S.ob, ob + 32 + k, L.ob_omega[k]);
  }
  stm(S.gc, gcb, L.pos.x); stm(S.gc, gcb + 1, L.pos.y); stm(S.gc, gcb + 2, L.pos.z);
  stm(S.gc, gcb + 3, L.qw); stm(S.gc, gcb + 4, L.qx); stm(S.gc, gcb + 5, L.qy); stm(S.gc, gcb + 6, L.qz);
  stm(S.gv, gvb, L.vw.x); stm(S.gv, gvb + 1, L.vw.y); stm(S.gv, gvb + 2, L.vw.z);
  stm(S.gv, gvb + 3, L.ww.x); stm(S.gv, gvb + 4, L.ww.y); stm(S.gv, gvb + 5, L.ww.z);
  stm(S.t0, env, L.t0); stm_i(S.frame_idx, env, L.frame); stm_u(S.episode, env, L.episode); stm(S.up_height, env, L.up_height);
  stm(S.ob, ob + 3, L.ob_phase[0]); stm(S.ob, ob + 4, L.ob_phase[1]);
  if (IRRL_CRUTIAL(P)) {
    vi sb = env * 9;
    stm(S.sphere, sb, L.sp.x); stm(S.sphere, sb + 1, L.sp.y); stm(S.sphere, sb + 2, L.sp.z);
    stm(S.sphere, sb + 3, L.sv.x); stm(S.sphere, sb + 4, L.sv.y); stm(S.sphere, sb + 5, L.sv.z);
    stm(S.sphere, sb + 6, L.srad); stm(S.sphere, sb + 7, L.smass); stm(S.sphere, sb + 8, i2f(L.sdyn));
  }
  if (P.obs_filter) {
#pragma unroll
    for (int k = 0; k < 5; k++) stm(S.ob_last, ob + k, L.obl_env[k]);
#pragma unroll
    for (int k = 0; k < 6; k++) stm(S.ob_last, ob + 29 + k, L.obl_env[5 + k]);
  }
  if (store_model) {
    stm(S.material, env * 3, L.m.mu); stm(S.material, env * 3 + 1, L.m.rest); stm(S.material, env * 3 + 2, L.m.rest_thr);
    stm(S.mass, env * 13, L.m.m0);
    vi cb = env * 39;
    stm(S.com, cb, L.m.com0.x); stm(S.com, cb + 1, L.m.com0.y); stm(S.com, cb + 2, L.m.com0.z);
    stm(S.thigh_dz, env, L.m.dz);
  }
  IRRL_MASKED_END
}

// ENV:1248-1268 + obs scaling ENV:375-393: writes this lane's share of the scaled [N,35] row.  The scaling divides by
// constants (1, 5 / 35 / 40, 0.7, 3): multiplications by their reciprocals here, one rounding away from the quotient.
IRRL_DEV void observe_write(const EnvParams &P, vi env, vi leg, vm valid, const EnvLane &L, float *ob_out);
IRRL_DEV void observe_lane(const EnvParams &P, vi env, vi leg, vm valid, EnvLane &L, float *ob_out) {
  if (P.obs_filter) {  // filter touches obs[5:35]
    float al = P.obs_filter_alpha;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      L.ob_q[k] = L.ob_q[k] * al + L.obl_q[k] * (1.0f - al);
      L.ob_qd[k] = L.ob_qd[k] * al + L.obl_qd[k] * (1.0f - al);
      L.ob_post[k] = L.ob_post[k] * al + L.obl_env[5 + k] * (1.0f - al);
      L.ob_omega[k] = L.ob_omega[k] * al + L.obl_env[8 + k] * (1.0f - al);
      L.obl_q[k] = L.ob_q[k]; L.obl_qd[k] = L.ob_qd[k]; L.obl_env[5 + k] = L.ob_post[k]; L.obl_env[8 + k] = L.ob_omega[k];
      L.obl_env[k] = L.ob_cmd[k];
    }
    L.obl_env[3] = L.ob_phase[0]; L.obl_env[4] = L.ob_phase[1];
  }
  observe_write(P, env, leg, valid, L, ob_out);
}
// the scaled row itself (no state is touched): observe_lane's stores; the persistent MlpPolicy rollout kernel calls it a second time with its
// wave's LDS scratch as `ob_out` and the robot's index inside the wave as `env` (env_kernels.hip)
IRRL_DEV void observe_write(const EnvParams &P, vi env, vi leg, vm valid, const EnvLane &L, float *ob_out) {
  vm lead = valid & (leg == 0);
  vi ob = env * 35;
  const float ijstd[3] = {1.0f / 5.0f, 1.0f / 35.0f, 1.0f / 40.0f};
  vf nominal[3] = {L.m.sy * P.abad, -0.78f, 1.57f};
  IRRL_MASKED_BEGIN(valid)
#pragma unroll
  for (int k = 0; k < 3; k++) {
    stm(ob_out, ob + 5 + leg * 3 + k, L.ob_q[k] - nominal[k]);
    stm(ob_out, ob + 17 + leg * 3 + k, L.ob_qd[k] * ijstd[k]);
  }
  IRRL_MASKED_END
  IRRL_MASKED_BEGIN(lead)
#pragma unroll
  for (int k = 0; k < 3; k++) {
    stm(ob_out, ob + 29 + k, (L.ob_post[k] - ((k == 2) ? 1.0f : 0.0f)) * (1.0f / 0.7f));
    stm(ob_out, ob + 32 + k, L.ob_omega[k] * (1.0f / 3.0f));
  }
  stm(ob_out, ob + 0, L.ob_cmd[0] - (P.Vx + 0.0f) / 2.0f);
  stm(ob_out, ob + 1, L.ob_cmd[1] - (P.Vy + -P.Vy) / 2.0f);
  stm(ob_out, ob + 2, L.ob_cmd[2] - (P.Omega + -P.Omega) / 2.0f);
  stm(ob_out, ob + 3, L.ob_phase[0]);
  stm(ob_out, ob + 4, L.ob_phase[1]);
  IRRL_MASKED_END
}

// ---------------------------------------------------------------------------------------------
// whole-step bodies (one call per lane); the __global__ wrappers live in env_kernels.hip
// ---------------------------------------------------------------------------------------------
// VEC:268-278 + 352-372 around ENV:692-809
struct NoStepHook { IRRL_DEV void operator()() const {} };
// `before_substeps` runs once between the step prologue (all of this step's global loads are behind it) and the substep loop:
// the fused env + policy kernel starts its LDS prefetch of the policy weights there (env_kernels.hip)
// RULE: the per-contact rule of the pool (EnvParams::contact_rule), a compile-time constant of the instantiation the launcher picks
// step_compute: the step on a lane context that is ALREADY in registers -- everything between load_lane and store_lane.  `tail(L)` runs at the
// end of the epilogue, inside its sub-lane-0 region: step_body stores the context there; the multi-step kernels, which keep the context in
// registers from one step to the next (env_kernels.hip), pass nothing and call lane_carry() behind it.
struct NoStepTail { IRRL_DEV void operator()(const EnvLane &, vf, vm) const {} };
// where a lane's three action components come from: the action batch in memory (row-major [N, 12]), or registers the caller filled ahead of
// time (the multi-step kernel requests step k + 1's row while step k runs)
struct ActionRow { const float *p; IRRL_DEV vf get(vi env, vi leg, int k) const { return ld(p, env * 12 + leg * 3 + k); } };
struct ActionRegs { vf a[3]; IRRL_DEV vf get(vi, vi, int k) const { return a[k]; } };
template <int RULE, class Hook = NoStepHook, class Tail = NoStepTail, class Action = ActionRow>
IRRL_DEV void step_compute(const EnvParams &P, EnvLane &L, vi env, vi leg, vm valid, const Action action, float *ob_out,
                           float *reward_out, uint8_t *done_out, float *extra_out, Hook before_substeps = Hook(), Tail tail = Tail()) {
#ifdef IRRL_PROFILE_WAVES
  L.prof_ranksteps = 0; L.prof_flags = 0;
#endif
  IRRL_MARK("step_prologue");
  vu envu = to_u(env) + P.env_id_offset;   // RNG address: the GLOBAL env id
  // ENV:700-708
  vf pT[3];
#ifdef IRRL_L16
  // noisy configurations: all of this step's draws in one Philox evaluation per lane (StepNoise above)
  const bool predraw = P.obs_noise != 0.0f;
  rng4 draw;
  draw.u0 = 0.0f; draw.u1 = 0.0f; draw.u2 = 0.0f; draw.u3 = 0.0f;
  if (predraw) {
    const vi sub = sub_id();
    const vu lg = to_u(leg);
    const vu purpose = vsel_u(sub == 0, IRRL_P_OBS_JOINT + lg, vsel_u(sub == 1, IRRL_P_OBS_JVEL + lg, vsel_u(sub == 2, IRRL_P_OBS_NORMAL + lg,
                              P.shared_noise ? vu(IRRL_P_ACTION_NOISE) : IRRL_P_ACTION_NOISE + lg)));
    draw = philox_u01(P.seed, envu, L.episode, to_u(L.frame), purpose);
  }
#else
  const bool predraw = false;
#endif
  {
    vf an[3] = {0.0f, 0.0f, 0.0f};
    if (P.action_noise != 0.0f) {
#ifdef IRRL_L16
      if (predraw) {
        rng4 r;
        r.u0 = sub_bcast<3>(draw.u0); r.u1 = sub_bcast<3>(draw.u1); r.u2 = sub_bcast<3>(draw.u2); r.u3 = sub_bcast<3>(draw.u3);
        if (P.shared_noise) {
          an[0] = an[1] = an[2] = 2.0f * r.u0 - 1.0f;
        } else {
          vf u[16];
          legs_gather16(r, u);
#pragma unroll
          for (int k = 0; k < 3; k++) an[k] = 2.0f * pick_leg(u, leg, k) - 1.0f;
        }
      } else
#endif
      if (P.shared_noise) {
        rng4 r = philox_u01(P.seed, envu, L.episode, to_u(L.frame), IRRL_P_ACTION_NOISE);
        an[0] = an[1] = an[2] = 2.0f * r.u0 - 1.0f;
      } else {
        vf u[16];
        legs_rng16(P.seed, envu, L.episode, to_u(L.frame), IRRL_P_ACTION_NOISE, u);
#pragma unroll
        for (int k = 0; k < 3; k++) an[k] = 2.0f * pick_leg(u, leg, k) - 1.0f;
      }
    }
    vf nominal[3] = {L.m.sy * P.abad, -0.78f, 1.57f};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      vf p = action.get(env, leg, k) * 1.0f + nominal[k];
      p = (1.0f - P.filter_para) * p + P.filter_para * L.ptl[k];
      p = p * (P.action_noise * an[k]) + p;
      pT[k] = p; L.ptl[k] = p;
    }
  }
  if (IRRL_CRUTIAL(P)) {
    // ENV:731-740: every int(5 period / control_dt) frames the meteorite is parked above the robot; on every other frame a
    // parked one is released
    const float kf = (float)P.attack_every;
    const vf fr = i2f(L.frame);
    const vm park = (fr - v_floor(fr / kf) * kf) < 0.5f;   // frame % K == 0 (exact in f32)
    sphere_place(P, L, L.pos, env_time(P, L), park);
    sphere_release(L, !park);
  }
  if (P.state_disturbance) {
    // ENV:743-748, 912-940 (Manual evaluation runs): every 10 gait periods the base state is kicked -- z, the four
    // quaternion components, v_z and the roll / pitch rates get uniform(-1, 1) noise scaled 0.03 | 0.1 | 0.1 | 0.3 times 0.5.
    // The quaternion is re-normalised here (the reference hands the perturbed one to RaiSim's setState).
    const float kf = (float)P.disturb_every;
    const vf fr = i2f(L.frame);
    const vm fire = (fr - v_floor(fr / kf) * kf) < 0.5f;   // frame % K == 0 (exact in f32)
    if (wave_any(fire)) {
      rng4 a = philox_u01(P.seed, envu, L.episode, to_u(L.frame), IRRL_P_DISTURB);
      rng4 b = philox_u01(P.seed, envu, L.episode, to_u(L.frame), IRRL_P_DISTURB + 1u);
      const float r = 0.5f;
      vf pz = L.pos.z + 0.03f * (2.0f * a.u0 - 1.0f) * r;
      vf qw = L.qw + 0.1f * (2.0f * a.u1 - 1.0f) * r, qx = L.qx + 0.1f * (2.0f * a.u2 - 1.0f) * r;
      vf qy = L.qy + 0.1f * (2.0f * a.u3 - 1.0f) * r, qz = L.qz + 0.1f * (2.0f * b.u0 - 1.0f) * r;
      vf inv = v_rsqrt(qw * qw + qx * qx + qy * qy + qz * qz);
      L.pos.z = vsel(fire, pz, L.pos.z);
      L.qw = vsel(fire, qw * inv, L.qw); L.qx = vsel(fire, qx * inv, L.qx); L.qy = vsel(fire, qy * inv, L.qy); L.qz = vsel(fire, qz * inv, L.qz);
      L.vw.z = vsel(fire, L.vw.z + 0.1f * (2.0f * b.u1 - 1.0f) * r, L.vw.z);
      L.ww.x = vsel(fire, L.ww.x + 0.3f * (2.0f * b.u2 - 1.0f) * r, L.ww.x);
      L.ww.y = vsel(fire, L.ww.y + 0.3f * (2.0f * b.u3 - 1.0f) * r, L.ww.y);
    }
  }
  before_substeps();
  // (the shipped settings' kernels: eight substeps as a constant, the loop unrolled by two -- the end of a substep and the start of the next
  // share basic blocks: multi-step kernel 27.9 -> 27.66 us per step, same box)
  const int n_substeps = IRRL_SOLVER_FIXED(RULE) ? IRRL_SHIPPED_SUBSTEPS : P.loop_count;
_Pragma("unroll 2")
  for (int i = 0; i < n_substeps; i++) physics_substep<RULE>(P, L, pT);
  IRRL_MARK("epi_noise");
  // The epilogue is per-leg work: with four sub-lanes per leg it would be executed four times over.  Only sub-lane 0
  // (the lane that owns the stores) runs it -- same issue time, a quarter of the active lanes, which is what the
  // power-limited clock of a fully occupied chip responds to.  All cross-leg DPP traffic below is between sub-lanes 0.
#ifdef IRRL_L16
  StepNoise sn;
  if (predraw) {   // all lanes still active: sub-lane 0 collects what sub-lanes 1 and 2 drew
    sn.joint = draw;
    sn.jvel.u0 = sub_bcast<1>(draw.u0); sn.jvel.u1 = sub_bcast<1>(draw.u1); sn.jvel.u2 = sub_bcast<1>(draw.u2); sn.jvel.u3 = sub_bcast<1>(draw.u3);
    sn.normal.u0 = sub_bcast<2>(draw.u0); sn.normal.u1 = sub_bcast<2>(draw.u1); sn.normal.u2 = sub_bcast<2>(draw.u2); sn.normal.u3 = sub_bcast<2>(draw.u3);
  }
  const StepNoise *pre = predraw ? &sn : nullptr;
#else
  const StepNoise *pre = nullptr;
#endif
  IRRL_SUB0_ONLY_BEGIN
  IRRL_MARK("epi_obs");
#ifndef IRRL_AB_NO_OBS
  update_observation(P, L, envu, pre);
#endif
  IRRL_MARK("epi_reward");
  vf extra[6];
#ifndef IRRL_AB_NO_REWARD
  vf rew = reward_update(P, L, extra);
#else
  vf rew = L.pos.z;
  for (int j = 0; j < 6; j++) extra[j] = L.pos.z;
#endif
  IRRL_MARK("epi_reset");
  // VEC:358-371: termination is decided on the post-physics state (the command / reference update below does not touch what it reads)
  vm done = (L.pos.z < 0.15f) | (L.pos.z > 0.65f) | (L.ob_post[2] < 0.5f);
#ifdef IRRL_AB_NO_RESET
  done = done & vm(false);
#endif
  if (wave_any(done & valid)) {
#ifdef IRRL_PROFILE_WAVES
    L.prof_flags |= 1;
#endif
    EnvLane Rn = L;
    reset_lane_head(P, Rn, envu);
    select_lane(done, Rn, L);
    rew = vsel(done, rew + P.c_term, rew);
  }
  IRRL_MARK("epi_cmd_gait");
  // the end of a step (ENV:784-785) and the end of a reset (ENV:627-629) are the same three statements: once for everybody
#ifndef IRRL_AB_NO_CMD
  reset_lane_tail(P, L, envu);
#else
  contact_obs_update(P, L);
  L.frame = L.frame + 1;
#endif
  IRRL_MARK("epi_observe");
  observe_lane(P, env, leg, valid, L, ob_out);
  vm lead = valid & (leg == 0);
  IRRL_MASKED_BEGIN(lead)
  stm(reward_out, env, rew);
  stm_u8(done_out, env, vsel_i(done, 1, 0));
#pragma unroll
  for (int j = 0; j < 6; j++) stm(extra_out, env * 6 + j, extra[j]);
#ifdef IRRL_PROFILE_WAVES
  stm(extra_out, env * 6 + 3, (float)L.prof_ranksteps); stm(extra_out, env * 6 + 4, (float)L.prof_flags);
#endif
  IRRL_MASKED_END
  IRRL_MARK("epi_store_context");
  tail(L, rew, done);      // (the step's reward and termination flag next to the final context)
  IRRL_SUB0_ONLY_END
  IRRL_MARK("end");
}

template <int RULE, class Hook = NoStepHook>
IRRL_DEV void step_body(const EnvParams &P, const EnvState &S, vi env, vi leg, vm valid, const float *action, float *ob_out,
                        float *reward_out, uint8_t *done_out, float *extra_out, Hook before_substeps = Hook()) {
  EnvLane L;
  IRRL_MARK("load_context");
  load_lane(P, S, env, leg, L, true);
  step_compute<RULE>(P, L, env, leg, valid, ActionRow{action}, ob_out, reward_out, done_out, extra_out, before_substeps,
                     [&](const EnvLane &Lf, vf, vm) { store_lane(P, S, env, leg, valid, Lf, P.randomize_per_episode != 0); });
}

// THE LANE CONTEXT CARRIED FROM STEP k TO STEP k + 1 IN REGISTERS (round 5: the multi-step kernels of env_kernels.hip load it once in front of
// their step loop and store it once behind it).  Three things make the carried context equal, bit for bit, to what store_lane + load_lane(for_step)
// would have handed over: (1) the per-ENV words (base pose and velocity, command, clock, model of the trunk ...) are stored by the lane of leg 0
// and read back by every lane of the robot -- the legs' own copies may differ from leg 0's in the last bit (each leg sums the partner legs'
// contributions in its own order) -- so leg 0's words are broadcast to the robot's lanes; (2) with 16 lanes per robot the epilogue ran on
// sub-lane 0 of every quad only, so for the per-LEG words sub-lane 0's are broadcast to the quad; one DPP move per word either way;
// (3) the words load_lane does not fetch for a step -- the step overwrites them before it reads them -- are cleared the same way.
IRRL_DEV void lane_carry(EnvLane &L) {
#define IRRL_BE(x) x = legs_bcast<0>(x)
#define IRRL_BEI(x) x = legs_bcast_i<0>(x)
#define IRRL_BEU(x) x = legs_bcast_u<0>(x)
#define IRRL_BE3(v) do { IRRL_BE(v.x); IRRL_BE(v.y); IRRL_BE(v.z); } while (0)
  IRRL_BE3(L.pos); IRRL_BE(L.qw); IRRL_BE(L.qx); IRRL_BE(L.qy); IRRL_BE(L.qz); IRRL_BE3(L.vw); IRRL_BE3(L.ww);
#pragma unroll
  for (int k = 0; k < 3; k++) { IRRL_BE(L.cmd[k]); IRRL_BE(L.cmdf[k]); }
#pragma unroll
  for (int k = 0; k < 11; k++) IRRL_BE(L.obl_env[k]);
  IRRL_BE(L.t0); IRRL_BEI(L.frame); IRRL_BEU(L.episode); IRRL_BE(L.up_height);
  IRRL_BE(L.m.m0); IRRL_BE3(L.m.com0); IRRL_BE(L.m.mu); IRRL_BE(L.m.rest); IRRL_BE(L.m.rest_thr); IRRL_BE(L.m.dz);
  IRRL_BE3(L.sp); IRRL_BE3(L.sv); IRRL_BE(L.srad); IRRL_BE(L.smass); IRRL_BEI(L.sdyn);
#undef IRRL_BE
#undef IRRL_BEI
#undef IRRL_BEU
#undef IRRL_BE3
#ifdef IRRL_L16
#define IRRL_BC(x) x = sub_bcast<0>(x)
#define IRRL_BC3(v) do { IRRL_BC(v.x); IRRL_BC(v.y); IRRL_BC(v.z); } while (0)
#pragma unroll
  for (int k = 0; k < 3; k++) {
    IRRL_BC(L.q[k]); IRRL_BC(L.qd[k]); IRRL_BC(L.ptl[k]); IRRL_BC(L.tql[k]); IRRL_BC(L.jr[k]); IRRL_BC(L.jrl[k]); IRRL_BC(L.jdr[k]); IRRL_BC(L.eer[k]);
    IRRL_BC(L.lamw[k]); IRRL_BC(L.obl_q[k]); IRRL_BC(L.obl_qd[k]);
  }
  L.in_contact = sub_bcast_i<0>(L.in_contact); L.ccount = sub_bcast_u<0>(L.ccount);
  IRRL_BC(L.m.mA); IRRL_BC(L.m.mT); IRRL_BC(L.m.mS); IRRL_BC3(L.m.comA); IRRL_BC3(L.m.comT); IRRL_BC3(L.m.comS);
#undef IRRL_BC
#undef IRRL_BC3
#endif
#pragma unroll
  for (int k = 0; k < 3; k++) { L.tq[k] = 0.0f; L.ob_cmd[k] = 0.0f; L.ob_post[k] = 0.0f; L.ob_omega[k] = 0.0f; L.ob_q[k] = 0.0f; L.ob_qd[k] = 0.0f; }
  L.contact = 0.0f; L.ob_phase[0] = 0.0f; L.ob_phase[1] = 0.0f;
  L.bodyLinVel = mk3(0.0f, 0.0f, 0.0f); L.bodyAngVel = mk3(0.0f, 0.0f, 0.0f);
  // (4) every carried word is handed to the next step as an OPAQUE register, the way a load would hand it over: left visible, the optimizer
  // sees which words a step does not change (the model, most of the time the command) -- it hoists their products out of the step loop and
  // contracts multiply-adds differently from the one-step kernel, whose results then differ in the last bit (measured: the first carried
  // build was 3 % faster and NOT bit-identical; the promise of the multi-step entry points is bit-identity)
#define IRRL_OQ(x) IRRL_OPAQUE(x)
#define IRRL_OQ3(v) do { IRRL_OQ(v.x); IRRL_OQ(v.y); IRRL_OQ(v.z); } while (0)
#pragma unroll
  for (int k = 0; k < 3; k++) {
    IRRL_OQ(L.q[k]); IRRL_OQ(L.qd[k]); IRRL_OQ(L.ptl[k]); IRRL_OQ(L.tql[k]); IRRL_OQ(L.jr[k]); IRRL_OQ(L.jrl[k]); IRRL_OQ(L.jdr[k]); IRRL_OQ(L.eer[k]);
    IRRL_OQ(L.lamw[k]); IRRL_OQ(L.cmd[k]); IRRL_OQ(L.cmdf[k]); IRRL_OQ(L.obl_q[k]); IRRL_OQ(L.obl_qd[k]);
  }
#pragma unroll
  for (int k = 0; k < 11; k++) IRRL_OQ(L.obl_env[k]);
  IRRL_OQ(L.in_contact); IRRL_OQ(L.ccount);
  IRRL_OQ3(L.pos); IRRL_OQ(L.qw); IRRL_OQ(L.qx); IRRL_OQ(L.qy); IRRL_OQ(L.qz); IRRL_OQ3(L.vw); IRRL_OQ3(L.ww);
  IRRL_OQ(L.t0); IRRL_OQ(L.frame); IRRL_OQ(L.episode); IRRL_OQ(L.up_height);
  IRRL_OQ(L.m.mA); IRRL_OQ(L.m.mT); IRRL_OQ(L.m.mS); IRRL_OQ3(L.m.comA); IRRL_OQ3(L.m.comT); IRRL_OQ3(L.m.comS); IRRL_OQ(L.m.m0); IRRL_OQ3(L.m.com0);
  IRRL_OQ(L.m.mu); IRRL_OQ(L.m.rest); IRRL_OQ(L.m.rest_thr); IRRL_OQ(L.m.dz);
#undef IRRL_OQ
#undef IRRL_OQ3
}

// VEC:145-194 per env: constructor randomisation (ENV:435-477) + first reset
IRRL_DEV void init_body(const EnvParams &P, const EnvState &S, vi env, vi leg, vm valid) {
  EnvLane L;
  // zero-initialised members of a fresh ENVIRONMENT (ENV:2045-2051, 412-418)
#pragma unroll
  for (int k = 0; k < 3; k++) {
    L.q[k] = 0.0f; L.qd[k] = 0.0f; L.ptl[k] = 0.0f; L.tql[k] = 0.0f; L.tq[k] = 0.0f; L.jr[k] = 0.0f; L.jrl[k] = 0.0f; L.jdr[k] = 0.0f;
    L.eer[k] = 0.0f; L.lamw[k] = 0.0f; L.cmd[k] = 0.0f; L.cmdf[k] = 0.0f; L.ob_cmd[k] = 0.0f; L.ob_post[k] = 0.0f; L.ob_omega[k] = 0.0f;
    L.ob_q[k] = 0.0f; L.ob_qd[k] = 0.0f; L.obl_q[k] = 0.0f; L.obl_qd[k] = 0.0f;
  }
#pragma unroll
  for (int k = 0; k < 11; k++) L.obl_env[k] = 0.0f;
  L.ob_phase[0] = 0.0f; L.ob_phase[1] = 0.0f;
  L.in_contact = 0; L.contact = 0.0f; L.ccount = 0u;
  L.pos = mk3(0.0f, 0.0f, 0.0f); L.qw = 1.0f; L.qx = 0.0f; L.qy = 0.0f; L.qz = 0.0f;
  L.vw = mk3(0.0f, 0.0f, 0.0f); L.ww = mk3(0.0f, 0.0f, 0.0f);
  L.t0 = 0.0f; L.frame = 0; L.episode = 0u; L.up_height = P.up_height_max;
  L.sp = mk3(0.0f, 0.0f, 0.0f); L.sv = mk3(0.0f, 0.0f, 0.0f); L.srad = 0.0f; L.smass = 0.0f; L.sdyn = 0;
  L.bodyLinVel = mk3(0.0f, 0.0f, 0.0f); L.bodyAngVel = mk3(0.0f, 0.0f, 0.0f);
  if (P.stochastic) model_randomize(L.m, leg, P.seed, to_u(env) + P.env_id_offset, 0u); else model_nominal(L.m, leg);
  L.jr[0] = L.m.sy * P.abad;  // ENV:415-418
  reset_lane(P, L, to_u(env) + P.env_id_offset);
  store_lane(P, S, env, leg, valid, L, true);
}

// VEC:201-207: reset every env, then observe
IRRL_DEV void reset_body(const EnvParams &P, const EnvState &S, vi env, vi leg, vm valid, float *ob_out) {
  EnvLane L;
  load_lane(P, S, env, leg, L);
  reset_lane(P, L, to_u(env) + P.env_id_offset);
  observe_lane(P, env, leg, valid, L, ob_out);
  store_lane(P, S, env, leg, valid, L, P.randomize_per_episode != 0);
}

// VEC:209-212
IRRL_DEV void observe_body(const EnvParams &P, const EnvState &S, vi env, vi leg, vm valid, float *ob_out) {
  EnvLane L;
  load_lane(P, S, env, leg, L);
  observe_lane(P, env, leg, valid, L, ob_out);
  if (P.obs_filter) store_lane(P, S, env, leg, valid, L, false);
}

// Diagnostics (ENV:1375-1402): world-frame inverse mass matrix (column-major [18x18]) and nonlinear term.
// M_w^-1 = T M_B^-1 T^T with  M_B^-1 = [[S^-1, -S^-1 D^T], [-D S^-1, C^-1 + D S^-1 D^T]].
IRRL_DEV void dynamics_probe_body(const EnvParams &P, const EnvState &S, vi env, vi leg, vm valid, float *minv_out, float *nonlin_out) {
  EnvLane L;
  load_lane(P, S, env, leg, L);
  rot3 R = quat_to_rot(L.qw, L.qx, L.qy, L.qz);
  v3 wB = rot_tmul(R, L.ww);
  LegKin k = leg_fk(L.m, L.q[0], L.q[1], L.q[2]);
  LegDyn D;
  leg_dynamics(L.m, k, L.qd, wB, IRRL_GRAV * R.r2, D);
  vm lead = valid & (leg == 0);
  if (nonlin_out) {
    v3 fw = rot_mul(R, mk3(D.bias_b[0], D.bias_b[1], D.bias_b[2])), nw = rot_mul(R, mk3(D.bias_b[3], D.bias_b[4], D.bias_b[5]));
    st_if(lead, nonlin_out, env * 18 + 0, fw.x); st_if(lead, nonlin_out, env * 18 + 1, fw.y); st_if(lead, nonlin_out, env * 18 + 2, fw.z);
    st_if(lead, nonlin_out, env * 18 + 3, nw.x); st_if(lead, nonlin_out, env * 18 + 4, nw.y); st_if(lead, nonlin_out, env * 18 + 5, nw.z);
#pragma unroll
    for (int j = 0; j < 3; j++) st_if(valid, nonlin_out, env * 18 + 6 + leg * 3 + j, D.bias_l[j]);
  }
  if (minv_out) {
    // columns of M_B^-1 by unit right-hand sides, then rotate the base rows/columns into the world frame.
    // Column c of M_w^-1 = T M_B^-1 T^T e_c.  For a base column, T^T e_c is a row of R spread over the base
    // block; for a joint column it is e_c itself.
    vi base = env * 324;
    for (int c = 0; c < 18; c++) {
      vf rb[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, rl[3] = {0.0f, 0.0f, 0.0f};
      if (c < 3) {
        v3 row = (c == 0) ? R.r0 : ((c == 1) ? R.r1 : R.r2);  // T^T e_c = R^T e_c
        rb[0] = row.x; rb[1] = row.y; rb[2] = row.z;
      } else if (c < 6) {
        v3 row = (c == 3) ? R.r0 : ((c == 4) ? R.r1 : R.r2);
        rb[3] = row.x; rb[4] = row.y; rb[5] = row.z;
      } else {
        int jl = (c - 6) / 3, jk = (c - 6) % 3;
        vf one = vsel(leg == jl, 1.0f, 0.0f);
        rl[0] = (jk == 0) ? one : vf(0.0f); rl[1] = (jk == 1) ? one : vf(0.0f); rl[2] = (jk == 2) ? one : vf(0.0f);
      }
      vf xb[6], xl[3];
      solve_M(D, rb, rl, xb, xl);
      v3 lw = rot_mul(R, mk3(xb[0], xb[1], xb[2])), aw = rot_mul(R, mk3(xb[3], xb[4], xb[5]));
      st_if(lead, minv_out, base + c * 18 + 0, lw.x); st_if(lead, minv_out, base + c * 18 + 1, lw.y); st_if(lead, minv_out, base + c * 18 + 2, lw.z);
      st_if(lead, minv_out, base + c * 18 + 3, aw.x); st_if(lead, minv_out, base + c * 18 + 4, aw.y); st_if(lead, minv_out, base + c * 18 + 5, aw.z);
#pragma unroll
      for (int j = 0; j < 3; j++) st_if(valid, minv_out, base + c * 18 + 6 + leg * 3 + j, xl[j]);
    }
  }
}

}  // namespace IRRL_CORE_NS
