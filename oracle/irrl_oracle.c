/*
 * irrl_oracle.c -- CPU restatement (oracle) of the FlexibleRobotRaisimGym env.step() hot path.
 * TEST INFRASTRUCTURE ONLY (see irrl_oracle.h for the rules and the parity status).
 *
 * Written so that it compiles both as C99 (gcc) and as C++ (g++ -x c++) -- the latter is used
 * by tools/flopcount to substitute ORC_REAL with an operation-counting class.
 *
 * Reference aliases: ENV = .../BlackPanther_V55/Environment.hpp, VEC = VectorizedEnvironment.hpp,
 * URDF = .../urdf/black_panther.urdf, PPO = flex_gym/algo/ppo2/ppo2.py.
 *
 * Section map
 *   1  small vector helpers
 *   2  Philox4x32-10 counter RNG (build-defined; the reference's rand()/Eigen::setRandom streams are
 *      process-global and thread-racy, SURVEY 7.3, so "identical seeds" means this generator)
 *   3  curve helpers, IK, torque clamp             (ENV:61-156, 1273-1312, 1687-1751)
 *   4  robot model                                  (URDF numbers, ENV:435-477 randomisation)
 *   5  rigid-body dynamics + contact (build's own formulation)
 *   6  task logic: gait reference, observation, reward, reset, step (ENV:547-809, 956-1109, 1444-1578)
 *   7  vector-env API (VEC:145-372) and probes
 */
#include "irrl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef ORC_REAL
#define ORC_REAL double
#endif
typedef ORC_REAL real;

#ifndef ORC_CUSTOM_MATH /* the flop-count build supplies overloads instead */
#define R_SQRT(x) ((real)sqrt((double)(x)))
#define R_SIN(x) ((real)sin((double)(x)))
#define R_COS(x) ((real)cos((double)(x)))
#define R_ASIN(x) ((real)asin((double)(x)))
#define R_ACOS(x) ((real)acos((double)(x)))
#define R_FLOOR(x) ((real)floor((double)(x)))
#define R_EXP(x) ((real)exp((double)(x)))
#define R_LOG(x) ((real)log((double)(x)))
#define R_FMOD(x, y) ((real)fmod((double)(x), (double)(y)))
#define R_FABS(x) ((real)fabs((double)(x)))
#define R_TO_DOUBLE(x) ((double)(x))
#endif
#define RC(x) ((real)(x))

/* ENV:45 -- the reference's PI is this truncated literal, and every phase/IK formula uses it. */
#define REF_PI 3.1415926

#define NB 13  /* moving bodies after RaiSim merges fixed joints (ENV:449) */
#define NV 18
#define NQ 19
#define NLEG 4

/* ------------------------------------------------------------------ 1. vector helpers */
static inline void v3_set(real *o, real a, real b, real c) { o[0] = a; o[1] = b; o[2] = c; }
static inline void v3_copy(real *o, const real *a) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
static inline void v3_add(real *o, const real *a, const real *b) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static inline void v3_sub(real *o, const real *a, const real *b) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline void v3_scale(real *o, const real *a, real s) { o[0] = a[0] * s; o[1] = a[1] * s; o[2] = a[2] * s; }
static inline void v3_axpy(real *o, real s, const real *a) { o[0] += s * a[0]; o[1] += s * a[1]; o[2] += s * a[2]; }
static inline real v3_dot(const real *a, const real *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void v3_cross(real *o, const real *a, const real *b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
/* o = A(3x3 row-major) * v */
static inline void m3_mulv(real *o, const real *A, const real *v) {
  real x = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
  real y = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
  real z = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
/* o = A^T * v */
static inline void m3_tmulv(real *o, const real *A, const real *v) {
  real x = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
  real y = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
  real z = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3_mul(real *o, const real *A, const real *B) {
  real t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  for (int i = 0; i < 9; i++) o[i] = t[i];
}
/* o = A * B^T */
static inline void m3_mul_bt(real *o, const real *A, const real *B) {
  real t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
  for (int i = 0; i < 9; i++) o[i] = t[i];
}
/* quaternion (w,x,y,z) -> rotation matrix body->world, row-major */
static void quat_to_rot(const real *q, real *R) {
  real w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = RC(1) - RC(2) * (y * y + z * z); R[1] = RC(2) * (x * y - w * z); R[2] = RC(2) * (x * z + w * y);
  R[3] = RC(2) * (x * y + w * z); R[4] = RC(1) - RC(2) * (x * x + z * z); R[5] = RC(2) * (y * z - w * x);
  R[6] = RC(2) * (x * z - w * y); R[7] = RC(2) * (y * z + w * x); R[8] = RC(1) - RC(2) * (x * x + y * y);
}
/* solve 3x3 SPD-ish system A x = b by cofactors (A row-major) */
static void m3_solve(const real *A, const real *b, real *x) {
  real c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  real det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  real id = RC(1) / det;
  real c10 = A[2] * A[7] - A[1] * A[8], c11 = A[0] * A[8] - A[2] * A[6], c12 = A[1] * A[6] - A[0] * A[7];
  real c20 = A[1] * A[5] - A[2] * A[4], c21 = A[2] * A[3] - A[0] * A[5], c22 = A[0] * A[4] - A[1] * A[3];
  real x0 = (c00 * b[0] + c10 * b[1] + c20 * b[2]) * id;
  real x1 = (c01 * b[0] + c11 * b[1] + c21 * b[2]) * id;
  real x2 = (c02 * b[0] + c12 * b[1] + c22 * b[2]) * id;
  x[0] = x0; x[1] = x1; x[2] = x2;
}

/* ------------------------------------------------------------------ 2. Philox4x32-10 */
static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
static void philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
  uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
  uint32_t k0 = key_in[0], k1 = key_in[1];
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* RNG purposes: every random draw of the path has a fixed (purpose, slot) address so that the CPU
 * oracle and the GPU kernels consume bit-identical uniforms. counter = (env, episode, step, purpose),
 * key = (seed, 0x49525231 "IRR1"). */
enum {
  P_DR_MATERIAL = 1, P_DR_MASS = 2 /* ..5 */, P_DR_COM = 6 /* ..15 */, P_DR_THIGH = 16,
  P_RESET_TIME = 20, P_RESET_CMD = 21, P_RESET_JOINT = 22, P_RESET_JOINT_IND = 23 /* ..28 */,
  P_RESET_BASE = 29, P_RESET_XY = 30,
  P_ACTION_NOISE = 40 /* ..42 */, P_OBS_JOINT = 44 /* ..46 */, P_OBS_JVEL = 47 /* ..49 */,
  P_OBS_NORMAL = 50 /* ..52 */, P_CMD = 56, P_DISTURB = 60 /* ..61 */
};
typedef struct { uint32_t seed, env, episode, step; } rng_addr;
static void rng_u01x4(const rng_addr *a, uint32_t purpose, real u[4]) {
  uint32_t ctr[4], key[2], o[4];
  ctr[0] = a->env; ctr[1] = a->episode; ctr[2] = a->step; ctr[3] = purpose;
  key[0] = a->seed; key[1] = 0x49525231u;
  philox4x32_10(ctr, key, o);
  for (int i = 0; i < 4; i++) u[i] = RC((double)(o[i] >> 8) * (1.0 / 16777216.0)); /* [0,1), exact in f32 */
}
/* n values in [0,1) from consecutive purposes */
static void rng_fill_u01(const rng_addr *a, uint32_t purpose, int n, real *out) {
  real u[4];
  for (int i = 0; i < n; i++) {
    if ((i & 3) == 0) rng_u01x4(a, purpose + (uint32_t)(i >> 2), u);
    out[i] = u[i & 3];
  }
}
/* Box-Muller: normal k uses uniforms (2k, 2k+1) of the stream */
static void rng_fill_normal(const rng_addr *a, uint32_t purpose, int n, real *out) {
  real u[16];
  rng_fill_u01(a, purpose, 2 * n, u);
  for (int k = 0; k < n; k++) {
    real r = R_SQRT(RC(-2.0) * R_LOG(RC(1.0) - u[2 * k]));
    out[k] = r * R_COS(RC(6.283185307179586) * u[2 * k + 1]);
  }
}

/* ------------------------------------------------------------------ 3. curve helpers, IK, clamp */
/* ENV:86-91 */
static void cubic_bezier(const real *p0, const real *pf, real s, real *o) {
  real bez = s * s * s + RC(3.0) * (s * s * (RC(1.0) - s));
  for (int i = 0; i < 3; i++) o[i] = p0[i] + bez * (pf[i] - p0[i]);
}
/* ENV:96-99 */
static real gauss_bump(real x, real width, real height) {
  return height * R_EXP(-(x - width / RC(2)) * (x - width / RC(2)) / (RC(2) * (width / RC(6)) * (width / RC(6))));
}
/* ENV:104-113 */
static void bezier2(const real *p0, const real *pf, real s, real height, real *o) {
  real bez = s * s * s + RC(3.0) * (s * s * (RC(1.0) - s));
  o[0] = p0[0] + bez * (pf[0] - p0[0]);
  o[1] = p0[1] + bez * (pf[1] - p0[1]);
  o[2] = p0[2] + gauss_bump(s, RC(1.0), height);
}
/* ENV:118-136 */
static real smooth_function(real phase, real slope, real lam) {
  real f = R_FMOD(phase, RC(1.0)), t;
  if (f < lam) t = (R_SIN(f / lam * RC(2) * RC(REF_PI)) * slope) + RC(0.5);
  else t = (-R_SIN((f - lam) / (RC(1.0) - lam) * RC(2) * RC(REF_PI)) * slope) + RC(0.5);
  if (t > RC(1.0)) return RC(1.0);
  if (t < RC(0.0)) return RC(0.0);
  return t;
}
/* ENV:138-156 */
static real smooth_function2(real phase, real slope, real lam) {
  real f = R_FMOD(phase, RC(1.0)), t;
  if (f < lam) t = (R_SIN(f / lam * RC(2) * RC(REF_PI)) * slope) + RC(0.5);
  else t = (-R_SIN((f - lam) / (RC(1.0) - lam) * RC(2) * RC(REF_PI)) * slope) + RC(0.5);
  if (t > RC(1.0)) return RC(0.0);
  if (t < RC(0.0)) return RC(1.0);
  return RC(1.0) - t;
}
/* ENV:71-81 */
static real sampling_reshape(real r) {
  if (r < RC(0.5) && r > RC(0)) return r * RC(4.0) / RC(3.0);
  return (RC(2.0) * r + RC(1.0)) / RC(3.0);
}
/* ENV:1687-1751. theta keeps its previous content in a slot whose asin/acos argument is out of
 * range (the reference only prints "errorN"). Returns a bitmask of the error branches taken. */
static int inverse_kinematics(real x, real y, real z, real l_hip, real l_thigh, real l_calf, real max_len,
                              int is_right, real *theta) {
  int err = 0;
  real ll = R_SQRT(x * x + y * y + z * z);
  if (ll > max_len) {
    x = x * (max_len - RC(1e-5)) / ll;
    y = y * (max_len - RC(1e-5)) / ll;
    z = z * (max_len - RC(1e-5)) / ll;
  }
  real temp, temp1, temp2;
  if (is_right) temp = (-z * l_hip - R_SQRT(y * y * (z * z + y * y - l_hip * l_hip))) / (z * z + y * y);
  else temp = (z * l_hip + R_SQRT(y * y * (z * z + y * y - l_hip * l_hip))) / (z * z + y * y);
  /* ENV:1703 uses integer abs() in the right branch?  No: <cmath>/<stdlib.h> overloads make
   * abs(double) the floating version under C++14, so both branches test |temp| <= 1. */
  if (R_FABS(temp) <= RC(1)) theta[0] = R_ASIN(temp); else err |= 1;
  real lr = R_SQRT(x * x + y * y + z * z - l_hip * l_hip);
  lr = (lr > (l_thigh + l_calf)) ? (l_thigh + l_calf - RC(1e-4)) : lr;
  temp = (l_thigh * l_thigh + l_calf * l_calf - lr * lr) / RC(2) / l_thigh / l_calf + RC(1e-5);
  if (R_FABS(temp) <= RC(1)) theta[2] = -(RC(REF_PI) - R_ACOS(temp)); else err |= 2;
  temp1 = x / lr;
  temp2 = (lr * lr + l_thigh * l_thigh - l_calf * l_calf) / RC(2) / lr / l_thigh - RC(1e-5);
  if (R_FABS(temp1) <= RC(1) && R_FABS(temp2) <= RC(1)) theta[1] = R_ACOS(temp2) - R_ASIN(temp1); else err |= 4;
  return err;
}
/* ENV:1273-1312: speed-dependent torque limits; knees (i%3==2) carry the 1.55 ratio (a float
 * literal in the reference: 1.55f widened to double). */
static void torque_clamp(real *tau, const real *qd, real tmax, real wc, real wmax, real *upper, real *lower) {
  real r = tmax / (wmax - wc);
  for (int i = 0; i < 12; i++) {
    real ratio = ((i + 1) % 3 == 0) ? RC((double)1.55f) : RC((double)1.0f);
    real up = (qd[i] * ratio > wc) ? (tmax - (qd[i] * ratio - wc) * r) : tmax;
    up = up * ratio;
    real low = (qd[i] * ratio < -wc) ? ((-wmax - qd[i] * ratio) / (-wmax + wc) * -tmax) : -tmax;
    low = low * ratio;
    if (upper) upper[i] = up;
    if (lower) lower[i] = low;
    real t = tau[i];
    t = (t < up) ? t : up;   /* fmin(torque, up) */
    t = (t > low) ? t : low; /* fmax(., low)     */
    tau[i] = t;
  }
}

/* ------------------------------------------------------------------ 4. robot model */
typedef struct {
  real mass[NB];
  real com[NB][3];     /* COM in the body's own frame */
  real inertia[NB][9]; /* about the COM, body frame, row-major */
  real jpos[NB][3];    /* joint origin in the parent frame (body 0 unused) */
  real rotor[12], damping[12];
  real mu, rest, rest_thr; /* default material (ENV:433,440-442) */
  real thigh_dz;           /* ENV:472-476 */
} robot_model;

static const int k_parent[NB] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11};

/* URDF numbers (black_panther.urdf:18-165 for FR; the other legs mirror signs). */
static void model_nominal(robot_model *m) {
  memset(m, 0, sizeof(*m));
  m->mass[0] = RC(3.72); /* URDF:18 (dummy_mass has zero mass, URDF:35) */
  v3_set(m->com[0], RC(0), RC(0), RC(-0.003));
  m->inertia[0][0] = RC(0.016269); m->inertia[0][4] = RC(0.050813); m->inertia[0][8] = RC(0.060989);
  for (int l = 0; l < NLEG; l++) {
    real sf = (l < 2) ? RC(1) : RC(-1);       /* front / hind */
    real sy = (l % 2 == 0) ? RC(-1) : RC(1);  /* right legs sit at -y */
    int a = 1 + 3 * l, t = 2 + 3 * l, s = 3 + 3 * l;
    /* abduct_xx: URDF:50-65 */
    v3_set(m->jpos[a], sf * RC(0.212), sy * RC(0.051), RC(0));
    m->mass[a] = RC(0.54);
    v3_set(m->com[a], sf * RC(0.058), sy * RC(0.00485), RC(0));
    m->inertia[a][0] = RC(0.000391); m->inertia[a][4] = RC(0.000739); m->inertia[a][8] = RC(0.000488);
    /* thigh_xx: URDF:78-93 */
    v3_set(m->jpos[t], RC(0), sy * RC(0.085), RC(0));
    m->mass[t] = RC(0.636);
    v3_set(m->com[t], RC(0), -sy * RC(0.019), RC(-0.01865));
    m->inertia[t][0] = RC(0.001724); m->inertia[t][4] = RC(0.001907); m->inertia[t][8] = RC(0.000468);
    m->inertia[t][5] = m->inertia[t][7] = -sy * RC(0.000228);
    /* shank_xx (URDF:104-119) with toe_xx merged through the fixed joint at (0,0,-0.19) (URDF:130-165) */
    v3_set(m->jpos[s], RC(0), RC(0), RC(-0.201));
    {
      real m1 = RC(0.064), z1 = RC(-0.0865), m2 = RC(0.05), z2 = RC(-0.19);
      real mt = m1 + m2, zc = (m1 * z1 + m2 * z2) / mt;
      real d1 = z1 - zc, d2 = z2 - zc;
      m->mass[s] = mt;
      v3_set(m->com[s], RC(0), RC(0), zc);
      m->inertia[s][0] = RC(0.000716) + m1 * d1 * d1 + RC(0.000025) + m2 * d2 * d2;
      m->inertia[s][4] = RC(0.000721) + m1 * d1 * d1 + RC(0.000025) + m2 * d2 * d2;
      m->inertia[s][8] = RC(0.000012) + RC(0.000025);
    }
    for (int k = 0; k < 3; k++) {
      m->rotor[3 * l + k] = (k == 2) ? RC(0.008966) : RC(0.003708); /* URDF:56,84,110 */
      m->damping[3 * l + k] = RC(0.01);
    }
  }
  m->mu = RC(0.6); m->rest = RC(0.2); m->rest_thr = RC(0.01); /* ENV:433 */
  m->thigh_dz = RC(0);
}

/* ENV:435-477: material, 13 body masses, 13 COM offsets, one shared shank-joint z offset. */
static void model_randomize(robot_model *m, const rng_addr *a) {
  real u[4], um[16], uc[40];
  model_nominal(m);
  rng_u01x4(a, P_DR_MATERIAL, u);
  m->mu = u[0] * RC(0.6) + RC(0.4);
  m->rest = u[1] * RC(0.3);
  m->rest_thr = u[2] * RC(2.0);
  rng_fill_u01(a, P_DR_MASS, 13, um);
  for (int i = 0; i < NB; i++) {
    real f = (um[i] - RC(0.5)) / RC(0.5) * RC(0.15) + RC(1.0); /* mass_distrubance_ratio ENV:2069 */
    m->mass[i] = m->mass[i] * f;
  }
  rng_fill_u01(a, P_DR_COM, 39, uc);
  for (int i = 0; i < NB; i++)
    for (int k = 0; k < 3; k++) m->com[i][k] += (RC(2.0) * uc[3 * i + k] - RC(1.0)) * RC(0.02); /* ENV:463-465,2070 */
  rng_u01x4(a, P_DR_THIGH, u);
  m->thigh_dz = (u[0] - RC(0.5)) / RC(0.5) * RC(0.01); /* ENV:472, 2071 */
  for (int l = 0; l < NLEG; l++) m->jpos[3 + 3 * l][2] += m->thigh_dz;
}

/* ------------------------------------------------------------------ 5. dynamics + contact
 * Build-defined formulation (RaiSim is closed source):
 *   generalized velocity  u~ = [R^T v, R^T w, qd]  (base-frame components of the world-frame gv),
 *   generalized accel     a  = physical accelerations in base-frame components,
 *   M_B(q) a + b(q, u~) = tau_B + sum_l J_l^T f_l ,   M_world = T M_B T^T,  T = diag(R, R, I).
 *   CRBA through composite (mass, first moment, inertia about the base origin); bias by classical
 *   recursive Newton-Euler with gravity folded into the base acceleration; rotor inertia on diag(M);
 *   joint damping explicit; semi-implicit Euler: velocity first (with contact impulses), then position.
 *   Contacts: 4 toe spheres (r = 0.0275, URDF:148) against the plane z = 0, hard contact solved at
 *   velocity level by block Gauss-Seidel (fixed sweep count, fixed leg order FR,FL,HR,HL, warm start),
 *   Coulomb cone mu, restitution e above the threshold speed, no positional correction (ERP 0, ENV:246).
 */
#define TOE_RADIUS 0.0275
#define TOE_Z (-0.19)
#define GRAV 9.81

typedef struct {
  real R[NB][9];  /* body i -> base frame */
  real p[NB][3];  /* origin of body i, base-frame components, relative to the base origin */
  real s[NB][3];  /* joint axis of body i (base-frame components) */
  real c[NB][3];  /* COM position rel. base origin */
} kin_t;

static void forward_kinematics(const robot_model *m, const real *q /*12*/, kin_t *k) {
  static const real I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; i++) k->R[0][i] = RC(I3[i]);
  v3_set(k->p[0], RC(0), RC(0), RC(0));
  v3_set(k->s[0], RC(0), RC(0), RC(0));
  for (int i = 1; i < NB; i++) {
    int par = k_parent[i];
    int jt = (i - 1) % 3; /* 0 abad (+x), 1 hip (-y), 2 knee (-y) */
    real ang = q[i - 1], cs = R_COS(ang), sn = R_SIN(ang);
    real Rj[9], ax[3];
    if (jt == 0) { /* rotation about +x */
      Rj[0] = RC(1); Rj[1] = RC(0); Rj[2] = RC(0);
      Rj[3] = RC(0); Rj[4] = cs; Rj[5] = -sn;
      Rj[6] = RC(0); Rj[7] = sn; Rj[8] = cs;
      v3_set(ax, RC(1), RC(0), RC(0));
    } else { /* rotation by ang about -y == Ry(-ang) */
      Rj[0] = cs; Rj[1] = RC(0); Rj[2] = -sn;
      Rj[3] = RC(0); Rj[4] = RC(1); Rj[5] = RC(0);
      Rj[6] = sn; Rj[7] = RC(0); Rj[8] = cs;
      v3_set(ax, RC(0), RC(-1), RC(0));
    }
    real off[3];
    m3_mulv(off, k->R[par], m->jpos[i]);
    v3_add(k->p[i], k->p[par], off);
    m3_mulv(k->s[i], k->R[par], ax);
    m3_mul(k->R[i], k->R[par], Rj);
  }
  for (int i = 0; i < NB; i++) {
    real rc[3];
    m3_mulv(rc, k->R[i], m->com[i]);
    v3_add(k->c[i], k->p[i], rc);
  }
}

/* M_B (18x18 row-major). Index: 0-2 linear, 3-5 angular, 6+j joint j. */
static void mass_matrix_B(const robot_model *m, const kin_t *k, real *M) {
  real cm[NB], ch[NB][3], cI[NB][9]; /* composite mass, first moment, inertia about base origin */
  for (int i = 0; i < NB; i++) {
    real RI[9], IB[9];
    m3_mul(RI, k->R[i], m->inertia[i]);
    m3_mul_bt(IB, RI, k->R[i]);
    const real *c = k->c[i];
    real cc = v3_dot(c, c);
    cm[i] = m->mass[i];
    v3_scale(ch[i], c, m->mass[i]);
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++)
        cI[i][3 * a + b] = IB[3 * a + b] + m->mass[i] * (((a == b) ? cc : RC(0)) - c[a] * c[b]);
  }
  for (int i = NB - 1; i >= 1; i--) {
    int par = k_parent[i];
    cm[par] += cm[i];
    v3_add(ch[par], ch[par], ch[i]);
    for (int a = 0; a < 9; a++) cI[par][a] += cI[i][a];
  }
  for (int i = 0; i < NV * NV; i++) M[i] = RC(0);
  /* base block */
  for (int a = 0; a < 3; a++) M[a * NV + a] = cm[0];
  {
    const real *h = ch[0];
    real hx[9] = {RC(0), -h[2], h[1], h[2], RC(0), -h[0], -h[1], h[0], RC(0)}; /* [h]x */
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        M[(3 + a) * NV + b] = hx[3 * a + b];      /* ang <- lin */
        M[a * NV + (3 + b)] = -hx[3 * a + b];     /* lin <- ang */
        M[(3 + a) * NV + (3 + b)] = cI[0][3 * a + b];
      }
  }
  for (int i = 1; i < NB; i++) {
    int col = 6 + (i - 1);
    const real *s = k->s[i], *pj = k->p[i];
    real tmp[3], P[3], L[3], sxp[3], hxs[3];
    /* P = s x (h - m p_j) */
    v3_scale(tmp, pj, cm[i]);
    v3_sub(tmp, ch[i], tmp);
    v3_cross(P, s, tmp);
    /* L_O = I_O s - h x (s x p_j) */
    m3_mulv(L, cI[i], s);
    v3_cross(sxp, s, pj);
    v3_cross(hxs, ch[i], sxp);
    v3_sub(L, L, hxs);
    for (int a = 0; a < 3; a++) {
      M[a * NV + col] = M[col * NV + a] = P[a];
      M[(3 + a) * NV + col] = M[col * NV + (3 + a)] = L[a];
    }
    /* ancestors (and self) on the same leg: M[k,j] = s_k . (L_O - p_k x P) */
    for (int anc = i; anc >= 1; anc = k_parent[anc]) {
      real pxP[3], Lk[3];
      v3_cross(pxP, k->p[anc], P);
      v3_sub(Lk, L, pxP);
      real v = v3_dot(k->s[anc], Lk);
      int row = 6 + (anc - 1);
      M[row * NV + col] = v;
      M[col * NV + row] = v;
      if (k_parent[anc] == 0) break;
    }
    M[col * NV + col] += m->rotor[i - 1];
  }
}

/* bias b (18): M_B a + b = tau. wB = base angular velocity (base comps), gB = R^T (0,0,-9.81). */
static void bias_forces_B(const robot_model *m, const kin_t *k, const real *wB, const real *qd, const real *gB, real *b) {
  real w[NB][3], al[NB][3], a[NB][3], f[NB][3], n[NB][3], rc[NB][3];
  v3_copy(w[0], wB);
  v3_set(al[0], RC(0), RC(0), RC(0));
  v3_scale(a[0], gB, RC(-1)); /* gravity trick */
  for (int i = 0; i < NB; i++) {
    if (i > 0) {
      int par = k_parent[i];
      real sq[3], d[3], t1[3], t2[3];
      v3_scale(sq, k->s[i], qd[i - 1]);
      v3_add(w[i], w[par], sq);
      v3_cross(t1, w[par], sq);
      v3_add(al[i], al[par], t1);
      v3_sub(d, k->p[i], k->p[par]);
      v3_cross(t1, al[par], d);
      v3_cross(t2, w[par], d);
      v3_cross(t2, w[par], t2);
      v3_add(a[i], a[par], t1);
      v3_add(a[i], a[i], t2);
    }
    real ac[3], t1[3], t2[3], RI[9], IB[9], Iw[3];
    v3_sub(rc[i], k->c[i], k->p[i]);
    v3_cross(t1, al[i], rc[i]);
    v3_cross(t2, w[i], rc[i]);
    v3_cross(t2, w[i], t2);
    v3_add(ac, a[i], t1);
    v3_add(ac, ac, t2);
    v3_scale(f[i], ac, m->mass[i]);
    m3_mul(RI, k->R[i], m->inertia[i]);
    m3_mul_bt(IB, RI, k->R[i]);
    m3_mulv(n[i], IB, al[i]);
    m3_mulv(Iw, IB, w[i]);
    v3_cross(t1, w[i], Iw);
    v3_add(n[i], n[i], t1);
    /* moment about the body's own origin */
    v3_cross(t1, rc[i], f[i]);
    v3_add(n[i], n[i], t1);
  }
  for (int i = NB - 1; i >= 1; i--) {
    int par = k_parent[i];
    real d[3], t1[3];
    b[6 + (i - 1)] = v3_dot(k->s[i], n[i]);
    v3_sub(d, k->p[i], k->p[par]);
    v3_cross(t1, d, f[i]);
    v3_add(n[par], n[par], n[i]);
    v3_add(n[par], n[par], t1);
    v3_add(f[par], f[par], f[i]);
  }
  for (int a_ = 0; a_ < 3; a_++) { b[a_] = f[0][a_]; b[3 + a_] = n[0][a_]; }
}

/* dense Cholesky M = L L^T (in place, lower), returns 0 on success */
static int chol_factor(real *A, int n) {
  for (int j = 0; j < n; j++) {
    real d = A[j * n + j];
    for (int k2 = 0; k2 < j; k2++) d -= A[j * n + k2] * A[j * n + k2];
    if (!(d > RC(0))) return 1;
    d = R_SQRT(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      real v = A[i * n + j];
      for (int k2 = 0; k2 < j; k2++) v -= A[i * n + k2] * A[j * n + k2];
      A[i * n + j] = v / d;
    }
  }
  return 0;
}
static void chol_solve(const real *L, int n, real *x) {
  for (int i = 0; i < n; i++) {
    real v = x[i];
    for (int k2 = 0; k2 < i; k2++) v -= L[i * n + k2] * x[k2];
    x[i] = v / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    real v = x[i];
    for (int k2 = i + 1; k2 < n; k2++) v -= L[k2 * n + i] * x[k2];
    x[i] = v / L[i * n + i];
  }
}

/* one-contact solve: velocity without own impulse c, local Delassus G (3x3), unit normal n,
 * target normal speed vstar, friction mu.  The build's first rule (ContactSolver bit 0 clear). */
static void solve_contact_dir(const real *G, const real *c, const real *n, real vstar, real mu, real *lam) {
  real cn = v3_dot(c, n) - vstar;
  if (cn >= RC(0)) { v3_set(lam, RC(0), RC(0), RC(0)); return; }
  real rhs[3], l[3], Gn[3];
  for (int i = 0; i < 3; i++) rhs[i] = -(c[i] - vstar * n[i]);
  m3_solve(G, rhs, l);
  m3_mulv(Gn, G, n);
  real nGn = v3_dot(n, Gn);
  real ln = v3_dot(l, n);
  real lt[3];
  for (int i = 0; i < 3; i++) lt[i] = l[i] - ln * n[i];
  real lt2 = v3_dot(lt, lt);
  /* the sticking impulse is admissible only if it pushes and stays inside the friction cone */
  if (ln > RC(0) && lt2 <= mu * mu * ln * ln) { v3_copy(lam, l); return; }
  /* otherwise the pressing contact slides: friction mu lam_n along the direction in which the sticking impulse would have
   * pushed (it opposes the slip), normal velocity condition kept exact (up to the cap below) */
  real lt2c = lt2 > RC(1e-30) ? lt2 : RC(1e-30);
  real inv = RC(1) / R_SQRT(lt2c), w[3], Gw[3];
  for (int i = 0; i < 3; i++) w[i] = n[i] + mu * lt[i] * inv;
  m3_mulv(Gw, G, w);
  real nGw = v3_dot(n, Gw);
  /* n.G w -> 0 is the jamming (Painleve) corner, where the exact sliding impulse diverges: the normal impulse is capped
   * at 5 times the frictionless one (continuous, bounded; the uncancelled approach velocity is left to the next substep) */
  real den = nGw > RC(0.2) * nGn ? nGw : RC(0.2) * nGn;
  v3_scale(lam, w, -cn / den);
}

/* one-contact solve, the PUBLISHED rule of the reference's physics engine (ContactSolver bit 0 set).  RaiSim (ENV:768 world_->integrate(),
 * closed source) resolves contacts with the per-contact iteration of Hwangbo, Lee & Hutter, "Per-Contact Iteration Method for
 * Solving Contact Dynamics", RA-L 2018: every single-contact problem is solved EXACTLY under Signorini's
 * condition, the Coulomb cone and the maximum-dissipation principle --
 *   opening  (c.n >= v*):                                    lam = 0
 *   sticking (lam_s = -G^-1 (c - v* n) pushes, inside cone): lam = lam_s
 *   slipping: the point of {cone boundary} x {v_n+ = v*} that minimises the post-impact kinetic energy
 *             h(lam) = 1/2 lam^T G lam + lam^T (c - v* n).
 * The paper finds that point by bisection on the polar angle of the conic; here it is found in closed form up to a scalar
 * root (same point; tests/test_contact_model_gap.py compares against a brute-force scan + bisection in numpy):
 *   contact frame (t1, t2, n); eliminate lam_n with the normal condition, lam_n = alpha + beta . x, alpha = -(c.n - v*) / G_nn,
 *   beta = -G_tn / G_nn, x = tangential impulse  =>  h = 1/2 x^T A x + b^T x with A = G_tt - G_tn G_tn^T / G_nn (Schur complement),
 *   b = c_t + alpha G_tn; its free minimiser x* = -A^-1 b is the sticking impulse.  The cone section |x| <= mu (alpha + beta . x)
 *   is, in xi = x / alpha, the FIXED ellipse (xi - xi_c)^T P (xi - xi_c) <= 1 with P = (1 - mu^2 beta beta^T) s / mu^2,
 *   xi_c = mu^2 beta / s, s = 1 - mu^2 |beta|^2 (focus at the origin, eccentricity mu |beta|).  Minimising the A-metric distance
 *   to xi* over the ellipse is a 2x2 trust-region problem: (A + gamma P)(xi - xi_c) = A (xi* - xi_c) with gamma >= 0 from the
 *   scalar equation  (xi - xi_c)^T P (xi - xi_c) = 1,  solved by Newton steps on  p(gamma) / sqrt(r(gamma)) = sigma
 *   (the reciprocal form: concave and increasing, so Newton is monotone after its first step from any start;
 *   p = det(A + gamma P), r = quadratic in gamma, sigma = |xi* - xi_c|).  Start: the root of the isotropic problem along the
 *   direction d of the far point, gamma_0 = a (sigma sqrt(rho) - 1) / rho with a = d^T A d, rho = d^T P d; MD_NEWTON = 2 steps
 *   from there reach 2e-9 of the converged point on 1100 captured sliding problems (from gamma = 0 it takes 4).
 * Caps (never active on the shipped configurations, where mu |beta| <= 0.86 over 1.2e5 captured contact problems):
 *   mu |beta| > sqrt(1 - MD_SMIN): the conic is a parabola / hyperbola (jamming corner); beta is shortened in the cone section to
 *   the ellipse of eccentricity sqrt(1 - MD_SMIN);  sigma > MD_SIGMAX (a barely pressing contact sliding fast: alpha -> 0, the
 *   impulse is ~1e-4 of a pressing one): the far point is pulled in to MD_SIGMAX;  mu < MD_MUMIN: frictionless, lam = alpha n.
 * The HIP kernels run this same algorithm (csrc/env_core.hpp solve_contact_md), step for step. */
#define MD_NEWTON 2
#define MD_SMIN 0.04
#define MD_SIGMAX 1.0e4
#define MD_MUMIN 1.0e-6
static void solve_contact_md(const real *G, const real *c, const real *n, real vstar, real mu_in, real *lam) {
  const real cn = v3_dot(c, n) - vstar;
  if (cn >= RC(0)) { v3_set(lam, RC(0), RC(0), RC(0)); return; }
  const real mu = mu_in > RC(MD_MUMIN) ? mu_in : RC(MD_MUMIN);
  /* branch-free orthonormal frame around n (Duff et al. 2017) */
  const real sg = n[2] >= RC(0) ? RC(1) : RC(-1);
  const real fa = RC(-1) / (sg + n[2]), fb = n[0] * n[1] * fa;
  const real t1[3] = {RC(1) + sg * n[0] * n[0] * fa, sg * fb, -sg * n[0]};
  const real t2[3] = {fb, sg + n[1] * n[1] * fa, -n[1]};
  real Gt1[3], Gt2[3], Gn[3];
  m3_mulv(Gt1, G, t1); m3_mulv(Gt2, G, t2); m3_mulv(Gn, G, n);
  const real g1 = v3_dot(t1, Gn), g2 = v3_dot(t2, Gn), ignn = RC(1) / v3_dot(n, Gn);
  const real A11 = v3_dot(t1, Gt1) - g1 * g1 * ignn, A12 = v3_dot(t1, Gt2) - g1 * g2 * ignn, A22 = v3_dot(t2, Gt2) - g2 * g2 * ignn;
  const real be1 = -g1 * ignn, be2 = -g2 * ignn;
  const real detA = A11 * A22 - A12 * A12, idetA = RC(1) / detA;
  /* sticking impulse */
  const real alpha = -cn * ignn;
  const real b1 = v3_dot(c, t1) + alpha * g1, b2 = v3_dot(c, t2) + alpha * g2;
  const real x1 = -(A22 * b1 - A12 * b2) * idetA, x2 = -(A11 * b2 - A12 * b1) * idetA;
  const real ln = alpha + be1 * x1 + be2 * x2;
  if (ln > RC(0) && x1 * x1 + x2 * x2 <= mu * mu * ln * ln) {
    for (int i = 0; i < 3; i++) lam[i] = x1 * t1[i] + x2 * t2[i] + ln * n[i];
    return;
  }
  if (mu_in <= RC(MD_MUMIN)) { v3_scale(lam, n, alpha); return; }
  /* slipping: the fixed ellipse of the cone section in xi = x / alpha */
  const real mb2 = mu * mu * (be1 * be1 + be2 * be2);
  const real shrink = mb2 <= RC(1.0 - MD_SMIN) ? RC(1) : R_SQRT(RC(1.0 - MD_SMIN) / mb2);
  const real e1 = shrink * mu * be1, e2 = shrink * mu * be2;       /* mu beta (capped) */
  const real s = RC(1) - (e1 * e1 + e2 * e2), is = RC(1) / s, ims = s / (mu * mu);
  const real P11 = (RC(1) - e1 * e1) * ims, P12 = -e1 * e2 * ims, P22 = (RC(1) - e2 * e2) * ims;
  const real detP = P11 * P22 - P12 * P12, mix = A22 * P11 - RC(2) * A12 * P12 + A11 * P22;
  const real zc1 = mu * e1 * is, zc2 = mu * e2 * is;
  /* the far point relative to the centre, as direction d and distance sigma */
  const real ia = RC(1) / alpha;
  real d1 = x1 * ia - zc1, d2 = x2 * ia - zc2;
  const real s2 = d1 * d1 + d2 * d2;
  const real isg = RC(1) / R_SQRT(s2 > RC(1e-30) ? s2 : RC(1e-30));
  const real sig = s2 * isg < RC(MD_SIGMAX) ? s2 * isg : RC(MD_SIGMAX);
  d1 *= isg; d2 *= isg;
  /* (A + gamma P)^-1 A d = (u + gamma q) / p(gamma):  u = adj(A) A d = detA d,  q = adj(P) A d */
  const real w1 = A11 * d1 + A12 * d2, w2 = A12 * d1 + A22 * d2;
  const real u1 = detA * d1, u2 = detA * d2;
  const real q1 = P22 * w1 - P12 * w2, q2 = P11 * w2 - P12 * w1;
  const real Pu1 = P11 * u1 + P12 * u2, Pu2 = P12 * u1 + P22 * u2, Pq1 = P11 * q1 + P12 * q2, Pq2 = P12 * q1 + P22 * q2;
  const real c0 = u1 * Pu1 + u2 * Pu2, c1 = RC(2) * (q1 * Pu1 + q2 * Pu2), c2 = q1 * Pq1 + q2 * Pq2;
  const real rho = c0 * idetA * idetA, g0 = (w1 * d1 + w2 * d2) * (sig * R_SQRT(rho) - RC(1)) / rho;
  real gam = g0 > RC(0) ? g0 : RC(0);
  for (int it = 0; it < MD_NEWTON; it++) {
    const real p = (detP * gam + mix) * gam + detA, r = (c2 * gam + c1) * gam + c0;
    const real dp = RC(2) * detP * gam + mix, dr = RC(2) * c2 * gam + c1;
    gam += r * (sig * R_SQRT(r) - p) / (dp * r - RC(0.5) * p * dr);
  }
  const real kk = sig / ((detP * gam + mix) * gam + detA);
  const real X1 = alpha * ((u1 + gam * q1) * kk + zc1), X2 = alpha * ((u2 + gam * q2) * kk + zc2);
  const real lnn = alpha + be1 * X1 + be2 * X2;                      /* normal velocity condition exact */
  for (int i = 0; i < 3; i++) lam[i] = X1 * t1[i] + X2 * t2[i] + lnn * n[i];
}
#define solve_contact(RULE, G, c, n, vstar, mu, lam) ((RULE) ? solve_contact_md(G, c, n, vstar, mu, lam) : solve_contact_dir(G, c, n, vstar, mu, lam))

/* ---- height field (Terrain: True).  RaiSim's Perlin terrain (ENV:254-264) is closed source; the spec is the
 * build's own: improved Perlin noise, LCG-shuffled permutation seeded by seedd,
 * h = zScale * sum_o gain^o noise(freq lac^o (x,y)), 5000 x 500 samples over 500 m x 20 m centred at the origin,
 * bilinear sampling, normal from the cell gradient.  Written independently of csrc/irrl_terrain.hpp. ---- */
#define HF_NX 5000
#define HF_NY 500
#define HF_XSIZE 500.0
#define HF_YSIZE 20.0
static double pn_fade(double t) { return t * t * t * (t * (t * 6.0 - 15.0) + 10.0); }
static double pn_lerp(double t, double a, double b) { return a + t * (b - a); }
static double pn_grad(int hash, double x, double y, double z) {
  int h = hash & 15;
  double u = (h < 8) ? x : y;
  double v = (h < 4) ? y : ((h == 12 || h == 14) ? x : z);
  return (((h & 1) == 0) ? u : -u) + (((h & 2) == 0) ? v : -v);
}
static double pn_noise2(const int *perm, double x, double y) {
  double fx = floor(x), fy = floor(y);
  int X = ((int)fx) & 255, Y = ((int)fy) & 255;
  x -= fx; y -= fy;
  double u = pn_fade(x), v = pn_fade(y), w = pn_fade(0.0);
  int A = perm[X] + Y, AA = perm[A], AB = perm[A + 1], B = perm[X + 1] + Y, BA = perm[B], BB = perm[B + 1];
  double lo = pn_lerp(v, pn_lerp(u, pn_grad(perm[AA], x, y, 0.0), pn_grad(perm[BA], x - 1, y, 0.0)),
                      pn_lerp(u, pn_grad(perm[AB], x, y - 1, 0.0), pn_grad(perm[BB], x - 1, y - 1, 0.0)));
  double hi = pn_lerp(v, pn_lerp(u, pn_grad(perm[AA + 1], x, y, -1.0), pn_grad(perm[BA + 1], x - 1, y, -1.0)),
                      pn_lerp(u, pn_grad(perm[AB + 1], x, y - 1, -1.0), pn_grad(perm[BB + 1], x - 1, y - 1, -1.0)));
  return pn_lerp(w, lo, hi);
}
static float *make_heightfield(uint32_t seed) {
  int base[256], perm[512];
  for (int i = 0; i < 256; i++) base[i] = i;
  uint32_t s = seed * 747796405u + 2891336453u;
  for (int i = 255; i > 0; i--) {
    s = s * 1664525u + 1013904223u;
    int j = (int)((s >> 8) % (uint32_t)(i + 1));
    int t = base[i]; base[i] = base[j]; base[j] = t;
  }
  for (int i = 0; i < 512; i++) perm[i] = base[i & 255];
  float *H = (float *)malloc(sizeof(float) * (size_t)HF_NX * HF_NY);
  const double dx = HF_XSIZE / (HF_NX - 1), dy = HF_YSIZE / (HF_NY - 1);
  for (int i = 0; i < HF_NX; i++) {
    double x = -0.5 * HF_XSIZE + i * dx;
    for (int j = 0; j < HF_NY; j++) {
      double y = -0.5 * HF_YSIZE + j * dy;
      double f = 1.0, a = 1.0, h = 0.0;
      for (int o = 0; o < 3; o++) { h += a * pn_noise2(perm, x * f, y * f); f *= 2.0; a *= 0.25; }
      H[(size_t)i * HF_NY + j] = (float)(0.1 * h);
    }
  }
  return H;
}
/* height and unit normal at world (x, y); flat ground when H == NULL */
static void terrain_sample(const float *H, real x, real y, real *h, real *n) {
  if (!H) { *h = RC(0); n[0] = RC(0); n[1] = RC(0); n[2] = RC(1); return; }
  const real inv_dx = RC((HF_NX - 1) / HF_XSIZE), inv_dy = RC((HF_NY - 1) / HF_YSIZE);
  real fx = (x + RC(0.5 * HF_XSIZE)) * inv_dx, fy = (y + RC(0.5 * HF_YSIZE)) * inv_dy;
  if (fx < RC(0)) fx = RC(0);
  if (fx > RC(HF_NX - 1.001)) fx = RC(HF_NX - 1.001);
  if (fy < RC(0)) fy = RC(0);
  if (fy > RC(HF_NY - 1.001)) fy = RC(HF_NY - 1.001);
  int i = (int)R_TO_DOUBLE(fx), j = (int)R_TO_DOUBLE(fy);
  real tx = fx - RC(i), ty = fy - RC(j);
  real h00 = RC((double)H[(size_t)i * HF_NY + j]), h01 = RC((double)H[(size_t)i * HF_NY + j + 1]);
  real h10 = RC((double)H[(size_t)(i + 1) * HF_NY + j]), h11 = RC((double)H[(size_t)(i + 1) * HF_NY + j + 1]);
  real a = h00 + ty * (h01 - h00), b = h10 + ty * (h11 - h10);
  *h = a + tx * (b - a);
  real dhdx = (b - a) * inv_dx;
  real dhdy = ((h01 - h00) + tx * ((h11 - h10) - (h01 - h00))) * inv_dy;
  real inv = RC(1) / R_SQRT(dhdx * dhdx + dhdy * dhdy + RC(1));
  n[0] = -dhdx * inv; n[1] = -dhdy * inv; n[2] = inv;
}

/* ------------------------------------------------------------------ 6. task logic */
typedef struct {
  /* physics state (RaiSim conventions) */
  real gc[NQ], gv[NV];
  real pTargetLast[12];
  real torque_last[12]; /* NORMALISED torque of the previous reward evaluation (ENV:1511-1515) */
  real torque[12];      /* last applied joint torque (clamped) */
  real jointRef[12], jointRefLast[12], jointDotRef[12], eeRef[12];
  real command[3], command_filtered[3];
  real t0;             /* episode start time (ENV:557) */
  int32_t frame_idx;   /* control steps since reset; current_time = t0 + frame_idx*dt */
  uint32_t episode;    /* ENV:554 itera */
  real up_height;
  real contact[4];      /* ENV:1116-1194 */
  real lam_w[4][3];     /* contact impulses of the last substep, world components */
  int32_t in_contact[4];/* contact-list membership of the last substep */
  real force_norm[4], vel_norm[4];
  real ob[35], ob_last[35];
  real bodyLinVel[3], bodyAngVel[3], Rwb[9];
  real rew_terms[8];
  real extra[6];
  real gen_force[NV]; /* last generalized force handed to the integrator (world comps for the base) */
  long gs_sweeps, gs_substeps; /* statistics: contact sweeps executed / substeps */
  long gs_hist[64];            /* statistics: substeps (with at least one contact) by sweeps taken (last bin: >= 63) */
  long box_hits;               /* statistics: (trunk-box corner, substep) pairs in contact */
  long box_substeps, box_sweeps; /* substeps with at least one corner in contact, and the sweeps they took */
  /* the meteorite of Crutial: True (ENV:273-284, 815-861): the CubeNum spheres are created at the same point with the same
   * radius, mass and velocity (cube_place_radius = 0, ENV:1976), so they are carried as ONE sphere of CubeNum times the mass */
  real sph_p[3], sph_v[3], sph_rad, sph_mass;
  int32_t sph_dyn;             /* raisim::BodyType: 0 STATIC (parked, no dynamics), 1 DYNAMIC (released) */
  long sph_hits;               /* statistics: substeps in which the sphere touched the trunk box */
  real box_lam[8][3];          /* corner impulses of the previous substep of THIS control step (warm start; base components) */
  int box_was_active[8];
  robot_model model;
} env_t;

struct orc_env {
  orc_cfg cfg;
  int n;
  env_t *envs;
  real obMean[35], obStd[35];
  real phase[4];
  real Kp[12], Kd[12];
  real filter_para, obs_filter_alpha;
  real max_len;
  double flops;
  float *height; /* NULL = flat ground */
  float *ref;    /* reference-trajectory table [ref_rows, 30] (ManualTraj: False), NULL otherwise */
  int ref_rows;
  /* contact-problem probe (tests): the last substep's toe contact problem of env `probe_env` (-1: off) */
  int probe_env;
  double probe_G[12 * 12], probe_cfree[12], probe_n[12], probe_vstar[4], probe_lam[12];
  int probe_active[4];
};

static real env_time(const orc_env *h, const env_t *e) { return e->t0 + RC(e->frame_idx) * RC(h->cfg.control_dt); }

static const double L_HIP = 0.085, L_THIGH = 0.209, L_CALF = 0.2175; /* ENV:1949-1952 */

/* ENV:1756-1890. cfg scalars passed explicitly so the unit probe can reuse it. */
static void gait_generator_manual(const orc_cfg *cfg, const real *phase, real max_len, const real *cmd_f, real t,
                                  int is_first, real *jointRefLast, real *jointRef, real *jointDotRef, real *eeRef,
                                  real *up_height) {
  static const real ee_off[12] = {RC(0.19), RC(-0.058), RC(0), RC(0.19), RC(0.058), RC(0),
                                  RC(-0.19), RC(-0.058), RC(0), RC(-0.19), RC(0.058), RC(0)}; /* ENV:331-334 */
  real lam = RC(cfg->lam), period = RC(cfg->period), stand = RC(cfg->stand_height), dt = RC(cfg->control_dt);
  real l_hip = RC(L_HIP), l_thigh = RC(L_THIGH), l_calf = RC(L_CALF);
  real temp[3] = {RC(0), RC(0), RC(0)}; /* shared across legs and across both passes (ENV:1766) */
  real gait_step = cmd_f[0] * lam * period;
  if (cfg->WILDCAT) gait_step = -gait_step;
  real side_step = cmd_f[1] * lam * period;
  real rot_step = cmd_f[2] * period * RC(0.4);
  if (cfg->HeightVariable) { /* ENV:1779-1792 */
    real ratio = R_FABS(cmd_f[0]) / RC(cfg->Vx);
    if (cfg->Vy > 0) { real r2 = R_FABS(cmd_f[1]) / RC(cfg->Vy); ratio = (ratio > r2) ? ratio : r2; }
    if (cfg->Omega > 0) { real r3 = R_FABS(cmd_f[2] / RC(cfg->Omega)); ratio = (ratio > r3) ? ratio : r3; }
    *up_height = (ratio > RC(0.1)) ? RC(cfg->up_height) : ratio * RC(cfg->up_height);
  }
  real toff[4];
  toff[0] = -l_hip + RC(cfg->LeanFront); toff[1] = l_hip - RC(cfg->LeanFront);
  toff[2] = -l_hip + RC(cfg->LeanHind);  toff[3] = l_hip - RC(cfg->LeanHind);
  for (int pass = is_first ? 0 : 1; pass < 2; pass++) {
    for (int i = 0; i < 4; i++) {
      real rp = t + phase[i] * period - ((pass == 0) ? dt : RC(0));
      rp = R_FMOD(rp, period) / period;
      real anti = (i < 2) ? RC(1.0) : RC(-1.0);
      real p0[3], pf[3], toe[3];
      if (rp < lam) {
        real r = rp / lam;
        v3_set(p0, gait_step / RC(2), side_step / RC(2) + anti * rot_step / RC(2), -stand);
        v3_set(pf, -gait_step / RC(2), -side_step / RC(2) + -anti * rot_step / RC(2), -stand);
        cubic_bezier(p0, pf, r, toe);
      } else {
        real r = (rp - lam) / (RC(1.0) - lam);
        v3_set(pf, gait_step / RC(2), side_step / RC(2) + anti * rot_step / RC(2), -stand);
        v3_set(p0, -gait_step / RC(2), -side_step / RC(2) + -anti * rot_step / RC(2), -stand);
        bezier2(p0, pf, r, *up_height, toe);
      }
      inverse_kinematics(toe[0], toe[1] + toff[i], toe[2], l_hip, l_thigh, l_calf, max_len, (i == 0 || i == 2), temp);
      if (pass == 0) {
        jointRefLast[3 * i + 0] = temp[0]; jointRefLast[3 * i + 1] = -temp[1]; jointRefLast[3 * i + 2] = -temp[2];
      } else {
        jointRef[3 * i + 0] = temp[0]; jointRef[3 * i + 1] = -temp[1]; jointRef[3 * i + 2] = -temp[2];
        eeRef[3 * i + 0] = toe[0]; eeRef[3 * i + 1] = toe[1]; eeRef[3 * i + 2] = toe[2];
      }
    }
  }
  for (int j = 0; j < 12; j++) {
    jointDotRef[j] = (jointRef[j] - jointRefLast[j]) / dt;
    jointRefLast[j] = jointRef[j];
    eeRef[j] = eeRef[j] + ee_off[j];
  }
}

static int ref_traj_mode(const orc_cfg *c) { return !c->ManualTraj && !c->Manual; }
/* row frame_idx of the table (ENV:972,1102,1670: ref.row(frame_idx)), clamped to the table */
static const float *ref_row(const orc_env *h, const env_t *e) {
  static const float zero_row[30] = {0};
  if (!h->ref) return zero_row;
  int f = e->frame_idx;
  if (f < 0) f = 0;
  if (f > h->ref_rows - 1) f = h->ref_rows - 1;
  return h->ref + (size_t)f * 30;
}
/* ENV:1010-1109: ManualTraj branch, and the reference-trajectory branch ENV:1100-1107 + gait_generator() ENV:1667-1671. */
static void command_obs_update(orc_env *h, env_t *e, int env_id, int flag_reset) {
  const orc_cfg *c = &h->cfg;
  if (c->Manual) return;
  if (ref_traj_mode(c)) {
    const float *row = ref_row(h, e);
    for (int i = 0; i < 3; i++) { e->ob[i] = RC((double)row[27 + i]); e->command_filtered[i] = e->ob[i]; }
    for (int j = 0; j < 12; j++) { e->jointRef[j] = RC((double)row[j]); e->jointDotRef[j] = RC((double)row[12 + j]); }
    return;
  }
  rng_addr a = {(uint32_t)c->seedd, (uint32_t)(env_id + c->EnvIdOffset), e->episode, (uint32_t)e->frame_idx};
  real u[4];
  rng_u01x4(&a, flag_reset ? P_RESET_CMD : P_CMD, u);
  real temp = u[0];
  if (temp < RC(0.5) / (RC(c->max_time) / RC(c->control_dt)) || flag_reset) {
    temp = u[1];
    /* ENV:1039-1045: "for (auto cmd : command) cmd = 0" iterates by value -> no effect. */
    if (RC(0.2) < temp && temp <= RC(0.7)) {
      real v = u[2];
      e->command[0] = v * RC(c->Vx) + (RC(1.0) - v) * RC(0.0); /* Vx_min stays 0 (ENV:1607,2054) */
    } else if (RC(0.7) < temp && temp <= RC(0.85)) {
      real v = u[2];
      e->command[1] = v * RC(c->Vy) + (RC(1.0) - v) * RC(-c->Vy);
    } else {
      real v = u[2];
      e->command[2] = v * RC(c->Omega) + (RC(1.0) - v) * RC(-c->Omega);
    }
  }
  if (flag_reset) {
    for (int i = 0; i < 3; i++) e->command_filtered[i] = e->command[i];
  } else {
    for (int i = 0; i < 3; i++)
      e->command_filtered[i] = e->command_filtered[i] * RC(0.995) + e->command[i] * (RC(1) - RC(0.995)); /* ENV:2043 */
  }
  for (int i = 0; i < 3; i++) e->ob[i] = e->command_filtered[i];
  gait_generator_manual(c, h->phase, h->max_len, e->command_filtered, env_time(h, e), flag_reset, e->jointRefLast,
                        e->jointRef, e->jointDotRef, e->eeRef, &e->up_height);
}

/* ENV:1116-1194 */
static void contact_obs_update(orc_env *h, env_t *e) {
  const orc_cfg *c = &h->cfg;
  if (!c->TimeBasedContact) {
    for (int i = 0; i < 4; i++) e->contact[i] = e->in_contact[i] ? RC(1.0) : RC(0.0);
  } else {
    for (int i = 0; i < 4; i++) {
      /* ENV:1172-1185 computes this phase in float */
      float rp = (float)(R_TO_DOUBLE(env_time(h, e)) + R_TO_DOUBLE(h->phase[i]) * c->period);
      rp = (float)(fmod((double)rp, c->period) / c->period);
      e->contact[i] = (rp < c->lam) ? RC(1.0) : RC(0.0);
    }
  }
}

/* ENV:956-1004 */
static void update_observation(orc_env *h, env_t *e, int env_id) {
  const orc_cfg *c = &h->cfg;
  rng_addr a = {(uint32_t)c->seedd, (uint32_t)(env_id + c->EnvIdOffset), e->episode, (uint32_t)e->frame_idx};
  for (int i = 0; i < 35; i++) e->ob[i] = RC(0); /* ENV:960 zeroes all 35; obs[0:3] is rewritten by command_obs_update */
  real t = env_time(h, e);
  if (ref_traj_mode(c)) { /* ENV:972 */
    const float *row = ref_row(h, e);
    e->ob[3] = RC((double)row[25]);
    e->ob[4] = RC((double)row[26]);
  } else {
    e->ob[3] = R_SIN(RC(2) * RC(REF_PI) * t / RC(c->period));
    e->ob[4] = R_COS(RC(2) * RC(REF_PI) * t / RC(c->period));
  }
  real nf = RC(c->ObsNoise);
  real uj[12], uv[12], nn[6];
  int noisy = (c->ObsNoise != 0.0);
  if (noisy) {
    rng_fill_u01(&a, P_OBS_JOINT, 12, uj);
    rng_fill_u01(&a, P_OBS_JVEL, 12, uv);
    rng_fill_normal(&a, P_OBS_NORMAL, 6, nn);
  } else {
    for (int i = 0; i < 12; i++) { uj[i] = RC(0.5); uv[i] = RC(0.5); }
    for (int i = 0; i < 6; i++) nn[i] = RC(0);
  }
  for (int j = 0; j < 12; j++) {
    e->ob[5 + j] = ((RC(2) * uj[j] - RC(1)) * RC(0.002) * nf) + e->gc[7 + j];  /* ENV:979,1988 */
    e->ob[17 + j] = ((RC(2) * uv[j] - RC(1)) * RC(0.8) * nf) + e->gv[6 + j];    /* ENV:983,1989 */
  }
  quat_to_rot(&e->gc[3], e->Rwb);
  for (int k = 0; k < 3; k++) e->ob[29 + k] = e->Rwb[6 + k] + (nn[k] * RC(0.02)) * nf; /* ENV:994-996,1997 */
  m3_tmulv(e->bodyLinVel, e->Rwb, &e->gv[0]);
  m3_tmulv(e->bodyAngVel, e->Rwb, &e->gv[3]);
  for (int k = 0; k < 3; k++) e->ob[32 + k] = e->bodyAngVel[k] + nf * (nn[3 + k] * RC(0.5)); /* ENV:1001-1003,1999 */
}

/* toe frame ("toe_xx_joint") world position / velocity */
static void toe_world(const env_t *e, const kin_t *k, int l, real *pos_w, real *vel_w, real *pos_B) {
  int s = 3 + 3 * l;
  real off[3] = {RC(0), RC(0), RC(TOE_Z)}, x[3], v[3], t1[3];
  m3_mulv(x, k->R[s], off);
  v3_add(x, x, k->p[s]);
  if (pos_B) v3_copy(pos_B, x);
  if (pos_w) { m3_mulv(pos_w, e->Rwb, x); v3_add(pos_w, pos_w, &e->gc[0]); }
  if (vel_w) {
    real vB[3], wB[3];
    m3_tmulv(vB, e->Rwb, &e->gv[0]);
    m3_tmulv(wB, e->Rwb, &e->gv[3]);
    v3_cross(v, wB, x);
    v3_add(v, v, vB);
    for (int b = 1 + 3 * l; b <= s; b++) {
      real d[3];
      v3_sub(d, x, k->p[b]);
      v3_cross(t1, k->s[b], d);
      v3_axpy(v, e->gv[6 + (b - 1)], t1);
    }
    m3_mulv(vel_w, e->Rwb, v);
  }
}

/* ENV:1199-1243 */
static void contact_information_update(orc_env *h, env_t *e) {
  kin_t k;
  forward_kinematics(&e->model, &e->gc[7], &k);
  for (int l = 0; l < 4; l++) {
    real n2 = v3_dot(e->lam_w[l], e->lam_w[l]);
    e->force_norm[l] = e->in_contact[l] ? R_SQRT(n2) / RC(h->cfg.control_dt) : RC(0); /* control_dt quirk ENV:1208 */
    real v[3];
    toe_world(e, &k, l, NULL, v, NULL);
    e->vel_norm[l] = R_SQRT(v3_dot(v, v));
  }
}

/* ENV:1444-1548 */
static real deep_mimic_reward(orc_env *h, env_t *e) {
  const orc_cfg *c = &h->cfg;
  kin_t k;
  forward_kinematics(&e->model, &e->gc[7], &k);
  real ee2 = RC(0);
  for (int l = 0; l < 4; l++) {
    real xB[3];
    toe_world(e, &k, l, NULL, NULL, xB); /* R^T (p_toe - p_body) == base-frame toe position */
    for (int a = 0; a < 3; a++) { real d = xB[a] - e->eeRef[3 * l + a]; ee2 += d * d; }
  }
  real EE = RC(c->EndEffectorRewardCoeff) * R_EXP(RC(-40) * ee2);
  real dz = e->gc[2] - RC(c->stand_height);
  real BC = RC(c->BodyPosRewardCoeff) * R_EXP(RC(-80) * (dz * dz));
  real BA = RC(c->BodyAttitudeRewardCoeff) * R_EXP(RC(-80) * (e->ob[29] * e->ob[29] + e->ob[30] * e->ob[30]));
  real j2 = RC(0), jd2 = RC(0);
  for (int j = 0; j < 12; j++) {
    real d = e->jointRef[j] - e->gc[7 + j]; j2 += d * d;
    real dd = e->jointDotRef[j] - e->gv[6 + j]; jd2 += dd * dd;
  }
  real JR = RC(c->JointRewardCoeff) * RC(0.25) * R_EXP(RC(-2.0) * j2);
  real JD = RC(c->JointRewardCoeff) * RC(0.75) * R_EXP(-RC(c->control_dt) * jd2);
  real lref[3] = {e->command_filtered[0], e->command_filtered[1], RC(0)};
  if (c->WILDCAT) lref[0] = -lref[0];
  real aref[3] = {RC(0), RC(0), e->command_filtered[2]};
  real lv2 = RC(0), av2 = RC(0);
  for (int a = 0; a < 3; a++) {
    real d1 = e->bodyLinVel[a] - lref[a]; lv2 += d1 * d1;
    real d2 = e->bodyAngVel[a] - aref[a]; av2 += d2 * d2;
  }
  real VR = RC(c->VelRewardCoeff) / RC(2) * R_EXP(RC(-2) * lv2) + RC(c->VelRewardCoeff) / RC(2) * R_EXP(RC(-2) * av2);
  static const real tlim[3] = {RC(18), RC(18), RC(27)}; /* ENV:354 */
  real tn2 = RC(0), td2 = RC(0), tn[12];
  for (int j = 0; j < 12; j++) {
    tn[j] = e->torque[j] / tlim[j % 3];
    tn2 += tn[j] * tn[j];
    real d = tn[j] - e->torque_last[j]; td2 += d * d;
  }
  real TR = RC(c->TorqueCoeff) / RC(2.0) * R_EXP(RC(-0.1) * tn2) +
            RC(c->TorqueCoeff) / RC(2.0) * R_EXP(RC(-0.1) / RC(c->control_dt) * td2);
  for (int j = 0; j < 12; j++) e->torque_last[j] = tn[j]; /* ENV:1511,1515: the NORMALISED torque is what is kept */
  real cr = RC(0), t = env_time(h, e);
  for (int i = 0; i < 4; i++) {
    real rp = t + h->phase[i] * RC(c->period);
    rp = R_FMOD(rp, RC(c->period)) / RC(c->period);
    cr += RC(4) * e->vel_norm[i] * e->vel_norm[i] * smooth_function(rp, RC(2), RC(c->lam));
    cr += RC(2) * (e->force_norm[i] / RC(12.5)) * (e->force_norm[i] / RC(12.5)) * smooth_function2(rp, RC(2), RC(c->lam));
  }
  real CR = RC(c->ContactCoeff) * R_EXP(RC(-2) * cr);
  e->rew_terms[0] = EE; e->rew_terms[1] = BC; e->rew_terms[2] = BA; e->rew_terms[3] = JR;
  e->rew_terms[4] = JD; e->rew_terms[5] = VR; e->rew_terms[6] = TR; e->rew_terms[7] = CR;
  return EE + BC + JR + JD + VR + BA + TR + CR; /* summation order of ENV:1546-1547 */
}

/* ENV:1553-1578 */

/* ---- the meteorite (Crutial: True) ----
 * ENV:815-842 meteoriteAttack(true): the spheres are re-created STATIC at gc_ + (0.05, 0, 1.0) with radius
 * (t/5 + 1) * cube_len and mass t/5 + 0.2 each ("steel"); ENV:845-858 meteoriteAttack(false) on a parked sphere: DYNAMIC with
 * velocity (gv_[0], gv_[1], -5). */
#define CUBE_LEN 0.08 /* ENV:1974 */
static void sphere_place(const orc_cfg *c, env_t *e, const real *base_pos, real t) {
  e->sph_p[0] = base_pos[0] + RC(0.05); e->sph_p[1] = base_pos[1]; e->sph_p[2] = base_pos[2] + RC(1.0);
  v3_set(e->sph_v, RC(0), RC(0), RC(0));
  e->sph_rad = (t / RC(5.0) + RC(1.0)) * RC(CUBE_LEN);
  e->sph_mass = (t / RC(5.0) + RC(0.2)) * RC((double)(c->CubeNum > 0 ? c->CubeNum : 1));
  e->sph_dyn = 0;
}
static void sphere_release(env_t *e) {
  if (e->sph_dyn) return;
  e->sph_dyn = 1;
  v3_set(e->sph_v, e->gv[0], e->gv[1], RC(-5.0));
}

static int is_terminal(const env_t *e) { return (e->gc[2] < RC(0.15) || e->gc[2] > RC(0.65) || e->ob[31] < RC(0.5)); }

/* one physics substep: PD + clamp (ENV:761-767) then the build's integrate() (stands in for ENV:768) */
static void physics_substep(orc_env *h, env_t *e, const real *pTarget) {
  const orc_cfg *c = &h->cfg;
  const robot_model *m = &e->model;
  real dt = RC(c->simulation_dt);
  real *q = &e->gc[7], *qd = &e->gv[6];
  const int rule_md = c->ContactSolver & 1;
  /* PD law + 1 % blend with torque_last + clamp */
  real tau[12];
  for (int j = 0; j < 12; j++) {
    real t = (pTarget[j] - q[j]) * h->Kp[j] - qd[j] * h->Kd[j];
    tau[j] = RC(0.99) * t + (RC(1.0) - RC(0.99)) * e->torque_last[j];
  }
  torque_clamp(tau, qd, RC(c->MotorMaxTorque), RC(c->MotorCriticalSpeed), RC(c->MotorMaxSpeed), NULL, NULL);
  for (int j = 0; j < 12; j++) e->torque[j] = tau[j];
  for (int a = 0; a < 6; a++) e->gen_force[a] = RC(0);
  for (int j = 0; j < 12; j++) e->gen_force[6 + j] = tau[j];

  real R[9];
  quat_to_rot(&e->gc[3], R);
  kin_t k;
  forward_kinematics(m, q, &k);
  real M[NV * NV], L[NV * NV], b[NV], u[NV], rhs[NV];
  mass_matrix_B(m, &k, M);
  real vB[3], wB[3], gB[3], gw[3] = {RC(0), RC(0), RC(-GRAV)};
  m3_tmulv(vB, R, &e->gv[0]);
  m3_tmulv(wB, R, &e->gv[3]);
  m3_tmulv(gB, R, gw);
  bias_forces_B(m, &k, wB, qd, gB, b);
  for (int a = 0; a < 3; a++) { u[a] = vB[a]; u[3 + a] = wB[a]; }
  for (int j = 0; j < 12; j++) u[6 + j] = qd[j];
  memcpy(L, M, sizeof(M));
  chol_factor(L, NV);
  for (int a = 0; a < 6; a++) rhs[a] = -b[a];
  for (int j = 0; j < 12; j++) rhs[6 + j] = tau[j] - m->damping[j] * qd[j] - b[6 + j];
  chol_solve(L, NV, rhs);
  real ufree[NV];
  for (int i = 0; i < NV; i++) ufree[i] = u[i] + dt * rhs[i];

  /* contact detection + Jacobians.  Contacts 0..3: the toe spheres; 4..11: the eight corners of the trunk's collision box
   * (URDF:26 box 0.3 x 0.2 x 0.1 centred on the base origin, collision body "body/0" ENV:242) as points against the same
   * ground with the same (default) material -- the ("steel","steel") pair of ENV:244 only concerns steel-steel pairs.
   * Corner i: x = +-0.15 (bit 2 set: -), y = +-0.1 (bit 1 set: -), z = -+0.05 (bit 0 set: +).  Gauss-Seidel order: toes
   * FR, FL, HR, HL iterate (below); the corners follow in ONE pass of sequential impulses, 0..7, cold start. */
  enum { NC = 12 };
  int active[NC];
  real J[NC][3][NV], MiJt[NC][3][NV], cfree[NC][3], vstar[NC], lamB[NC][3], nBl[NC][3];
  static const real box_half[3] = {RC(0.15), RC(0.1), RC(0.05)};
  for (int ci = 0; ci < NC; ci++) {
    real xc[3], cw[3], radius;
    int l = ci, s = 3 + 3 * ci;
    if (ci < 4) {
      real off[3] = {RC(0), RC(0), RC(TOE_Z)};
      m3_mulv(xc, k.R[s], off);
      v3_add(xc, xc, k.p[s]);
      radius = RC(TOE_RADIUS);
    } else {
      int b = ci - 4;
      xc[0] = (b & 4) ? -box_half[0] : box_half[0];
      xc[1] = (b & 2) ? -box_half[1] : box_half[1];
      xc[2] = (b & 1) ? box_half[2] : -box_half[2];
      radius = RC(0);
    }
    m3_mulv(cw, R, xc);
    /* sphere (or point) against the locally planar ground: distance of the centre to the tangent plane minus the radius */
    real hgt, nw[3];
    terrain_sample(h->height, e->gc[0] + cw[0], e->gc[1] + cw[1], &hgt, nw);
    real gap = (e->gc[2] + cw[2] - hgt) * nw[2] - radius;
    active[ci] = (gap <= RC(0));
    if (!active[ci]) { v3_set(lamB[ci], RC(0), RC(0), RC(0)); continue; }
    real *nB = nBl[ci];
    m3_tmulv(nB, R, nw);
    real x[3];
    for (int a = 0; a < 3; a++) x[a] = xc[a] - radius * nB[a];
    for (int r = 0; r < 3; r++) for (int cc = 0; cc < NV; cc++) J[ci][r][cc] = RC(0);
    for (int r = 0; r < 3; r++) J[ci][r][r] = RC(1);
    /* -[x]x */
    J[ci][0][4] = x[2]; J[ci][0][5] = -x[1];
    J[ci][1][3] = -x[2]; J[ci][1][5] = x[0];
    J[ci][2][3] = x[1]; J[ci][2][4] = -x[0];
    if (ci < 4)
      for (int bdy = 1 + 3 * l; bdy <= s; bdy++) {
        real d[3], col[3];
        v3_sub(d, x, k.p[bdy]);
        v3_cross(col, k.s[bdy], d);
        for (int r = 0; r < 3; r++) J[ci][r][6 + (bdy - 1)] = col[r];
      }
    real vpre[3];
    for (int r = 0; r < 3; r++) {
      real a1 = RC(0), a2 = RC(0);
      for (int cc = 0; cc < NV; cc++) { a1 += J[ci][r][cc] * u[cc]; a2 += J[ci][r][cc] * ufree[cc]; }
      vpre[r] = a1; cfree[ci][r] = a2;
      for (int cc = 0; cc < NV; cc++) MiJt[ci][r][cc] = J[ci][r][cc];
      chol_solve(L, NV, MiJt[ci][r]);
    }
    real vn = v3_dot(vpre, nB);
    vstar[ci] = (vn < -m->rest_thr) ? (-m->rest * vn) : RC(0);
    /* warm start: previous impulse if the foot was already in the contact list */
    if (ci < 4 && e->in_contact[ci]) m3_tmulv(lamB[ci], R, e->lam_w[ci]); else v3_set(lamB[ci], RC(0), RC(0), RC(0));
  }
  real G[4][4][9];
  for (int la = 0; la < 4; la++)
    for (int lb = 0; lb < 4; lb++)
      if (active[la] && active[lb])
        for (int r = 0; r < 3; r++)
          for (int r2 = 0; r2 < 3; r2++) {
            real acc = RC(0);
            for (int cc = 0; cc < NV; cc++) acc += J[la][r][cc] * MiJt[lb][r2][cc];
            G[la][lb][3 * r + r2] = acc;
          }
  int sweeps_done = 0;
  /* THE PER-CONTACT ITERATION OVER THE TOES.  ContactSolver bit 1 is the ORDER inside a sweep: clear = Gauss-Seidel FR, FL, HR, HL
   * (the published method); set = the four toes update simultaneously from the sweep's starting iterate -- they couple only
   * through the heavy base, the off-diagonal Delassus blocks are a fraction of the diagonal ones, and this converges almost as
   * fast as Gauss-Seidel (2.4 against 2.2 sweeps per substep, same fixed point: measured) while the lock-step
   * kernels need ONE solve per sweep instead of one per contact.  Bit 0 is the per-contact RULE (solve_contact_md / _dir). */
  int group_of[4], n_groups;
  if (c->ContactSolver & 2) { for (int l = 0; l < 4; l++) group_of[l] = 0; n_groups = 1; }
  else { for (int l = 0; l < 4; l++) group_of[l] = l; n_groups = 4; }
  real dlam[4][3] = {{RC(0)}};
  for (int it = 0; it < c->ContactIterations; it++) {
    real d2 = RC(0), l2 = RC(0);
    for (int g = 0; g < n_groups; g++) {
      real newl[4][3];
      for (int l = 0; l < 4; l++) {
        if (!active[l] || group_of[l] != g) continue;
        real cv[3];
        v3_copy(cv, cfree[l]);
        for (int lb = 0; lb < 4; lb++) {
          if (lb == l || !active[lb]) continue;
          real t[3];
          m3_mulv(t, G[l][lb], lamB[lb]);
          v3_add(cv, cv, t);
        }
        solve_contact(rule_md, G[l][l], cv, nBl[l], vstar[l], m->mu, newl[l]);
      }
      for (int l = 0; l < 4; l++) {
        if (!active[l] || group_of[l] != g) continue;
        for (int a = 0; a < 3; a++) {
          real dd = newl[l][a] - lamB[l][a];
          dlam[l][a] = dd;
          lamB[l][a] += dd;
          d2 += dd * dd; l2 += lamB[l][a] * lamB[l][a];
        }
      }
    }
    /* build-defined early exit (same rule in the kernels, evaluated per wave there) */
    if (c->ContactTolerance > 0 && n_groups == 1 && c->ContactExit == 1) {
      /* PREDICTED exit (orc_cfg::ContactExit): would the NEXT sweep still move the impulses by more than the tolerance?  Its change is
       * estimated contact by contact from the velocity change this sweep's dlambda causes at the contact, answered linearly. */
      if (it + 1 >= c->ContactIterations) { it++; sweeps_done = it; goto gs_done; }
      real p2 = RC(0);
      for (int l = 0; l < 4; l++) {
        if (!active[l]) continue;
        real dc[3] = {RC(0), RC(0), RC(0)}, ans[3];
        for (int lb = 0; lb < 4; lb++) {
          if (lb == l || !active[lb]) continue;
          real t[3];
          m3_mulv(t, G[l][lb], dlam[lb]);
          v3_add(dc, dc, t);
        }
        m3_solve(G[l][l], dc, ans);
        p2 += v3_dot(ans, ans);
      }
      if (p2 <= RC(c->ContactTolerance * c->ContactTolerance) * l2 + RC(1e-20)) { it++; sweeps_done = it; goto gs_done; }
    } else if (c->ContactTolerance > 0 && d2 <= RC(c->ContactTolerance * c->ContactTolerance) * l2 + RC(1e-20)) { it++; sweeps_done = it; goto gs_done; }
  }
  sweeps_done = c->ContactIterations;
gs_done:
  e->gs_sweeps += sweeps_done;
  e->gs_substeps += 1;
  { int anyc = 0; for (int l = 0; l < 4; l++) anyc |= active[l]; if (anyc) e->gs_hist[sweeps_done < 63 ? sweeps_done : 63] += 1; }
  { int anyb = 0; for (int l = 4; l < NC; l++) anyb |= active[l]; if (anyb) { e->box_substeps += 1; e->box_sweeps += sweeps_done; } }
  if (h->probe_env >= 0 && e == &h->envs[h->probe_env]) {   /* tests: hand out the toe contact problem and its solution */
    for (int la = 0; la < 4; la++) {
      h->probe_active[la] = active[la];
      h->probe_vstar[la] = active[la] ? R_TO_DOUBLE(vstar[la]) : 0.0;
      for (int r = 0; r < 3; r++) {
        h->probe_cfree[3 * la + r] = active[la] ? R_TO_DOUBLE(cfree[la][r]) : 0.0;
        h->probe_n[3 * la + r] = active[la] ? R_TO_DOUBLE(nBl[la][r]) : 0.0;
        h->probe_lam[3 * la + r] = R_TO_DOUBLE(lamB[la][r]);
        for (int lb = 0; lb < 4; lb++) for (int r2 = 0; r2 < 3; r2++)
          h->probe_G[(3 * la + r) * 12 + 3 * lb + r2] = (active[la] && active[lb]) ? R_TO_DOUBLE(G[la][lb][3 * r + r2]) : 0.0;
      }
    }
  }
  for (int l = 0; l < 4; l++) {
    e->in_contact[l] = active[l];
    if (!active[l]) { v3_set(e->lam_w[l], RC(0), RC(0), RC(0)); continue; }
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < NV; cc++) ufree[cc] += MiJt[l][r][cc] * lamB[l][r];
    m3_mulv(e->lam_w[l], R, lamB[l]);
  }
  /* THE TRUNK-BOX CORNERS: one pass of sequential impulses behind the toe iteration, corners 0..7 in order, cold start.  Each
   * touching corner sees the velocity the toes (and the corners before it) have produced, solves its own single-contact problem
   * exactly (same rule as a toe: restitution target, Coulomb cone) and is applied at once; nothing is re-iterated, so what a
   * later corner does to an earlier one -- and to the toes -- is left to the next 0.25 ms substep.  A corner reaches the ground
   * only on a robot that is falling over (it terminates at 60 degrees of tilt or 0.15 m of height) or on rough terrain: cheap
   * and momentum-consistent matters more there than a converged coupled solve. */
  for (int l = 4; l < NC; l++) {
    if (!active[l]) continue;
    real Gc[9], cv[3], lam[3];
    for (int r = 0; r < 3; r++) {
      real acc = RC(0);
      for (int cc = 0; cc < NV; cc++) acc += J[l][r][cc] * ufree[cc];
      cv[r] = acc;
      for (int r2 = 0; r2 < 3; r2++) {
        real g = RC(0);
        for (int cc = 0; cc < NV; cc++) g += J[l][r][cc] * MiJt[l][r2][cc];
        Gc[3 * r + r2] = g;
      }
    }
    solve_contact(rule_md, Gc, cv, nBl[l], vstar[l], m->mu, lam);
    for (int r = 0; r < 3; r++)
      for (int cc = 0; cc < NV; cc++) ufree[cc] += MiJt[l][r][cc] * lam[r];
    e->box_hits += 1;
  }
  /* THE METEORITE (Crutial: True), released spheres only (a parked one hangs 1 m above where the base was; nothing reaches it).
   * One pass of sequential impulses behind the corners, same single-contact rule:
   *   sphere - trunk box ("steel"-"steel": mu 0, e 0.95, threshold 0.001, ENV:244): closest point of the box (URDF:26) to the
   *     sphere centre; the trunk side of the Delassus block is J M^-1 J^T of that point, the sphere side I / m_s;
   *   sphere - ground (default material pair of this env);
   * then the sphere integrates like everything else (semi-implicit Euler).  Sphere - leg contacts are not modelled. */
  if (c->Crutial && e->sph_dyn) {
    real sv_pre[3], ms_inv = RC(1) / e->sph_mass;
    v3_copy(sv_pre, e->sph_v);
    e->sph_v[2] -= RC(GRAV) * dt;
    real dw[3], cB[3], qB[3], dd[3];
    v3_sub(dw, e->sph_p, &e->gc[0]);
    m3_tmulv(cB, R, dw);
    for (int a = 0; a < 3; a++) { qB[a] = cB[a] > box_half[a] ? box_half[a] : (cB[a] < -box_half[a] ? -box_half[a] : cB[a]); dd[a] = cB[a] - qB[a]; }
    real dist2 = v3_dot(dd, dd);
    if (dist2 < e->sph_rad * e->sph_rad) {
      real nB[3], Js[3][NV], MiJs[3][NV], Gc[9], cv[3], lam[3], svB[3], pre[3], tmp[3], svB_pre[3];
      if (dist2 > RC(1e-18)) { real inv = RC(1) / R_SQRT(dist2); for (int a = 0; a < 3; a++) nB[a] = -dd[a] * inv; }
      else v3_set(nB, RC(0), RC(0), RC(-1)); /* centre inside the box: the sphere leaves through the top face */
      for (int r = 0; r < 3; r++) for (int cc = 0; cc < NV; cc++) Js[r][cc] = RC(0);
      /* point velocity v + w x q: rows [ I | -[q]x ] */
      Js[0][0] = Js[1][1] = Js[2][2] = RC(1);
      Js[0][4] = qB[2];  Js[0][5] = -qB[1];
      Js[1][3] = -qB[2]; Js[1][5] = qB[0];
      Js[2][3] = qB[1];  Js[2][4] = -qB[0];
      for (int r = 0; r < 3; r++) { memcpy(MiJs[r], Js[r], sizeof(Js[r])); chol_solve(L, NV, MiJs[r]); }
      m3_tmulv(svB, R, e->sph_v);
      m3_tmulv(svB_pre, R, sv_pre);
      for (int r = 0; r < 3; r++) {
        real acc = RC(0);
        for (int cc = 0; cc < NV; cc++) acc += Js[r][cc] * ufree[cc];
        cv[r] = acc - svB[r];
        for (int r2 = 0; r2 < 3; r2++) {
          real g = RC(0);
          for (int cc = 0; cc < NV; cc++) g += Js[r][cc] * MiJs[r2][cc];
          Gc[3 * r + r2] = g + (r == r2 ? ms_inv : RC(0));
        }
      }
      v3_cross(tmp, wB, qB);
      for (int a = 0; a < 3; a++) pre[a] = vB[a] + tmp[a] - svB_pre[a];
      real vn = v3_dot(pre, nB);
      real vs = vn < RC(-0.001) ? RC(-0.95) * vn : RC(0);
      solve_contact(rule_md, Gc, cv, nB, vs, RC(0), lam);
      for (int r = 0; r < 3; r++)
        for (int cc = 0; cc < NV; cc++) ufree[cc] += MiJs[r][cc] * lam[r];
      for (int a = 0; a < 3; a++) svB[a] -= lam[a] * ms_inv;
      m3_mulv(e->sph_v, R, svB);
      e->sph_hits += 1;
    }
    {
      real hgt, nw[3];
      if (h->height) terrain_sample(h->height, e->sph_p[0], e->sph_p[1], &hgt, nw);
      else { hgt = RC(0); v3_set(nw, RC(0), RC(0), RC(1)); }
      if ((e->sph_p[2] - hgt) * nw[2] - e->sph_rad <= RC(0)) {
        real Gc[9] = {ms_inv, RC(0), RC(0), RC(0), ms_inv, RC(0), RC(0), RC(0), ms_inv}, lam[3];
        real vn = v3_dot(sv_pre, nw);
        real vs = vn < -m->rest_thr ? -m->rest * vn : RC(0);
        solve_contact(rule_md, Gc, e->sph_v, nw, vs, m->mu, lam);
        v3_axpy(e->sph_v, ms_inv, lam);
      }
    }
    v3_axpy(e->sph_p, dt, e->sph_v);
  }
  /* back to world-frame gv, then positions (semi-implicit Euler) */
  m3_mulv(&e->gv[0], R, &ufree[0]);
  m3_mulv(&e->gv[3], R, &ufree[3]);
  for (int j = 0; j < 12; j++) { qd[j] = ufree[6 + j]; q[j] += dt * qd[j]; }
  for (int a = 0; a < 3; a++) e->gc[a] += dt * e->gv[a];
  {
    real *qq = &e->gc[3], w0 = qq[0], x0 = qq[1], y0 = qq[2], z0 = qq[3];
    real hx = RC(0.5) * dt * e->gv[3], hy = RC(0.5) * dt * e->gv[4], hz = RC(0.5) * dt * e->gv[5];
    /* q+ = normalize(q + 0.5 dt (0,w) (x) q), world-frame w => left multiplication */
    real w1 = w0 - hx * x0 - hy * y0 - hz * z0;
    real x1 = x0 + hx * w0 + hy * z0 - hz * y0;
    real y1 = y0 - hx * z0 + hy * w0 + hz * x0;
    real z1 = z0 + hx * y0 - hy * x0 + hz * w0;
    real inv = RC(1) / R_SQRT(w1 * w1 + x1 * x1 + y1 * y1 + z1 * z1);
    qq[0] = w1 * inv; qq[1] = x1 * inv; qq[2] = y1 * inv; qq[3] = z1 * inv;
  }
}

/* ENV:547-635 */
static void env_reset(orc_env *h, env_t *e, int env_id) {
  const orc_cfg *c = &h->cfg;
  e->episode++;
  e->frame_idx = 0;
  rng_addr a = {(uint32_t)c->seedd, (uint32_t)(env_id + c->EnvIdOffset), e->episode, 0u};
  real u[4];
  if (c->RandomizePerEpisode && c->StochasticDynamics) model_randomize(&e->model, &a);
  rng_u01x4(&a, P_RESET_TIME, u);
  e->t0 = c->Manual ? RC(0.0) : u[0];
  if (ref_traj_mode(c)) { /* ENV:538-539, 571: frame_max = rows / 2, frame_len = max_time / control_dt */
    int span = h->ref_rows / 2 - (int)(c->max_time / c->control_dt) - 10;
    e->frame_idx = span > 0 ? (int)R_TO_DOUBLE(R_FLOOR(RC((double)span) * sampling_reshape(u[1]))) : 0;
  }
  /* ENV:608-612: the meteorite is parked above gc_ -- which still holds the base position of the state BEFORE this reset
   * (the new state is set further down, ENV:617-623) -- sized by the new episode's start time */
  if (c->Crutial) sphere_place(c, e, &e->gc[0], e->t0);
  for (int i = 0; i < 3; i++) e->command_filtered[i] = RC(0);
  for (int j = 0; j < 12; j++) e->torque_last[j] = RC(0);
  command_obs_update(h, e, env_id, 1);
  contact_obs_update(h, e);
  /* initial state (ENV:581-623) */
  real gc0[NQ], gv0[NV];
  real abad = RC(c->abad);
  real nominal[12] = {-abad, RC(-0.78), RC(1.57), abad, RC(-0.78), RC(1.57), -abad, RC(-0.78), RC(1.57), abad, RC(-0.78), RC(1.57)};
  for (int i = 0; i < NQ; i++) gc0[i] = RC(0);
  for (int i = 0; i < NV; i++) gv0[i] = RC(0);
  gc0[2] = RC(0.35); gc0[3] = RC(1);
  if (c->Manual) {
    for (int j = 0; j < 12; j++) gc0[7 + j] = nominal[j];
  } else {
    real nj[12], nv[12], nb[4], uxy[4];
    if (c->SharedNoiseScalar) {
      rng_u01x4(&a, P_RESET_JOINT, u);
      for (int j = 0; j < 12; j++) { nj[j] = RC(2) * u[0] - RC(1); nv[j] = RC(2) * u[1] - RC(1); }
    } else {
      real t24[24];
      rng_fill_u01(&a, P_RESET_JOINT_IND, 24, t24);
      for (int j = 0; j < 12; j++) { nj[j] = RC(2) * t24[j] - RC(1); nv[j] = RC(2) * t24[12 + j] - RC(1); }
    }
    for (int j = 0; j < 12; j++) {
      gc0[7 + j] = e->jointRef[j] * (nj[j] * RC(0.3)) + e->jointRef[j];
      gv0[6 + j] = e->jointDotRef[j] * (nv[j] * RC(0.3)) + e->jointDotRef[j];
    }
    rng_u01x4(&a, P_RESET_BASE, nb);
    gv0[0] = e->command_filtered[0] * ((RC(2) * nb[0] - RC(1)) * RC(0.2) + RC(1.0));
    if (c->WILDCAT) gv0[0] = -gv0[0];
    gv0[1] = e->command_filtered[1] * ((RC(2) * nb[1] - RC(1)) * RC(0.2) + RC(1.0));
    gv0[5] = e->command_filtered[2] * ((RC(2) * nb[2] - RC(1)) * RC(0.2) + RC(1.0));
    rng_u01x4(&a, P_RESET_XY, uxy);
    gc0[0] = uxy[0] * RC(5.0) + (RC(1.0) - uxy[0]) * RC(-5.0);
    gc0[1] = uxy[1] * RC(5.0) + (RC(1.0) - uxy[1]) * RC(-5.0);
  }
  memcpy(e->gc, gc0, sizeof(gc0));
  memcpy(e->gv, gv0, sizeof(gv0));
  update_observation(h, e, env_id);
  memcpy(e->ob_last, e->ob, sizeof(e->ob));
  contact_obs_update(h, e);
  command_obs_update(h, e, env_id, 0);
  e->frame_idx++;
}

/* ENV:692-809 */
static real env_step(orc_env *h, env_t *e, int env_id, const float *action) {
  const orc_cfg *c = &h->cfg;
  real abad = RC(c->abad);
  real nominal[12] = {-abad, RC(-0.78), RC(1.57), abad, RC(-0.78), RC(1.57), -abad, RC(-0.78), RC(1.57), abad, RC(-0.78), RC(1.57)};
  real pT[12];
  rng_addr a = {(uint32_t)c->seedd, (uint32_t)(env_id + c->EnvIdOffset), e->episode, (uint32_t)e->frame_idx};
  real an[12];
  if (c->ActionNoise != 0.0) {
    if (c->SharedNoiseScalar) { real u[4]; rng_u01x4(&a, P_ACTION_NOISE, u); for (int j = 0; j < 12; j++) an[j] = RC(2) * u[0] - RC(1); }
    else { real t12[12]; rng_fill_u01(&a, P_ACTION_NOISE, 12, t12); for (int j = 0; j < 12; j++) an[j] = RC(2) * t12[j] - RC(1); }
  } else for (int j = 0; j < 12; j++) an[j] = RC(0);
  for (int j = 0; j < 12; j++) {
    real p = RC((double)action[j]) * RC(1.0) + nominal[j];
    p = (RC(1.0) - h->filter_para) * p + h->filter_para * e->pTargetLast[j];
    p = p * (RC(c->ActionNoise) * an[j]) + p;
    pT[j] = p;
    e->pTargetLast[j] = p;
  }
  if (c->Crutial) { /* ENV:731-740: every 5 gait periods the meteorite is parked above the robot, one control step later it is released */
    int K = (int)(5.0 * c->period / c->control_dt);
    if (K > 0 && e->frame_idx % K == 0) sphere_place(c, e, &e->gc[0], env_time(h, e));
    else sphere_release(e);
  }
  if (c->ForceDisturbance && c->Manual) { /* ENV:743-748 -> state_disturbance ENV:912-940 */
    int K = (int)(c->period / c->control_dt * 10.0);
    if (K > 0 && e->frame_idx % K == 0) {
      real ua[4], ub[4];
      rng_u01x4(&a, P_DISTURB, ua);
      rng_u01x4(&a, P_DISTURB + 1, ub);
      const real r = RC(0.5);
      e->gc[2] += RC(0.03) * (RC(2) * ua[0] - RC(1)) * r;
      e->gc[3] += RC(0.1) * (RC(2) * ua[1] - RC(1)) * r;
      e->gc[4] += RC(0.1) * (RC(2) * ua[2] - RC(1)) * r;
      e->gc[5] += RC(0.1) * (RC(2) * ua[3] - RC(1)) * r;
      e->gc[6] += RC(0.1) * (RC(2) * ub[0] - RC(1)) * r;
      real nq = R_SQRT(e->gc[3] * e->gc[3] + e->gc[4] * e->gc[4] + e->gc[5] * e->gc[5] + e->gc[6] * e->gc[6]);
      for (int k = 3; k < 7; k++) e->gc[k] = e->gc[k] / nq; /* build-defined: unit quaternion before it reaches the integrator */
      e->gv[2] += RC(0.1) * (RC(2) * ub[1] - RC(1)) * r;
      e->gv[3] += RC(0.3) * (RC(2) * ub[2] - RC(1)) * r;
      e->gv[4] += RC(0.3) * (RC(2) * ub[3] - RC(1)) * r;
    }
  }
  int loop = (int)(c->control_dt / c->simulation_dt + 1e-10); /* ENV:711 */
  for (int i = 0; i < loop; i++) physics_substep(h, e, pT);
  update_observation(h, e, env_id);
  contact_information_update(h, e);
  real r = deep_mimic_reward(h, e);
  command_obs_update(h, e, env_id, 0);
  contact_obs_update(h, e);
  e->frame_idx += 1;
  return r;
}

static void env_update_extra(env_t *e) { /* ENV:942-950, order fixed by this build */
  e->extra[0] = e->rew_terms[0]; e->extra[1] = e->rew_terms[1]; e->extra[2] = e->gc[2];
  e->extra[3] = e->rew_terms[2]; e->extra[4] = e->rew_terms[3]; e->extra[5] = e->rew_terms[5];
}

/* ENV:1248-1268 */
static void env_observe(orc_env *h, env_t *e, float *ob) {
  if (h->cfg.ObsFilter) {
    for (int i = 5; i < 35; i++) e->ob[i] = e->ob[i] * h->obs_filter_alpha + e->ob_last[i] * (RC(1.0) - h->obs_filter_alpha);
    memcpy(e->ob_last, e->ob, sizeof(e->ob));
  }
  for (int i = 0; i < 35; i++) ob[i] = (float)R_TO_DOUBLE((e->ob[i] - h->obMean[i]) / h->obStd[i]);
}

/* ------------------------------------------------------------------ 7. vector-env API */
static void obs_scaling(const orc_cfg *c, real *mean, real *std) {
  real abad = RC(c->abad);
  real nominal[12] = {-abad, RC(-0.78), RC(1.57), abad, RC(-0.78), RC(1.57), -abad, RC(-0.78), RC(1.57), abad, RC(-0.78), RC(1.57)};
  static const double jstd[3] = {5.0, 35.0, 40.0};
  for (int i = 0; i < 35; i++) { mean[i] = RC(0); std[i] = RC(1); }
  mean[0] = (RC(c->Vx) + RC(0.0)) / RC(2); /* Vx_min = 0 */
  mean[1] = (RC(c->Vy) + RC(-c->Vy)) / RC(2);
  mean[2] = (RC(c->Omega) + RC(-c->Omega)) / RC(2);
  for (int j = 0; j < 12; j++) { mean[5 + j] = nominal[j]; std[17 + j] = RC(jstd[j % 3]); }
  mean[31] = RC(1.0);
  for (int k = 0; k < 3; k++) { std[29 + k] = RC(0.7); std[32 + k] = RC(3.0); }
}

orc_env *orc_create(const orc_cfg *cfg) {
  if (!cfg || cfg->num_envs <= 0) return NULL;
  orc_env *h = (orc_env *)calloc(1, sizeof(orc_env));
  if (h) h->probe_env = -1;
  h->cfg = *cfg;
  if (h->cfg.ContactIterations <= 0) h->cfg.ContactIterations = 6;
  h->n = cfg->num_envs;
  h->envs = (env_t *)calloc((size_t)h->n, sizeof(env_t));
  obs_scaling(cfg, h->obMean, h->obStd);
  switch (cfg->GaitType) { /* ENV:398-409 */
    case 0: h->phase[0] = RC(0.5); h->phase[1] = RC(0.0); h->phase[2] = RC(0.0); h->phase[3] = RC(0.5); break;
    case 1: h->phase[0] = RC(0.5); h->phase[1] = RC(0.5); h->phase[2] = RC(0.0); h->phase[3] = RC(0.0); break;
    case 2: h->phase[0] = RC(0.0); h->phase[1] = RC(0.25); h->phase[2] = RC(0.5); h->phase[3] = RC(0.75); break;
    default: break; /* ENV: phase_ stays zero */
  }
  for (int j = 0; j < 12; j++) { /* ENV:338-350 */
    real ratio = (j % 3 == 0) ? RC(cfg->AbadRatio) : RC(1.0);
    h->Kp[j] = RC(cfg->Stiffness) * ratio;
    h->Kd[j] = RC(cfg->Damping) * ratio;
  }
  /* ENV:396,423-427: evaluated in the constructor, i.e. with the base-class default control_dt_ = 0.01
   * (BASE:111) because setControlTimeStep runs afterwards (VEC:150-152). */
  h->filter_para = cfg->Filter ? RC(1 - cfg->Freq * 0.01) : RC(0);
  h->obs_filter_alpha = cfg->ObsFilter ? RC(2.0 * 3.14 * 0.01 * 20.0 / (2.0 * 3.14 * 0.01 * 20.0 + 1.0)) : RC(1.0);
  h->max_len = R_SQRT(RC(L_HIP) * RC(L_HIP) + (RC(L_CALF) + RC(L_THIGH)) * (RC(L_CALF) + RC(L_THIGH)));
  h->height = cfg->Terrain ? make_heightfield((uint32_t)cfg->seedd) : NULL;
  return h;
}
void orc_destroy(orc_env *h) { if (h) { free(h->envs); free(h->height); free(h->ref); free(h); } }
/* ENV:1895 set_ref: [rows, cols >= 30] row-major f32; call before orc_init */
int orc_set_ref(orc_env *h, const float *table, int rows, int cols) {
  if (!ref_traj_mode(&h->cfg) || !table || rows < 2 || cols < 30) return 1;
  free(h->ref);
  h->ref = (float *)malloc((size_t)rows * 30 * sizeof(float));
  for (int r = 0; r < rows; r++) memcpy(h->ref + (size_t)r * 30, table + (size_t)r * cols, 30 * sizeof(float));
  h->ref_rows = rows;
  return 0;
}
int orc_num_envs(const orc_env *h) { return h->n; }
void orc_box_stats(const orc_env *h, long out[3]) { out[0] = out[1] = out[2] = 0; for (int i = 0; i < h->n; i++) { out[0] += h->envs[i].box_hits; out[1] += h->envs[i].box_substeps; out[2] += h->envs[i].box_sweeps; } }
void orc_sweep_histogram(const orc_env *h, long out[64]) { for (int k = 0; k < 64; k++) { out[k] = 0; for (int i = 0; i < h->n; i++) out[k] += h->envs[i].gs_hist[k]; } }
void orc_set_probe(orc_env *h, int env_id) { h->probe_env = env_id; }
void orc_get_probe(const orc_env *h, double *G, double *cfree, double *n, double *vstar, double *lam, int *active) {
  memcpy(G, h->probe_G, sizeof(h->probe_G)); memcpy(cfree, h->probe_cfree, sizeof(h->probe_cfree)); memcpy(n, h->probe_n, sizeof(h->probe_n));
  memcpy(vstar, h->probe_vstar, sizeof(h->probe_vstar)); memcpy(lam, h->probe_lam, sizeof(h->probe_lam)); memcpy(active, h->probe_active, sizeof(h->probe_active));
}
void orc_sphere_info(orc_env *h, float *out) { /* ENV:1423-1436 GetSphereInfo: centre of cubes[0] and its radius */
  for (int i = 0; i < h->n; i++) {
    const env_t *e = &h->envs[i];
    for (int k = 0; k < 3; k++) out[4 * i + k] = (float)R_TO_DOUBLE(e->sph_p[k]);
    out[4 * i + 3] = (float)R_TO_DOUBLE(e->sph_rad);
  }
}
long orc_sphere_hits(const orc_env *h) { long t = 0; for (int i = 0; i < h->n; i++) t += h->envs[i].sph_hits; return t; }
long orc_box_hits(const orc_env *h) { long t = 0; for (int i = 0; i < h->n; i++) t += h->envs[i].box_hits; return t; }
int orc_real_bytes(void) { return (int)sizeof(real); }

void orc_init(orc_env *h) {
  for (int i = 0; i < h->n; i++) {
    env_t *e = &h->envs[i];
    memset(e, 0, sizeof(*e));
    e->up_height = RC(h->cfg.up_height);
    real abad = RC(h->cfg.abad);
    real jr[12] = {-abad, RC(0), RC(0), abad, RC(0), RC(0), -abad, RC(0), RC(0), abad, RC(0), RC(0)}; /* ENV:415-418 */
    memcpy(e->jointRef, jr, sizeof(jr));
    rng_addr a = {(uint32_t)h->cfg.seedd, (uint32_t)(i + h->cfg.EnvIdOffset), 0u, 0u};
    if (h->cfg.StochasticDynamics) model_randomize(&e->model, &a); else model_nominal(&e->model);
    env_reset(h, e, i);
  }
}

void orc_reset(orc_env *h, float *ob) {
  for (int i = 0; i < h->n; i++) env_reset(h, &h->envs[i], i);
  orc_observe(h, ob);
}
void orc_observe(orc_env *h, float *ob) { for (int i = 0; i < h->n; i++) env_observe(h, &h->envs[i], ob + 35 * i); }

void orc_step(orc_env *h, const float *action, float *ob, float *reward, uint8_t *done, float *extra) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic)
#endif
  for (int i = 0; i < h->n; i++) { /* VEC:273-277, 352-372 */
    env_t *e = &h->envs[i];
    real r = env_step(h, e, i, action + 12 * i);
    float rf = (float)R_TO_DOUBLE(r);
    int d = is_terminal(e);
    float term = d ? (float)h->cfg.terminalRewardCoeff : 0.f;
    env_update_extra(e);
    for (int j = 0; j < 6; j++) extra[6 * i + j] = (float)R_TO_DOUBLE(e->extra[j]);
    if (d) { env_reset(h, e, i); rf += term; }
    reward[i] = rf;
    done[i] = (uint8_t)d;
    env_observe(h, e, ob + 35 * i);
  }
}
void orc_is_terminal(orc_env *h, uint8_t *done) { for (int i = 0; i < h->n; i++) done[i] = (uint8_t)is_terminal(&h->envs[i]); }
void orc_set_seed(orc_env *h, int seed) { h->cfg.seedd = seed; }
/* VEC:328-331 setControlTimeStep: every later use of control_dt_ (substep count ENV:711, time, cadences ENV:733,747, rewards) sees the new value */
void orc_set_control_dt(orc_env *h, double dt) { if (dt > 0) h->cfg.control_dt = dt; }

void orc_origin_state(orc_env *h, float *out) { /* ENV:1317-1325 */
  for (int i = 0; i < h->n; i++) {
    env_t *e = &h->envs[i];
    float *o = out + 41 * i;
    for (int k = 0; k < NQ; k++) o[k] = (float)R_TO_DOUBLE(e->gc[k]);
    for (int k = 0; k < NV; k++) o[NQ + k] = (float)R_TO_DOUBLE(e->gv[k]);
    for (int k = 0; k < 4; k++) o[NQ + NV + k] = (float)R_TO_DOUBLE(e->contact[k]);
  }
}
void orc_reference_state(orc_env *h, float *out) { /* ENV:1339-1345 (the VEC:223-226 dispatch bug is the shim's business) */
  for (int i = 0; i < h->n; i++) {
    env_t *e = &h->envs[i];
    for (int k = 0; k < 12; k++) { out[24 * i + k] = (float)R_TO_DOUBLE(e->jointRef[k]); out[24 * i + 12 + k] = (float)R_TO_DOUBLE(e->jointDotRef[k]); }
  }
}
void orc_joint_effort(orc_env *h, float *out) { for (int i = 0; i < h->n; i++) for (int k = 0; k < 12; k++) out[12 * i + k] = (float)R_TO_DOUBLE(h->envs[i].gen_force[6 + k]); }
void orc_generalized_force(orc_env *h, float *out) { for (int i = 0; i < h->n; i++) for (int k = 0; k < NV; k++) out[NV * i + k] = (float)R_TO_DOUBLE(h->envs[i].gen_force[k]); }

static void world_mass_matrix(const robot_model *m, const real *gc, real *Mw, real *Linv_opt) {
  kin_t k; real R[9], M[NV * NV];
  quat_to_rot(&gc[3], R);
  forward_kinematics(m, &gc[7], &k);
  mass_matrix_B(m, &k, M);
  /* M_w = T M_B T^T, T = diag(R, R, I) */
  real T[NV * NV];
  for (int i = 0; i < NV * NV; i++) T[i] = RC(0);
  for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) { T[a * NV + b] = R[3 * a + b]; T[(3 + a) * NV + 3 + b] = R[3 * a + b]; }
  for (int j = 6; j < NV; j++) T[j * NV + j] = RC(1);
  real TM[NV * NV];
  for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) { real acc = RC(0); for (int k2 = 0; k2 < NV; k2++) acc += T[i * NV + k2] * M[k2 * NV + j]; TM[i * NV + j] = acc; }
  for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) { real acc = RC(0); for (int k2 = 0; k2 < NV; k2++) acc += TM[i * NV + k2] * T[j * NV + k2]; Mw[i * NV + j] = acc; }
  (void)Linv_opt;
}
void orc_inverse_mass_matrix(orc_env *h, float *out) { /* ENV:1375-1391, column-major flatten (symmetric anyway) */
  for (int i = 0; i < h->n; i++) {
    real Mw[NV * NV], col[NV];
    world_mass_matrix(&h->envs[i].model, h->envs[i].gc, Mw, NULL);
    chol_factor(Mw, NV);
    for (int c2 = 0; c2 < NV; c2++) {
      for (int r = 0; r < NV; r++) col[r] = (r == c2) ? RC(1) : RC(0);
      chol_solve(Mw, NV, col);
      for (int r = 0; r < NV; r++) out[324 * i + c2 * NV + r] = (float)R_TO_DOUBLE(col[r]);
    }
  }
}
static void world_nonlinear(const robot_model *m, const real *gc, const real *gv, real *hw) {
  kin_t k; real R[9], wB[3], gB[3], gw[3] = {RC(0), RC(0), RC(-GRAV)}, b[NV];
  quat_to_rot(&gc[3], R);
  forward_kinematics(m, &gc[7], &k);
  m3_tmulv(wB, R, &gv[3]);
  m3_tmulv(gB, R, gw);
  bias_forces_B(m, &k, wB, &gv[6], gB, b);
  m3_mulv(&hw[0], R, &b[0]);
  m3_mulv(&hw[3], R, &b[3]);
  for (int j = 6; j < NV; j++) hw[j] = b[j];
}
void orc_nonlinear(orc_env *h, float *out) { /* ENV:1396-1402 */
  for (int i = 0; i < h->n; i++) {
    real hw[NV];
    world_nonlinear(&h->envs[i].model, h->envs[i].gc, h->envs[i].gv, hw);
    for (int k = 0; k < NV; k++) out[NV * i + k] = (float)R_TO_DOUBLE(hw[k]);
  }
}
void orc_set_contact_coeff(orc_env *h, const float *in) { /* ENV:1407-1418 */
  for (int i = 0; i < h->n; i++) {
    h->envs[i].model.mu = RC((double)in[3 * i]); h->envs[i].model.rest = RC((double)in[3 * i + 1]); h->envs[i].model.rest_thr = RC((double)in[3 * i + 2]);
  }
}

/* flat state layout shared with the product's irrl_env_get_state/set_state (include/irrl_env.h) */
enum { S_GC = 0, S_GV = 19, S_PTL = 37, S_TQL = 49, S_TQ = 61, S_JR = 73, S_JRL = 85, S_JDR = 97, S_EER = 109,
       S_CMD = 121, S_CMDF = 124, S_T0 = 127, S_FRAME = 128, S_EPISODE = 129, S_UPH = 130, S_CONTACT = 131,
       S_LAMW = 135, S_INCONTACT = 147, S_MATERIAL = 151, S_MASS = 154, S_COM = 167, S_THIGH = 206, S_OB = 207,
       S_OBLAST = 242, S_SPHERE = 277 /* pos 3, vel 3, radius, mass, dynamic */, S_END = 286 };

void orc_get_state(orc_env *h, double *out) {
  for (int i = 0; i < h->n; i++) {
    env_t *e = &h->envs[i]; double *o = out + (size_t)ORC_STATE_DIM * i;
    for (int k = 0; k < ORC_STATE_DIM; k++) o[k] = 0.0;
    for (int k = 0; k < NQ; k++) o[S_GC + k] = R_TO_DOUBLE(e->gc[k]);
    for (int k = 0; k < NV; k++) o[S_GV + k] = R_TO_DOUBLE(e->gv[k]);
    for (int k = 0; k < 12; k++) {
      o[S_PTL + k] = R_TO_DOUBLE(e->pTargetLast[k]); o[S_TQL + k] = R_TO_DOUBLE(e->torque_last[k]); o[S_TQ + k] = R_TO_DOUBLE(e->torque[k]);
      o[S_JR + k] = R_TO_DOUBLE(e->jointRef[k]); o[S_JRL + k] = R_TO_DOUBLE(e->jointRefLast[k]); o[S_JDR + k] = R_TO_DOUBLE(e->jointDotRef[k]);
      o[S_EER + k] = R_TO_DOUBLE(e->eeRef[k]); o[S_LAMW + k] = R_TO_DOUBLE(e->lam_w[k / 3][k % 3]);
    }
    for (int k = 0; k < 3; k++) { o[S_CMD + k] = R_TO_DOUBLE(e->command[k]); o[S_CMDF + k] = R_TO_DOUBLE(e->command_filtered[k]); }
    o[S_T0] = R_TO_DOUBLE(e->t0); o[S_FRAME] = (double)e->frame_idx; o[S_EPISODE] = (double)e->episode; o[S_UPH] = R_TO_DOUBLE(e->up_height);
    for (int k = 0; k < 4; k++) { o[S_CONTACT + k] = R_TO_DOUBLE(e->contact[k]); o[S_INCONTACT + k] = (double)e->in_contact[k]; }
    o[S_MATERIAL] = R_TO_DOUBLE(e->model.mu); o[S_MATERIAL + 1] = R_TO_DOUBLE(e->model.rest); o[S_MATERIAL + 2] = R_TO_DOUBLE(e->model.rest_thr);
    for (int k = 0; k < NB; k++) { o[S_MASS + k] = R_TO_DOUBLE(e->model.mass[k]); for (int a = 0; a < 3; a++) o[S_COM + 3 * k + a] = R_TO_DOUBLE(e->model.com[k][a]); }
    o[S_THIGH] = R_TO_DOUBLE(e->model.thigh_dz);
    for (int k = 0; k < 35; k++) { o[S_OB + k] = R_TO_DOUBLE(e->ob[k]); o[S_OBLAST + k] = R_TO_DOUBLE(e->ob_last[k]); }
    for (int k = 0; k < 3; k++) { o[S_SPHERE + k] = R_TO_DOUBLE(e->sph_p[k]); o[S_SPHERE + 3 + k] = R_TO_DOUBLE(e->sph_v[k]); }
    o[S_SPHERE + 6] = R_TO_DOUBLE(e->sph_rad); o[S_SPHERE + 7] = R_TO_DOUBLE(e->sph_mass); o[S_SPHERE + 8] = (double)e->sph_dyn;
  }
}
void orc_set_state(orc_env *h, const double *in) {
  for (int i = 0; i < h->n; i++) {
    env_t *e = &h->envs[i]; const double *o = in + (size_t)ORC_STATE_DIM * i;
    for (int k = 0; k < NQ; k++) e->gc[k] = RC(o[S_GC + k]);
    for (int k = 0; k < NV; k++) e->gv[k] = RC(o[S_GV + k]);
    for (int k = 0; k < 12; k++) {
      e->pTargetLast[k] = RC(o[S_PTL + k]); e->torque_last[k] = RC(o[S_TQL + k]); e->torque[k] = RC(o[S_TQ + k]);
      e->jointRef[k] = RC(o[S_JR + k]); e->jointRefLast[k] = RC(o[S_JRL + k]); e->jointDotRef[k] = RC(o[S_JDR + k]);
      e->eeRef[k] = RC(o[S_EER + k]); e->lam_w[k / 3][k % 3] = RC(o[S_LAMW + k]);
    }
    for (int k = 0; k < 3; k++) { e->command[k] = RC(o[S_CMD + k]); e->command_filtered[k] = RC(o[S_CMDF + k]); }
    e->t0 = RC(o[S_T0]); e->frame_idx = (int32_t)o[S_FRAME]; e->episode = (uint32_t)o[S_EPISODE]; e->up_height = RC(o[S_UPH]);
    for (int k = 0; k < 4; k++) { e->contact[k] = RC(o[S_CONTACT + k]); e->in_contact[k] = (int32_t)o[S_INCONTACT + k]; }
    model_nominal(&e->model);
    e->model.mu = RC(o[S_MATERIAL]); e->model.rest = RC(o[S_MATERIAL + 1]); e->model.rest_thr = RC(o[S_MATERIAL + 2]);
    for (int k = 0; k < NB; k++) { e->model.mass[k] = RC(o[S_MASS + k]); for (int a = 0; a < 3; a++) e->model.com[k][a] = RC(o[S_COM + 3 * k + a]); }
    e->model.thigh_dz = RC(o[S_THIGH]);
    for (int l = 0; l < NLEG; l++) e->model.jpos[3 + 3 * l][2] += e->model.thigh_dz;
    for (int k = 0; k < 35; k++) { e->ob[k] = RC(o[S_OB + k]); e->ob_last[k] = RC(o[S_OBLAST + k]); }
    for (int k = 0; k < 3; k++) { e->sph_p[k] = RC(o[S_SPHERE + k]); e->sph_v[k] = RC(o[S_SPHERE + 3 + k]); }
    e->sph_rad = RC(o[S_SPHERE + 6]); e->sph_mass = RC(o[S_SPHERE + 7]); e->sph_dyn = (int32_t)o[S_SPHERE + 8];
    quat_to_rot(&e->gc[3], e->Rwb);
    m3_tmulv(e->bodyLinVel, e->Rwb, &e->gv[0]);
    m3_tmulv(e->bodyAngVel, e->Rwb, &e->gv[3]);
  }
}

/* ---- unit probes (double in/out) ---- */
/* the single-contact solves for the tests: G row-major 3x3, c, n, out lam; rule 1 = published (solve_contact_md), 0 = the build's first rule */
void orc_solve_contact(int rule, const double G[9], const double c[3], const double n[3], double vstar, double mu, double lam[3]) {
  real g[9], cc[3], nn[3], l[3];
  for (int i = 0; i < 9; i++) g[i] = RC(G[i]);
  for (int i = 0; i < 3; i++) { cc[i] = RC(c[i]); nn[i] = RC(n[i]); }
  solve_contact(rule, g, cc, nn, RC(vstar), RC(mu), l);
  for (int i = 0; i < 3; i++) lam[i] = R_TO_DOUBLE(l[i]);
}
void orc_cubic_bezier(const double p0[3], const double pf[3], double s, double out[3]) {
  real a[3] = {RC(p0[0]), RC(p0[1]), RC(p0[2])}, b[3] = {RC(pf[0]), RC(pf[1]), RC(pf[2])}, o[3];
  cubic_bezier(a, b, RC(s), o);
  for (int i = 0; i < 3; i++) out[i] = R_TO_DOUBLE(o[i]);
}
double orc_gauss(double x, double width, double height) { return R_TO_DOUBLE(gauss_bump(RC(x), RC(width), RC(height))); }
void orc_bezier2(const double p0[3], const double pf[3], double s, double hgt, double out[3]) {
  real a[3] = {RC(p0[0]), RC(p0[1]), RC(p0[2])}, b[3] = {RC(pf[0]), RC(pf[1]), RC(pf[2])}, o[3];
  bezier2(a, b, RC(s), RC(hgt), o);
  for (int i = 0; i < 3; i++) out[i] = R_TO_DOUBLE(o[i]);
}
double orc_smooth_function(double p, double s, double l) { return R_TO_DOUBLE(smooth_function(RC(p), RC(s), RC(l))); }
double orc_smooth_function2(double p, double s, double l) { return R_TO_DOUBLE(smooth_function2(RC(p), RC(s), RC(l))); }
double orc_sampling_reshape(double r) { return R_TO_DOUBLE(sampling_reshape(RC(r))); }
int orc_inverse_kinematics(double x, double y, double z, double l_hip, double l_thigh, double l_calf, double max_len,
                           int is_right, double theta[3]) {
  real th[3] = {RC(theta[0]), RC(theta[1]), RC(theta[2])};
  int err = inverse_kinematics(RC(x), RC(y), RC(z), RC(l_hip), RC(l_thigh), RC(l_calf), RC(max_len), is_right, th);
  for (int i = 0; i < 3; i++) theta[i] = R_TO_DOUBLE(th[i]);
  return err;
}
void orc_torque_clamp(const double tau_in[12], const double qd[12], double tau_max, double w_crit, double w_max,
                      double tau_out[12], double upper[12], double lower[12]) {
  real t[12], v[12], up[12], lo[12];
  for (int i = 0; i < 12; i++) { t[i] = RC(tau_in[i]); v[i] = RC(qd[i]); }
  torque_clamp(t, v, RC(tau_max), RC(w_crit), RC(w_max), up, lo);
  for (int i = 0; i < 12; i++) { tau_out[i] = R_TO_DOUBLE(t[i]); upper[i] = R_TO_DOUBLE(up[i]); lower[i] = R_TO_DOUBLE(lo[i]); }
}
void orc_gait_reference(const orc_cfg *cfg, const double cmd_f[3], double t, int is_first, double jointRefLast[12],
                        double jointRef[12], double jointDotRef[12], double eeRef[12], double *up_height_io) {
  real ph[4] = {RC(0), RC(0), RC(0), RC(0)};
  switch (cfg->GaitType) {
    case 0: ph[0] = RC(0.5); ph[3] = RC(0.5); break;
    case 1: ph[0] = RC(0.5); ph[1] = RC(0.5); break;
    case 2: ph[1] = RC(0.25); ph[2] = RC(0.5); ph[3] = RC(0.75); break;
    default: break;
  }
  real ml = R_SQRT(RC(L_HIP) * RC(L_HIP) + (RC(L_CALF) + RC(L_THIGH)) * (RC(L_CALF) + RC(L_THIGH)));
  real c3[3] = {RC(cmd_f[0]), RC(cmd_f[1]), RC(cmd_f[2])}, jrl[12], jr[12], jdr[12], ee[12], uh = RC(*up_height_io);
  for (int i = 0; i < 12; i++) { jrl[i] = RC(jointRefLast[i]); jr[i] = RC(jointRef[i]); jdr[i] = RC(0); ee[i] = RC(0); }
  gait_generator_manual(cfg, ph, ml, c3, RC(t), is_first, jrl, jr, jdr, ee, &uh);
  for (int i = 0; i < 12; i++) { jointRefLast[i] = R_TO_DOUBLE(jrl[i]); jointRef[i] = R_TO_DOUBLE(jr[i]); jointDotRef[i] = R_TO_DOUBLE(jdr[i]); eeRef[i] = R_TO_DOUBLE(ee[i]); }
  *up_height_io = R_TO_DOUBLE(uh);
}
void orc_obs_scaling(const orc_cfg *cfg, double mean[35], double std[35]) {
  real m[35], s[35];
  obs_scaling(cfg, m, s);
  for (int i = 0; i < 35; i++) { mean[i] = R_TO_DOUBLE(m[i]); std[i] = R_TO_DOUBLE(s[i]); }
}
/* PPO:554-568 */
void orc_gae(int T, int N, const float *rewards, const float *values, const uint8_t *dones, const float *last_values,
             const uint8_t *last_dones, float gamma, float lam, float *adv, float *returns) {
  for (int n = 0; n < N; n++) {
    float last = 0.f;
    for (int t = T - 1; t >= 0; t--) {
      float nonterm, nextv;
      if (t == T - 1) { nonterm = 1.0f - (float)last_dones[n]; nextv = last_values[n]; }
      else { nonterm = 1.0f - (float)dones[(size_t)(t + 1) * N + n]; nextv = values[(size_t)(t + 1) * N + n]; }
      float delta = rewards[(size_t)t * N + n] + gamma * nextv * nonterm - values[(size_t)t * N + n];
      last = delta + gamma * lam * nonterm * last;
      adv[(size_t)t * N + n] = last;
      returns[(size_t)t * N + n] = last + values[(size_t)t * N + n];
    }
  }
}
void orc_mass_matrix_world(const double gc[19], double M[324]) {
  robot_model m; real g[NQ], Mw[NV * NV];
  model_nominal(&m);
  for (int i = 0; i < NQ; i++) g[i] = RC(gc[i]);
  world_mass_matrix(&m, g, Mw, NULL);
  for (int i = 0; i < NV * NV; i++) M[i] = R_TO_DOUBLE(Mw[i]);
}
void orc_nonlinear_world(const double gc[19], const double gv[18], double hout[18]) {
  robot_model m; real g[NQ], v[NV], hw[NV];
  model_nominal(&m);
  for (int i = 0; i < NQ; i++) g[i] = RC(gc[i]);
  for (int i = 0; i < NV; i++) v[i] = RC(gv[i]);
  world_nonlinear(&m, g, v, hw);
  for (int i = 0; i < NV; i++) hout[i] = R_TO_DOUBLE(hw[i]);
}
void orc_toe_kinematics(const double gc[19], const double gv[18], double pos[12], double vel[12]) {
  env_t e; kin_t k;
  memset(&e, 0, sizeof(e));
  model_nominal(&e.model);
  for (int i = 0; i < NQ; i++) e.gc[i] = RC(gc[i]);
  for (int i = 0; i < NV; i++) e.gv[i] = RC(gv[i]);
  quat_to_rot(&e.gc[3], e.Rwb);
  forward_kinematics(&e.model, &e.gc[7], &k);
  for (int l = 0; l < 4; l++) {
    real p[3], v[3];
    toe_world(&e, &k, l, p, v, NULL);
    for (int a = 0; a < 3; a++) { pos[3 * l + a] = R_TO_DOUBLE(p[a]); vel[3 * l + a] = R_TO_DOUBLE(v[a]); }
  }
}
void orc_rng_u01(uint32_t seed, uint32_t env, uint32_t episode, uint32_t step, uint32_t purpose, double out[4]) {
  rng_addr a = {seed, env, episode, step}; real u[4];
  rng_u01x4(&a, purpose, u);
  for (int i = 0; i < 4; i++) out[i] = R_TO_DOUBLE(u[i]);
}
double orc_last_step_flops(const orc_env *h) { return h->flops; }
int orc_heightfield(const orc_env *h, float *out, int *nx, int *ny) {
  *nx = HF_NX; *ny = HF_NY;
  if (!h->height) return 0;
  if (out) memcpy(out, h->height, sizeof(float) * (size_t)HF_NX * HF_NY);
  return 1;
}
void orc_terrain_sample(const orc_env *h, double x, double y, double out[4]) {
  real hh, n[3];
  terrain_sample(h->height, RC(x), RC(y), &hh, n);
  out[0] = R_TO_DOUBLE(hh); out[1] = R_TO_DOUBLE(n[0]); out[2] = R_TO_DOUBLE(n[1]); out[3] = R_TO_DOUBLE(n[2]);
}
double orc_mean_contact_sweeps(const orc_env *h) {
  double a = 0, b = 0;
  for (int i = 0; i < h->n; i++) { a += (double)h->envs[i].gs_sweeps; b += (double)h->envs[i].gs_substeps; }
  return b > 0 ? a / b : 0.0;
}
