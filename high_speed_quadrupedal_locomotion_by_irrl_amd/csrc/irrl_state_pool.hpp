// irrl_state_pool.hpp -- host-side description of the device state pool: one allocation, 256-byte
// aligned sub-arrays, and the flat per-env "checkpoint" layout exchanged through
// irrl_env_get_state / irrl_env_set_state (include/irrl_env.h, IRRL_S_* offsets).
//
// The reference keeps this state scattered over N heap-allocated ENVIRONMENT objects (members at
// Environment.hpp:1904-2086) plus one raisim::World each; here it is a structure of arrays sized for
// HBM residency: 286 words (1.1 KB) per robot, 36 MB for 32 768 robots.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "env_params.h"

namespace irrl_host {

enum { IRRL_STATE_DIM_ = 288 };
// flat layout offsets (doubles per env) -- keep in sync with include/irrl_env.h
enum {
  FS_GC = 0, FS_GV = 19, FS_PTL = 37, FS_TQL = 49, FS_TQ = 61, FS_JR = 73, FS_JRL = 85, FS_JDR = 97, FS_EER = 109,
  FS_CMD = 121, FS_CMDF = 124, FS_T0 = 127, FS_FRAME = 128, FS_EPISODE = 129, FS_UPH = 130, FS_CONTACT = 131,
  FS_LAMW = 135, FS_INCONTACT = 147, FS_MATERIAL = 151, FS_MASS = 154, FS_COM = 167, FS_THIGH = 206, FS_OB = 207,
  FS_OBLAST = 242, FS_SPHERE = 277, FS_END = 286
};

struct StatePool {
  int n = 0;
  size_t bytes = 0;
  size_t off[25] = {0};

  static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

  explicit StatePool(int n_envs = 0) : n(n_envs) {
    const int words[25] = {19, 18, 12, 12, 12, 12, 12, 12, 12, 12, 4, 4, 3, 3, 1, 1, 1, 1, 3, 13, 39, 1, 35, 35, 9};
    size_t o = 0;
    for (int i = 0; i < 25; i++) { off[i] = o; o = align256(o + (size_t)words[i] * 4u * (size_t)(n > 0 ? n : 0)); }
    bytes = o;
  }
  EnvState view(void *base) const {
    char *b = (char *)base;
    EnvState S;
    S.gc = (float *)(b + off[0]); S.gv = (float *)(b + off[1]); S.ptarget_last = (float *)(b + off[2]);
    S.torque_last = (float *)(b + off[3]); S.torque = (float *)(b + off[4]); S.joint_ref = (float *)(b + off[5]);
    S.joint_ref_last = (float *)(b + off[6]); S.joint_dot_ref = (float *)(b + off[7]); S.ee_ref = (float *)(b + off[8]);
    S.lam_w = (float *)(b + off[9]); S.in_contact = (int32_t *)(b + off[10]); S.contact = (float *)(b + off[11]);
    S.command = (float *)(b + off[12]); S.command_filtered = (float *)(b + off[13]); S.t0 = (float *)(b + off[14]);
    S.frame_idx = (int32_t *)(b + off[15]); S.episode = (uint32_t *)(b + off[16]); S.up_height = (float *)(b + off[17]);
    S.material = (float *)(b + off[18]); S.mass = (float *)(b + off[19]); S.com = (float *)(b + off[20]);
    S.thigh_dz = (float *)(b + off[21]); S.ob = (float *)(b + off[22]); S.ob_last = (float *)(b + off[23]);
    S.sphere = (float *)(b + off[24]);
    S.contact_count = nullptr;   // diagnostic, allocated separately by the C-ABI
    return S;
  }
  // host mirror -> flat [n, 288] doubles
  void pack(const void *host_base, double *out) const {
    EnvState S = view(const_cast<void *>(host_base));
    for (int e = 0; e < n; e++) {
      double *o = out + (size_t)e * IRRL_STATE_DIM_;
      for (int k = 0; k < IRRL_STATE_DIM_; k++) o[k] = 0.0;
      for (int k = 0; k < 19; k++) o[FS_GC + k] = S.gc[e * 19 + k];
      for (int k = 0; k < 18; k++) o[FS_GV + k] = S.gv[e * 18 + k];
      for (int k = 0; k < 12; k++) {
        o[FS_PTL + k] = S.ptarget_last[e * 12 + k]; o[FS_TQL + k] = S.torque_last[e * 12 + k]; o[FS_TQ + k] = S.torque[e * 12 + k];
        o[FS_JR + k] = S.joint_ref[e * 12 + k]; o[FS_JRL + k] = S.joint_ref_last[e * 12 + k]; o[FS_JDR + k] = S.joint_dot_ref[e * 12 + k];
        o[FS_EER + k] = S.ee_ref[e * 12 + k]; o[FS_LAMW + k] = S.lam_w[e * 12 + k];
      }
      for (int k = 0; k < 3; k++) { o[FS_CMD + k] = S.command[e * 3 + k]; o[FS_CMDF + k] = S.command_filtered[e * 3 + k]; o[FS_MATERIAL + k] = S.material[e * 3 + k]; }
      o[FS_T0] = S.t0[e]; o[FS_FRAME] = S.frame_idx[e]; o[FS_EPISODE] = S.episode[e]; o[FS_UPH] = S.up_height[e];
      for (int k = 0; k < 4; k++) { o[FS_CONTACT + k] = S.contact[e * 4 + k]; o[FS_INCONTACT + k] = S.in_contact[e * 4 + k]; }
      for (int k = 0; k < 13; k++) o[FS_MASS + k] = S.mass[e * 13 + k];
      for (int k = 0; k < 39; k++) o[FS_COM + k] = S.com[e * 39 + k];
      o[FS_THIGH] = S.thigh_dz[e];
      for (int k = 0; k < 35; k++) { o[FS_OB + k] = S.ob[e * 35 + k]; o[FS_OBLAST + k] = S.ob_last[e * 35 + k]; }
      for (int k = 0; k < 9; k++) o[FS_SPHERE + k] = S.sphere[e * 9 + k];
    }
  }
  void unpack(const double *in, void *host_base) const {
    EnvState S = view(host_base);
    for (int e = 0; e < n; e++) {
      const double *o = in + (size_t)e * IRRL_STATE_DIM_;
      for (int k = 0; k < 19; k++) S.gc[e * 19 + k] = (float)o[FS_GC + k];
      for (int k = 0; k < 18; k++) S.gv[e * 18 + k] = (float)o[FS_GV + k];
      for (int k = 0; k < 12; k++) {
        S.ptarget_last[e * 12 + k] = (float)o[FS_PTL + k]; S.torque_last[e * 12 + k] = (float)o[FS_TQL + k]; S.torque[e * 12 + k] = (float)o[FS_TQ + k];
        S.joint_ref[e * 12 + k] = (float)o[FS_JR + k]; S.joint_ref_last[e * 12 + k] = (float)o[FS_JRL + k]; S.joint_dot_ref[e * 12 + k] = (float)o[FS_JDR + k];
        S.ee_ref[e * 12 + k] = (float)o[FS_EER + k]; S.lam_w[e * 12 + k] = (float)o[FS_LAMW + k];
      }
      for (int k = 0; k < 3; k++) { S.command[e * 3 + k] = (float)o[FS_CMD + k]; S.command_filtered[e * 3 + k] = (float)o[FS_CMDF + k]; S.material[e * 3 + k] = (float)o[FS_MATERIAL + k]; }
      S.t0[e] = (float)o[FS_T0]; S.frame_idx[e] = (int32_t)o[FS_FRAME]; S.episode[e] = (uint32_t)o[FS_EPISODE]; S.up_height[e] = (float)o[FS_UPH];
      for (int k = 0; k < 4; k++) { S.contact[e * 4 + k] = (float)o[FS_CONTACT + k]; S.in_contact[e * 4 + k] = (int32_t)o[FS_INCONTACT + k]; }
      for (int k = 0; k < 13; k++) S.mass[e * 13 + k] = (float)o[FS_MASS + k];
      for (int k = 0; k < 39; k++) S.com[e * 39 + k] = (float)o[FS_COM + k];
      S.thigh_dz[e] = (float)o[FS_THIGH];
      for (int k = 0; k < 35; k++) { S.ob[e * 35 + k] = (float)o[FS_OB + k]; S.ob_last[e * 35 + k] = (float)o[FS_OBLAST + k]; }
      for (int k = 0; k < 9; k++) S.sphere[e * 9 + k] = (float)o[FS_SPHERE + k];
    }
  }
};

}  // namespace irrl_host
