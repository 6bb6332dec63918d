// emu_main.cpp -- TEST-ONLY host emulation of the env kernels.
//
// Compiles the product's kernel source (csrc/env_core.hpp) against tests/host_emulation/lanes_cpu.hpp,
// where a "quad" is four array slots stepped in lockstep.  Purpose: let the CPU test-suite (no GPU in
// the build container) exercise the exact lane algorithm -- Schur-complement dynamics, DPP-style quad
// reductions, masked reset -- against the f64 oracle before the kernels ever reach an MI355X.
// It is NOT a product path: nothing in the package loads this library, and the C-ABI has no CPU mode.
#include "lanes_cpu.hpp"

#include "env_core.hpp"
#include "irrl_config.hpp"
#include "irrl_state_pool.hpp"
#include "irrl_terrain.hpp"
#include "irrl_csv.hpp"

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

struct Emu {
  EnvParams P;
  irrl_host::StatePool pool;
  std::vector<char> mem;
  std::vector<float> height;
  std::vector<float> ref;
  std::vector<uint32_t> counters;     // EnvState::contact_count (the product allocates it beside the pool: irrl_env_abi.hip)
  EnvState S;
};
static std::string g_err;

static inline vi lanes_env(int e) { return vi(e); }
// the lanes that own the stores of a robot: all of them with one lane per leg, sub-lane 0 of every leg with four
static inline vm store_mask() {
#if IRRL_EMU_W == 16
  return lanes::sub_id() == 0;
#else
  return vm(true);
#endif
}

// `count` steps with the lane context CARRIED IN REGISTERS from step to step -- the step loop of the product's multi-step kernels
// (env_kernels.hip: irrl_steps_persistent_kernel and the persistent rollout kernels): load once, { step_compute; lane_carry } x count, store
// once.  Outputs [count, N, .].  poison != 0 (16 lanes per robot): behind every step the context of sub-lanes 1-3 of every quad is overwritten
// with garbage first -- on the GPU those lanes did not run the epilogue, so whatever lane_carry does not re-broadcast from sub-lane 0 is stale.
template <int RULE>
static void steps_carried(Emu *h, int count, const float *action_rows, float *ob, float *reward, uint8_t *done, float *extra, int poison) {
  const size_t n = (size_t)h->P.n_envs;
  for (int e = 0; e < h->P.n_envs; e++) {
    const vi env = lanes_env(e), leg = lanes::leg_id();
    const vm mask = store_mask();
    irrl::EnvLane L;
    irrl::load_lane(h->P, h->S, env, leg, L, true);
    for (int k = 0; k < count; k++) {
      if (k > 0) {
        if (getenv("IRRL_EMU_CARRY_DEBUG")) {
          irrl::store_lane(h->P, h->S, env, leg, mask, L, true);
          irrl::EnvLane L2;
          irrl::load_lane(h->P, h->S, env, leg, L2, true);
          irrl::EnvLane L1 = L;
          irrl::lane_carry(L1);
          const uint32_t *a = (const uint32_t *)&L1, *b = (const uint32_t *)&L2;
          for (size_t i = 0; i < sizeof(irrl::EnvLane) / 4; i++)
            if (a[i] != b[i]) { fprintf(stderr, "env %d step %d: word %zu (vector %zu, lane %zu) carried %08x loaded %08x\n", e, k, i, i / W, i % W, a[i], b[i]); break; }
        }
        irrl::lane_carry(L);      // (between steps only: the final store wants the last step's own torque / contact / observation words)
      }
      irrl::step_compute<RULE>(h->P, L, env, leg, mask, irrl::ActionRow{action_rows + (size_t)k * n * 12}, ob + (size_t)k * n * 35, reward + (size_t)k * n,
                               done + (size_t)k * n, extra + (size_t)k * n * 6);
#if IRRL_EMU_W == 16
      if (poison) {
        static_assert(sizeof(irrl::EnvLane) % (4 * W) == 0, "EnvLane is made of 32-bit lane vectors");
        uint32_t *w = (uint32_t *)&L;
        for (size_t i = 0; i < sizeof(irrl::EnvLane) / 4; i++)
          if ((i % W) & 3) w[i] = 0x7fc00000u ^ (uint32_t)(i * 2654435761u);
        irrl::model_signs(L.m, leg);      // (per-lane constants, not part of the carried state)
      }
#endif
    }
    irrl::store_lane(h->P, h->S, env, leg, mask, L, true);
  }
}
extern "C" {
const char *emu_last_error() { return g_err.c_str(); }
void *emu_create(const char *cfg_yaml) {
  g_err.clear();
  irrl_host::Config c;
  Emu *h = new Emu();
  if (!c.parse(cfg_yaml, g_err) || !irrl_host::build_params(c, h->P, g_err)) { delete h; return nullptr; }
  h->pool = irrl_host::StatePool(h->P.n_envs);
  h->mem.assign(h->pool.bytes, 0);
  h->S = h->pool.view(h->mem.data());
  h->counters.assign((size_t)h->P.n_envs * 4, 0u);
  h->S.contact_count = h->counters.data();
  if (h->P.terrain) {
    irrl_host::generate_heightfield(irrl_host::TerrainSpec(), h->P.seed, h->height);
    h->P.height = h->height.data();
    float hmax = 0.0f;
    for (float v : h->height) hmax = v > hmax ? v : hmax;
    h->P.hf_max = hmax;
  }
  return h;
}
void emu_destroy(void *hv) { delete (Emu *)hv; }
// the product's host CSV reader (csrc/irrl_csv.hpp), reachable without a GPU: out == NULL queries the shape
int emu_read_csv(const char *path, float *out, int *rows, int *cols) {
  std::vector<float> data;
  std::string e;
  if (!irrl_host::read_csv_f32(path, data, *rows, *cols, e)) { g_err = e; return 1; }
  if (out) std::memcpy(out, data.data(), data.size() * sizeof(float));
  return 0;
}
int emu_set_ref(void *hv, const float *table, int rows, int cols) {
  Emu *h = (Emu *)hv;
  if (!h->P.ref_traj || rows < 2 || cols < 30) return 1;
  h->ref.resize((size_t)rows * 30);
  for (int r = 0; r < rows; r++) std::memcpy(&h->ref[(size_t)r * 30], table + (size_t)r * cols, 30 * sizeof(float));
  h->P.ref = h->ref.data();
  h->P.ref_rows = rows;
  return 0;
}
int emu_num_envs(void *hv) { return ((Emu *)hv)->P.n_envs; }
void emu_init(void *hv) {
  Emu *h = (Emu *)hv;
  for (int e = 0; e < h->P.n_envs; e++) irrl::init_body(h->P, h->S, lanes_env(e), lanes::leg_id(), store_mask());
}
void emu_reset(void *hv, float *ob) {
  Emu *h = (Emu *)hv;
  for (int e = 0; e < h->P.n_envs; e++) irrl::reset_body(h->P, h->S, lanes_env(e), lanes::leg_id(), store_mask(), ob);
}
void emu_observe(void *hv, float *ob) {
  Emu *h = (Emu *)hv;
  for (int e = 0; e < h->P.n_envs; e++) irrl::observe_body(h->P, h->S, lanes_env(e), lanes::leg_id(), store_mask(), ob);
}
void emu_step(void *hv, const float *action, float *ob, float *reward, uint8_t *done, float *extra) {
  Emu *h = (Emu *)hv;
  for (int e = 0; e < h->P.n_envs; e++)
    if (h->P.contact_rule) irrl::step_body<1>(h->P, h->S, lanes_env(e), lanes::leg_id(), store_mask(), action, ob, reward, done, extra);
    else irrl::step_body<0>(h->P, h->S, lanes_env(e), lanes::leg_id(), store_mask(), action, ob, reward, done, extra);
}
void emu_steps_carried(void *hv, int count, const float *action_rows, float *ob, float *reward, uint8_t *done, float *extra, int poison) {
  Emu *h = (Emu *)hv;
  if (h->P.contact_rule) steps_carried<1>(h, count, action_rows, ob, reward, done, extra, poison);
  else steps_carried<0>(h, count, action_rows, ob, reward, done, extra, poison);
}
void emu_probe(void *hv, float *minv, float *nonlin) {
  Emu *h = (Emu *)hv;
  for (int e = 0; e < h->P.n_envs; e++)
    irrl::dynamics_probe_body(h->P, h->S, lanes_env(e), lanes::leg_id(), store_mask(), minv, nonlin);
}
void emu_get_state(void *hv, double *out) { Emu *h = (Emu *)hv; h->pool.pack(h->mem.data(), out); }
void emu_set_state(void *hv, const double *in) { Emu *h = (Emu *)hv; h->pool.unpack(in, h->mem.data()); }
void emu_get_params(void *hv, EnvParams *out) { *out = ((Emu *)hv)->P; }
int emu_heightfield(void *hv, float *out) {
  Emu *h = (Emu *)hv;
  if (h->height.empty()) return 0;
  if (out) std::memcpy(out, h->height.data(), h->height.size() * sizeof(float));
  return 1;
}
}
