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

#include <string>
#include <vector>

struct Emu {
  EnvParams P;
  irrl_host::StatePool pool;
  std::vector<char> mem;
  std::vector<float> height;
  std::vector<float> ref;
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
