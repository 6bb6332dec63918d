// irrl_env_abi.hip -- C-ABI (include/irrl_env.h) over the gfx950 env kernels.
//
// Host side of the drop-in boundary: owns the device state pool (the role of
// VectorizedEnvironment<ENVIRONMENT>, VectorizedEnvironment.hpp:127-382), parses the YAML string the
// reference hands over, and launches the kernels stream-ordered.  No CPU compute path exists here:
// without a usable gfx950 device irrl_env_create fails.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "env_params.h"
#include "lstm_kernels.hip"
#include "mlp_update.hpp"
#include "mlp_bf16.hpp"
#include "mlp_bf16_pc.hpp"
#include "ppo_optim.hpp"

// The env kernels are compiled in two lane layouts from the same source (csrc/env_kernels.hip, see build.py):
//   _l16  16 lanes per robot (lanes_hip16.hpp): 4 robots per wave -- fills all 1024 SIMDs at 4096 robots, shortest step
//   _l4    4 lanes per robot (lanes_hip.hpp):  16 robots per wave -- 2.3x fewer instructions per robot, the better
//          throughput once the pool is large enough to occupy the chip on its own
//   _l4w2  the same, compiled for two waves per SIMD (256 registers each): pools with more 4-lane waves than SIMDs (> 16 384 robots)
#define IRRL_DECLARE_KERNELS(sfx)                                                                                            \
  extern "C" __global__ void irrl_step_kernel##sfx(EnvParams, EnvState, const float *, float *, float *, uint8_t *, float *);   \
  extern "C" __global__ void irrl_step_kernel_dir##sfx(EnvParams, EnvState, const float *, float *, float *, uint8_t *, float *); \
  extern "C" __global__ void irrl_step_kernel_md##sfx(EnvParams, EnvState, const float *, float *, float *, uint8_t *, float *); \
  extern "C" __global__ void irrl_step_kernel_crutial##sfx(EnvParams, EnvState, const float *, float *, float *, uint8_t *, float *); \
  extern "C" __global__ void irrl_step_kernel_crutial_md##sfx(EnvParams, EnvState, const float *, float *, float *, uint8_t *, float *); \
  extern "C" __global__ void irrl_steps_persistent_kernel##sfx(EnvParams, EnvState, const float *, int, int, int, float *, float *, uint8_t *, float *, int); \
  extern "C" __global__ void irrl_steps_persistent_kernel_flat##sfx(EnvParams, EnvState, const float *, int, int, int, float *, float *, uint8_t *, float *, int); \
  extern "C" __global__ void irrl_step_kernel_flat##sfx(EnvParams, EnvState, const float *, float *, float *, uint8_t *, float *);   \
  extern "C" __global__ void irrl_init_kernel##sfx(EnvParams, EnvState);                                                       \
  extern "C" __global__ void irrl_reset_kernel##sfx(EnvParams, EnvState, float *);                                             \
  extern "C" __global__ void irrl_observe_kernel##sfx(EnvParams, EnvState, float *);                                           \
  extern "C" __global__ void irrl_probe_kernel##sfx(EnvParams, EnvState, float *, float *);
IRRL_DECLARE_KERNELS(_l16)
IRRL_DECLARE_KERNELS(_l4)
IRRL_DECLARE_KERNELS(_l4w2)
extern "C" __global__ void irrl_terminal_kernel(EnvParams, EnvState, uint8_t *);
extern "C" __global__ void irrl_step_policy_kernel_l16(EnvParams, EnvState, const float *, float *, float *, uint8_t *, float *, PolicyStepArgs);
extern "C" __global__ void irrl_rollout_persistent_kernel_l16(EnvParams, EnvState, float *, float *, uint8_t *, float *, PolicyStepArgs, int);
extern "C" __global__ void irrl_rollout_persistent_actor_kernel_l16(EnvParams, EnvState, float *, float *, uint8_t *, float *, PolicyStepArgs, int);
extern "C" __global__ void irrl_rollout_persistent_actor_wave_kernel_l16(EnvParams, EnvState, float *, float *, uint8_t *, float *, PolicyStepArgs, int);
extern "C" __global__ void irrl_rollout_persistent_mlp_kernel_l16(EnvParams, EnvState, float *, float *, uint8_t *, float *, PolicyStepArgs, int);

#include "irrl_config.hpp"
#include "irrl_state_pool.hpp"
#include "irrl_terrain.hpp"
#include "irrl_csv.hpp"
#include "../../include/irrl_env.h"

#include <cmath>
#include <string>
#include <vector>

static thread_local std::string g_err;

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) {                                                                             \
      g_err = std::string(#expr) + " failed: " + hipGetErrorString(e_);                                \
      return 1;                                                                                         \
    }                                                                                                   \
  } while (0)

struct irrl_env {
  EnvParams P;
  irrl_host::Config cfg;
  irrl_host::StatePool pool;
  int device = 0;
  hipStream_t stream = nullptr;
  void *d_pool = nullptr;
  EnvState S;
  // device-side I/O used by the *_host entry points
  float *d_action = nullptr, *d_ob = nullptr, *d_reward = nullptr, *d_extra = nullptr, *d_scratch = nullptr;
  uint8_t *d_done = nullptr;
  float *d_height = nullptr;  // shared height field (Terrain: True)
  float *d_ref = nullptr;     // reference-trajectory table [rows, 30] (ManualTraj: False)
  uint32_t *d_counters = nullptr;  // [N,4] toe-substeps in contact (EnvState::contact_count, diagnostic)
  void *d_snapshot = nullptr;      // shadow copy of the pool (irrl_env_snapshot / irrl_env_restore), allocated on first use
  std::vector<float> h_height;
  // pinned host staging
  char *h_pinned = nullptr;
  size_t pinned_bytes = 0;
  bool initialised = false;
  int lanes = 16;  // lanes per robot of the kernels this pool launches (16 or 4)
  int waves2 = 0;  // 4-lane layout only: launch the kernels compiled for two waves per SIMD (_l4w2)
  std::string resource_dir;
};

static const char *const kExtraNames[IRRL_EXTRA_DIM] = {"EndEffectorReward(0.15)", "Height_Keep_Reward(0.1)", "base height",
                                                        "Balance_Keep_Reward(0.1)", "JointReward(0.65)", "VelocityReward(0.2)"};

// launch shape of the lane kernels: `waves_per_block()` waves per workgroup (1 by default; IRRL_WAVES_PER_BLOCK=4 packs
// the four SIMDs of a CU into one workgroup), IRRL_ROBOTS_PER_WAVE robots per wave
static int waves_per_block() {
  static int w = [] { const char *e = getenv("IRRL_WAVES_PER_BLOCK"); int v = e ? atoi(e) : 1; return (v == 1 || v == 2 || v == 4) ? v : 1; }();
  return w;
}
static inline dim3 quad_block() { return dim3((unsigned)(64 * waves_per_block())); }
static inline dim3 lane_grid(const irrl_env *h, int n) {
  const int per_block = (64 / h->lanes) * waves_per_block();
  return dim3((unsigned)((n + per_block - 1) / per_block));
}
// layout choice: 16 lanes per robot (4 robots per wave) while the pool's waves fit the device's SIMDs -- one resident wave each: the kernels take
// 335-363 registers --, 4 lanes per robot (16 robots per wave) beyond: 4096 robots on an MI355X.  (Rounds 2-5 drew the line at 6144 robots; measured in
// round 6, 4097-6144 robots ran a second round of 16-lane waves: 59.3 us per step against 41.2 us in the 4-lane layout,
// profiles/r06_lane_layout_threshold.log.)  IRRL_LANES_PER_ROBOT=4|16 overrides (used by the tests to cover both layouts on small pools)
static int device_simds(int device) {
  int cus = 256;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
  return cus * 4;
}
static int pick_lanes(int n_envs, int device) {
  const char *e = getenv("IRRL_LANES_PER_ROBOT");
  if (e) { int v = atoi(e); if (v == 4 || v == 16) return v; }
  return ((n_envs + 3) / 4 <= device_simds(device)) ? 16 : 4;
}
// 4-lane pools: two waves per SIMD once there are more waves than SIMDs (16 robots per wave: 16 384 robots on an MI355X); IRRL_L4_WAVES=1|2
// overrides (used by the tests to run the _l4w2 kernels on small pools)
static int pick_waves2(int n_envs, int device) {
  const char *e = getenv("IRRL_L4_WAVES");
  if (e) { int v = atoi(e); if (v == 1 || v == 2) return v == 2; }
  return (n_envs + 15) / 16 > device_simds(device);
}
#define IRRL_LAUNCH(h, name, grid, ...)                                                                                   \
  do {                                                                                                                    \
    if ((h)->lanes == 16) hipLaunchKernelGGL(name##_l16, grid, quad_block(), 0, (h)->stream, __VA_ARGS__);                 \
    else if ((h)->waves2) hipLaunchKernelGGL(name##_l4w2, grid, quad_block(), 0, (h)->stream, __VA_ARGS__);                \
    else hipLaunchKernelGGL(name##_l4, grid, quad_block(), 0, (h)->stream, __VA_ARGS__);                                   \
  } while (0)

// the step kernel: one instantiation per (Crutial, per-contact rule) -- env_kernels.hip -- and, for the default pool kind (no meteorite, published
// rule), one with the shipped solver settings compiled in (simultaneous sweeps + predicted exit + a tolerance above zero + at most six sweeps + eight substeps per control step: env_core.hpp IRRL_SOLVER_FIXED) next to the
// one that reads them from EnvParams; the multi-step and rollout kernels exist for the former only (the launchers fall back)
// flat ground compiled in (env_core.hpp IRRL_FLAT_GROUND): the step kernel and the multi-step kernel exist in that form too.  (Not in the
// instrumented build of tools/wave_spread.py, whose per-wave clock sits in irrl_step_kernel.)
static inline bool flat_kernels(const irrl_env *h) {
#ifdef IRRL_PROFILE_WAVES
  (void)h;
  return false;
#else
  return !h->P.terrain;
#endif
}
static inline bool shipped_solver(const irrl_env *h) {
  return h->P.contact_jacobi != 0 && h->P.contact_exit != 0 && h->P.contact_tol > 0.0f && h->P.contact_iters == IRRL_SHIPPED_SWEEP_CAP &&
         h->P.loop_count == IRRL_SHIPPED_SUBSTEPS;
}
#define IRRL_LAUNCH_STEP(h, grid, ...)                                                                                      \
  do {                                                                                                                    \
    if ((h)->P.crutial) {                                                                                                   \
      if ((h)->P.contact_rule) IRRL_LAUNCH(h, irrl_step_kernel_crutial_md, grid, __VA_ARGS__);                               \
      else IRRL_LAUNCH(h, irrl_step_kernel_crutial, grid, __VA_ARGS__);                                                      \
    } else if ((h)->P.contact_rule) {                                                                                       \
      if (shipped_solver(h) && flat_kernels(h)) IRRL_LAUNCH(h, irrl_step_kernel_flat, grid, __VA_ARGS__);                   \
      else if (shipped_solver(h)) IRRL_LAUNCH(h, irrl_step_kernel, grid, __VA_ARGS__);                                      \
      else IRRL_LAUNCH(h, irrl_step_kernel_md, grid, __VA_ARGS__);                                                          \
    }                                                                                                                       \
    else IRRL_LAUNCH(h, irrl_step_kernel_dir, grid, __VA_ARGS__);                                                            \
  } while (0)

extern "C" {

const char *irrl_last_error(void) { return g_err.c_str(); }
#ifndef IRRL_SRC_HASH
#define IRRL_SRC_HASH "unknown"
#endif
// the build recipe (build.py) bakes a content hash of csrc/ + include/ + the compiler flags into the library, so a
// prebuilt .so that does not match the sources next to it is detected without relying on file times
const char *irrl_version(void) { return "gfx950;irrl-env r3;irrl-src-hash:" IRRL_SRC_HASH; }

irrl_env *irrl_env_create(const char *resource_dir, const char *cfg_yaml, int device) {
  g_err.clear();
  if (!cfg_yaml) { g_err = "cfg_yaml is NULL"; return nullptr; }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    g_err = "no HIP device visible: this library is gfx950 (MI355X) only and has no CPU path";
    return nullptr;
  }
  if (device < 0 || device >= count) { g_err = "device ordinal out of range"; return nullptr; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { g_err = "hipGetDeviceProperties failed"; return nullptr; }
  if (std::string(prop.gcnArchName).find("gfx950") != 0) {
    g_err = std::string("device arch is ") + prop.gcnArchName + ", this library is built for gfx950 only";
    return nullptr;
  }
  irrl_env *h = new irrl_env();
  h->device = device;
  h->resource_dir = resource_dir ? resource_dir : "";
  if (!h->cfg.parse(cfg_yaml, g_err) || !irrl_host::build_params(h->cfg, h->P, g_err)) { delete h; return nullptr; }
  h->pool = irrl_host::StatePool(h->P.n_envs);
  h->lanes = pick_lanes(h->P.n_envs, device);
  h->waves2 = (h->lanes == 4) ? pick_waves2(h->P.n_envs, device) : 0;
  const size_t n = (size_t)h->P.n_envs;
  bool ok = hipSetDevice(device) == hipSuccess && hipMalloc(&h->d_pool, h->pool.bytes) == hipSuccess &&
            hipMemset(h->d_pool, 0, h->pool.bytes) == hipSuccess && hipMalloc((void **)&h->d_action, n * 12 * 4) == hipSuccess &&
            hipMalloc((void **)&h->d_ob, n * (35 * 4 + 4 + 6 * 4 + 1)) == hipSuccess &&   // ob | reward | extra | done: ONE D2H copy per host step
            hipMalloc((void **)&h->d_scratch, n * 342 * 4) == hipSuccess &&
            hipMalloc((void **)&h->d_counters, n * 4 * 4) == hipSuccess && hipMemset(h->d_counters, 0, n * 4 * 4) == hipSuccess;
  h->pinned_bytes = n * 342 * 4 + h->pool.bytes + 4096;
  ok = ok && hipHostMalloc((void **)&h->h_pinned, h->pinned_bytes, hipHostMallocDefault) == hipSuccess;
  if (ok && h->P.terrain) {
    irrl_host::generate_heightfield(irrl_host::TerrainSpec(), h->P.seed, h->h_height);
    const size_t hb = h->h_height.size() * sizeof(float);
    ok = hipMalloc((void **)&h->d_height, hb) == hipSuccess && hipMemcpy(h->d_height, h->h_height.data(), hb, hipMemcpyHostToDevice) == hipSuccess;
    h->P.height = h->d_height;
    float hmax = 0.0f;
    for (float v : h->h_height) hmax = v > hmax ? v : hmax;
    h->P.hf_max = hmax;
  }
  if (ok) {
    h->d_reward = h->d_ob + n * 35;
    h->d_extra = h->d_reward + n;
    h->d_done = (uint8_t *)(h->d_extra + n * 6);
  }
  if (!ok) { g_err = "device / pinned allocation failed"; irrl_env_destroy(h); return nullptr; }
  h->S = h->pool.view(h->d_pool);
  h->S.contact_count = h->d_counters;
  if (h->P.ref_traj) {
    // VEC:158-169: the table named by cfg["RefTraj"]; a missing file is only a console message there (and a crash at the
    // first step) -- here the pool is created, and init() refuses to run until irrl_env_set_ref_host supplied a table
    auto it = h->cfg.kv.find("RefTraj");
    std::vector<float> tab;
    int rows = 0, cols = 0;
    std::string e;
    if (it != h->cfg.kv.end() && irrl_host::read_csv_f32(it->second, tab, rows, cols, e)) {
      if (irrl_env_set_ref_host(h, tab.data(), rows, cols) != 0) { irrl_env_destroy(h); return nullptr; }
    }
  }
  return h;
}

// ENV:1895 set_ref (VEC:173-176): reference-trajectory table, [rows, cols >= 30] row-major f32; only the first 30 columns
// are kept (theta 12 | theta_dot 12 | z | phase 2 | cmd 3)
int irrl_env_set_ref_host(irrl_env *h, const float *table, int rows, int cols) {
  if (!h->P.ref_traj) { g_err = "this pool generates its own reference (ManualTraj / Manual): no table is used"; return 1; }
  if (!table || rows < 2 || cols < 30) { g_err = "reference table needs at least 2 rows and 30 columns (Environment.hpp:17-21)"; return 1; }
  std::vector<float> packed((size_t)rows * 30);
  for (int r = 0; r < rows; r++) std::memcpy(&packed[(size_t)r * 30], table + (size_t)r * cols, 30 * sizeof(float));
  HIP_TRY(hipSetDevice(h->device));
  if (h->d_ref) { (void)hipFree(h->d_ref); h->d_ref = nullptr; }
  HIP_TRY(hipMalloc((void **)&h->d_ref, packed.size() * sizeof(float)));
  HIP_TRY(hipMemcpy(h->d_ref, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
  h->P.ref = h->d_ref;
  h->P.ref_rows = rows;
  return 0;
}

void irrl_env_destroy(irrl_env *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->d_pool) (void)hipFree(h->d_pool);
  if (h->d_action) (void)hipFree(h->d_action);
  if (h->d_ob) (void)hipFree(h->d_ob);   // reward / extra / done live in the same allocation
  if (h->d_scratch) (void)hipFree(h->d_scratch);
  if (h->d_height) (void)hipFree(h->d_height);
  if (h->d_ref) (void)hipFree(h->d_ref);
  if (h->d_counters) (void)hipFree(h->d_counters);
  if (h->d_snapshot) (void)hipFree(h->d_snapshot);
  if (h->h_pinned) (void)hipHostFree(h->h_pinned);
  delete h;
}

int irrl_env_set_stream(irrl_env *h, void *hip_stream) { h->stream = (hipStream_t)hip_stream; return 0; }

int irrl_env_init(irrl_env *h) {
  if (h->P.ref_traj && !h->P.ref) {
    g_err = "ManualTraj: False needs the reference-trajectory table: RefTraj file not readable and irrl_env_set_ref_host not called";
    return 1;
  }
  HIP_TRY(hipSetDevice(h->device));
  IRRL_LAUNCH(h, irrl_init_kernel, lane_grid(h, h->P.n_envs), h->P, h->S);
  HIP_TRY(hipGetLastError());
  h->initialised = true;
  return 0;
}

int irrl_env_num_envs(const irrl_env *h) { return h->P.n_envs; }
int irrl_env_lanes_per_robot(const irrl_env *h) { return h->lanes; }
int irrl_env_waves_per_simd(const irrl_env *h) { return h->waves2 ? 2 : 1; }
int irrl_env_ob_dim(const irrl_env *) { return IRRL_OB_DIM; }
int irrl_env_action_dim(const irrl_env *) { return IRRL_ACTION_DIM; }
int irrl_env_extra_dim(const irrl_env *) { return IRRL_EXTRA_DIM; }
const char *irrl_env_extra_name(const irrl_env *, int j) { return (j >= 0 && j < IRRL_EXTRA_DIM) ? kExtraNames[j] : ""; }

// The device-pointer entry points launch on h->stream, which belongs to h->device: make that device current when the
// caller's is another one (a process driving pools on several GPUs).  hipGetDevice is a thread-local read, so the common
// single-device case costs no driver call.
static int use_device(irrl_env *h) {
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess && cur == h->device) return 0;
  HIP_TRY(hipSetDevice(h->device));
  return 0;
}
static int need_init(irrl_env *h) {
  if (!h->initialised) { g_err = "irrl_env_init has not been called"; return 1; }
  return 0;
}

int irrl_env_step(irrl_env *h, const float *action, float *ob, float *reward, uint8_t *done, float *extra) {
  if (need_init(h)) return 1;
  if (use_device(h)) return 1;
  IRRL_LAUNCH_STEP(h, lane_grid(h, h->P.n_envs), h->P, h->S, action, ob, reward, done, extra);
  HIP_TRY(hipGetLastError());
  return 0;
}

// `count` steps from an action table, one launch per step.  out_rows != 0: ob / reward / done / extra are [count, N, .] tables and step k
// fills row k (what `count` step() calls of the reference return, VEC:268-278 / RaisimGymVecEnv.py:26-52); 0: [N, .] arrays, overwritten
static int step_rows_impl(irrl_env *h, int count, const float *action_rows, int n_rows, int first_row, float *ob, float *reward,
                          uint8_t *done, float *extra, int out_rows, const char *who) {
  if (need_init(h)) return 1;
  if (count < 0 || n_rows <= 0 || first_row < 0) { g_err = std::string(who) + ": count >= 0, n_rows > 0, first_row >= 0"; return 1; }
  if (!action_rows || !ob || !reward || !done || !extra) { g_err = std::string(who) + ": NULL argument"; return 1; }
  if (use_device(h)) return 1;
  const size_t row = (size_t)h->P.n_envs * 12, orow = out_rows ? (size_t)h->P.n_envs : (size_t)0;
  for (int k = 0; k < count; k++) {
    const float *action = action_rows + row * (size_t)((first_row + k) % n_rows);
    IRRL_LAUNCH_STEP(h, lane_grid(h, h->P.n_envs), h->P, h->S, action, ob + orow * 35 * (size_t)k, reward + orow * (size_t)k, done + orow * (size_t)k,
                     extra + orow * 6 * (size_t)k);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}
// the same `count` steps as ONE launch (env_kernels.hip irrl_steps_persistent_kernel: a wave walks its robots through all of them, no
// grid-wide boundary between steps); pools the kernel is not instantiated for (meteorite, first contact rule) take the launch-per-step path
static int step_rows_persistent_impl(irrl_env *h, int count, const float *action_rows, int n_rows, int first_row, float *ob, float *reward,
                                     uint8_t *done, float *extra, int out_rows, const char *who) {
  if (need_init(h)) return 1;
  if (count < 0 || n_rows <= 0 || first_row < 0) { g_err = std::string(who) + ": count >= 0, n_rows > 0, first_row >= 0"; return 1; }
  if (!action_rows || !ob || !reward || !done || !extra) { g_err = std::string(who) + ": NULL argument"; return 1; }
  if (h->P.crutial || !h->P.contact_rule || !shipped_solver(h)) return step_rows_impl(h, count, action_rows, n_rows, first_row, ob, reward, done, extra, out_rows, who);
  if (use_device(h)) return 1;
  if (count > 0) {
    if (flat_kernels(h)) IRRL_LAUNCH(h, irrl_steps_persistent_kernel_flat, lane_grid(h, h->P.n_envs), h->P, h->S, action_rows, n_rows, first_row, count, ob, reward, done, extra, out_rows);
    else IRRL_LAUNCH(h, irrl_steps_persistent_kernel, lane_grid(h, h->P.n_envs), h->P, h->S, action_rows, n_rows, first_row, count, ob, reward, done, extra, out_rows);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int irrl_env_step_rows(irrl_env *h, int count, const float *action_rows, int n_rows, int first_row, float *ob, float *reward,
                       uint8_t *done, float *extra) {
  return step_rows_impl(h, count, action_rows, n_rows, first_row, ob, reward, done, extra, 0, "irrl_env_step_rows");
}
int irrl_env_step_rows_persistent(irrl_env *h, int count, const float *action_rows, int n_rows, int first_row, float *ob, float *reward,
                                  uint8_t *done, float *extra) {
  return step_rows_persistent_impl(h, count, action_rows, n_rows, first_row, ob, reward, done, extra, 0, "irrl_env_step_rows_persistent");
}
int irrl_env_step_rows_out(irrl_env *h, int count, const float *action_rows, int n_rows, int first_row, float *ob_rows, float *reward_rows,
                           uint8_t *done_rows, float *extra_rows) {
  return step_rows_impl(h, count, action_rows, n_rows, first_row, ob_rows, reward_rows, done_rows, extra_rows, 1, "irrl_env_step_rows_out");
}
int irrl_env_step_rows_persistent_out(irrl_env *h, int count, const float *action_rows, int n_rows, int first_row, float *ob_rows,
                                      float *reward_rows, uint8_t *done_rows, float *extra_rows) {
  return step_rows_persistent_impl(h, count, action_rows, n_rows, first_row, ob_rows, reward_rows, done_rows, extra_rows, 1, "irrl_env_step_rows_persistent_out");
}

// which `fuse` modes of irrl_lstm_rollout exist for THIS pool and network: 1 = the mode runs as described, 0 = it does not (fuse 1 / 2 then fall
// back to two launches per step inside the call; fuse 3 is refused -- its caller must evaluate the critic itself, so it has to know beforehand)
static bool rollout_one_tile(const irrl_env *h, int hid) { return h->lanes == 16 && !h->P.crutial && h->P.contact_rule && shipped_solver(h) && hid == 48; }
int irrl_lstm_rollout_supports(irrl_env *h, int hid, int fuse) {
  if (!h) { g_err = "irrl_lstm_rollout_supports: NULL handle"; return -1; }
  if (fuse == 0) return 1;
  if (fuse < 0 || fuse > 3) return 0;
  return rollout_one_tile(h, hid) ? 1 : 0;
}

int irrl_lstm_rollout(irrl_env *h, int steps, int hid, int ob_dim, int act_dim, float *obs, uint8_t *dones, const float *states_in,
                      float *states_out, const float *const *lstm_w, const float *pi_w, const float *pi_b, const float *vf_w,
                      const float *vf_b, const float *logstd, const float *noise, int rng_on, unsigned rng_seed, long long rng_step,
                      const long long *rng_base, int env_id_offset, float *action, float *clipped, float *value, float *neglogp, long long row, float *mb_obs,
                      float *mb_actions, float *mb_values, float *mb_neglogp, uint8_t *mb_dones, float *mb_rewards,
                      float *env_reward, float *env_extra, int fuse, void *hip_stream) {
  if (need_init(h)) return 1;
  if (steps < 0 || row < 0 || act_dim != 12 || ob_dim != 35) { g_err = "irrl_lstm_rollout: steps >= 0, row >= 0, ob 35, act 12"; return 1; }
  if (!(mb_obs && mb_actions && mb_values && mb_neglogp && mb_dones)) { g_err = "irrl_lstm_rollout: the rollout buffers are mandatory"; return 1; }
  HIP_TRY(hipSetDevice(h->device));
  h->stream = (hipStream_t)hip_stream;
  const int n = h->P.n_envs;
  auto noise_at = [&](int k) { return noise ? noise + (size_t)k * (size_t)n * (size_t)act_dim : nullptr; };
  // step k's policy part alone (k == 0, and every step on the two-launch path)
  auto policy = [&](int k) {
    return irrl_lstm_policy_step(hid, ob_dim, act_dim, n, obs, dones, k == 0 ? states_in : states_out, states_out, lstm_w, pi_w, pi_b, vf_w, vf_b,
                                 logstd, noise_at(k), rng_on, rng_seed, rng_step + k, rng_base, env_id_offset, action, clipped, value, neglogp, row + k, mb_obs,
                                 mb_actions, mb_values, mb_neglogp, mb_dones, mb_rewards, env_reward, hip_stream);
  };
  // fuse != 0: one launch per step, env.step k together with the policy step k + 1 (16-lane layout = one MFMA M-tile per four env
  // waves, the reference's 48-unit network, pools without the meteorite).  Bit-identical to the two-launch sequence, and measured
  // SLOWER on MI355X (62.9 against 58.2 us per step at 4096 envs): kept as an option, not the default.
  const bool one_tile = rollout_one_tile(h, hid);   // what the combined kernels are instantiated for
  // fuse == 2: THE WHOLE ROLLOUT AS ONE PERSISTENT LAUNCH (irrl_rollout_persistent_kernel_l16): a workgroup loops over all steps for its
  // 16 robots -- no grid-wide boundary between steps, layer-0 weights fetched into LDS once.  Same device code per step as the
  // other two paths, bit-identical buffers.
  // fuse == 3: the persistent launch with the CRITIC OFF THE PER-STEP PATH (irrl_rollout_persistent_actor_kernel_l16): the actor stack alone per
  // step, all of its LSTM operands resident in LDS; `value` / mb_values and the critic's half of `states` are NOT written -- the caller
  // evaluates the critic over the recorded observations afterwards (ppo2.Runner).  Everything else bit-identical to the other paths.
  if ((fuse == 2 || fuse == 3) && one_tile && steps > 0) {
    // the same argument checks the other two paths get from irrl_lstm_policy_step (a bad caller gets rc = 1, not an out-of-bounds access)
    if (!(obs && dones && states_in && states_out && lstm_w && pi_w && pi_b && vf_w && vf_b && logstd && action && clipped && value && neglogp &&
          env_reward && env_extra)) { g_err = "irrl_lstm_rollout: NULL argument on the persistent path"; return 1; }
    for (int i = 0; i < 12; i++)
      if (!lstm_w[i]) { g_err = "irrl_lstm_rollout: the LSTM weight table has a NULL entry"; return 1; }
    if (!mb_rewards) { g_err = "irrl_lstm_rollout: the persistent path writes the reward rows (mb_rewards is mandatory)"; return 1; }
    PolicyStepArgs a;
    a.obs = obs; a.dones = dones; a.states_in = states_in; a.states_out = states_out;
    for (int i = 0; i < 12; i++) a.w[i] = lstm_w[i];
    a.pi_w = pi_w; a.pi_b = pi_b; a.vf_w = vf_w; a.vf_b = vf_b; a.logstd = logstd; a.noise = noise;
    a.action = action; a.clipped = clipped; a.value = value; a.neglogp = neglogp;
    a.row = row; a.rng_base = rng_base;
    a.mb_obs = mb_obs; a.mb_actions = mb_actions; a.mb_values = mb_values; a.mb_neglogp = mb_neglogp; a.mb_dones = mb_dones;
    a.mb_rewards = mb_rewards; a.prev_reward = mb_rewards ? env_reward : nullptr;
    a.rng_step = rng_step; a.rng_seed = rng_seed; a.rng_on = rng_on; a.env_id_offset = (unsigned)env_id_offset;
    a.N = n; a.ob_dim = ob_dim; a.act_dim = act_dim;
    if (fuse == 3) {
      // the actor as each wave's own work (round 5, second half); IRRL_ACTOR_WAVES=0: the workgroup-wide actor step (same buffers, bit for bit)
      const char *aw = getenv("IRRL_ACTOR_WAVES");
      if (aw && aw[0] == '0')
        hipLaunchKernelGGL(irrl_rollout_persistent_actor_kernel_l16, dim3((n + 15) / 16), dim3(256), 0, h->stream, h->P, h->S, obs, env_reward, dones, env_extra, a, steps);
      else
        hipLaunchKernelGGL(irrl_rollout_persistent_actor_wave_kernel_l16, dim3((n + 15) / 16), dim3(256), 0, h->stream, h->P, h->S, obs, env_reward, dones, env_extra, a, steps);
    } else
      hipLaunchKernelGGL(irrl_rollout_persistent_kernel_l16, dim3((n + 15) / 16), dim3(256), 0, h->stream, h->P, h->S, obs, env_reward, dones, env_extra, a, steps);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (fuse == 3) { g_err = "irrl_lstm_rollout: fuse = 3 (critic off the per-step path) exists for the persistent kernel only: 16 lanes per robot, hid 48, no Crutial, published contact rule"; return 1; }
  const bool fused = fuse == 1 && one_tile && steps > 1;
  if (steps > 0 && policy(0) != 0) { g_err = "irrl_lstm_rollout: policy step refused its arguments"; return 1; }
  for (int k = 0; k < steps; k++) {
    if (fused && k + 1 < steps) {
      PolicyStepArgs a;
      a.obs = obs; a.dones = dones; a.states_in = states_out; a.states_out = states_out;
      for (int i = 0; i < 12; i++) a.w[i] = lstm_w[i];
      a.pi_w = pi_w; a.pi_b = pi_b; a.vf_w = vf_w; a.vf_b = vf_b; a.logstd = logstd; a.noise = noise_at(k + 1);
      a.action = action; a.clipped = clipped; a.value = value; a.neglogp = neglogp;
      a.row = row + k + 1; a.rng_base = rng_base;
      a.mb_obs = mb_obs; a.mb_actions = mb_actions; a.mb_values = mb_values; a.mb_neglogp = mb_neglogp; a.mb_dones = mb_dones;
      a.mb_rewards = mb_rewards; a.prev_reward = mb_rewards ? env_reward : nullptr;
      a.rng_step = rng_step + k + 1; a.rng_seed = rng_seed; a.rng_on = rng_on; a.env_id_offset = (unsigned)env_id_offset;
      a.N = n; a.ob_dim = ob_dim; a.act_dim = act_dim;
      hipLaunchKernelGGL(irrl_step_policy_kernel_l16, dim3((n + 15) / 16), dim3(256), 0, h->stream, h->P, h->S, (const float *)clipped, obs,
                         env_reward, dones, env_extra, a);
    } else {
      IRRL_LAUNCH_STEP(h, lane_grid(h, n), h->P, h->S, (const float *)clipped, obs, env_reward, dones, env_extra);
      if (k + 1 < steps && policy(k + 1) != 0) { g_err = "irrl_lstm_rollout: policy step refused its arguments"; return 1; }
    }
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// The MlpPolicy twin of irrl_lstm_rollout (no recurrent state).  fuse == 2 (and a pool the combined kernel is instantiated for: 16 lanes
// per robot, no meteorite, published contact rule, 64 hidden units, 35 observations): the whole rollout as ONE persistent launch
// (irrl_rollout_persistent_mlp_kernel_l16); otherwise 2 x steps launches back to back.  Bit-identical buffers either way.
int irrl_mlp_rollout(irrl_env *h, int steps, int hid, int ob_dim, int act_dim, float *obs, uint8_t *dones, const float *const *mlp_w, const float *pi_w,
                     const float *pi_b, const float *vf_w, const float *vf_b, const float *logstd, const float *noise, int rng_on, unsigned rng_seed,
                     long long rng_step, const long long *rng_base, int env_id_offset, float *action, float *clipped, float *value, float *neglogp,
                     long long row, float *mb_obs, float *mb_actions, float *mb_values, float *mb_neglogp, uint8_t *mb_dones, float *mb_rewards,
                     float *env_reward, float *env_extra, int fuse, void *hip_stream) {
  if (need_init(h)) return 1;
  if (steps < 0 || row < 0 || act_dim != 12 || ob_dim != 35 || hid != 64) { g_err = "irrl_mlp_rollout: steps >= 0, row >= 0, ob 35, act 12, hid 64"; return 1; }
  if (!(mb_obs && mb_actions && mb_values && mb_neglogp && mb_dones)) { g_err = "irrl_mlp_rollout: the rollout buffers are mandatory"; return 1; }
  if (!(obs && dones && mlp_w && pi_w && pi_b && vf_w && vf_b && logstd && action && clipped && value && neglogp && env_reward && env_extra)) {
    g_err = "irrl_mlp_rollout: NULL argument"; return 1;
  }
  for (int i = 0; i < 8; i++)
    if (!mlp_w[i]) { g_err = "irrl_mlp_rollout: the weight table has a NULL entry"; return 1; }
  HIP_TRY(hipSetDevice(h->device));
  h->stream = (hipStream_t)hip_stream;
  const int n = h->P.n_envs;
  auto noise_at = [&](int k) { return noise ? noise + (size_t)k * (size_t)n * (size_t)act_dim : nullptr; };
  const bool one_tile = h->lanes == 16 && !h->P.crutial && h->P.contact_rule && shipped_solver(h);
  if (fuse == 2 && one_tile && steps > 0) {
    if (!mb_rewards) { g_err = "irrl_mlp_rollout: the persistent path writes the reward rows (mb_rewards is mandatory)"; return 1; }
    PolicyStepArgs a;
    a.obs = obs; a.dones = dones; a.states_in = nullptr; a.states_out = nullptr;
    for (int i = 0; i < 12; i++) a.w[i] = i < 8 ? mlp_w[i] : nullptr;
    a.pi_w = pi_w; a.pi_b = pi_b; a.vf_w = vf_w; a.vf_b = vf_b; a.logstd = logstd; a.noise = noise;
    a.action = action; a.clipped = clipped; a.value = value; a.neglogp = neglogp;
    a.row = row; a.rng_base = rng_base;
    a.mb_obs = mb_obs; a.mb_actions = mb_actions; a.mb_values = mb_values; a.mb_neglogp = mb_neglogp; a.mb_dones = mb_dones;
    a.mb_rewards = mb_rewards; a.prev_reward = env_reward;
    a.rng_step = rng_step; a.rng_seed = rng_seed; a.rng_on = rng_on; a.env_id_offset = (unsigned)env_id_offset;
    a.N = n; a.ob_dim = ob_dim; a.act_dim = act_dim;
    hipLaunchKernelGGL(irrl_rollout_persistent_mlp_kernel_l16, dim3((n + 15) / 16), dim3(256), 0, h->stream, h->P, h->S, obs, env_reward, dones, env_extra, a, steps);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  for (int k = 0; k < steps; k++) {
    if (irrl_mlp_policy_step(hid, ob_dim, act_dim, n, obs, dones, mlp_w, pi_w, pi_b, vf_w, vf_b, logstd, noise_at(k), rng_on, rng_seed, rng_step + k, rng_base,
                             env_id_offset, action, clipped, value, neglogp, row + k, mb_obs, mb_actions, mb_values, mb_neglogp, mb_dones, mb_rewards, env_reward,
                             hip_stream) != 0) { g_err = "irrl_mlp_rollout: policy step refused its arguments"; return 1; }
    IRRL_LAUNCH_STEP(h, lane_grid(h, n), h->P, h->S, (const float *)clipped, obs, env_reward, dones, env_extra);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// pinned staging layout: action | ob | reward | extra | done
static void staging(irrl_env *h, float **a, float **o, float **r, float **x, uint8_t **d) {
  const size_t n = (size_t)h->P.n_envs;
  char *p = h->h_pinned;
  *a = (float *)p; p += n * 12 * 4;
  *o = (float *)p; p += n * 35 * 4;
  *r = (float *)p; p += n * 4;
  *x = (float *)p; p += n * 6 * 4;
  *d = (uint8_t *)p;
}

static int step_host_impl(irrl_env *h, int n_step, const float *action, float *ob, float *reward, uint8_t *done, float *extra) {
  if (need_init(h)) return 1;
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = (size_t)n_step;
  float *pa, *po, *pr, *px; uint8_t *pd;
  staging(h, &pa, &po, &pr, &px, &pd);
  std::memcpy(pa, action, n * 12 * 4);
  HIP_TRY(hipMemcpyAsync(h->d_action, pa, n * 12 * 4, hipMemcpyHostToDevice, h->stream));
  EnvParams P = h->P;
  P.n_envs = n_step;
  IRRL_LAUNCH_STEP(h, lane_grid(h, n_step), P, h->S, (const float *)h->d_action, h->d_ob, h->d_reward, h->d_done, h->d_extra);
  HIP_TRY(hipGetLastError());
  if (n_step == h->P.n_envs) {
    // device outputs and their pinned staging are laid out alike (ob | reward | extra | done): one copy
    HIP_TRY(hipMemcpyAsync(po, h->d_ob, n * (35 * 4 + 4 + 6 * 4 + 1), hipMemcpyDeviceToHost, h->stream));
  } else {
    HIP_TRY(hipMemcpyAsync(po, h->d_ob, n * 35 * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(pr, h->d_reward, n * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(px, h->d_extra, n * 6 * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(pd, h->d_done, n, hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::memcpy(ob, po, n * 35 * 4);
  std::memcpy(reward, pr, n * 4);
  std::memcpy(extra, px, n * 6 * 4);
  std::memcpy(done, pd, n);
  return 0;
}
int irrl_env_step_host(irrl_env *h, const float *action, float *ob, float *reward, uint8_t *done, float *extra) {
  return step_host_impl(h, h->P.n_envs, action, ob, reward, done, extra);
}
int irrl_env_test_step_host(irrl_env *h, const float *action, float *ob, float *reward, uint8_t *done, float *extra) {
  return step_host_impl(h, 1, action, ob, reward, done, extra);
}

int irrl_env_reset(irrl_env *h, float *ob) {
  if (need_init(h)) return 1;
  if (use_device(h)) return 1;
  IRRL_LAUNCH(h, irrl_reset_kernel, lane_grid(h, h->P.n_envs), h->P, h->S, ob);
  HIP_TRY(hipGetLastError());
  return 0;
}
int irrl_env_observe(irrl_env *h, float *ob) {
  if (need_init(h)) return 1;
  if (use_device(h)) return 1;
  IRRL_LAUNCH(h, irrl_observe_kernel, lane_grid(h, h->P.n_envs), h->P, h->S, ob);
  HIP_TRY(hipGetLastError());
  return 0;
}
int irrl_env_is_terminal(irrl_env *h, uint8_t *done) {
  if (need_init(h)) return 1;
  if (use_device(h)) return 1;
  hipLaunchKernelGGL(irrl_terminal_kernel, dim3((unsigned)((h->P.n_envs + 255) / 256)), dim3(256), 0, h->stream, h->P, h->S, done);
  HIP_TRY(hipGetLastError());
  return 0;
}

static int d2h(irrl_env *h, void *dst, const void *src, size_t bytes) {
  HIP_TRY(hipSetDevice(h->device));
  if (bytes > h->pinned_bytes) { g_err = "staging buffer too small"; return 1; }
  HIP_TRY(hipMemcpyAsync(h->h_pinned, src, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::memcpy(dst, h->h_pinned, bytes);
  return 0;
}
int irrl_env_reset_host(irrl_env *h, float *ob) {
  if (irrl_env_reset(h, h->d_ob)) return 1;
  return d2h(h, ob, h->d_ob, (size_t)h->P.n_envs * 35 * 4);
}
int irrl_env_observe_host(irrl_env *h, float *ob) {
  if (irrl_env_observe(h, h->d_ob)) return 1;
  return d2h(h, ob, h->d_ob, (size_t)h->P.n_envs * 35 * 4);
}
int irrl_env_is_terminal_host(irrl_env *h, uint8_t *done) {
  if (irrl_env_is_terminal(h, h->d_done)) return 1;
  return d2h(h, done, h->d_done, (size_t)h->P.n_envs);
}

int irrl_env_set_seed(irrl_env *h, int seed) { h->P.seed = (uint32_t)seed; return 0; }
int irrl_env_set_simulation_dt(irrl_env *h, double dt) {
  if (!(dt > 0)) { g_err = "simulation dt must be positive"; return 1; }
  h->P.sim_dt = (float)dt;
  h->P.loop_count = (int32_t)((double)h->P.control_dt / dt + 1e-6);
  if (h->P.loop_count < 1) h->P.loop_count = 1;
  return 0;
}
int irrl_env_set_control_dt(irrl_env *h, double dt) {
  if (!(dt > 0)) { g_err = "control dt must be positive"; return 1; }
  h->P.control_dt = (float)dt;
  h->P.loop_count = (int32_t)(dt / (double)h->P.sim_dt + 1e-6);
  if (h->P.loop_count < 1) h->P.loop_count = 1;
  irrl_host::derive_cadences(h->cfg, dt, h->P);   // attack_every / disturb_every follow the live control_dt like ENV:733,747
  irrl_host::derive_params(h->P);
  return 0;
}
int irrl_env_close(irrl_env *) { return 0; }
int irrl_env_curriculum_update(irrl_env *) { return 0; }

// ---- diagnostics getters ----
static int host_pool(irrl_env *h, std::vector<char> &mirror) {
  if (need_init(h)) return 1;
  HIP_TRY(hipSetDevice(h->device));
  mirror.resize(h->pool.bytes);
  HIP_TRY(hipMemcpyAsync(h->h_pinned, h->d_pool, h->pool.bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::memcpy(mirror.data(), h->h_pinned, h->pool.bytes);
  return 0;
}
int irrl_env_origin_state_host(irrl_env *h, float *out) {
  std::vector<char> m;
  if (host_pool(h, m)) return 1;
  EnvState S = h->pool.view(m.data());
  for (int e = 0; e < h->P.n_envs; e++) {
    float *o = out + (size_t)e * 41;
    for (int k = 0; k < 19; k++) o[k] = S.gc[e * 19 + k];
    for (int k = 0; k < 18; k++) o[19 + k] = S.gv[e * 18 + k];
    for (int k = 0; k < 4; k++) o[37 + k] = S.contact[e * 4 + k];
  }
  return 0;
}
int irrl_env_reference_state_host(irrl_env *h, float *out) {
  std::vector<char> m;
  if (host_pool(h, m)) return 1;
  EnvState S = h->pool.view(m.data());
  for (int e = 0; e < h->P.n_envs; e++)
    for (int k = 0; k < 12; k++) { out[(size_t)e * 24 + k] = S.joint_ref[e * 12 + k]; out[(size_t)e * 24 + 12 + k] = S.joint_dot_ref[e * 12 + k]; }
  return 0;
}
int irrl_env_joint_effort_host(irrl_env *h, float *out) {
  std::vector<char> m;
  if (host_pool(h, m)) return 1;
  EnvState S = h->pool.view(m.data());
  std::memcpy(out, S.torque, (size_t)h->P.n_envs * 12 * 4);
  return 0;
}
int irrl_env_generalized_force_host(irrl_env *h, float *out) {
  std::vector<char> m;
  if (host_pool(h, m)) return 1;
  EnvState S = h->pool.view(m.data());
  for (int e = 0; e < h->P.n_envs; e++) {
    for (int k = 0; k < 6; k++) out[(size_t)e * 18 + k] = 0.0f;  // no base wrench (ForceDisturbance never fires, ENV:751)
    for (int k = 0; k < 12; k++) out[(size_t)e * 18 + 6 + k] = S.torque[e * 12 + k];
  }
  return 0;
}
static int probe(irrl_env *h, float *minv_host, float *nonlin_host) {
  if (need_init(h)) return 1;
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = (size_t)h->P.n_envs;
  float *d_minv = h->d_scratch, *d_nl = h->d_scratch + n * 324;
  IRRL_LAUNCH(h, irrl_probe_kernel, lane_grid(h, h->P.n_envs), h->P, h->S, minv_host ? d_minv : (float *)nullptr, nonlin_host ? d_nl : (float *)nullptr);
  HIP_TRY(hipGetLastError());
  if (minv_host && d2h(h, minv_host, d_minv, n * 324 * 4)) return 1;
  if (nonlin_host && d2h(h, nonlin_host, d_nl, n * 18 * 4)) return 1;
  return 0;
}
int irrl_env_inverse_mass_matrix_host(irrl_env *h, float *out) { return probe(h, out, nullptr); }
int irrl_env_nonlinear_host(irrl_env *h, float *out) { return probe(h, nullptr, out); }
int irrl_env_set_contact_coeff_host(irrl_env *h, const float *in) {
  if (need_init(h)) return 1;
  HIP_TRY(hipSetDevice(h->device));
  const size_t bytes = (size_t)h->P.n_envs * 3 * 4;
  std::memcpy(h->h_pinned, in, bytes);
  HIP_TRY(hipMemcpyAsync(h->S.material, h->h_pinned, bytes, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}
int irrl_env_sphere_info_host(irrl_env *h, float *out) {   // Environment.hpp:1423-1436: centre of cubes[0] and its radius
  if (h && !h->P.crutial) { g_err = "GetSphereInfo: Please make sure the [Flag_Crucial] is True (Environment.hpp:1434)"; return 1; }
  if (need_init(h)) return 1;
  std::vector<float> sph((size_t)h->P.n_envs * 9);
  if (d2h(h, sph.data(), h->S.sphere, sph.size() * 4)) return 1;
  for (int e = 0; e < h->P.n_envs; e++) {
    for (int k = 0; k < 3; k++) out[4 * e + k] = sph[9 * e + k];
    out[4 * e + 3] = sph[9 * e + 6];
  }
  return 0;
}

int irrl_env_get_state_host(irrl_env *h, double *out) {
  std::vector<char> m;
  if (host_pool(h, m)) return 1;
  h->pool.pack(m.data(), out);
  return 0;
}
int irrl_env_set_state_host(irrl_env *h, const double *in) {
  std::vector<char> m;
  if (host_pool(h, m)) return 1;
  h->pool.unpack(in, m.data());
  std::memcpy(h->h_pinned, m.data(), h->pool.bytes);
  HIP_TRY(hipMemcpyAsync(h->d_pool, h->h_pinned, h->pool.bytes, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}
int irrl_env_heightfield_host(irrl_env *h, float *out, int *nx, int *ny) {
  if (nx) *nx = h->P.hf_nx;
  if (ny) *ny = h->P.hf_ny;
  if (!h->P.terrain) { g_err = "this pool runs on the plane z = 0 (Terrain: False)"; return 1; }
  if (out) std::memcpy(out, h->h_height.data(), h->h_height.size() * sizeof(float));
  return 0;
}
// device-side snapshot of the whole state pool and its restoration, stream-ordered on the pool's stream (no host copy): lets a
// caller run throw-away steps (e.g. the warm-up before a hipGraph capture) and continue from where it was
int irrl_env_snapshot(irrl_env *h) {
  if (need_init(h)) return 1;
  if (use_device(h)) return 1;
  if (!h->d_snapshot) HIP_TRY(hipMalloc(&h->d_snapshot, h->pool.bytes + (size_t)h->P.n_envs * 16));
  HIP_TRY(hipMemcpyAsync(h->d_snapshot, h->d_pool, h->pool.bytes, hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync((char *)h->d_snapshot + h->pool.bytes, h->d_counters, (size_t)h->P.n_envs * 16, hipMemcpyDeviceToDevice, h->stream));
  return 0;
}
int irrl_env_restore(irrl_env *h) {
  if (need_init(h)) return 1;
  if (use_device(h)) return 1;
  if (!h->d_snapshot) { g_err = "irrl_env_restore: no snapshot has been taken"; return 1; }
  HIP_TRY(hipMemcpyAsync(h->d_pool, h->d_snapshot, h->pool.bytes, hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_counters, (char *)h->d_snapshot + h->pool.bytes, (size_t)h->P.n_envs * 16, hipMemcpyDeviceToDevice, h->stream));
  return 0;
}
// stream-ordered variant: sums into d_out[3] (device, unsigned long long) on the pool's stream, no host synchronisation
__global__ void irrl_counters_kernel(int n, const uint32_t *episode, const uint32_t *cc, const int32_t *frame, unsigned long long *out) {
  __shared__ unsigned long long s[3][256];
  unsigned long long a = 0, b = 0, c = 0;
  for (int i = (int)threadIdx.x; i < n; i += 256) { a += episode[i]; c += (unsigned long long)frame[i]; }
  for (int i = (int)threadIdx.x; i < 4 * n; i += 256) b += cc[i];
  s[0][threadIdx.x] = a; s[1][threadIdx.x] = b; s[2][threadIdx.x] = c;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { s[0][threadIdx.x] += s[0][threadIdx.x + w]; s[1][threadIdx.x] += s[1][threadIdx.x + w]; s[2][threadIdx.x] += s[2][threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = s[0][0]; out[1] = s[1][0]; out[2] = s[2][0]; }
}
int irrl_env_counters(irrl_env *h, unsigned long long *d_out) {
  if (need_init(h)) return 1;
  if (use_device(h)) return 1;
  hipLaunchKernelGGL(irrl_counters_kernel, dim3(1), dim3(256), 0, h->stream, h->P.n_envs, (const uint32_t *)h->S.episode, (const uint32_t *)h->d_counters,
                     (const int32_t *)h->S.frame_idx, d_out);
  HIP_TRY(hipGetLastError());
  return 0;
}
// diagnostic counters summed over the pool: out[0] = episodes started (init + every reset), out[1] = toe-substeps spent in
// the contact list, out[2] = control steps since the last reset summed over the envs (frame_idx)
int irrl_env_counters_host(irrl_env *h, unsigned long long *out) {
  if (need_init(h)) return 1;
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = (size_t)h->P.n_envs;
  std::vector<uint32_t> ep(n), cc(n * 4);
  std::vector<int32_t> fr(n);
  HIP_TRY(hipDeviceSynchronize());   // steps may have been replayed from a graph on another stream than h->stream
  HIP_TRY(hipMemcpy(ep.data(), h->S.episode, n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(cc.data(), h->d_counters, n * 16, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(fr.data(), h->S.frame_idx, n * 4, hipMemcpyDeviceToHost));
  unsigned long long a = 0, b = 0, c = 0;
  for (size_t i = 0; i < n; i++) { a += ep[i]; c += (unsigned long long)fr[i]; }
  for (size_t i = 0; i < n * 4; i++) b += cc[i];
  out[0] = a; out[1] = b; out[2] = c;
  return 0;
}
double irrl_env_cfg_value(const irrl_env *h, const char *key) {
  double d = NAN;
  std::string e;
  bool b;
  auto it = h->cfg.kv.find(key);
  if (it == h->cfg.kv.end()) return NAN;
  if (irrl_host::Config::to_bool(it->second, b)) return b ? 1.0 : 0.0;
  if (!h->cfg.get_double(key, d, e)) return NAN;
  return d;
}

// ---- PMC calibration: a copy with the env kernels' access width (one dword per lane), known byte count ----
__global__ void irrl_calib_copy_kernel(const float *__restrict__ src, float *__restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
int irrl_calib_copy_dword(const float *src, float *dst, size_t n, void *hip_stream) {
  hipLaunchKernelGGL(irrl_calib_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, src, dst, n);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- synthetic action stream of the benchmark: a = clip(sigma * N(0,1), -1, 1) from Philox4x32-10 keyed
// (seed, 'ACT1'), counter (env, step, block, 0): block j of (env, step) gives actions 4j .. 4j+3 by Box-Muller on the
// uniform pairs (u0,u1) -> (cos, sin), (u2,u3) -> (cos, sin).  Rows are [step - step0][env - env0][12].
__global__ void irrl_bench_actions_kernel(unsigned seed, int env0, int n_envs, long long step0, int n_steps, float sigma, float *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n_steps * n_envs * 3;
  if (i >= total) return;
  const int j = (int)(i % 3);
  const size_t se = i / 3;
  const int e = (int)(se % (size_t)n_envs);
  const long long s = step0 + (long long)(se / (size_t)n_envs);
  unsigned c0 = (unsigned)(env0 + e), c1 = (unsigned)s, c2 = (unsigned)j, c3 = 0u;
  unsigned k0 = seed, k1 = 0x41435431u;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned hi0 = __umulhi(c0, 0xD2511F53u), lo0 = c0 * 0xD2511F53u;
    const unsigned hi1 = __umulhi(c2, 0xCD9E8D57u), lo1 = c2 * 0xCD9E8D57u;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const float sc = 1.0f / 16777216.0f;
  const float u0 = (float)(c0 >> 8) * sc, u1 = (float)(c1 >> 8) * sc, u2 = (float)(c2 >> 8) * sc, u3 = (float)(c3 >> 8) * sc;
  const float ra = sqrtf(-2.0f * logf(1.0f - u0)), rb = sqrtf(-2.0f * logf(1.0f - u2));
  const float aa = 6.283185307179586f * u1, ab = 6.283185307179586f * u3;
  float z[4] = {ra * cosf(aa), ra * sinf(aa), rb * cosf(ab), rb * sinf(ab)};
  float *o = out + se * 12 + 4 * j;
#pragma unroll
  for (int k = 0; k < 4; k++) o[k] = fminf(fmaxf(sigma * z[k], -1.0f), 1.0f);
}
int irrl_bench_actions(unsigned seed, int env0, int n_envs, long long step0, int n_steps, float sigma, float *out, void *hip_stream) {
  if (n_envs <= 0 || n_steps <= 0 || !out) { g_err = "irrl_bench_actions: empty request"; return 1; }
  const size_t total = (size_t)n_steps * n_envs * 3;
  hipLaunchKernelGGL(irrl_bench_actions_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, seed, env0, n_envs, step0,
                     n_steps, sigma, out);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- GAE reverse scan (ppo2.py:554-568): one lane per env, coalesced [T,N] rows ----
__global__ void irrl_gae_kernel(int T, int N, const float *__restrict__ rewards, const float *__restrict__ values,
                                const uint8_t *__restrict__ dones, const float *__restrict__ last_values,
                                const uint8_t *__restrict__ last_dones, float gamma, float lam, float *__restrict__ adv,
                                float *__restrict__ returns) {
#pragma clang fp contract(off)  // keep the reference's operation order exactly (no FMA fusion): bit-equal to the oracle
  int n = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (n >= N) return;
  float last = 0.0f;
  float nextv = last_values[n];
  float nonterm = 1.0f - (float)last_dones[n];
  // the scan is serial, its operands are not: eight steps' rows are requested together (one lane per env = 64 waves on the whole chip:
  // requested one step at a time the scan runs at the latency of a load per step, 390 ns x 750)
  constexpr int B = 8;
  for (int t0 = T - 1; t0 >= 0; t0 -= B) {
    float v[B], r[B];
    uint8_t d[B];
#pragma unroll
    for (int j = 0; j < B; j++) {
      const int t = t0 - j >= 0 ? t0 - j : 0;
      const size_t i = (size_t)t * N + n;
      v[j] = values[i]; r[j] = rewards[i]; d[j] = dones[i];
    }
#pragma unroll
    for (int j = 0; j < B; j++) {
      if (t0 - j < 0) break;
      const size_t i = (size_t)(t0 - j) * N + n;
      float delta = r[j] + gamma * nextv * nonterm - v[j];
      last = delta + gamma * lam * nonterm * last;
      adv[i] = last;
      returns[i] = last + v[j];
      nextv = v[j];
      nonterm = 1.0f - (float)d[j];
    }
  }
}
int irrl_gae(int T, int N, const float *rewards, const float *values, const uint8_t *dones, const float *last_values,
             const uint8_t *last_dones, float gamma, float lam, float *adv, float *returns, void *hip_stream) {
  if (T <= 0 || N <= 0) { g_err = "irrl_gae: empty rollout"; return 1; }
  hipLaunchKernelGGL(irrl_gae_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, (hipStream_t)hip_stream, T, N, rewards, values, dones,
                     last_values, last_dones, gamma, lam, adv, returns);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- PPO2 clipped-surrogate loss (kernel template: csrc/lstm_kernels.hip, irrl_ppo_loss_kernel) ----
int irrl_ppo_loss(size_t M, int act_dim, const float *mean, const float *logstd, const float *vpred, const float *actions,
                  const float *returns, const float *old_values, const float *old_neglogp, const float *adv_stats, float cliprange,
                  float vf_coef, float *d_mean, float *d_vpred, float *partials, int n_blocks, void *hip_stream) {
  if (M == 0 || n_blocks <= 0) { g_err = "irrl_ppo_loss: empty batch"; return 1; }
  const float inv_m = 1.0f / (float)M;
  hipStream_t s = (hipStream_t)hip_stream;
  if (act_dim == 12) hipLaunchKernelGGL(irrl_ppo_loss_kernel<12>, dim3((unsigned)n_blocks), dim3(256), 0, s, M, mean, logstd, vpred, actions, returns, old_values, old_neglogp, adv_stats, cliprange, vf_coef, inv_m, d_mean, d_vpred, partials);
  else { g_err = "irrl_ppo_loss: act_dim must be 12"; return 1; }
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- heads + PPO2 loss in one launch (kernel: csrc/lstm_kernels.hip, irrl_ppo_heads_loss_kernel) ----
int irrl_ppo_heads_loss(size_t M, int act_dim, int hid, const float *h_pi, const float *h_v, const float *pi_w, const float *pi_b, const float *vf_w,
                        const float *vf_b, const float *logstd, const float *actions, const float *returns, const float *old_values,
                        const float *old_neglogp, const float *adv_stats, float cliprange, float vf_coef, float *d_hpi, float *d_hv, float *mean_out,
                        float *value_out, float *partials, int n_blocks, void *hip_stream) {
  if (M == 0 || n_blocks <= 0) { g_err = "irrl_ppo_heads_loss: empty batch"; return 1; }
  if (act_dim != 12 || hid != 48) { g_err = "irrl_ppo_heads_loss: built for 48-unit latents and 12 actions"; return 1; }
  const float inv_m = 1.0f / (float)M;
  hipLaunchKernelGGL((irrl_ppo_heads_loss_kernel<12, 48>), dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)hip_stream, M, h_pi, h_v, pi_w, pi_b, vf_w, vf_b,
                     logstd, actions, returns, old_values, old_neglogp, adv_stats, cliprange, vf_coef, inv_m, d_hpi, d_hv, mean_out, value_out, partials);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- MlpPolicy: gradients of one PPO2 minibatch, one launch per network (kernel: csrc/mlp_update.hpp) ----
int irrl_mlp_ppo_grads(int kind, size_t n, const int64_t *idx, int ob_dim, int hid, int act_dim, const float *obs, const float *actions,
                       const float *returns, const float *old_values, const float *old_neglogp, const float *w1, const float *b1, const float *w2,
                       const float *b2, const float *w3, const float *b3, const float *logstd, const float *adv_stats, float cliprange, float vf_coef,
                       float *partials, int n_blocks, void *hip_stream) {
  if (n == 0 || n_blocks <= 0) { g_err = "irrl_mlp_ppo_grads: empty batch"; return 1; }
  if (ob_dim != IRRL_MLP_OB || hid != IRRL_MLP_H || act_dim != 12) { g_err = "irrl_mlp_ppo_grads: built for 35 observations, [64, 64] hidden units and 12 actions"; return 1; }
  if (kind != 0 && kind != 1) { g_err = "irrl_mlp_ppo_grads: kind is 0 (policy network) or 1 (value network)"; return 1; }
  MlpUpdateArgs a;
  a.n = n; a.idx = idx; a.obs = obs; a.actions = actions; a.returns = returns; a.old_values = old_values; a.old_neglogp = old_neglogp; a.rec = nullptr;
  a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.logstd = logstd; a.adv_stats = adv_stats;
  a.cliprange = cliprange; a.vf_coef = vf_coef; a.inv_n = 1.0f / (float)n; a.partials = partials;
  if (kind == 0) hipLaunchKernelGGL(irrl_mlp_ppo_kernel<0>, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)hip_stream, a);
  else hipLaunchKernelGGL(irrl_mlp_ppo_kernel<1>, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)hip_stream, a);
  HIP_TRY(hipGetLastError());
  return 0;
}
int irrl_mlp_ppo_partial_len(void) { return IRRL_MLP_P; }

// the same gradients with every product formed as three bf16 plane products on the matrix cores (kernel: csrc/mlp_bf16.hpp); same arguments,
// same partial-sum rows.  returns 0 on success
static int mlp_bf16_launch(const char *who, int kind, bool use_rec, size_t n, int n_blocks, const MlpUpdateArgs &a, void *hip_stream) {
  if (n == 0 || n_blocks <= 0) { g_err = std::string(who) + ": empty batch"; return 1; }
  if (kind != 0 && kind != 1) { g_err = std::string(who) + ": kind is 0 (policy network) or 1 (value network)"; return 1; }
  // the opt-in belongs to the CURRENT device (a process may drive several GPUs): remembered per device ordinal
  static IrrlPerDeviceFlag allowed_on;     // function-local static with a constructor: initialised once, thread-safe (C++11)
  int dev_ = 0;
  if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= IRRL_MAX_DEVICES) { g_err = std::string(who) + ": no current device"; return 1; }
  int &allowed = allowed_on.v[dev_];
  if (allowed < 0) {   // the weight planes and the waves' images exceed the 64 KB a kernel gets without asking (gfx950 has 160 KB per CU)
    const void *ks[4] = {(const void *)irrl_mlp_ppo_bf16_kernel<0, false>, (const void *)irrl_mlp_ppo_bf16_kernel<1, false>,
                         (const void *)irrl_mlp_ppo_bf16_kernel<0, true>, (const void *)irrl_mlp_ppo_bf16_kernel<1, true>};
    const void *kp[4] = {(const void *)irrl_mlp_ppo_bf16_pc_kernel<0, false>, (const void *)irrl_mlp_ppo_bf16_pc_kernel<1, false>,
                         (const void *)irrl_mlp_ppo_bf16_pc_kernel<0, true>, (const void *)irrl_mlp_ppo_bf16_pc_kernel<1, true>};
    allowed = 0;
    for (int i = 0; i < 4; i++) {
      if (hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, mlp_bf16_lds_bytes()) != hipSuccess) allowed = 1;
      if (hipFuncSetAttribute(kp[i], hipFuncAttributeMaxDynamicSharedMemorySize, mlp_bf16_pc_lds_bytes()) != hipSuccess) allowed = 1;
    }
  }
  if (allowed != 0) { g_err = std::string(who) + ": the device refused the kernel's LDS size"; return 1; }
  // IRRL_MLP_WAVES=4: the one-wave-per-SIMD kernel of mlp_bf16.hpp; default: producer / consumer wave pairs (mlp_bf16_pc.hpp) -- same partial rows, bit for bit
  // (read per call: the parity test runs both kernels in one process)
  const char *waves_env = getenv("IRRL_MLP_WAVES");
  const bool pairs = !(waves_env && waves_env[0] == '4');
  const dim3 grid((unsigned)n_blocks), block(pairs ? 512 : 256);
  hipStream_t st = (hipStream_t)hip_stream;
  if (pairs) {
    if (use_rec) {
      if (kind == 0) hipLaunchKernelGGL((irrl_mlp_ppo_bf16_pc_kernel<0, true>), grid, block, mlp_bf16_pc_lds_bytes(), st, a);
      else hipLaunchKernelGGL((irrl_mlp_ppo_bf16_pc_kernel<1, true>), grid, block, mlp_bf16_pc_lds_bytes(), st, a);
    } else {
      if (kind == 0) hipLaunchKernelGGL((irrl_mlp_ppo_bf16_pc_kernel<0, false>), grid, block, mlp_bf16_pc_lds_bytes(), st, a);
      else hipLaunchKernelGGL((irrl_mlp_ppo_bf16_pc_kernel<1, false>), grid, block, mlp_bf16_pc_lds_bytes(), st, a);
    }
  } else if (use_rec) {
    if (kind == 0) hipLaunchKernelGGL((irrl_mlp_ppo_bf16_kernel<0, true>), grid, block, mlp_bf16_lds_bytes(), st, a);
    else hipLaunchKernelGGL((irrl_mlp_ppo_bf16_kernel<1, true>), grid, block, mlp_bf16_lds_bytes(), st, a);
  } else {
    if (kind == 0) hipLaunchKernelGGL((irrl_mlp_ppo_bf16_kernel<0, false>), grid, block, mlp_bf16_lds_bytes(), st, a);
    else hipLaunchKernelGGL((irrl_mlp_ppo_bf16_kernel<1, false>), grid, block, mlp_bf16_lds_bytes(), st, a);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int irrl_mlp_ppo_grads_bf16(int kind, size_t n, const int64_t *idx, int ob_dim, int hid, int act_dim, const float *obs, const float *actions,
                            const float *returns, const float *old_values, const float *old_neglogp, const float *w1, const float *b1, const float *w2,
                            const float *b2, const float *w3, const float *b3, const float *logstd, const float *adv_stats, float cliprange, float vf_coef,
                            float *partials, int n_blocks, void *hip_stream) {
  if (ob_dim != IRRL_MLP_OB || hid != IRRL_MLP_H || act_dim != 12) { g_err = "irrl_mlp_ppo_grads_bf16: built for 35 observations, [64, 64] hidden units and 12 actions"; return 1; }
  MlpUpdateArgs a;
  a.n = n; a.idx = idx; a.obs = obs; a.actions = actions; a.returns = returns; a.old_values = old_values; a.old_neglogp = old_neglogp; a.rec = nullptr;
  a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.logstd = logstd; a.adv_stats = adv_stats;
  a.cliprange = cliprange; a.vf_coef = vf_coef; a.inv_n = 1.0f / (float)n; a.partials = partials;
  return mlp_bf16_launch("irrl_mlp_ppo_grads_bf16", kind, false, n, n_blocks, a, hip_stream);
}

// the five per-sample arrays of the flat rollout -> packed 256-byte records (csrc/mlp_update.hpp IRRL_MLP_REC), once per update
int irrl_mlp_pack_records(size_t n, const float *obs, const float *actions, const float *returns, const float *old_values, const float *old_neglogp,
                          float *rec, void *hip_stream) {
  if (n == 0) { g_err = "irrl_mlp_pack_records: empty batch"; return 1; }
  if (!obs || !actions || !returns || !old_values || !old_neglogp || !rec) { g_err = "irrl_mlp_pack_records: NULL argument"; return 1; }
  if ((uintptr_t)rec & 255u) { g_err = "irrl_mlp_pack_records: rec must be 256-byte aligned (a record = two 128-byte lines)"; return 1; }
  hipLaunchKernelGGL(irrl_mlp_pack_kernel, dim3(4096), dim3(256), 0, (hipStream_t)hip_stream, n, obs, actions, returns, old_values, old_neglogp, rec);
  HIP_TRY(hipGetLastError());
  return 0;
}
int irrl_mlp_record_floats(void) { return IRRL_MLP_REC; }

// irrl_mlp_ppo_grads_bf16 reading the minibatch's samples out of the packed records: same arithmetic on the same values, bit-identical rows
int irrl_mlp_ppo_grads_bf16_rec(int kind, size_t n, const int64_t *idx, const float *rec, const float *w1, const float *b1, const float *w2,
                                const float *b2, const float *w3, const float *b3, const float *logstd, const float *adv_stats, float cliprange,
                                float vf_coef, float *partials, int n_blocks, void *hip_stream) {
  if (!rec || ((uintptr_t)rec & 255u)) { g_err = "irrl_mlp_ppo_grads_bf16_rec: rec is NULL or not 256-byte aligned"; return 1; }
  MlpUpdateArgs a;
  a.n = n; a.idx = idx; a.obs = nullptr; a.actions = nullptr; a.returns = nullptr; a.old_values = nullptr; a.old_neglogp = nullptr; a.rec = rec;
  a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.logstd = logstd; a.adv_stats = adv_stats;
  a.cliprange = cliprange; a.vf_coef = vf_coef; a.inv_n = 1.0f / (float)n; a.partials = partials;
  return mlp_bf16_launch("irrl_mlp_ppo_grads_bf16_rec", kind, true, n, n_blocks, a, hip_stream);
}

// sums[3] = (sum a, sum a^2, n) of a = returns[r] - old_values[r] over the minibatch's rows, in double; scratch: [2 * n_blocks] doubles;
// stats (may be NULL): (mean, population std) of a as floats, what the loss kernels take as adv_stats when there is one rank
int irrl_adv_moments(size_t n, const int64_t *idx, const float *returns, const float *old_values, double *scratch, int n_blocks, double *sums,
                     float *stats, void *hip_stream) {
  if (n == 0 || n_blocks <= 0) { g_err = "irrl_adv_moments: empty batch"; return 1; }
  hipLaunchKernelGGL(irrl_adv_moments_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)hip_stream, idx, n, returns, old_values, scratch, (size_t)1);
  hipLaunchKernelGGL(irrl_adv_moments_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, scratch, n_blocks, n, sums, stats);
  HIP_TRY(hipGetLastError());
  return 0;
}
// the same moments with the advantages read out of the packed records (word 51 = return - old value, formed in f32 by the pack kernel)
int irrl_adv_moments_rec(size_t n, const int64_t *idx, const float *rec, double *scratch, int n_blocks, double *sums, float *stats, void *hip_stream) {
  if (n == 0 || n_blocks <= 0 || !rec) { g_err = "irrl_adv_moments_rec: empty batch"; return 1; }
  hipLaunchKernelGGL(irrl_adv_moments_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)hip_stream, idx, n, rec + 51, (const float *)nullptr, scratch,
                     (size_t)IRRL_MLP_REC);
  hipLaunchKernelGGL(irrl_adv_moments_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, scratch, n_blocks, n, sums, stats);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---- the tail of an optimizer step on flat buffers (kernels: csrc/ppo_optim.hpp; ppo2.py:182-197 clip_by_global_norm + Adam) ----
int irrl_clip_adam(int n, float *theta, const float *grad, float *m, float *v, float grad_scale, float max_norm, float lr, float beta1, float beta2,
                   float eps, long long step, float *norm_out, void *hip_stream) {
  if (n <= 0 || !theta || !grad || !m || !v) { g_err = "irrl_clip_adam: empty parameter set"; return 1; }
  if (step < 1) { g_err = "irrl_clip_adam: step counts from 1"; return 1; }
  if (((uintptr_t)theta | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15u) { g_err = "irrl_clip_adam: buffers must be 16-byte aligned"; return 1; }
  ClipAdamArgs a;
  a.n = n; a.theta = theta; a.g = grad; a.m = m; a.v = v; a.grad_scale = grad_scale; a.max_norm = max_norm;
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.step_size = (float)((double)lr / (1.0 - std::pow((double)beta1, (double)step)));
  a.inv_bc2_sqrt = (float)(1.0 / std::sqrt(1.0 - std::pow((double)beta2, (double)step)));
  a.norm_out = norm_out;
  hipLaunchKernelGGL(irrl_clip_adam_kernel, dim3(1), dim3(1024), 0, (hipStream_t)hip_stream, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

int irrl_sum_rows_scatter(const float *part, int nmat, int rows, int cols, const int *map, const float *add, float *out, void *hip_stream) {
  if (nmat <= 0 || rows <= 0 || cols <= 0 || !part || !map || !out) { g_err = "irrl_sum_rows_scatter: empty request"; return 1; }
  hipLaunchKernelGGL(irrl_sum_rows_scatter_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)nmat), dim3(256), 0, (hipStream_t)hip_stream, part, rows,
                     cols, map, add, out);
  HIP_TRY(hipGetLastError());
  return 0;
}

int irrl_random_permutation(long long n, unsigned seed, unsigned counter, long long *out, void *hip_stream) {
  if (n <= 0 || n > (1ll << 30) || !out) { g_err = "irrl_random_permutation: 0 < n <= 2^30"; return 1; }
  int bits = 1;
  while ((1ll << bits) < n) bits++;
  const int half_bits = (bits + 1) / 2 < 1 ? 1 : (bits + 1) / 2;
  hipLaunchKernelGGL(irrl_random_permutation_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, (uint32_t)n, half_bits, seed,
                     counter, out);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"
