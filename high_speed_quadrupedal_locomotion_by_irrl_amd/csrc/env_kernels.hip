// env_kernels.hip -- __global__ wrappers of the lane bodies in env_core.hpp (gfx950 only); compiled once per lane layout (build.py).
// Launch shape of the lane kernels: 64-thread workgroups = one wave = 4 robots (16 lanes per robot, one DPP row each: pools <= 6144) or 16 robots
// (4 lanes per robot, one DPP quad each).  At N = 4096 the 16-lane layout is 1024 workgroups -> one wave on every SIMD of the 256 CUs; the kernels
// keep the whole robot state in registers across the 8 substeps (no LDS; scratch only in the reset branch), so __launch_bounds__(256, 1) lets the
// allocator use the full 512-register budget of a SIMD that hosts a single wave.  The rollout kernels (policy in the same launch) run 256-thread
// workgroups: four env waves = 16 robots per workgroup.
#ifndef IRRL_LANES_PER_ROBOT
#define IRRL_LANES_PER_ROBOT 16
#endif
#if IRRL_LANES_PER_ROBOT == 16
#include "lanes_hip16.hpp"
#define IRRL_ROBOTS_PER_WAVE 4
#else
#include "lanes_hip.hpp"
#define IRRL_ROBOTS_PER_WAVE 16
#endif
#include "env_core.hpp"
// second instantiation of the lane bodies with the meteorite (Crutial: True) compiled out: pools without it -- every benchmark
// and training configuration -- run the step kernel built from this one (measured: the run-time-flag version costs 0.5 us of
// the 38 us step even with the flag off: 16 more VGPRs live across the substeps and the branches in the schedule)
#undef IRRL_CORE_NS
#undef IRRL_CRUTIAL
#define IRRL_CORE_NS irrl_plain
#define IRRL_CRUTIAL(P) false
#include "env_core.hpp"

#if IRRL_LANES_PER_ROBOT == 16
#include "policy_step.hpp"   // the LSTM policy's rollout step (device code shared with lstm_kernels.hip)
#endif

// This file is compiled three times (build.py): once per lane layout, kernel names suffixed _l16 / _l4, and the 4-lane layout once more for
// TWO waves per SIMD (_l4w2, -DIRRL_L4_WAVES2: 256 registers per wave, ~130-220 of the step kernels' values in scratch).  Pools of more than
// 16 384 robots are more 4-lane waves than the chip has SIMDs; two resident waves issue 1.35 x what one does (HISTORY.md Appendix C, 2a), which
// pays for the spills: 32 768 envs 394 -> 450 M env-steps/s, 131 072 envs 394 -> 482 M, 16 384 envs (one wave per SIMD either way) 393 -> 386 M
// -- the launcher takes _l4w2 above 16 384 robots only (profiles/r06_ab_l4_two_waves_per_simd_same_box.log).  Same source, same arithmetic.
#if IRRL_LANES_PER_ROBOT == 16
#define IRRL_K(name) name##_l16
#elif defined(IRRL_L4_WAVES2)
#define IRRL_K(name) name##_l4w2
#else
#define IRRL_K(name) name##_l4
#endif

// XCD-aware block -> robots mapping.  The hardware hands consecutive workgroups to the 8 XCDs round-robin (workgroup b runs on XCD
// b % 8), and every XCD has its own L2.  With the identity mapping the four robots of workgroup b and those of b + 1 -- neighbours
// in every array of the pool, 16 B apart in a per-env scalar array -- sit on different XCDs, so one 128-byte line is fetched from
// HBM by up to eight L2s.  irrl_xcd_block() renumbers the blocks so that XCD x owns ONE contiguous range of robots (a bijection of
// [0, gridDim.x) for any grid size): a line is then fetched by one L2 (two at a range boundary).  Results do not change -- only
// which wave computes which robot.
__device__ __forceinline__ int irrl_xcd_block() {
#ifdef IRRL_NO_XCD_SWIZZLE   /* A/B switch of tools/build_variants.py */
  return (int)blockIdx.x;
#endif
  const int nb = (int)gridDim.x, b = (int)blockIdx.x;
  const int x = b & 7, j = b >> 3;                  // XCD of this workgroup, its rank among that XCD's workgroups
  const int q = nb >> 3, r = nb & 7;                // XCD x gets q workgroups, + 1 if x < r
  return x * q + (x < r ? x : r) + j;
}
// env_: robot of this lane; leg_: its leg; valid_: this lane owns the stores of (robot, leg) -- with 16 lanes per robot
// that is sub-lane 0 of each quad.  Idle rows shadow the last robot with their stores masked.  BLK: the (renumbered) block index.
#if IRRL_LANES_PER_ROBOT == 16
#define IRRL_LANE_PROLOGUE_B(BLK)                                                            \
  const int lane_ = (int)(threadIdx.x & 63u);                                                \
  const int wave_ = (int)((BLK) * (blockDim.x >> 6) + (threadIdx.x >> 6));                    \
  int env_ = wave_ * 4 + (lane_ >> 4);                                                       \
  const int leg_ = (lane_ >> 2) & 3;                                                         \
  const bool valid_ = (env_ < P.n_envs) && ((lane_ & 3) == 0);                               \
  if (env_ >= P.n_envs) env_ = P.n_envs - 1;
#else
#define IRRL_LANE_PROLOGUE_B(BLK)                              \
  const int lane_ = (int)(threadIdx.x & 63u);                  \
  int env_ = (int)((BLK) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (lane_ >> 2); \
  const int leg_ = lane_ & 3;                                  \
  const bool valid_ = env_ < P.n_envs;                         \
  if (!valid_) env_ = P.n_envs - 1; /* idle quads shadow the last robot; their stores are masked */
#endif
// P / S of a kernel body: the by-value arguments named in the kernarg segment (lanes_hip*.hpp: their fields are read where they are used,
// with scalar loads, instead of all at the kernel's entry) -- or, A/B switch of tools/build_variants.py, the arguments themselves
#ifndef IRRL_NO_PARAMS_KERNARG
#define IRRL_BIND_ARGS                                         \
  const EnvParams &P = irrl_kernarg<EnvParams>(0);             \
  const EnvState &S = irrl_kernarg<EnvState>((unsigned)((sizeof(EnvParams) + alignof(EnvState) - 1) / alignof(EnvState) * alignof(EnvState)));
#define IRRL_PARAMS_REFRESH(P) irrl_refresh(P)          /* once per step of the multi-step kernels */
// the rollout kernels' PolicyStepArgs (behind P, S and NPTR pointers): A0 names it, the multi-step kernels take a step's own copy from it per step
#define IRRL_KERNARG_ALIGN(off, T) (((off) + alignof(T) - 1) / alignof(T) * alignof(T))
#define IRRL_BIND_POLICY_ARGS_N(A0, A_, NPTR)                                                                                      \
  const PolicyStepArgs &A0 = irrl_kernarg<PolicyStepArgs>((unsigned)IRRL_KERNARG_ALIGN(                                            \
      IRRL_KERNARG_ALIGN(sizeof(EnvParams), EnvState) + sizeof(EnvState) + (NPTR) * sizeof(void *), PolicyStepArgs));
#define IRRL_BIND_POLICY_ARGS(A0, A_) IRRL_BIND_POLICY_ARGS_N(A0, A_, 4)
#else
#define IRRL_BIND_ARGS const EnvParams &P = P_; const EnvState &S = S_;
#define IRRL_PARAMS_REFRESH(P) (P)
#define IRRL_BIND_POLICY_ARGS_N(A0, A_, NPTR) const PolicyStepArgs &A0 = A_;
#define IRRL_BIND_POLICY_ARGS(A0, A_) IRRL_BIND_POLICY_ARGS_N(A0, A_, 4)
#endif
#define IRRL_LANE_PROLOGUE IRRL_LANE_PROLOGUE_B(irrl_xcd_block())          /* the stand-alone lane kernels */
#define IRRL_LANE_PROLOGUE_IDENTITY IRRL_LANE_PROLOGUE_B((int)blockIdx.x)  /* kernels whose policy part addresses robots by blockIdx */

// one wave per SIMD and its whole register file -- or (_l4w2, above) two waves per SIMD at 256 registers each
#if defined(IRRL_L4_WAVES2) && IRRL_LANES_PER_ROBOT == 4
#define IRRL_ENV_BOUNDS __launch_bounds__(256, 2)
#else
#define IRRL_ENV_BOUNDS __launch_bounds__(256, 1)
#endif

extern "C" {

// The step kernel exists once per (Crutial, per-contact rule): the launcher picks by EnvParams::crutial / ::contact_rule.  Suffix
// _md = the published maximum-dissipation rule of RaiSim's solver (ContactSolver bit 0, the shipped default), none = the build's
// first sliding rule.  (One kernel deciding by a run-time flag cost the OTHER rule's path 6 us of a 37 us step: both rules'
// per-substep constants were live across the sweep loop and the allocator paid for them in AGPR copies.)
#define IRRL_STEP_KERNEL(NAME, NS, RULE)                                                                                           \
  __global__ void IRRL_ENV_BOUNDS                                                                                        \
  IRRL_K(NAME)(EnvParams P_, EnvState S_, const float *action, float *ob, float *reward, uint8_t *done, float *extra) {             \
    IRRL_BIND_ARGS                                                                                                                 \
    IRRL_LANE_PROLOGUE                                                                                                             \
    NS::step_body<RULE>(P, S, env_, leg_, valid_, action, ob, reward, done, extra);                                                \
  }
IRRL_STEP_KERNEL(irrl_step_kernel_crutial, irrl, 0)
IRRL_STEP_KERNEL(irrl_step_kernel_crutial_md, irrl, 1)
IRRL_STEP_KERNEL(irrl_step_kernel_dir, irrl_plain, 0)
IRRL_STEP_KERNEL(irrl_step_kernel_flat, irrl_plain, IRRL_RULE_SHIPPED_FLAT)   // the default pool on flat ground (env_core.hpp IRRL_FLAT_GROUND)
IRRL_STEP_KERNEL(irrl_step_kernel_md, irrl_plain, 1)   // the published rule under the other solver settings (Gauss-Seidel order / confirming exit)

// the default pool: no meteorite, published rule
__global__ void IRRL_ENV_BOUNDS
IRRL_K(irrl_step_kernel)(EnvParams P_, EnvState S_, const float *action, float *ob, float *reward, uint8_t *done, float *extra) {
  IRRL_BIND_ARGS
  IRRL_LANE_PROLOGUE
#ifdef IRRL_PROFILE_WAVES   /* diagnostic build (tools/wave_spread.py): extra[env][5] <- this wave's duration in 100 MHz ticks */
  const unsigned long long t0_ = wall_clock64();
#endif
  irrl_plain::step_body<IRRL_RULE_SHIPPED>(P, S, env_, leg_, valid_, action, ob, reward, done, extra);
#ifdef IRRL_PROFILE_WAVES
  const unsigned long long t1_ = wall_clock64();
  if (valid_ && leg_ == 0) extra[env_ * 6 + 5] = (float)(t1_ - t0_);
#endif
}

#if IRRL_LANES_PER_ROBOT == 16
// ONE ROLLOUT STEP IN ONE LAUNCH: env.step of 16 robots (the workgroup's four waves, four robots each: the step kernel's body
// unchanged) and, behind a workgroup barrier, the LSTM policy's step on the observations those 16 robots just produced (one
// MFMA M-tile; policy_step.hpp with two virtual waves per wave).  Published contact rule only (the launcher falls back otherwise).  `action` is what the previous launch's policy part wrote
// for these robots (a.clipped), `ob` / `done` / `reward` are a.obs / a.dones / a.prev_reward: nothing a workgroup touches
// belongs to another workgroup, so the only synchronisation is the barrier.  Against two launches per step this removes a
// launch boundary and hides the layer-0 weight fetch -- and still MEASURES SLOWER (62.9 against 58.2 us per step at 4096 envs,
// tools/rollout_phases.py): the launch ends with its slowest workgroup, which pays the whole policy part behind its slowest
// robot, and four waves (three of them with two virtual waves of MFMA work each) take 18.5 us for what the stand-alone kernel's
// six waves do in 14.1.  Bit-identical results; an OPTION of irrl_lstm_rollout (fuse = 1), not the default.  (HID 48, ob 35.)
__global__ void IRRL_ENV_BOUNDS
irrl_step_policy_kernel_l16(EnvParams P_, EnvState S_, const float *action, float *ob, float *reward, uint8_t *done, float *extra, PolicyStepArgs a_) {
  IRRL_BIND_ARGS
  IRRL_BIND_POLICY_ARGS_N(a, a_, 5)
  __shared__ float hbuf[2][16 * 49];
  __shared__ float terms[16][17];
  __shared__ float head_w[48 * 17];
  __shared__ __attribute__((aligned(1024))) float lds_w[PolicyLdsImage<48>::FLOATS];   // 126 KiB: wh0 | wx0 of the actor and the critic stack
#ifdef IRRL_PROFILE_POLICY
  const unsigned long long pt0_ = wall_clock64();
#endif
  {
    IRRL_LANE_PROLOGUE_IDENTITY
    // the env part keeps no LDS and, between its prologue and its epilogue, issues no global load (flat ground): the layer-0
    // policy weights travel L2 -> LDS underneath the eight substeps
    irrl_plain::step_body<IRRL_RULE_SHIPPED>(P, S, env_, leg_, valid_, action, ob, reward, done, extra, [&]() { policy_prefetch_lds<48, 256>(a, lds_w); });
  }
#ifdef IRRL_PROFILE_POLICY
  const unsigned long long pt1_ = wall_clock64();
#else
  const unsigned long long pt0_ = 0, pt1_ = 0;
#endif
  __syncthreads();   // the workgroup's stores of obs / dones / reward are complete and visible to its own loads, the LDS image has landed
  policy_step_body<48, 9, 2, 256, true>(a, (int)blockIdx.x * 16, hbuf, terms, head_w, lds_w, pt0_, pt1_);
}
#endif

#if IRRL_LANES_PER_ROBOT == 16
// THE WHOLE ROLLOUT IN ONE LAUNCH (persistent): a workgroup owns 16 robots -- its four env waves, one MFMA M-tile of the policy --
// for all `steps` control steps: policy step k -> barrier -> env.step k -> barrier -> policy step k + 1 ...  Robots never interact
// (VEC:273) and the policy is per-robot, so NOTHING crosses workgroups: there is no grid-wide boundary between steps, a step costs
// its workgroup's own time instead of the slowest of 1024 waves (mean 35.8 us against 43.4 us for the step kernel at 4096 envs,
// tools/wave_spread.py), the 2 x steps launch boundaries are gone, and the layer-0 weights are fetched into LDS ONCE.  The code of a
// step is the fused kernel's above (same device functions, same order): obs / dones / states / clipped actions / rollout rows are
// bit-identical to the two-launch sequence.  Within a workgroup every global array is written and re-read by the same CU: the
// vector L1 is coherent at workgroup scope (non-tgsplit), the barriers' waits on the memory counters order the accesses.
__global__ void IRRL_ENV_BOUNDS
irrl_rollout_persistent_kernel_l16(EnvParams P_, EnvState S_, float *ob, float *reward, uint8_t *done, float *extra, PolicyStepArgs a_, int steps) {
  IRRL_BIND_ARGS
  IRRL_BIND_POLICY_ARGS(a, a_)
  __shared__ float hbuf[2][16 * 49];
  __shared__ float terms[16][17];
  __shared__ float head_w[48 * 17];
  __shared__ __attribute__((aligned(1024))) float lds_w[PolicyLdsImage<48>::FLOATS];   // 126 KiB: wh0 | wx0 of the actor and the critic stack
  policy_prefetch_lds<48, 256>(a, lds_w);
  const float *states_first = a.states_in, *noise0 = a.noise;
  const long long row0 = a.row, rng0 = a.rng_step;
  const size_t noise_stride = (size_t)a.N * (size_t)a.act_dim;
  __syncthreads();   // the LDS image has landed
#ifdef IRRL_PROFILE_PERSIST   /* diagnostic build (tools/persistent_phases.py): where a step goes, per wave, summed over the steps */
  unsigned long long ph_[4] = {0, 0, 0, 0}, pts_ = wall_clock64();
#define IRRL_PP_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = wall_clock64(); ph_[i] += n_ - pts_; pts_ = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define IRRL_PP_STAMP(i) do { } while (0)
#endif
  for (int k = 0; k < steps; k++) {
    // threadIdx.x made opaque once per iteration: every per-lane address below is then computed inside the loop (left to the
    // optimizer, the loop-invariant addresses of both parts -- hundreds of 64-bit values -- are hoisted and spilled)
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    PolicyStepArgs ak = IRRL_PARAMS_REFRESH(a);
    ak.row = row0 + k; ak.rng_step = rng0 + k;
    ak.noise = noise0 ? noise0 + (size_t)k * noise_stride : nullptr;
    ak.states_in = k == 0 ? states_first : a.states_out;
    policy_step_body<48, 9, 2, 256, true>(ak, (int)blockIdx.x * 16, hbuf, terms, head_w, lds_w, 0, 0, tid);
    IRRL_PP_STAMP(0);   // policy step
    __syncthreads();   // this workgroup's clipped actions (and the rollout rows) are stored and visible to its own loads
    IRRL_PP_STAMP(1);   // barrier behind the policy step
    {
      const int lane_ = tid & 63;
      const int wave_ = (int)blockIdx.x * 4 + (tid >> 6);
      int env_ = wave_ * 4 + (lane_ >> 4);
      const int leg_ = (lane_ >> 2) & 3;
      const bool valid_ = (env_ < P.n_envs) && ((lane_ & 3) == 0);
      if (env_ >= P.n_envs) env_ = P.n_envs - 1;
      irrl_plain::step_body<IRRL_RULE_SHIPPED>(IRRL_PARAMS_REFRESH(P), IRRL_PARAMS_REFRESH(S), env_, leg_, valid_, (const float *)ak.clipped, ob, reward, done, extra);
    }
    IRRL_PP_STAMP(2);   // env step of this wave's four robots
    __syncthreads();   // obs / dones / reward of step k are stored and visible: the next policy step reads them
    IRRL_PP_STAMP(3);   // barrier behind the env step: waiting for the workgroup's slowest wave
  }
#ifdef IRRL_PROFILE_PERSIST
  if ((threadIdx.x & 63u) == 0u) {
    const int wv = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (wv * 4 < P.n_envs) for (int i = 0; i < 4; i++) extra[(size_t)wv * 4 * 6 + i] = (float)ph_[i];
  }
#endif
}
#endif

#if IRRL_LANES_PER_ROBOT == 16
// THE LSTM ROLLOUT WITH THE CRITIC OFF THE PER-STEP PATH (round 5; irrl_lstm_rollout fuse = 3).  The value V(s_t) is a function of the
// observation history only -- nothing in the rollout depends on it until GAE -- so the per-step part runs the ACTOR stack alone and the
// caller evaluates the critic stack over the recorded [T, N, 35] observations afterwards with the update's sequence kernels (two launches
// for the whole rollout; ppo2.Runner).  What that buys per step: half the policy part's MFMAs and cells, and -- the LDS that held the critic's
// layer-0 operands now holds the actor's layer-1 operands -- NO weight fetched from L2 inside the step loop; the head weights are staged
// once.  Same device functions and per-element arithmetic as the full kernel above: actions, clipped actions, neglogp, observations,
// rewards, dones and the actor's LSTM state are bit-identical to every other rollout mode; `value` / `mb_values` are not written.
__global__ void IRRL_ENV_BOUNDS
irrl_rollout_persistent_actor_kernel_l16(EnvParams P_, EnvState S_, float *ob, float *reward, uint8_t *done, float *extra, PolicyStepArgs a_, int steps) {
  IRRL_BIND_ARGS
  IRRL_BIND_POLICY_ARGS(a, a_)
  __shared__ float hbuf[2][16 * 49];
  __shared__ float terms[16][17];
  __shared__ float head_w[48 * 17];
  __shared__ __attribute__((aligned(1024))) float lds_w[PolicyLdsImage<48>::FLOATS];   // wh0 | wx0 | wh1 | wx1 of the ACTOR stack
  policy_prefetch_lds_actor<48, 256>(a, lds_w);
  for (int i = (int)threadIdx.x; i < 48 * a.act_dim; i += 256) head_w[i] = a.pi_w[i];
  const float *states_first = a.states_in, *noise0 = a.noise;
  const long long row0 = a.row, rng0 = a.rng_step;
  const size_t noise_stride = (size_t)a.N * (size_t)a.act_dim;
  // the env part's lane context stays in registers across the steps (irrl_steps_persistent_kernel below): with one virtual wave per wave the
  // policy part leaves room for it (379 registers, no scratch beyond the reset branch's)
  const int lane0_ = (int)(threadIdx.x & 63u);
  int env0_ = ((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * 4 + (lane0_ >> 4);
  const int leg0_ = (lane0_ >> 2) & 3;
  const bool valid0_ = (env0_ < P.n_envs) && ((lane0_ & 3) == 0);
  if (env0_ >= P.n_envs) env0_ = P.n_envs - 1;
#ifndef IRRL_ACTOR_NO_CARRY      /* A/B switch of tools/build_variants.py */
  irrl_plain::EnvLane L;
  irrl_plain::load_lane(P, S, env0_, leg0_, L, true);
#endif
  __syncthreads();   // the LDS image and the head weights have landed
  for (int k = 0; k < steps; k++) {
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));     // (see irrl_rollout_persistent_kernel_l16: keeps the per-lane addresses inside the loop)
    PolicyStepArgs ak = IRRL_PARAMS_REFRESH(a);
    ak.row = row0 + k; ak.rng_step = rng0 + k;
    ak.noise = noise0 ? noise0 + (size_t)k * noise_stride : nullptr;
    ak.states_in = k == 0 ? states_first : a.states_out;
    policy_step_body<48, 9, 1, 256, true, true>(ak, (int)blockIdx.x * 16, hbuf, terms, head_w, lds_w, 0, 0, tid);
    __syncthreads();   // this workgroup's clipped actions (and the rollout rows) are stored and visible to its own loads
    {
      int env_ = env0_;
      asm volatile("" : "+v"(env_));
#ifndef IRRL_ACTOR_NO_CARRY
      if (k > 0) irrl_plain::lane_carry(L);
      irrl_plain::step_compute<IRRL_RULE_SHIPPED>(IRRL_PARAMS_REFRESH(P), L, env_, leg0_, valid0_, irrl_plain::ActionRow{(const float *)ak.clipped}, ob, reward, done, extra);
#else
      irrl_plain::step_body<IRRL_RULE_SHIPPED>(IRRL_PARAMS_REFRESH(P), IRRL_PARAMS_REFRESH(S), env_, leg0_, valid0_, (const float *)ak.clipped, ob, reward, done, extra);
#endif
    }
    __syncthreads();   // obs / dones / reward of step k are stored and visible: the next policy step reads them
  }
#ifndef IRRL_ACTOR_NO_CARRY
  if (steps > 0) {
    IRRL_SUB0_ONLY_BEGIN
    irrl_plain::store_lane(IRRL_PARAMS_REFRESH(P), IRRL_PARAMS_REFRESH(S), env0_, leg0_, valid0_, L, P.randomize_per_episode != 0);
    IRRL_SUB0_ONLY_END
  }
#endif
}
#endif

#if IRRL_LANES_PER_ROBOT == 16
// THE ACTOR-ONLY ROLLOUT WITH THE POLICY AS EACH WAVE'S OWN WORK (round 5, second half; irrl_lstm_rollout fuse = 3, the default form of it).
// The kernel above still runs the actor for a workgroup's 16 robots on 16-row MFMA tiles: three barriers per step and, in every step, the
// wait for the slowest of the workgroup's four env waves.  Here a wave runs the actor stack for ITS four robots (lstm_actor_wave_body,
// policy_step.hpp: v_mfma_f32_4x4x1, operands out of a transposed LDS image), keeps h of both layers in its LDS scratch and c in registers for
// the whole rollout, and hands observations / reward / done flag / clipped actions between its env step and its policy step through that
// scratch next to the stores to memory -- no workgroup barrier and no load behind a store inside the step loop.  Actions, neglogp,
// observations, rewards, dones and the actor's final LSTM state are bit-identical to every other rollout mode.
__global__ void IRRL_ENV_BOUNDS
irrl_rollout_persistent_actor_wave_kernel_l16(EnvParams P_, EnvState S_, float *ob, float *reward, uint8_t *done, float *extra, PolicyStepArgs a_, int steps) {
  IRRL_BIND_ARGS
  IRRL_BIND_POLICY_ARGS(a, a_)
  constexpr int HID = 48, NG = HID / 16, SD = 8 * HID;
  typedef LstmWaveLds<HID> LAY;
  __shared__ __attribute__((aligned(16))) float wsl[4][LAY::FLOATS];
  __shared__ float head_w[HID * 16];
  __shared__ __attribute__((aligned(16))) float lds_w[LstmWaveImage<HID>::FLOATS];      // the ACTOR's operands, [gate column][K] (policy_step.hpp)
  lstm_wave_image_stage<HID, 256>(a, lds_w);
  for (int i = (int)threadIdx.x; i < HID * a.act_dim; i += 256) head_w[i] = a.pi_w[i];
  const float *noise0 = a.noise;
  const long long row0 = a.row, rng0 = a.rng_step;
  const size_t noise_stride = (size_t)a.N * (size_t)a.act_dim;
  const int lane0_ = (int)(threadIdx.x & 63u);
  const int wave_ = (int)(threadIdx.x >> 6);
  const int e4_ = ((int)blockIdx.x * 4 + wave_) * 4;            // the wave's first robot
  const int rl_ = lane0_ >> 4;                                   // the robot this lane integrates (env part)
  int env0_ = e4_ + rl_;
  const int leg0_ = (lane0_ >> 2) & 3;
  const bool valid0_ = (env0_ < P.n_envs) && ((lane0_ & 3) == 0);
  if (env0_ >= P.n_envs) env0_ = P.n_envs - 1;
  irrl_plain::EnvLane L;
  irrl_plain::load_lane(P, S, env0_, leg0_, L, true);
  float *ws = wsl[wave_];
  // policy part: this lane's robot is l & 3, its unit inside a column group l >> 2
  const int pr_ = lane0_ & 3, pq_ = lane0_ >> 2;
  const bool pok_ = e4_ + pr_ < a.N;
  const int pe_ = pok_ ? e4_ + pr_ : a.N - 1;
  float cst[2][NG], bias[2][NG];
#pragma unroll
  for (int G = 0; G < NG; G++) {
    cst[0][G] = a.states_in[(size_t)pe_ * SD + 16 * G + pq_];
    cst[1][G] = a.states_in[(size_t)pe_ * SD + 2 * HID + 16 * G + pq_];
    bias[0][G] = a.w[2][64 * G + lane0_];
    bias[1][G] = a.w[5][64 * G + lane0_];
  }
  {   // the state of things in front of step 0, from memory: observations, done flags, the last reward, h of both layers
    const int n = ((a.N - e4_ < 4) ? a.N - e4_ : 4);
    for (int i = lane0_; i < 4 * 35; i += 64) ws[LAY::X + i] = (i < n * 35) ? a.obs[(size_t)e4_ * 35 + i] : 0.0f;
    for (int i = lane0_; i < 4 * HID; i += 64) {
      const int r = i / HID, k = i - r * HID;
      const int e = (e4_ + r < a.N) ? e4_ + r : a.N - 1;
      ws[LAY::H0 + i] = a.states_in[(size_t)e * SD + HID + k];
      ws[LAY::H1 + i] = a.states_in[(size_t)e * SD + 3 * HID + k];
    }
    if (lane0_ < 4) {
      const int e = (e4_ + lane0_ < a.N) ? e4_ + lane0_ : a.N - 1;
      ws[LAY::DON + lane0_] = a.dones[e] ? 1.0f : 0.0f;
      ws[LAY::REW + lane0_] = a.prev_reward ? a.prev_reward[e] : 0.0f;
    }
  }
  __syncthreads();   // the LDS image and the head weights have landed
  for (int k = 0; k < steps; k++) {
    int lane = lane0_;
    asm volatile("" : "+v"(lane));
    PolicyStepArgs ak = IRRL_PARAMS_REFRESH(a);
    ak.row = row0 + k; ak.rng_step = rng0 + k;
    ak.noise = noise0 ? noise0 + (size_t)k * noise_stride : nullptr;
    lstm_actor_wave_body<HID>(ak, e4_, ws, lds_w, head_w, lane, cst, bias);
    PS_WAVE_SYNC();    // this wave's clipped actions are in its scratch
    {
      int env_ = env0_;
      asm volatile("" : "+v"(env_));
      if (k > 0) irrl_plain::lane_carry(L);
      irrl_plain::ActionRegs act;
#pragma unroll
      for (int j = 0; j < 3; j++) act.a[j] = ws[LAY::ACT + rl_ * 12 + leg0_ * 3 + j];
      irrl_plain::step_compute<IRRL_RULE_SHIPPED, irrl_plain::NoStepHook>(
          IRRL_PARAMS_REFRESH(P), L, env_, leg0_, valid0_, act, ob, reward, done, extra, irrl_plain::NoStepHook(),
          [&](const irrl_plain::EnvLane &Lf, float rew, bool dn) {
            irrl_plain::observe_write(P, rl_, leg0_, valid0_, Lf, ws + LAY::X);
            if (valid0_ && leg0_ == 0) { ws[LAY::REW + rl_] = rew; ws[LAY::DON + rl_] = dn ? 1.0f : 0.0f; }
          });
    }
    PS_WAVE_SYNC();    // observations / done flags / rewards of step k are in the scratch
  }
  if (steps > 0) {
    IRRL_SUB0_ONLY_BEGIN
    irrl_plain::store_lane(IRRL_PARAMS_REFRESH(P), IRRL_PARAMS_REFRESH(S), env0_, leg0_, valid0_, L, P.randomize_per_episode != 0);
    IRRL_SUB0_ONLY_END
    if (pok_) {      // the actor's LSTM state behind the last step (the critic's half is the caller's: ppo2.Runner._critic_pass)
#pragma unroll
      for (int G = 0; G < NG; G++) {
        const int u = 16 * G + pq_;
        a.states_out[(size_t)pe_ * SD + u] = cst[0][G];
        a.states_out[(size_t)pe_ * SD + HID + u] = ws[LAY::H0 + pr_ * HID + u];
        a.states_out[(size_t)pe_ * SD + 2 * HID + u] = cst[1][G];
        a.states_out[(size_t)pe_ * SD + 3 * HID + u] = ws[LAY::H1 + pr_ * HID + u];
      }
    }
  }
}
#endif

#if IRRL_LANES_PER_ROBOT == 16
// THE SAME FOR MlpPolicy (BASELINE config 2's learner): the whole rollout in one launch.  The policy's 52 KB of weights and biases are copied
// to LDS ONCE (transposed: mlp_policy_stage_lds); a step of the policy part is then two MFMA chains on LDS operands + the heads (a few us
// against 8.7 us for the stand-alone launch, whose life is launch + weight fetch), and a step costs a WAVE its own time instead of the slowest
// of the 1024 env waves: since the second half of round 5 the policy of a wave's four robots is that wave's own work (see inside).
// Same device functions, same order: the buffers are bit-identical to the two-launch sequence.
__global__ void IRRL_ENV_BOUNDS
irrl_rollout_persistent_mlp_kernel_l16(EnvParams P_, EnvState S_, float *ob, float *reward, uint8_t *done, float *extra, PolicyStepArgs a_, int steps) {
  IRRL_BIND_ARGS
  IRRL_BIND_POLICY_ARGS(a, a_)
  __shared__ __attribute__((aligned(16))) float wsl[4][MlpWaveLds<64>::FLOATS];      // per wave: its four robots' scratch (policy_step.hpp)
  __shared__ float head_w[64 * 17];
  __shared__ __attribute__((aligned(16))) float wl[MlpLdsImage<64>::FLOATS];
  mlp_policy_stage_lds<64>(a, wl, head_w, (int)threadIdx.x, 256);
  const float *noise0 = a.noise;
  const long long row0 = a.row, rng0 = a.rng_step;
  const size_t noise_stride = (size_t)a.N * (size_t)a.act_dim;
  // the env part's lane context stays in registers across the steps (round 5; irrl_steps_persistent_kernel below): this policy's step needs
  // few registers (its weights and activations live in LDS), so the context survives it without spilling
  const int lane0_ = (int)(threadIdx.x & 63u);
  const int wave_ = (int)(threadIdx.x >> 6);
  const int e4_ = ((int)blockIdx.x * 4 + wave_) * 4;            // the wave's first robot
  const int rl_ = lane0_ >> 4;                                   // this lane's robot inside the wave
  int env0_ = e4_ + rl_;
  const int leg0_ = (lane0_ >> 2) & 3;
  const bool valid0_ = (env0_ < P.n_envs) && ((lane0_ & 3) == 0);
  if (env0_ >= P.n_envs) env0_ = P.n_envs - 1;
  irrl_plain::EnvLane L;
  irrl_plain::load_lane(P, S, env0_, leg0_, L, true);
  // Round 5: the policy of a wave's four robots is that wave's own work (mlp_policy_wave_body): no workgroup barrier in the step loop, no wait
  // for the slowest of the four env waves in every step -- and what a wave hands from its env step to its policy step and back (observations,
  // reward, done flag; clipped actions) goes through its LDS scratch next to the stores to memory, so no load inside the loop waits for a store.
  float *ws = wsl[wave_];
  {   // the state of things in front of step 0, from memory: observations, done flags, the last reward
    const int n = ((a.N - e4_ < 4) ? a.N - e4_ : 4);
    for (int i = lane0_; i < 4 * 35; i += 64) ws[MlpWaveLds<64>::X + i] = (i < n * 35) ? a.obs[(size_t)e4_ * 35 + i] : 0.0f;
    if (lane0_ < 4) {
      const int e = (e4_ + lane0_ < a.N) ? e4_ + lane0_ : a.N - 1;
      ws[MlpWaveLds<64>::DON + lane0_] = a.dones[e] ? 1.0f : 0.0f;
      ws[MlpWaveLds<64>::REW + lane0_] = a.prev_reward ? a.prev_reward[e] : 0.0f;
    }
  }
  __syncthreads();
  for (int k = 0; k < steps; k++) {
    int lane = lane0_;
    asm volatile("" : "+v"(lane));     // (see irrl_rollout_persistent_kernel_l16: keeps the per-lane addresses inside the loop)
    PolicyStepArgs ak = IRRL_PARAMS_REFRESH(a);
    ak.row = row0 + k; ak.rng_step = rng0 + k;
    ak.noise = noise0 ? noise0 + (size_t)k * noise_stride : nullptr;
    mlp_policy_wave_body<64, true, true>(ak, e4_, ws, wl, head_w, lane);
    PS_WAVE_SYNC();    // this wave's clipped actions are in its scratch
    {
      int env_ = env0_;
      asm volatile("" : "+v"(env_));
      if (k > 0) irrl_plain::lane_carry(L);
      irrl_plain::ActionRegs act;
#pragma unroll
      for (int j = 0; j < 3; j++) act.a[j] = ws[MlpWaveLds<64>::ACT + rl_ * 12 + leg0_ * 3 + j];
      irrl_plain::step_compute<IRRL_RULE_SHIPPED, irrl_plain::NoStepHook>(
          IRRL_PARAMS_REFRESH(P), L, env_, leg0_, valid0_, act, ob, reward, done, extra, irrl_plain::NoStepHook(),
          [&](const irrl_plain::EnvLane &Lf, float rew, bool dn) {
            // (inside the epilogue's sub-lane-0 region) the scaled observation row, the reward and the done flag once more, into the scratch
            irrl_plain::observe_write(P, rl_, leg0_, valid0_, Lf, ws + MlpWaveLds<64>::X);
            if (valid0_ && leg0_ == 0) { ws[MlpWaveLds<64>::REW + rl_] = rew; ws[MlpWaveLds<64>::DON + rl_] = dn ? 1.0f : 0.0f; }
          });
    }
    PS_WAVE_SYNC();    // observations / done flags / rewards of step k are in the scratch: the wave's next policy step reads them
  }
  if (steps > 0) {
    IRRL_SUB0_ONLY_BEGIN
    irrl_plain::store_lane(IRRL_PARAMS_REFRESH(P), IRRL_PARAMS_REFRESH(S), env0_, leg0_, valid0_, L, P.randomize_per_episode != 0);
    IRRL_SUB0_ONLY_END
  }
}
#endif

// `count` CONSECUTIVE env.step()s IN ONE LAUNCH (irrl_env_step_rows_persistent[_out]): step k takes action row (first_row + k) % n_rows of a table
// resident in HBM.  Robots never interact (VEC:273), so a wave simply walks its own robots through the `count` steps: no grid-wide
// boundary between steps -- a step costs a wave its own time instead of the slowest of the 1024 waves, and the launch boundaries are gone.
// Round 5: the robots' lane context STAYS IN REGISTERS from step to step (load_lane once in front of the loop, store_lane once behind it;
// lane_carry() between two steps hands the next step exactly the words a store + load would have: env_core.hpp) -- a step no longer starts
// behind a round trip of ~90 stores and ~90 loads per lane through the L2.  The body of a step is the step kernel's (step_compute, same
// order): states and outputs are bit-identical to `count` launches.  Default pool kind only (no meteorite, published rule); the launcher
// falls back otherwise.
}  // extern "C"
template <int RULE>
__device__ __forceinline__ void irrl_steps_persistent_body(const EnvParams &P_, const EnvState &S_, const float *action_rows, int n_rows, int first_row, int count,
                                                           float *ob, float *reward, uint8_t *done, float *extra, int out_rows) {
  (void)P_; (void)S_;   // (IRRL_BIND_ARGS names the kernel's first two arguments in the kernarg segment; the by-value copies serve the A/B build only)
  IRRL_BIND_ARGS
  const int blk = irrl_xcd_block();
  const size_t row = (size_t)P.n_envs * 12;
  // out_rows != 0: the outputs are [count, N, .] tables and step k fills row k -- the trajectory `count` step() calls of the reference
  // would have returned (VEC:268-278, RaisimGymVecEnv.py:26-52); 0: [N, .] arrays every step overwrites (the last step's survive)
  const size_t orow = out_rows ? (size_t)P.n_envs : (size_t)0;
  const int lane0_ = (int)(threadIdx.x & 63u);
#if IRRL_LANES_PER_ROBOT == 16
  int env0_ = (blk * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6)) * 4 + (lane0_ >> 4);
  const int leg0_ = (lane0_ >> 2) & 3;
  const bool valid0_ = (env0_ < P.n_envs) && ((lane0_ & 3) == 0);
  if (env0_ >= P.n_envs) env0_ = P.n_envs - 1;
#else
  int env0_ = (blk * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6)) * 16 + (lane0_ >> 2);
  const int leg0_ = lane0_ & 3;
  const bool valid0_ = env0_ < P.n_envs;
  if (!valid0_) env0_ = P.n_envs - 1;
#endif
  irrl_plain::EnvLane L;
  irrl_plain::load_lane(P, S, env0_, leg0_, L, true);
  // the action row of step k + 1 is requested while step k runs (three words per lane): a step does not start behind that round trip either
  irrl_plain::ActionRegs act_next;
  {
    const float *a0 = action_rows + row * (size_t)(first_row % n_rows) + (size_t)env0_ * 12 + leg0_ * 3;
    act_next.a[0] = a0[0]; act_next.a[1] = a0[1]; act_next.a[2] = a0[2];
  }
  for (int k = 0; k < count; k++) {
    // the lane's robot made opaque once per iteration: the per-lane addresses of the action row and of the output rows are then computed
    // inside the loop (hoisted, they are dozens of 64-bit values that spill)
    int env_ = env0_;
    asm volatile("" : "+v"(env_));
    if (k > 0) irrl_plain::lane_carry(L);
    const irrl_plain::ActionRegs act = act_next;
    {
      const int kn = (k + 1 < count) ? k + 1 : k;      // (behind the last step: that step's own row once more)
      const float *an = action_rows + row * (size_t)((first_row + kn) % n_rows) + (size_t)env_ * 12 + leg0_ * 3;
      act_next.a[0] = an[0]; act_next.a[1] = an[1]; act_next.a[2] = an[2];
    }
    irrl_plain::step_compute<RULE, irrl_plain::NoStepHook, irrl_plain::NoStepTail, irrl_plain::ActionRegs>(
        IRRL_PARAMS_REFRESH(P), L, env_, leg0_, valid0_, act, ob + orow * 35 * (size_t)k, reward + orow * (size_t)k, done + orow * (size_t)k, extra + orow * 6 * (size_t)k);
  }
  if (count > 0) {
    IRRL_SUB0_ONLY_BEGIN
    irrl_plain::store_lane(IRRL_PARAMS_REFRESH(P), IRRL_PARAMS_REFRESH(S), env0_, leg0_, valid0_, L, P.randomize_per_episode != 0);
    IRRL_SUB0_ONLY_END
  }
}
extern "C" {
// the multi-step kernel once with the run-time terrain test (rough ground) and once with flat ground compiled in (the launcher picks)
__global__ void IRRL_ENV_BOUNDS
IRRL_K(irrl_steps_persistent_kernel)(EnvParams P_, EnvState S_, const float *action_rows, int n_rows, int first_row, int count, float *ob, float *reward,
                                     uint8_t *done, float *extra, int out_rows) {
  irrl_steps_persistent_body<IRRL_RULE_SHIPPED>(P_, S_, action_rows, n_rows, first_row, count, ob, reward, done, extra, out_rows);
}
__global__ void IRRL_ENV_BOUNDS
IRRL_K(irrl_steps_persistent_kernel_flat)(EnvParams P_, EnvState S_, const float *action_rows, int n_rows, int first_row, int count, float *ob, float *reward,
                                          uint8_t *done, float *extra, int out_rows) {
  irrl_steps_persistent_body<IRRL_RULE_SHIPPED_FLAT>(P_, S_, action_rows, n_rows, first_row, count, ob, reward, done, extra, out_rows);
}

__global__ void IRRL_ENV_BOUNDS IRRL_K(irrl_init_kernel)(EnvParams P_, EnvState S_) {
  IRRL_BIND_ARGS
  IRRL_LANE_PROLOGUE
  irrl::init_body(P, S, env_, leg_, valid_);
}

__global__ void IRRL_ENV_BOUNDS IRRL_K(irrl_reset_kernel)(EnvParams P_, EnvState S_, float *ob) {
  IRRL_BIND_ARGS
  IRRL_LANE_PROLOGUE
  irrl::reset_body(P, S, env_, leg_, valid_, ob);
}

__global__ void IRRL_ENV_BOUNDS IRRL_K(irrl_observe_kernel)(EnvParams P_, EnvState S_, float *ob) {
  IRRL_BIND_ARGS
  IRRL_LANE_PROLOGUE
  irrl::observe_body(P, S, env_, leg_, valid_, ob);
}

__global__ void IRRL_ENV_BOUNDS IRRL_K(irrl_probe_kernel)(EnvParams P_, EnvState S_, float *minv, float *nonlin) {
  IRRL_BIND_ARGS
  IRRL_LANE_PROLOGUE
  irrl::dynamics_probe_body(P, S, env_, leg_, valid_, minv, nonlin);
}

#if IRRL_LANES_PER_ROBOT == 16
// isTerminalState (ENV:1553-1578) on the stored state; one thread per robot (layout independent: emitted once)
__global__ void irrl_terminal_kernel(EnvParams P_, EnvState S_, uint8_t *done) {
  IRRL_BIND_ARGS
  int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (e >= P.n_envs) return;
  float z = S.gc[e * 19 + 2], up = S.ob[e * 35 + 31];
  done[e] = (z < 0.15f || z > 0.65f || up < 0.5f) ? 1 : 0;
}
#endif

}  // extern "C"
