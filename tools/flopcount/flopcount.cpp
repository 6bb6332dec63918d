// flopcount.cpp -- exact algorithmic operation count of one env-step, emitted by the instrumented oracle
// ("replace the estimate with an exact operation count emitted by the CPU restatement").
// Build + run: see tools/flopcount/run.sh.  The count is for the ORACLE's dense formulation (13-body CRBA/RNEA,
// dense 18x18 Cholesky, dense Delassus); the kernels' Schur-complement formulation does less arithmetic, so using
// this number as "algorithmic flops" is conservative in the kernels' favour only if quoted as such -- DESIGN.md
// section 6 quotes both.
#include "count_real.hpp"
thread_local uint64_t CountReal::n = 0;
#define ORC_REAL CountReal
#include "../../oracle/irrl_oracle.c"

#include <cstdio>
#include <vector>

int main(int argc, char **argv) {
  orc_cfg c;
  memset(&c, 0, sizeof(c));
  c.num_envs = 64; c.num_threads = 1; c.simulation_dt = 0.00025; c.control_dt = 0.002; c.seedd = 1;
  c.abad = 0; c.period = 0.2; c.lam = 0.5; c.stand_height = 0.28; c.up_height = 0.08; c.down_height = 0; c.gait_step = 0.15;
  c.Vx = 5; c.Vy = 0; c.Omega = 1; c.LeanFront = 0; c.LeanHind = 0;
  c.ManualTraj = 1; c.WILDCAT = 1; c.StochasticDynamics = (argc > 1); c.terminalRewardCoeff = -1;
  c.BodyPosRewardCoeff = 0.2; c.BodyAttitudeRewardCoeff = 0.2; c.JointRewardCoeff = 0.4; c.VelRewardCoeff = 0.2; c.TorqueCoeff = 0.1; c.ContactCoeff = 0.1;
  c.Stiffness = 40; c.Stiffness_Low = 40; c.AbadRatio = 1; c.Damping = 1; c.Freq = 30; c.max_time = 1.5; c.CubeNum = 1; c.FPS = 60;
  c.ObsNoise = (argc > 1) ? 2.0 : 0.0; c.GaitType = 1; c.MotorMaxTorque = 18; c.MotorCriticalSpeed = 100; c.MotorMaxSpeed = 200;
  c.ContactIterations = 6; c.SharedNoiseScalar = 1; c.ContactTolerance = 1e-4;
  c.ContactSolver = getenv("IRRL_FLOPCOUNT_SOLVER") ? atoi(getenv("IRRL_FLOPCOUNT_SOLVER")) : 1;   /* 1 = the published method: Gauss-Seidel + the published per-contact rule */
  orc_env *h = orc_create(&c);
  orc_init(h);
  const int n = c.num_envs;
  std::vector<float> act(n * 12), ob(n * 35), rew(n), extra(n * 6);
  std::vector<uint8_t> done(n);
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 1.2f; };
  // 90 steps until the robots stand on the ground, then count 100 steps
  for (int k = 0; k < 90; k++) { for (auto &a : act) a = rnd(); orc_step(h, act.data(), ob.data(), rew.data(), done.data(), extra.data()); }
  CountReal::n = 0;
  const int K = 100;
  for (int k = 0; k < K; k++) { for (auto &a : act) a = rnd(); orc_step(h, act.data(), ob.data(), rew.data(), done.data(), extra.data()); }
  double per = (double)CountReal::n / ((double)K * n);
  printf("{\"contact_solver\": %d, \"flops_per_env_step\": %.1f, \"envs\": %d, \"steps\": %d, \"mean_contact_sweeps\": %.3f, \"convention\": \"add/sub/mul/div/sqrt/transcendental = 1\"}\n",
         c.ContactSolver, per, n, K, orc_mean_contact_sweeps(h));
  return 0;
}
