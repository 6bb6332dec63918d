#!/usr/bin/env python3
"""Phase time stamps of ONE workgroup of a rollout step (diagnostic build: tools/build_variants.py pol=-DIRRL_PROFILE_POLICY; the
policy code then writes 100 MHz stamps over neglogp[N/2 : N/2 + 9] (the middle workgroup's own entries) of the last launch).  Runs short direct rollouts of the PPO runner.
    IRRL_ENV_LIB=.../libirrl_env_pol.so [IRRL_ROLLOUT_FUSED=1] python tools/rollout_phases.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, yaml
import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy
from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2, Runner
from high_speed_quadrupedal_locomotion_by_irrl_amd.vec_env import TorchVecEnv
cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, "default_cfg.yaml")))["environment"]
cfg["num_envs"] = 4096
env = TorchVecEnv(FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg)))
T = 60
model = PPO2(policy=CustomLSTMPolicy, env=env, n_steps=T, nminibatches=1, noptepochs=1, seed=1)
runner = Runner(env, model, T, 0.99, 0.998)
rows = []
for k in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    runner.run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rows.append(runner._out[3][n // 2:n // 2 + 9].cpu().numpy() * 0.01)
t = np.median(np.array(rows[3:]), axis=0)
names = ["start", "loads issued", "L0 + recurrent L1 MFMAs (loads landed)", "barrier", "L0 cell + L1 input MFMAs", "L1 cell", "heads / sample / rows"]
print("fused kernel: this wave's env part %.2f us, policy part entered at %.2f us" % (t[7], t[8]) if t[8] > 0 else "stand-alone policy kernel")
for n, a, b in zip(names[1:], t[:6], t[1:7]):
    print("%-45s %6.2f us (at %.2f)" % (n, b - a, b))
print("rollout of %d steps incl. GAE and reset: %.2f ms wall (%.1f us per step)" % (T, 1e3 * dt, 1e6 * dt / T))
