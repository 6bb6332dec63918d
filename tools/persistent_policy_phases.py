#!/usr/bin/env python3
"""Phases of the POLICY PART of one step of the persistent LSTM rollout kernel (one workgroup, the last step of a rollout; diagnostic build
tools/build_variants.py pol=-DIRRL_PROFILE_POLICY):   IRRL_ENV_LIB=.../libirrl_env_pol.so python tools/persistent_policy_phases.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, yaml
import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy
from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2, Runner
from high_speed_quadrupedal_locomotion_by_irrl_amd.vec_env import TorchVecEnv
n, T = 4096, 200
cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, "default_cfg.yaml")))["environment"]
cfg["num_envs"] = n
env = TorchVecEnv(FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg)))
model = PPO2(policy=CustomLSTMPolicy, env=env, n_steps=T, nminibatches=1, noptepochs=1, seed=1)
r = Runner(env, model, T, 0.99, 0.998)
names = ["start", "loads issued", "L0 + recurrent L1 MFMAs (loads landed)", "barrier", "L0 cell + L1 input MFMAs", "L1 cell", "heads / sample / rows"]
for mode in (2, 0):
    r.rollout_one_launch_per_step = mode
    rows = []
    for _ in range(4):
        r.run()
        torch.cuda.synchronize()
        rows.append(r._out[3][n // 2:n // 2 + 7].cpu().numpy() * 0.01)
    t = np.median(np.array(rows[1:]), axis=0)
    print("persistent rollout kernel (4 waves, two virtual waves each)" if mode == 2 else "stand-alone policy step kernel (6 waves)")
    for nm, a, b in zip(names[1:], t[:-1], t[1:]):
        print("   %-45s %6.2f us (at %.2f)" % (nm, b - a, b))
