#!/usr/bin/env python3
"""Where a step of the PERSISTENT rollout kernel goes (irrl_rollout_persistent_kernel_l16), per wave, averaged over the steps.
Needs the diagnostic build (tools/build_variants.py persist=-DIRRL_PROFILE_PERSIST), which leaves every wave's four phase sums
(100 MHz ticks) in extraInfo:   IRRL_ENV_LIB=.../libirrl_env_persist.so python tools/persistent_phases.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, yaml
import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy
from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2, Runner
from high_speed_quadrupedal_locomotion_by_irrl_amd.vec_env import TorchVecEnv
n, T = 4096, 750
cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, "default_cfg.yaml")))["environment"]
cfg["num_envs"] = n
env = TorchVecEnv(FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg)))
model = PPO2(policy=CustomLSTMPolicy, env=env, n_steps=T, nminibatches=1, noptepochs=1, seed=1)
r = Runner(env, model, T, 0.99, 0.998)
r.rollout_one_launch_per_step = 2
for _ in range(2):
    r.run()
    torch.cuda.synchronize()
ph = env.extra.reshape(n // 4, 4, 6)[:, 0, :4].cpu().numpy() * 0.01 / T       # [waves, 4 phases] us per step
names = ["policy step", "barrier behind it", "env step (this wave)", "barrier behind it (slowest wave of the workgroup)"]
out = {"waves": int(ph.shape[0]), "steps": T, "us_per_step": {k: {"mean": float(ph[:, i].mean()), "min": float(ph[:, i].min()), "max": float(ph[:, i].max())} for i, k in enumerate(names)},
       "sum_of_means": float(ph.mean(0).sum())}
print(json.dumps(out))
