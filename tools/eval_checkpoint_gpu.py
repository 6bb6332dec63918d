"""Evaluate a checkpoint in TRAINING-mode envs on the GPU (same path as the rollout: fused policy step + env step), report
episode statistics with sampled and with deterministic actions.   usage: python tools/eval_checkpoint_gpu.py ckpt.pkl [envs] [cfg.yaml]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, yaml
import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
from high_speed_quadrupedal_locomotion_by_irrl_amd.vec_env import TorchVecEnv
from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2, Runner

ck = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, "default_cfg.yaml")
cfg = yaml.safe_load(open(cfg_path))["environment"]
cfg["num_envs"] = n
env = TorchVecEnv(FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg)))
model = PPO2.load(ck, env=env)
runner = Runner(env, model, 750, 0.99, 0.998)
for k in range(2):
    b = runner.run()
    r, l, c = env.pop_episode_stats()
    print("rollout %d (sampled actions, fused path): ep_reward_mean %.1f ep_len_mean %.1f episodes %d" % (k, r, l, c))
# deterministic, generic torch path, fresh LSTM state
pol = model.policy
obs = env.reset().clone()
states = pol.initial_state(n, env.device)
dones = torch.zeros(n, dtype=torch.bool, device=env.device)
ep = torch.zeros(n, device=env.device); lens = []
for t in range(750):
    a, _, states, _ = pol.step(obs, states, dones, deterministic=True)
    o, rwd, d = env.step(a.clamp(-1, 1))
    obs = o.clone(); dones = d.clone(); ep += 1
    if bool(d.any()):
        lens += ep[d].tolist(); ep[d] = 0
print("deterministic: %d episodes ended in 750 steps x %d envs, mean length of ended %.1f" % (len(lens), n, float(np.mean(lens)) if lens else float('nan')))
