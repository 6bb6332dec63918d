#!/usr/bin/env python3
"""us per env.step() of K steps issued as K launches (irrl_env_step_rows) and as ONE persistent launch (irrl_env_step_rows_persistent), same
pool state, HIP events on the launch stream:  python tools/persistent_rows_time.py [--envs 4096] [--cfg bp5_imitation.yaml]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml
import ctypes as C
import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--cfg", default="bp5_imitation.yaml")
a = ap.parse_args()
n = a.envs
cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, a.cfg)))["environment"]
cfg["num_envs"] = n
env = FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg)); env.init()
dev = torch.device("cuda", 0)
rows = 4096
acts = torch.empty(rows, n, 12, device=dev)
lib = _lib.load()
_lib.check(lib.irrl_bench_actions(1, 0, n, 0, rows, 0.3, C.c_void_p(acts.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
ob = torch.zeros(n, 35, device=dev); rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.bool, device=dev); extra = torch.zeros(n, 6, device=dev)
env.step_rows(1000, acts, 0, ob, rew, done, extra)
cur = 1000
for K in (20, 200, 2000):
    for mode in ("launches", "persistent", "launches", "persistent"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        env.step_rows(300, acts, cur % rows, ob, rew, done, extra); cur += 300
        e0.record()
        env.step_rows(K, acts, cur % rows, ob, rew, done, extra, persistent=(mode == "persistent")); cur += K
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / K
        print("%d envs, K = %4d, %-10s: %.2f us per step = %.1f M env-steps/s" % (n, K, mode, us, n / us), flush=True)
