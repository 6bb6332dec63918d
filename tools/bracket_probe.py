#!/usr/bin/env python3
"""Where does the wall clock of a short synchronised burst of env steps go?  20 steps issued (a) by one irrl_env_step_rows call, (b) by 20
ctypes calls; host time stamps after the launches returned, after the completion event was seen by polling, after torch.cuda.synchronize().
    python tools/bracket_probe.py [--reps 8] [--steps 20]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml
import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, "bp5_imitation.yaml")))["environment"]
cfg["num_envs"] = 4096
env = FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg)); env.init()
dev = torch.device("cuda", 0)
n = 4096
acts = torch.clamp(0.3 * torch.randn(64, n, 12, device=dev), -1, 1)
ob = torch.zeros(n, 35, device=dev); rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.bool, device=dev); extra = torch.zeros(n, 6, device=dev)
env.step_rows(1200, acts, 0, ob, rew, done, extra)
torch.cuda.synchronize()
for mode in ("rows", "python", "rows_poll", "rows_sleep"):
    for rep in range(a.reps):
        call = env.step_rows_call(a.steps, acts, 0, ob, rew, done, extra)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if mode == "rows_sleep":
            time.sleep(0.002)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        if mode == "python":
            for k in range(a.steps):
                env.step(acts[k], ob, rew, done, extra)
        else:
            call()
        e1.record()
        t1 = time.perf_counter()
        t2 = t1
        if mode == "rows_poll":
            while not e1.query():
                pass
            t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print("%-10s rep %d: launches returned %7.1f us, polled done %7.1f us, synchronize returned %7.1f us; events %7.1f us (%.2f us/step)"
              % (mode, rep, 1e6 * (t1 - t0), 1e6 * (t2 - t0), 1e6 * (t3 - t0), 1e3 * e0.elapsed_time(e1), 1e3 * e0.elapsed_time(e1) / a.steps), flush=True)
