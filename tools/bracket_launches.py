#!/usr/bin/env python3
"""Per-LAUNCH durations inside a short synchronised bracket of env steps (the driver's --steps 20), by what the host thread does while the GPU
works (blocked in synchronize / polling the closing event) and by what came before (idle for a few ms / straight behind 300 steps).
    python tools/bracket_launches.py [--reps 6] [--steps 20]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml
import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, "bp5_imitation.yaml")))["environment"]
cfg["num_envs"] = 4096
env = FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg)); env.init()
dev = torch.device("cuda", 0)
n, K = 4096, a.steps
acts = torch.clamp(0.3 * torch.randn(64, n, 12, device=dev), -1, 1)
ob = torch.zeros(n, 35, device=dev); rew = torch.zeros(n, device=dev); done = torch.zeros(n, dtype=torch.bool, device=dev); extra = torch.zeros(n, 6, device=dev)
env.step_rows(1200, acts, 0, ob, rew, done, extra)
torch.cuda.synchronize()
for before in ("idle", "hot"):
    for host in ("blocked", "polling"):
        for rep in range(a.reps):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
            c0 = env.counters()
            if before == "hot":
                env.step_rows(300, acts, 0, ob, rew, done, extra)
            else:
                time.sleep(0.005)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            evs[0].record()
            for k in range(K):
                env.step(acts[k], ob, rew, done, extra)
                evs[k + 1].record()
            if host == "polling":
                while not evs[K].query():
                    pass
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            d = [1e3 * evs[k].elapsed_time(evs[k + 1]) for k in range(K)]
            print("%-4s %-8s rep %d: wall %6.1f us, events %6.1f us = %.2f us/step; per launch: %s" %
                  (before, host, rep, 1e6 * (t1 - t0), 1e3 * evs[0].elapsed_time(evs[K]), 1e3 * evs[0].elapsed_time(evs[K]) / K, " ".join("%.1f" % x for x in d)), flush=True)
