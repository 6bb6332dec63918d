#!/usr/bin/env python3
"""Soak: STEPS env steps as a few persistent launches against the same steps one launch at a time, same action table -- pools and outputs compared
bit for bit at the end of every chunk (in-step resets, per-episode randomisation on rough ground):  python tools/persistent_soak.py [steps] [cfg] [envs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, yaml
import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cfg_name = sys.argv[2] if len(sys.argv) > 2 else "bp5_terrain.yaml"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, cfg_name)))["environment"]
cfg["num_envs"] = n
envs = [FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg)) for _ in range(2)]
for e in envs:
    e.init()
g = torch.Generator(device="cuda").manual_seed(3)
table = (0.5 * torch.randn(512, n, 12, device="cuda", generator=g)).clamp(-1, 1)
outs = [(torch.zeros(n, 35, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, dtype=torch.bool, device="cuda"), torch.zeros(n, 6, device="cuda")) for _ in range(2)]
chunk, done, t_p, t_r = 2500, 0, 0.0, 0.0
while done < steps:
    k = min(chunk, steps - done)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    envs[0].step_rows(k, table, done % 512, *outs[0], persistent=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    envs[1].step_rows(k, table, done % 512, *outs[1])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    t_p += t1 - t0; t_r += t2 - t1; done += k
    same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])) and np.array_equal(envs[0].get_state(), envs[1].get_state())
    print("%6d steps: %s   episodes started %d   finite %s" % (done, "bit-identical" if same else "DIFFERENT", envs[0].counters()[0], bool(torch.isfinite(outs[0][0]).all())), flush=True)
    assert same
print("%s, %d envs, %d steps: persistent launches %.2f us per step, one launch per step %.2f us per step" % (cfg_name, n, steps, 1e6 * t_p / steps, 1e6 * t_r / steps))
