#!/usr/bin/env python3
"""Soak run of a Crutial pool: 4096 envs x 3000 steps of strong random actions (robots fall, spheres hit trunks, ground, get
re-parked); every output must stay finite and the sphere inside a sane box around its robot."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, yaml
import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
n = 4096
for cfg_name in ("default_cfg.yaml", "bp5_terrain.yaml"):
    cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, cfg_name)))["environment"]
    cfg.update(num_envs=n, Crutial=True, CubeNum=6)
    env = FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg)); env.init()
    dev = torch.device("cuda")
    ob, rew = torch.zeros(n, 35, device=dev), torch.zeros(n, device=dev)
    done, extra = torch.zeros(n, dtype=torch.bool, device=dev), torch.zeros(n, 6, device=dev)
    g = torch.Generator(device="cuda").manual_seed(3)
    hits = 0
    for k in range(3000):
        a = torch.randn(n, 12, device=dev, generator=g).clamp(-1, 1)
        env.step(a, ob, rew, done, extra)
        if k % 250 == 249:
            assert torch.isfinite(ob).all() and torch.isfinite(rew).all() and torch.isfinite(extra).all(), k
            st = env.get_state()
            sph = st[:, 277:286]
            assert np.isfinite(st).all()
            rel = sph[:, 0:3] - st[:, 0:3]
            assert np.abs(sph[:, 3:6]).max() < 60.0, np.abs(sph[:, 3:6]).max()
            hits += int((sph[:, 8] == 1).sum())
    info = np.zeros((n, 4), np.float32); env.GetSphereInfo(info)
    print(cfg_name, "ok: 3000 steps finite; released spheres seen at the checkpoints:", hits, "max |sphere velocity| %.1f m/s" % np.abs(sph[:, 3:6]).max(), "radius range", info[:, 3].min(), info[:, 3].max())
