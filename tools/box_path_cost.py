#!/usr/bin/env python3
"""What does a control step cost when robots lie on a corner of the trunk box?  All 4096 robots are put low and tilted onto a
bottom corner (tests/parity_lib.tilt_onto_box_corner), ONE env.step() is timed with HIP events, repeated from fresh tilted
states; the same with the robots 10 cm higher (no corner touches) is the reference.  The difference is the price of the rare
contact-block instantiation (csrc/env_core.hpp), which sets the length of every step in which some wave holds a falling robot."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import numpy as np
    import torch
    import parity_lib as PL
    from conftest import load_env_cfg
    from hip_env import HipVecEnv
    n = 4096
    env = HipVecEnv(load_env_cfg("bp5_imitation.yaml", num_envs=n))
    a = np.zeros((n, 12), np.float32)
    for _ in range(80):
        env.step(a)
    base = env.get_state()
    rng = np.random.RandomState(0)
    act = torch.zeros(n, 12, device="cuda")
    ob, rew = torch.zeros(n, 35, device="cuda"), torch.zeros(n, device="cuda")
    done, extra = torch.zeros(n, dtype=torch.bool, device="cuda"), torch.zeros(n, 6, device="cuda")
    out = {}
    for name, dz in (("on_a_corner", 0.0), ("same_pose_10cm_higher", 0.10)):
        ts, dones = [], 0
        for rep in range(12):
            st = PL.tilt_onto_box_corner(base, rep, rng)
            st[:, 2] += dz
            env.set_state(PL.f32_round_state(st))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            env.impl.step(act, ob, rew, done, extra)
            e1.record()
            torch.cuda.synchronize()
            ts.append(1e3 * e0.elapsed_time(e1))
            dones += int(done.sum())
        out[name] = {"us_per_step_median": float(np.median(ts[2:])), "dones_per_step": dones / 12.0}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
