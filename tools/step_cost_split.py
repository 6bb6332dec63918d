#!/usr/bin/env python3
"""Where does the time of one env.step() go?  T(loop_count) = fixed + loop_count * substep: times the step kernel (HIP events,
hipGraph of 200 steps) for 1 / 2 / 4 / 8 physics substeps per control step (irrl_env_set_control_dt changes loop_count, the
physics time step stays 0.25 ms), from landed states, and prints the fitted fixed (prologue + epilogue) and per-substep cost.
    python tools/step_cost_split.py [--envs 4096]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--cfg", default="bp5_imitation.yaml")
    a = ap.parse_args()
    import ctypes as C
    import numpy as np
    import torch
    import yaml
    import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
    from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
    from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
    cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, a.cfg)))["environment"]
    n = a.envs
    cfg["num_envs"] = n
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    rows = 700
    actions = torch.empty(rows, n, 12, device=dev)
    _lib.check(lib.irrl_bench_actions(1, 0, n, 0, rows, 0.3, C.c_void_p(actions.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    ob, rew = torch.zeros(n, 35, device=dev), torch.zeros(n, device=dev)
    done, extra = torch.zeros(n, dtype=torch.bool, device=dev), torch.zeros(n, 6, device=dev)
    out = {}
    for loops in (8, 4, 2, 1):
        env = FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg))
        env.init()
        for k in range(400):                      # land with the full step
            env.step(actions[k], ob, rew, done, extra)
        env.setControlTimeStep(0.00025 * loops)   # loop_count = loops from here on (the task clock runs slower; irrelevant for timing)
        for k in range(50):
            env.step(actions[400 + k], ob, rew, done, extra)
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for k in range(200):
                env.step(actions[450 + k], ob, rew, done, extra)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        c0 = env.counters()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        c1 = env.counters()
        out[loops] = {"us_per_step": 1e3 * e0.elapsed_time(e1) / 200, "contact_fraction": (c1[1] - c0[1]) / float(4 * loops * n * 200)}
        del env
    t8, t1 = out[8]["us_per_step"], out[1]["us_per_step"]
    sub = (t8 - t1) / 7.0
    print(json.dumps({"envs": n, "per_loop_count": out, "substep_us": sub, "fixed_us": t1 - sub,
                      "note": "fixed = launch + state load/store + prologue + epilogue (observation, reward, command, gait reference + IK, termination)"}))


if __name__ == "__main__":
    main()
