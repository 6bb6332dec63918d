#!/usr/bin/env python3
"""How uneven are the 1024 waves of one step launch?  With one wave per SIMD the launch lasts as long as its SLOWEST wave.
Needs the diagnostic build (tools/build_variants.py prof=-DIRRL_PROFILE_WAVES), which writes every wave's duration (100 MHz
ticks) into extraInfo[:, 5]:   IRRL_ENV_LIB=.../libirrl_env_prof.so python tools/wave_spread.py [--cfg default_cfg.yaml]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="bp5_imitation.yaml")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--sigma", type=float, default=0.3, help="action scale (1.0 ~ an untrained policy: robots fall all the time)")
    ap.add_argument("--solver", type=int, default=3)
    a = ap.parse_args()
    import ctypes as C
    import numpy as np
    import torch, yaml
    import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
    from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
    from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
    n = 4096
    cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, a.cfg)))["environment"]
    cfg["num_envs"] = n
    cfg["ContactSolver"] = a.solver
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    env = FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg))
    env.init()
    rows = 400 + a.steps
    actions = torch.empty(rows, n, 12, device=dev)
    _lib.check(lib.irrl_bench_actions(1, 0, n, 0, rows, a.sigma, C.c_void_p(actions.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    ob, rew = torch.zeros(n, 35, device=dev), torch.zeros(n, device=dev)
    done, extra = torch.zeros(n, dtype=torch.bool, device=dev), torch.zeros(n, 6, device=dev)
    for k in range(400):
        env.step(actions[k], ob, rew, done, extra)
    per = []
    for k in range(a.steps):
        env.step(actions[400 + k], ob, rew, done, extra)
        per.append(extra[::4, 3:6].clone())        # per wave (4 robots share it): rank steps, flags, duration
    torch.cuda.synchronize()
    raw = torch.stack(per).cpu().numpy()
    t = raw[:, :, 2] * 0.01       # us
    rs, fl = raw[:, :, 0], raw[:, :, 1].astype(np.int64)
    reset, box, sweeps = (fl & 1) != 0, (fl & 2) != 0, fl >> 8
    def grp(m):
        return {"share": float(m.mean()), "mean_us": float(t[m].mean()) if m.any() else None}
    slow = t >= np.percentile(t, 99)
    detail = {"no_reset_no_box": grp(~reset & ~box), "reset": grp(reset), "box": grp(box),
              "rank_steps_per_step": {"mean": float(rs.mean()), "p99": float(np.percentile(rs, 99)), "max": float(rs.max())},
              "sweeps_per_step": {"mean": float(sweeps.mean()), "p99": float(np.percentile(sweeps, 99)), "max": float(sweeps.max())},
              "slowest_1pct": {"reset_share": float(reset[slow].mean()), "box_share": float(box[slow].mean()), "rank_steps_mean": float(rs[slow].mean())},
              "us_per_rank_step_fit": float(np.polyfit(rs[~reset & ~box].ravel(), t[~reset & ~box].ravel(), 1)[0]),
              "us_at_zero_rank_steps_fit": float(np.polyfit(rs[~reset & ~box].ravel(), t[~reset & ~box].ravel(), 1)[1]),
              "slowest_wave_per_step": {"reset_share": float(np.mean([reset[i, np.argmax(t[i])] for i in range(t.shape[0])])),
                                        "box_share": float(np.mean([box[i, np.argmax(t[i])] for i in range(t.shape[0])])),
                                        "rank_steps_mean": float(np.mean([rs[i, np.argmax(t[i])] for i in range(t.shape[0])]))}}
    mx = t.max(1)
    # what ONE persistent launch over these steps sees (irrl_env_step_rows_persistent: a wave walks its robots through all the steps, nothing
    # waits per step): the launch lasts as long as the wave with the largest SUM over the steps
    per_wave_mean = t.mean(0)
    out = {"cfg": a.cfg, "sigma": a.sigma, "solver": a.solver, "waves": int(t.shape[1]), "steps": a.steps, "wave_us": {"mean": float(t.mean()), "p50": float(np.median(t)), "p90": float(np.percentile(t, 90)),
           "p99": float(np.percentile(t, 99)), "max_mean_over_steps": float(mx.mean())},
           "slowest_over_mean": float(mx.mean() / t.mean()),
           "one_persistent_launch_over_these_steps": {"slowest_wave_mean_us": float(per_wave_mean.max()), "slowest_over_mean": float(per_wave_mean.max() / t.mean())},
           "detail": detail}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
