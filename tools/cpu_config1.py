#!/usr/bin/env python3
"""BASELINE config 1 ("64 parallel envs, CPU reference path via run_bp_v5.py --train ... plumbing") and the CPU-side figures of
BASELINE.md section 3 -- a TOOL, not part of the product: the package has no CPU path.

The literal reference path (RaiSim + OpenMP + TensorFlow/stable-baselines) cannot run anywhere we control (closed binary absent,
SURVEY 8c); its stand-in is this build's own CPU restatement, the f64 oracle (oracle/), with the reference's threading model
(`#pragma omp parallel for schedule(dynamic)` over envs, VEC:273), driven through the SAME script surface a user runs:
scripts/run_bp_v5.py --train, whose `FlexibleGymEnv` / `Environment` names are bound to oracle-backed doubles here and whose
learner (ppo2.py, policies.py) runs on CPU tensors.  That substitution is stated in every line this tool prints.

Prints one JSON object:
  plumbing   2 PPO iterations (750-step rollouts of 64 envs, GAE, 10 epochs of full-length BPTT) through run_bp_v5.main
  env_rate   env-steps/s of the oracle at 64 and 4096 envs on all usable cores, and on ONE thread, f64 and f32 builds
             (bench.py's Philox action stream, after a 60-step landing pre-roll)
    python tools/cpu_config1.py [--iters 2] [--seconds 6]"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "scripts")):
    if p not in sys.path:
        sys.path.insert(0, p)


def cores_and_model():
    import bench
    return bench.usable_cores(), bench.cpu_model()


def env_rate(n, precision, threads, seconds, cfg):
    import numpy as np
    import oracle as O
    from bench_actions import bench_actions
    gomp = ctypes.CDLL("libgomp.so.1")
    gomp.omp_set_num_threads(threads)
    c = dict(cfg)
    c["num_envs"] = n
    env = O.OracleVecEnv(c, precision=precision)
    acts = bench_actions(1, 0, n, 0, 64, 0.3)
    t0 = time.perf_counter()
    for k in range(60):
        env.step(acts[k % 64])
    per = (time.perf_counter() - t0) / 60
    steps = int(max(5, min(750, seconds / max(per, 1e-6))))
    t0 = time.perf_counter()
    for k in range(steps):
        env.step(acts[(60 + k) % 64])
    dt = time.perf_counter() - t0
    return {"envs": n, "precision": precision, "threads": threads, "steps": steps, "env_steps_per_sec": n * steps / dt}


def plumbing(iters, cores, max_time=None):
    """scripts/run_bp_v5.py --train on CPU: the script's env classes are replaced by oracle-backed doubles"""
    import tempfile
    import torch
    import yaml
    import run_bp_v5 as script
    from oracle_torch_env import OracleTorchEnv
    import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
    cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, "default_cfg.yaml")))
    cfg["environment"].update(num_envs=64, render=False, num_threads=cores)
    if max_time is not None:     # tests: a short rollout (n_steps = max_time / control_dt) through the same plumbing
        cfg["environment"]["max_time"] = float(max_time)
    n_steps = int(round(float(cfg["environment"]["max_time"]) / float(cfg["environment"]["control_dt"])))
    tmp = tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False)
    yaml.safe_dump(cfg, tmp)
    tmp.close()
    script.FlexibleGymEnv = lambda rsc, cfg_yaml, device=None: yaml.safe_load(cfg_yaml)
    script.Environment = lambda env_cfg: OracleTorchEnv(env_cfg)
    torch.set_num_threads(cores)
    rows = []
    orig_learn = script.PPO2._learn_loop

    def learn_loop(self, *a, **k):
        orig_learn(self, *a, **k)
        rows.extend(self.log)
    script.PPO2._learn_loop = learn_loop
    t0 = time.perf_counter()
    script.main(["--train", "--cfg", tmp.name, "--max_iter", str(64 * n_steps * iters), "--save", "0"])
    dt = time.perf_counter() - t0
    os.unlink(tmp.name)
    script.PPO2._learn_loop = orig_learn
    return {"command": "scripts/run_bp_v5.py --train --cfg <default_cfg.yaml with num_envs 64> --max_iter %d" % (64 * n_steps * iters),
            "iterations": iters, "n_steps": n_steps, "wall_s": dt, "ppo_iters_per_sec": iters / dt, "samples_per_sec": 64 * n_steps * iters / dt,
            "log": [{k: r[k] for k in ("nupdates", "fps", "ep_reward_mean", "ep_len_mean", "explained_variance", "value_loss") if k in r} for r in rows]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=6.0, help="budget per env-rate sample")
    ap.add_argument("--skip-plumbing", action="store_true")
    a = ap.parse_args()
    import yaml
    import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
    cores, model = cores_and_model()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, "bp5_imitation.yaml")))["environment"]
    out = {"what": "CPU stand-in for the reference path: this build's f64 oracle + the PyTorch-CPU learner (the RaiSim path is closed source and absent)",
           "cpu_model": model, "usable_cores": cores, "env_rate": []}
    for n, prec, thr in ((64, "f64", cores), (4096, "f64", cores), (4096, "f32", cores), (256, "f64", 1), (256, "f32", 1)):
        out["env_rate"].append(env_rate(n, prec, thr, a.seconds, cfg))
    if not a.skip_plumbing:
        out["plumbing"] = plumbing(a.iters, cores)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
