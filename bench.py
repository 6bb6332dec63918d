#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the env.step() hot path on MI355X (BASELINE.json metric).

A "step" is one FlexibleGymEnv.step() over one batch of 4096 robots per GPU (8 physics substeps at 4 kHz +
observation + 8 reward terms + termination + masked in-step reset), with the action batch already resident
in HBM.  Workload at N=1: BASELINE config 2 ("4096 envs on 1xMI355X, imitation reward, pure env-step
kernel"), synthetic actions a = clip(0.3*N(0,1), -1, 1).  Multi-GPU: one process per GPU
(launched by torch.distributed.run), 4096 envs per rank, no data-path collective (envs never interact,
RaisimGymEnv.hpp:56) -> weak scaling; the only collectives are the timing barrier and the max-over-ranks.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel = irrl_step_kernel, HIP-event
timed inside this run), `roofline_fp32` (the VALU-FP32 view: this path is latency/issue bound, not HBM
bound), `cpu_baseline` (the f64 oracle with the reference's OpenMP-over-envs threading on the
GPU box's host cores, bounded sample, rank 0 / N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic cost of one env-step (SURVEY 8d / BASELINE.md section 4)
ALG_BYTES_PER_ENV_STEP = 1521.0
ALG_FLOPS_PER_ENV_STEP = 1.15e5  # instrumented oracle (tools/flopcount): 114 823 flop/env-step on this workload, 1.6 contact sweeps
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP32_PEAK_TFLOPS = 157.3


def usable_cores():
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota (a container can
    report 256 hardware threads while being throttled to a fraction of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(env_cfg, target_seconds):
    """Oracle (kind "port": the build's own CPU restatement; the RaiSim reference is closed source and absent)
    timed with `#pragma omp parallel for schedule(dynamic)` over envs like VectorizedEnvironment.hpp:273."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    cfg = dict(env_cfg)
    n = int(cfg["num_envs"])
    env = O.OracleVecEnv(cfg)
    rng = np.random.RandomState(1)
    acts = [np.clip(0.3 * rng.normal(size=(n, 12)), -1, 1).astype(np.float32) for _ in range(8)]
    t0 = time.perf_counter()
    for k in range(4):
        env.step(acts[k])
    per_step = (time.perf_counter() - t0) / 4
    steps = int(max(8, min(750, target_seconds / max(per_step, 1e-6))))
    t0 = time.perf_counter()
    for k in range(steps):
        env.step(acts[k % 8])
    dt = time.perf_counter() - t0
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d control steps (f64 oracle, OpenMP dynamic over envs, %d threads, %.1f s)" % (n, steps, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--cfg", default="bp5_imitation.yaml")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline sample budget (0 disables)")
    ap.add_argument("--ppo-iters", type=int, default=2, help="timed PPO iterations of the LSTM policy at N=1 (0 disables)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import yaml
    import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
    from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the env kernels have no CPU path")
    # IRRL_BENCH_BACKEND=gloo + IRRL_BENCH_ONE_DEVICE=1 let the multi-rank control flow be exercised on a 1-GPU box
    # (both ranks on cuda:0, barrier / max-reduce over gloo); the driver's runs use the defaults: one GPU per rank, RCCL
    backend = os.environ.get("IRRL_BENCH_BACKEND", "nccl")
    if os.environ.get("IRRL_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    with open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, args.cfg)) as f:
        env_cfg = yaml.safe_load(f)["environment"]
    env_cfg["num_envs"] = args.envs
    env_cfg["seedd"] = int(env_cfg.get("seedd", 1)) + 7919 * rank   # different robots on every rank
    n = args.envs
    env = FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(env_cfg), device=local_rank)
    env.init()

    # synthetic action pool resident in HBM (Philox-free here: torch generator seeded per rank)
    g = torch.Generator(device=dev)
    g.manual_seed(1 + rank)
    pool = [torch.clamp(0.3 * torch.randn(n, 12, device=dev, generator=g), -1, 1).contiguous() for _ in range(64)]
    ob = torch.zeros(n, 35, device=dev)
    rew = torch.zeros(n, device=dev)
    done = torch.zeros(n, dtype=torch.bool, device=dev)
    extra = torch.zeros(n, 6, device=dev)
    n_done = torch.zeros((), device=dev)

    def run(k_steps, count_done=False):
        for k in range(k_steps):
            env.step(pool[k % 64], ob, rew, done, extra)
            if count_done:
                n_done.add_(done.sum())

    run(args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(args.steps)
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps   # events on the stream the kernel is launched on
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # sanity on the timed work: finite outputs, episodes really terminate and reset inside the step
    run(50, count_done=True)
    torch.cuda.synchronize()
    assert torch.isfinite(ob).all() and torch.isfinite(rew).all(), "non-finite env outputs"

    if rank == 0:
        # HBM bytes per launch from the PMC passes (FETCH_SIZE x calibrated correction + WRITE_SIZE), collected with
        # rocprofv3 in separate runs (tools/gpu_pmc.sh) and committed under profiles/ -- not measurable in-process
        traffic, issue = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary_latest.json")))
            if int(pmc.get("envs", 4096)) == n and env.lanes_per_robot == 16:
                traffic = float(pmc["hbm_bytes_per_launch"]["total"])
                # how close the single resident wave per SIMD runs to its issue limit of one VALU instruction per 4 cycles
                issue = {"valu_insts_per_wave": pmc["derived"]["valu_insts_per_wave"], "cycles_per_valu_inst": pmc["derived"]["cycles_per_valu_inst"],
                         "frac_of_single_wave_issue_peak": 4.0 / pmc["derived"]["cycles_per_valu_inst"], "source": "profiles/pmc_summary_latest.json"}
        except Exception:
            pass
        total_env_steps = float(n) * world * args.steps
        value = total_env_steps / elapsed
        launch_s = kernel_ms * 1e-3
        ach_gbs = ALG_BYTES_PER_ENV_STEP * n / launch_s / 1e9
        ach_tf = ALG_FLOPS_PER_ENV_STEP * n / launch_s / 1e12
        out = {
            "metric": "env-steps/sec (4096 envs per MI355X)", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: %d envs/GPU x 1 env.step (8 substeps @4 kHz + obs + reward + "
                                   "termination + in-step reset), cfg %s, actions clip(0.3 N(0,1))" % (n, args.cfg),
                       "envs_per_gpu": n, "global_envs": n * world, "parallelism": "env-sharded x%d" % world,
                       "lanes_per_robot": env.lanes_per_robot},
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "irrl_step_kernel_l%d" % env.lanes_per_robot, "avg_launch_us": kernel_ms * 1e3,
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_ENV_STEP * n},
            "roofline_fp32": {"bound": "valu_fp32 (latency/issue bound: 1 wave per SIMD at 4096 envs)", "achieved": ach_tf,
                              "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / FP32_PEAK_TFLOPS,
                              "algorithmic_flops_per_launch": ALG_FLOPS_PER_ENV_STEP * n, "valu_issue": issue},
            "resets_in_50_steps": float(n_done.item()),
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(env_cfg, args.cpu_seconds)
        if world == 1 and args.ppo_iters > 0:
            # second half of BASELINE.json's metric ("PPO iters/sec"): the reference's training iteration (750-step rollout of
            # all envs with the 2x48 + 2x48 LSTM policy, GAE, 10 epochs of full-length BPTT, global reset) on this GPU;
            # reported beside the headline value, never mixed into it
            del env
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import ppo_bench
            out["ppo"] = ppo_bench.measure("lstm", n, 750, args.ppo_iters + 1, 10, "default_cfg.yaml", verbose=False)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
