#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the env.step() hot path on MI355X, and PPO iterations/sec (BASELINE.json metric).

A "step" is one FlexibleGymEnv.step() over one batch of 4096 robots per GPU (8 physics substeps at 4 kHz +
observation + 8 reward terms + termination + masked in-step reset), with the action batch already resident
in HBM.  Workload at N=1: BASELINE config 2 ("4096 envs on 1xMI355X, imitation reward, pure env-step
kernel"), synthetic actions a = clip(0.3*N(0,1), -1, 1) from Philox(seed=1, stream=env, counter=step)
(device generator `irrl_bench_actions`, numpy twin tools/bench_actions.py).

STEADY STATE: robots are reset 4 cm above their standing height, i.e. the first ~46 control steps after a reset
are free flight with the contact solve skipped.  Before --warmup the bench therefore ALWAYS runs an untimed
pre-roll (--preroll, default 1000, at least 200 steps: the robots land within ~60 steps, the first wave of falls and in-step
resets that follows has passed by step ~500) so that the timed region sees the stationary workload; it reads the
kernels' own counters (toe-substeps in the contact list, episodes started) around the timed region, reports
`contact_fraction_in_timed_region` / `resets_in_timed_region`, and FAILS when the region was free flight.
The last HOT_STEPS (300) untimed steps go out straight in front of the synchronize that opens the timed bracket (after --warmup and the
launch-mode probes), so that the GPU arrives there from work and not from milliseconds of idling on host set-up (bracket()).

Multi-GPU: `python bench.py --gpus N` with N > 1 starts N ranks by itself (python -m torch.distributed.run, one
rank per GPU, rendezvous on 127.0.0.1) unless it already runs under a launcher (WORLD_SIZE set).  4096 envs per
rank, no data-path collective in the env benchmark (envs never interact, RaisimGymEnv.hpp:56) -> weak scaling;
collectives there are only the timing barrier and the max-over-ranks.  The PPO leg runs on every rank with the
north_star's collectives in the loop (flat 283 KB gradient all-reduce + 3-float advantage moments per optimizer
step, ppo2.py) and reports whole-job iterations/s and samples/s.

Prints ONE JSON line (rank 0).  `value` = env-steps/s of the K timed steps issued the way --launch says (default: ONE persistent
launch, irrl_env_step_rows_persistent_out) with EVERY step's ob / reward / done / extraInfo stored to its own row in HBM.  Extra
objects: `roofline` (the binding one: FP32 VALU, dominant kernel HIP-event timed over the timed region on the stream it is launched
on; `traffic` = its HBM bytes from the PMC passes), `roofline_hbm` (the HBM view: this path is latency/issue bound, not HBM bound,
SURVEY 8d), `launch_modes` (the other ways of issuing the same steps, measured after the timed bracket), `cpu_baseline` (the f64 oracle
with the reference's OpenMP-over-envs threading on the GPU box's host cores + its single-thread rate, bounded sample, rank 0 / N=1 only),
`ppo` (second half of the metric: the LSTM policy of config 3; default precision of the update + `f32_level`: the same iteration with
the update at the f32 level) and `ppo_mlp` (the same with config 2's MlpPolicy learner).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic cost of one env-step (SURVEY 8d / BASELINE.md section 4)
ALG_BYTES_PER_ENV_STEP = 1521.0
ALG_FLOPS_PER_ENV_STEP = 1.18e5  # instrumented oracle (tools/flopcount), the published method (Gauss-Seidel + published per-contact rule): 118 364 flop/env-step on this workload, 1.6 contact sweeps (round 2, first rule: 114 823)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP32_PEAK_TFLOPS = 157.3
ACTION_SEED = 1
MIN_PREROLL = 200
HOT_STEPS = 300                 # untimed steps of the same workload right in front of a bracket's opening synchronize (see bracket())


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--preroll", type=int, default=1000, help="untimed env steps before --warmup (never fewer than %d): robots land" % MIN_PREROLL)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--cfg", default="bp5_imitation.yaml")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="override one key of the env configuration (YAML scalar), e.g. --set Crutial=true; the bench line names it")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline sample budget (0 disables)")
    ap.add_argument("--ppo-iters", type=int, default=5, help="timed PPO iterations of each learner leg, reported with min / median / max (0 disables)")
    ap.add_argument("--ppo-steps", type=int, default=750, help="rollout length of the PPO leg (the metric's is 750)")
    ap.add_argument("--ppo-epochs", type=int, default=10, help="optimisation epochs of the PPO leg (the metric's is 10)")
    ap.add_argument("--launch", choices=("persistent", "rows", "graph", "python"), default="persistent",
                    help="how the K timed steps are issued -- FIXED by this flag, nothing is probed or selected inside the measured run.  In every "
                         "mode EVERY step's ob / reward / done / extraInfo is stored to its own row of [K, N, .] tables in HBM (what K step() calls "
                         "of the reference return).  'persistent' (default) = ONE launch in which every wave walks its own robots through the K "
                         "steps (irrl_env_step_rows_persistent_out: robots never interact, so nothing has to wait for the slowest wave of a step), "
                         "'rows' = K back-to-back launches from one irrl_env_step_rows_out call, 'graph' = one hipGraph of K step-kernel nodes, "
                         "'python' = one ctypes call per step.  The modes that were not timed are reported as extras (`launch_modes`).")
    ap.add_argument("--no-mode-extras", action="store_true", help="skip the untimed extras that measure the other launch modes")
    ap.add_argument("--no-graph", dest="launch", action="store_const", const="python", help="same as --launch python")
    ap.add_argument("--check-steps", type=int, default=2000, help="extra untimed-for-`value` window after the timed region that "
                    "re-measures us/step over a longer run (0 disables); reported as `steady_state_check`")
    return ap.parse_args(argv)


def launched_by_torchrun():
    """A launcher's env:// rendezvous is COMPLETE: WORLD_SIZE, RANK and MASTER_PORT all set, as torch.distributed.run exports them.  A box that
    merely exports WORLD_SIZE=1 is not a launcher: the process stays single (or starts its own ranks for --gpus N)."""
    return all(k in os.environ for k in ("WORLD_SIZE", "RANK", "MASTER_PORT"))


def launch_ranks(args):
    """`--gpus N` outside a launcher: start N fresh rank processes and relay their output.  This parent has not imported
    torch and never touches the GPU (a process that has initialised the GPU must not be replaced / must not fork ranks)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(env_cfg, target_seconds):
    """Oracle (kind "port": the build's own CPU restatement; the RaiSim reference is closed source and absent)
    timed with `#pragma omp parallel for schedule(dynamic)` over envs like VectorizedEnvironment.hpp:273, on the same
    Philox action stream as the GPU (after the same kind of pre-roll: landed robots), then once more on ONE thread."""
    import ctypes
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import oracle as O
    from bench_actions import bench_actions
    cores = usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    cfg = dict(env_cfg)
    n = int(cfg["num_envs"])
    env = O.OracleVecEnv(cfg)
    gomp = ctypes.CDLL("libgomp.so.1")
    gomp.omp_set_num_threads(cores)
    acts = bench_actions(ACTION_SEED, 0, n, 0, 64, 0.3)
    t0 = time.perf_counter()
    for k in range(60):          # pre-roll: the robots land (46 steps of free flight after the reset)
        env.step(acts[k % 64])
    per_step = (time.perf_counter() - t0) / 60
    steps = int(max(8, min(750, 0.75 * target_seconds / max(per_step, 1e-6))))
    t0 = time.perf_counter()
    for k in range(steps):
        env.step(acts[(60 + k) % 64])
    dt = time.perf_counter() - t0
    multi = n * steps / dt
    # single thread: a slice of the pool, so that it stays inside the budget
    gomp.omp_set_num_threads(1)
    n1 = max(16, min(n, 256))
    cfg1 = dict(cfg)
    cfg1["num_envs"] = n1
    env1 = O.OracleVecEnv(cfg1)
    a1 = np.ascontiguousarray(acts[:, :n1])
    for k in range(60):
        env1.step(a1[k % 64])
    steps1 = int(max(8, min(750, 0.2 * target_seconds * multi / cores / n1)))
    t0 = time.perf_counter()
    for k in range(steps1):
        env1.step(a1[(60 + k) % 64])
    dt1 = time.perf_counter() - t0
    gomp.omp_set_num_threads(cores)
    return {"value": multi, "unit": "env-steps/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "single_thread_value": n1 * steps1 / dt1,
            "sample": "%d envs x %d control steps after a 60-step pre-roll (f64 oracle, OpenMP dynamic over envs, %d threads, %.1f s); "
                      "single thread: %d envs x %d steps (%.1f s)" % (n, steps, cores, dt, n1, steps1, dt1)}


def worker(args):
    import ctypes as C
    import torch
    import yaml
    import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
    from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
    from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv

    launched = launched_by_torchrun()
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the env kernels have no CPU path")
    # IRRL_BENCH_BACKEND=gloo + IRRL_BENCH_ONE_DEVICE=1 let the multi-rank control flow be exercised on a 1-GPU box
    # (both ranks on cuda:0, barrier / reductions over gloo); the driver's runs use the defaults: one GPU per rank, RCCL
    backend = os.environ.get("IRRL_BENCH_BACKEND", "nccl")
    if os.environ.get("IRRL_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    ranks_seen = 1
    if launched:
        # under a launcher, ALSO with one rank: the job then runs its barrier / reductions / PPO collectives over RCCL
        # on device tensors like an N-GPU job does -- the way to exercise the nccl path on a 1-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                      # proves every rank is in the communicator (RCCL over xGMI by default)
        ranks_seen = int(ones.item())

    with open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, args.cfg)) as f:
        env_cfg = yaml.safe_load(f)["environment"]
    for kv in args.set:
        key, _, val = kv.partition("=")
        if key not in env_cfg:
            raise SystemExit("--set %s: no such key in %s" % (key, args.cfg))
        env_cfg[key] = yaml.safe_load(val)
    n = args.envs
    env_cfg["num_envs"] = n
    env_cfg["EnvIdOffset"] = rank * n   # rank r owns the global env ids r * n .. (r + 1) * n - 1 of the one big pool (same seed everywhere)
    env = FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(env_cfg), device=local_rank)
    env.init()
    loop_count = int(round(float(env_cfg["control_dt"]) / float(env_cfg["simulation_dt"])))

    # synthetic action stream resident in HBM: row s = actions of global step s for this rank's envs (global env id =
    # rank * n + e).  Beyond 16384 steps the stream wraps (it would be 3 GB otherwise); the default run uses 5.5 k rows.
    preroll = max(MIN_PREROLL, args.preroll)
    total = preroll + args.warmup + 8 * (args.steps + HOT_STEPS) + 50 + args.check_steps
    rows = min(total, 16384)
    lib = _lib.load()
    actions = torch.empty(rows, n, 12, device=dev)
    _lib.check(lib.irrl_bench_actions(ACTION_SEED, rank * n, n, 0, rows, 0.3, C.c_void_p(actions.data_ptr()),
                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    ob = torch.zeros(n, 35, device=dev)
    rew = torch.zeros(n, device=dev)
    done = torch.zeros(n, dtype=torch.bool, device=dev)
    extra = torch.zeros(n, 6, device=dev)
    # the timed steps keep EVERY step's outputs: row k of these tables is what the k-th step() call of the reference returns
    # (VEC:268-278, RaisimGymVecEnv.py:26-52) -- 173 B per env-step delivered to HBM, nothing overwritten inside the bracket
    K = args.steps
    ob_rows = torch.zeros(K, n, 35, device=dev)
    rew_rows = torch.zeros(K, n, device=dev)
    done_rows = torch.zeros(K, n, dtype=torch.bool, device=dev)
    extra_rows = torch.zeros(K, n, 6, device=dev)
    cursor = [0]

    def run(k_steps, mode="rows"):
        s0 = cursor[0]
        if mode in ("rows", "persistent"):
            env.step_rows(k_steps, actions, s0 % rows, ob, rew, done, extra, persistent=(mode == "persistent"))
        else:
            for k in range(k_steps):
                env.step(actions[(s0 + k) % rows], ob, rew, done, extra)
        cursor[0] = s0 + k_steps

    run(preroll)                 # untimed, unconditional: robots land and the contact set becomes stationary
    run(args.warmup)
    torch.cuda.synchronize()
    # The K timed steps go out as K back-to-back launches from ONE C call (irrl_env_step_rows: action row = launch argument): no
    # per-step Python / ctypes latency inside the bracket, and the first kernel starts a few microseconds after the call -- with
    # the driver's --steps 20 the bracket is 0.8 ms long and a hipGraph launch (--launch graph: what the PPO rollout uses, 1500
    # nodes there) costs ~40 us before its first node runs.  Capturing records the launches without executing them.
    graph_cache = {}

    def run_kept(s0):
        """the K timed steps, one FlexibleGymEnv.step() call per step, every step's outputs into its own row"""
        for k in range(K):
            env.step(actions[(s0 + k) % rows], ob_rows[k], rew_rows[k], done_rows[k], extra_rows[k])

    def make_graph():
        if "g" not in graph_cache:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                run_kept(cursor[0])
            torch.cuda.synchronize()
            graph_cache["g"] = g
        return graph_cache["g"]

    def bracket(mode, events=None, barrier=False, counters=None, keep=True):
        """args.steps env steps issued in `mode` (keep: every step's outputs into its own row of the [K, N, .] tables; keep=False, an extra
        only: the [N, .] arrays every step overwrites), bracketed by (barrier +) synchronize on both sides -> wall seconds.  Completion is first
        seen by polling the closing event (hipEventQuery), then confirmed by torch.cuda.synchronize(): on some boxes of this pool a host
        thread BLOCKED in the synchronize behind a short burst is woken up milliseconds late (profiles/r04_burst_wakeup.log: 20 steps =
        0.9 ms of kernels, 2.0-3.4 ms of wall clock); a thread that polls is not.
        HOT_STEPS untimed steps of the same workload go out right in front of the opening synchronize -- the tail of the pre-roll: a GPU
        that has idled for a few milliseconds (set-up work on the host, the probe brackets) runs every launch of the next millisecond
        6 % slower than one that arrives from work (47.3 against 44.5 us per launch with an event between launches, same box,
        profiles/r04_bracket_launches.log), and --warmup 5 (0.2 ms) does not bring it back.  What the bracket then measures is the
        kernel at its working clock plus the start-up of the first launch behind a synchronize, which belongs to the contract."""
        g = make_graph() if mode == "graph" else None
        e0, e1 = events if events is not None else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        run(HOT_STEPS, "persistent" if mode == "persistent" else "rows")      # (the same kernel as the timed steps: its code is warm in the L2s)
        if counters is not None:
            env.counters_into(counters)       # stream-ordered, behind the hot steps: what the kernels have counted up to the timed region
        eh = torch.cuda.Event()
        eh.record()
        outs = (ob_rows, rew_rows, done_rows, extra_rows) if keep else (ob, rew, done, extra)
        call = (env.step_rows_call(args.steps, actions, cursor[0] % rows, *outs, persistent=(mode == "persistent"))
                if mode in ("rows", "persistent") else None)
        while not eh.query():        # the host thread arrives from work too: it polls through the 12 ms of the hot steps instead of sleeping in the synchronize
            pass
        if barrier and dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        e0.record()
        if mode in ("rows", "persistent"):
            call()
            cursor[0] += args.steps
        elif mode == "graph":
            g.replay()
            cursor[0] += args.steps
        else:
            run_kept(cursor[0])
            cursor[0] += args.steps
        e1.record()
        while not e1.query():
            pass
        torch.cuda.synchronize()
        return time.perf_counter() - t_start

    # HOW the K timed steps are issued is FIXED by --launch (default: persistent); nothing is probed or selected inside the measured run.
    # In every mode every step's outputs are stored to their own rows.  persistent: the engine's multi-step entry point takes the K action
    # rows resident in HBM and runs the K steps as ONE launch (irrl_env_step_rows_persistent_out): robots never interact (VEC:273), so every
    # wave walks its own robots through the K steps and nothing waits for the slowest wave of a step -- what the fused rollout kernels of the
    # learner do with the env part.  The other ways keep one launch per step: K back-to-back launches from one C call (rows), a hipGraph of
    # K nodes (~40 us before its first node runs), one ctypes call per step (the reference-shaped surface).  Those not timed are measured
    # AFTER the timed bracket and reported as extras (`launch_modes`).
    mode = args.launch
    if mode == "graph" and args.steps > 20000:
        raise SystemExit("bench.py: --launch graph with more than 20000 nodes is not supported; use --launch rows")
    # the kernels' counters are summed on the device, stream-ordered: no read-back (idle GPU) right before the timed region
    cnt0, cnt1 = torch.zeros(3, dtype=torch.int64, device=dev), torch.zeros(3, dtype=torch.int64, device=dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    elapsed = bracket(mode, (ev0, ev1), barrier=True, counters=cnt0)   # barrier + synchronize | K steps | synchronize (+ barrier below); the MAX over ranks is taken below
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    env.counters_into(cnt1)      # nothing has stepped the pool since the timed region ended
    torch.cuda.synchronize()
    kernel_ms = ev0.elapsed_time(ev1) / args.steps   # events on the stream the kernel is launched on
    c0, c1 = cnt0.tolist(), cnt1.tolist()
    resets = c1[0] - c0[0]
    contact_fraction = (c1[1] - c0[1]) / float(4 * loop_count * n * args.steps)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        agg = torch.tensor([float(resets), contact_fraction], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(agg)
        resets, contact_fraction = int(agg[0].item()), float(agg[1].item()) / world
    assert torch.isfinite(ob_rows).all() and torch.isfinite(rew_rows).all() and torch.isfinite(extra_rows).all(), "non-finite env outputs in the kept rows"
    # every one of the K rows was delivered: a step's observation carries the gait phase (sin, cos) in columns 3:5 (ENV:964-968), whose
    # squares sum to ~1 after un-scaling -- a row the kernel had not written would still hold the zeros it was allocated with
    assert bool((ob_rows[:, :, 3:5].abs().sum(dim=(1, 2)) > 0).all()), "a step's output row was not written"
    # the timed region must be the steady state (robots on the ground, episodes ending inside the step), not free flight
    if contact_fraction <= 0.0:
        raise SystemExit("bench.py: no toe was in contact during the timed region (free flight) -- not a valid measurement")
    if resets == 0 and n * world * args.steps >= 50000:
        raise SystemExit("bench.py: no episode ended inside the timed region of %d env-steps -- not the steady-state workload" % (n * world * args.steps))
    # EXTRAS, untimed for `value`: the same K steps issued in the other ways (every step's outputs kept, as in the timed bracket), two
    # brackets each, the faster one reported; and the persistent launch that keeps only the last step's outputs (round 4's headline form)
    launch_modes = None
    if not args.no_mode_extras and args.steps <= 20000:
        launch_modes = {}
        for cand in ("persistent", "rows", "graph", "python"):
            t_c = min(bracket(cand) for _ in range(2))
            launch_modes[cand] = {"us_per_step": t_c * 1e6 / args.steps, "env_steps_per_sec": float(n) * args.steps / t_c}
        t_c = min(bracket("persistent", keep=False) for _ in range(2))
        launch_modes["persistent_last_step_outputs_only"] = {"us_per_step": t_c * 1e6 / args.steps, "env_steps_per_sec": float(n) * args.steps / t_c}
        launch_modes["what"] = ("wall clock of the same synchronize | %d steps | synchronize bracket on this rank, best of two, measured AFTER the timed bracket; "
                                "all outputs of every step kept except in persistent_last_step_outputs_only (irrl_env_step_rows_persistent: the form "
                                "BENCH_r04 timed)" % args.steps)
    check = None
    if args.check_steps > 0:
        # a longer window behind the timed region, in FIVE consecutive parts: mean and spread of us / step (the driver's --steps 20
        # bracket is 0.8 ms long; this says what the same kernel does over thousands of steps and how much that moves)
        torch.cuda.synchronize()
        parts = 5
        per = max(1, args.check_steps // parts)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(parts + 1)]
        cc0 = env.counters()
        evs[0].record()
        for i in range(parts):
            run(per)
            evs[i + 1].record()
        torch.cuda.synchronize()
        cc1 = env.counters()
        us = sorted(1e3 * evs[i].elapsed_time(evs[i + 1]) / per for i in range(parts))
        pe0, pe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pe0.record()
        run(per * parts, "persistent")
        pe1.record()
        torch.cuda.synchronize()
        check = {"what": "one launch per step (irrl_env_step_rows), five consecutive windows; persistent_us_per_step: the same number of steps as ONE launch",
                 "persistent_us_per_step": 1e3 * pe0.elapsed_time(pe1) / (per * parts),
                 "steps": per * parts, "us_per_step": 1e3 * evs[0].elapsed_time(evs[parts]) / (per * parts),
                 "us_per_step_min_median_max": [us[0], us[parts // 2], us[-1]], "windows": parts,
                 "contact_fraction": (cc1[1] - cc0[1]) / float(4 * loop_count * n * per * parts), "resets": cc1[0] - cc0[0]}

    # the same K steps issued the way the reference's runner issues them: ONE FlexibleGymEnv.step() call (ctypes -> C-ABI) per
    # control step (RaisimGymVecEnv.py:31), wall clock around the loop -- reported beside `value`, never as `value`
    per_call = None
    if world == 1:
        run(HOT_STEPS)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        run(args.steps, "python")
        torch.cuda.synchronize()
        per_call_s = time.perf_counter() - tc
        per_call = {"value": float(n) * args.steps / per_call_s, "unit": "env-steps/s", "us_per_step": 1e6 * per_call_s / args.steps,
                    "what": "%d steps, one FlexibleGymEnv.step() call per step on device tensors (the reference-shaped call surface)" % args.steps}

    # ... and through the COMPILED boundary (native/_flexible_robot: the pybind11 class of raisim_gym.cpp:14-46 over the C-ABI) with the same
    # device tensors: what a reference-side caller who keeps its compiled module gets.  Its own pool (same configuration, same
    # action stream, pre-rolled to the steady state), max(K, 500) steps, wall clock around the loop.
    per_call_native = None
    if per_call is not None and os.environ.get("IRRL_BENCH_NATIVE", "1") != "0":
        try:
            import importlib.util
            from high_speed_quadrupedal_locomotion_by_irrl_amd import build as hip_build
            spec = importlib.util.spec_from_file_location("_flexible_robot", hip_build.pybind_module_path())   # (the name PyInit__flexible_robot is looked up under)
            nat_mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(nat_mod)
            nat = nat_mod.FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(env_cfg), local_rank)
            nat.init()
            for k in range(preroll):
                nat.step(actions[k % rows], ob, rew, done, extra)
            kn = max(args.steps, 500)
            torch.cuda.synchronize()
            tn = time.perf_counter()
            for k in range(kn):
                nat.step(actions[(preroll + k) % rows], ob, rew, done, extra)
            torch.cuda.synchronize()
            tn = time.perf_counter() - tn
            per_call_native = {"value": float(n) * kn / tn, "unit": "env-steps/s", "us_per_step": 1e6 * tn / kn,
                               "what": "%d steps, one step() call per step on device tensors through the compiled pybind11 module (__cuda_array_interface__)" % kn}
            del nat
        except Exception as exc:      # the compiled module is an optional second call surface
            per_call_native = {"error": "%s: %s" % (type(exc).__name__, exc)}

    out = None
    if rank == 0:
        # HBM bytes per launch from the PMC passes (FETCH_SIZE x calibrated correction + WRITE_SIZE), collected with
        # rocprofv3 in separate runs (tools/gpu.sh pmc) and committed under profiles/ -- not measurable in-process
        traffic, issue, pmc_note = None, None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary_latest.json")))
            # the counters describe ONE binary: they are reported only when the summary carries the version string of the
            # library this process loaded (tools/pmc_workload.py records it on the GPU box), never for a later kernel
            if pmc.get("library") != _lib.version():
                pmc_note = "profiles/pmc_summary_latest.json was measured on %s, this run is %s: counters withheld" % (pmc.get("library"), _lib.version())
            elif int(pmc.get("envs", 4096)) == n and env.lanes_per_robot == 16 and not args.set and args.cfg == "bp5_imitation.yaml":
                if mode == "persistent":      # counters of the persistent kernel WITH every step's outputs kept (per step of its launches) x the steps of THIS launch
                    pp = pmc.get("persistent")
                    if pp and pp.get("outputs_kept"):
                        traffic = float(pp["hbm_bytes_per_step"]["total"]) * args.steps
                        pd = pp["derived"]
                        issue = {"valu_insts_per_wave_per_step": pd["valu_insts_per_wave_per_step"], "cycles_per_valu_inst": pd["cycles_per_valu_inst"],
                                 "frac_of_single_wave_issue_peak": 4.0 / pd["cycles_per_valu_inst"], "l2_hit_rate": pd["l2_hit_rate"],
                                 "source": "profiles/pmc_summary_latest.json (irrl_steps_persistent_kernel_flat_l16, launches of %d steps, every step's outputs kept)" % pp["steps_per_launch"]}
                    else:
                        pmc_note = "profiles/pmc_summary_latest.json has no counters of the persistent kernel with every step's outputs kept"
                else:
                    traffic = float(pmc["hbm_bytes_per_launch"]["total"])
                    # how close the single resident wave per SIMD runs to its issue limit of one VALU instruction per 4 cycles
                    issue = {"valu_insts_per_wave": pmc["derived"]["valu_insts_per_wave"], "cycles_per_valu_inst": pmc["derived"]["cycles_per_valu_inst"],
                             "frac_of_single_wave_issue_peak": 4.0 / pmc["derived"]["cycles_per_valu_inst"], "source": "profiles/pmc_summary_latest.json"}
        except Exception as e:
            pmc_note = "no usable profiles/pmc_summary_latest.json (%s)" % e
        total_env_steps = float(n) * world * args.steps
        value = total_env_steps / elapsed
        launch_s = kernel_ms * 1e-3         # per STEP (HIP events around the K steps / K)
        spl = args.steps if mode == "persistent" else 1     # steps one launch of the dominant kernel processes
        ach_gbs = ALG_BYTES_PER_ENV_STEP * n / launch_s / 1e9
        ach_tf = ALG_FLOPS_PER_ENV_STEP * n / launch_s / 1e12
        out = {
            "metric": "env-steps/sec (4096 envs per MI355X)", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: %d envs/GPU x 1 env.step (8 substeps @4 kHz + obs + reward + "
                                   "termination + in-step reset), cfg %s%s, actions clip(0.3 N(0,1)) from Philox(seed 1, stream env, counter step), "
                                   "%d-step untimed pre-roll before the warm-up, %d more untimed steps straight in front of the bracket's opening synchronize"
                                   % (n, args.cfg, (" with " + ", ".join(args.set)) if args.set else "", preroll, HOT_STEPS),
                       "envs_per_gpu": n, "global_envs": n * world, "parallelism": "env-sharded x%d" % world,
                       "lanes_per_robot": env.lanes_per_robot, "waves_per_simd": env.waves_per_simd, "preroll": preroll, "hot_steps_before_bracket": HOT_STEPS,
                       "launch": mode,
                       "launch_what": {"persistent": "ONE launch: every wave walks its own robots through the %d steps (irrl_env_step_rows_persistent_out; "
                                                     "robots never interact, VEC:273 -- no grid-wide boundary between steps)" % args.steps,
                                       "rows": "%d back-to-back launches from one irrl_env_step_rows_out call" % args.steps,
                                       "graph": "one hipGraph of %d step-kernel nodes" % args.steps,
                                       "python": "one ctypes call per step"}[mode] + "; fixed by --launch, nothing probed or selected inside the run",
                       "outputs": "every one of the %d steps stores its ob [N,35] / reward [N] / done [N] / extraInfo [N,6] to its own row of [K,N,.] tables "
                                  "in HBM (173 B per env-step): what K step() calls of the reference return (VEC:268-278)" % args.steps},
            # the binding roofline of this path is the FP32 vector ALU (~78 flop per algorithmic byte), so THAT is `roofline`;
            # `traffic` = HBM bytes of one launch of the timed kernel from the PMC passes (hash-gated file under profiles/)
            "roofline": {"bound": "valu_fp32", "achieved": ach_tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / FP32_PEAK_TFLOPS,
                         "traffic": traffic,
                         "kernel": ("irrl_steps_persistent_kernel" if mode == "persistent" else "irrl_step_kernel") + ("" if env_cfg.get("Terrain") else "_flat")
                                   + "_l%d" % env.lanes_per_robot + ("w2" if env.waves_per_simd == 2 else ""),
                         "steps_per_launch": spl, "avg_launch_us": kernel_ms * 1e3 * spl, "avg_step_us": kernel_ms * 1e3,
                         "algorithmic_flops_per_launch": ALG_FLOPS_PER_ENV_STEP * n * spl, "algorithmic_bytes_per_launch": ALG_BYTES_PER_ENV_STEP * n * spl,
                         "valu_issue": issue,
                         "note": "latency / issue bound: one wave per SIMD at 4096 envs; flop constant = the instrumented oracle's exact count"},
            "roofline_hbm": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
                             "traffic": traffic, "algorithmic_bytes_per_launch": ALG_BYTES_PER_ENV_STEP * n * spl,
                             "counter_traffic_over_algorithmic": (traffic / (ALG_BYTES_PER_ENV_STEP * n * spl)) if traffic else None,
                             "note": "not the binding roofline; kept so that wasted re-reads would show"},
            "launch_modes": launch_modes,
            "contact_fraction_in_timed_region": contact_fraction, "resets_in_timed_region": resets,
            "steady_state_check": check, "rccl_ranks_seen": ranks_seen, "backend": backend if dist is not None else None,
            "library": lib.irrl_version().decode(), "pmc_note": pmc_note, "per_step_call": per_call, "per_step_call_compiled": per_call_native,
            "contact_solver": int(env_cfg.get("ContactSolver", 3)),
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(env_cfg, args.cpu_seconds)
    if args.ppo_iters > 0:
        # second half of BASELINE.json's metric ("PPO iters/sec"): the reference's training iteration (750-step rollout of
        # all envs with the 2x48 + 2x48 LSTM policy, GAE, 10 epochs of full-length BPTT, global reset) on every rank, gradients
        # and advantage moments all-reduced per optimizer step; reported beside the headline value, never mixed into it
        del env
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import ppo_bench
        def ppo_leg(policy, cfg_name, grad_note, precision=None):
            ppo = ppo_bench.measure(policy, n, args.ppo_steps, args.ppo_iters + 1, args.ppo_epochs, cfg_name, verbose=False, rank=rank, precision=precision)
            if dist is not None:
                t = torch.tensor([ppo["rollout_s"], ppo["update_s"], ppo["rollout_s"] + ppo["update_s"]], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ppo["rollout_s"], ppo["update_s"], it_s = (float(x) for x in t.tolist())
            else:
                it_s = ppo["rollout_s"] + ppo["update_s"]
            ppo.update({"world": world, "global_envs": n * world, "ppo_iters_per_sec": 1.0 / it_s, "samples_per_sec": n * world * args.ppo_steps / it_s,
                        "env_steps_per_sec_in_rollout": n * world * args.ppo_steps / ppo["rollout_s"], "cfg": cfg_name,
                        "collectives_per_optimizer_step": None if dist is None else grad_note})
            return ppo
        # every learner leg twice: at the learner's DEFAULT arithmetic (bf16x3: two bf16 planes per operand, ~2^-16 per product) and at the
        # f32 level (bf16x6 for the LSTM sequence kernels = three planes, ~2^-24; the exact-f32 MFMA kernels for the MlpPolicy gradients)
        f32_level = os.environ.get("IRRL_BENCH_F32_LEVEL", "1") != "0"
        ppo = ppo_leg("lstm", "default_cfg.yaml", "all-reduce of the flat gradient (283 KB) + 3-float advantage moments")
        if f32_level:
            ppo["f32_level"] = ppo_leg("lstm", "default_cfg.yaml", "all-reduce of the flat gradient (283 KB) + 3-float advantage moments", precision="bf16x6")
        # BASELINE config 2's learner beside it: MlpPolicy [64, 64] on the imitation-only config (4 minibatches x 10 epochs)
        ppo_mlp = ppo_leg("mlp", "bp5_imitation.yaml", "all-reduce of the flat gradient (56 KB) + 3-float advantage moments")
        if f32_level:
            ppo_mlp["f32_level"] = ppo_leg("mlp", "bp5_imitation.yaml", "all-reduce of the flat gradient (56 KB) + 3-float advantage moments", precision="f32")
        if out is not None:
            out["ppo"] = ppo
            out["ppo_mlp"] = ppo_mlp
    if out is not None:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and not launched_by_torchrun():
        raise SystemExit(launch_ranks(args))
    worker(args)


if __name__ == "__main__":
    main()
