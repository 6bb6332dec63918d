"""CPU-side checks of the measurement harness and the build recipe: the content-hash staleness rule of build.py, the
numpy twin of the benchmark's Philox action stream, and bench.py's self-launch of N ranks (no GPU work here)."""
import os
import shutil
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_prebuilt_library_carries_the_hash_of_the_sources_next_to_it():
    from high_speed_quadrupedal_locomotion_by_irrl_amd import build
    build.build()
    assert build.embedded_hash() == build.source_hash()
    assert not build.is_stale()


def test_touched_source_forces_a_rebuild(tmp_path):
    """staleness is decided by content: a one-byte change in any csrc/ file (or another flag set) makes the shipped .so stale,
    while copying the tree (new mtimes, same bytes) does not"""
    from high_speed_quadrupedal_locomotion_by_irrl_amd import build
    build.build()
    copy = tmp_path / "csrc"
    shutil.copytree(build.CSRC, str(copy), ignore=shutil.ignore_patterns("_obj"))
    assert not build.is_stale(csrc=str(copy))                       # fresh mtimes, identical content: reuse
    with open(str(copy / "env_core.hpp"), "a") as f:
        f.write("// touched\n")
    assert build.is_stale(csrc=str(copy))                           # content changed: rebuild
    assert build.is_stale(extra_flags=("-DIRRL_GS_FASTPATH",))      # different flags: rebuild
    assert build.embedded_hash(str(tmp_path / "missing.so")) is None


def test_version_string_names_the_hash():
    from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib, build
    v = _lib.load().irrl_version().decode()
    assert v.startswith("gfx950;") and v.endswith("irrl-src-hash:" + build.source_hash())


def test_action_stream_depends_only_on_seed_env_and_step():
    from bench_actions import bench_actions, philox4x32_10
    # Philox4x32-10 known-answer vectors (Random123 kat_vectors: counter / key all zero, and the pi digits case)
    r = philox4x32_10(np.uint32(0), np.uint32(0), np.uint32(0), np.uint32(0), 0, 0)
    assert [int(x) for x in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = philox4x32_10(np.uint32(0x243f6a88), np.uint32(0x85a308d3), np.uint32(0x13198a2e), np.uint32(0x03707344), 0xa4093822, 0x299f31d0)
    assert [int(x) for x in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    a = bench_actions(1, 0, 64, 0, 10)
    assert a.shape == (10, 64, 12) and a.dtype == np.float32 and np.abs(a).max() <= 1.0
    b = bench_actions(1, 32, 8, 4, 3)                      # a window of envs / steps: same numbers
    assert np.array_equal(b, a[4:7, 32:40])
    assert not np.array_equal(bench_actions(2, 0, 64, 0, 10), a)
    big = bench_actions(1, 0, 4096, 0, 8)
    assert abs(big.std() - 0.3) < 0.01 and abs(big.mean()) < 1e-3   # clip(0.3 N(0,1)): clipping at 3.3 sigma is invisible


def test_bench_gpus_n_launches_n_ranks_by_itself(monkeypatch):
    """`python bench.py --gpus N` outside a launcher starts torch.distributed.run with N ranks on 127.0.0.1 and relays its exit
    code; under a launcher (WORLD_SIZE, RANK and MASTER_PORT set) it runs the worker instead."""
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    for k in ("WORLD_SIZE", "RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    (cmd, env), = calls
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher: no second launch
    calls.clear()
    ran = []
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("MASTER_PORT", "29500")
    monkeypatch.setattr(bench, "worker", lambda args: ran.append(args.gpus))
    bench.main()
    assert ran == [4] and not calls
    # a shell that merely exports WORLD_SIZE (no RANK / MASTER_PORT) is NOT a launcher: one process stays single ...
    monkeypatch.delenv("RANK")
    monkeypatch.delenv("MASTER_PORT")
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert not bench.launched_by_torchrun()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    ran.clear()
    bench.main()
    assert ran == [1] and not calls
    # ... and --gpus 4 still starts its own ranks
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert len(calls) == 1


def test_baseline_config1_plumbing_runs_through_the_training_script_on_cpu():
    """BASELINE config 1 ("64 parallel envs, CPU reference path via run_bp_v5.py --train ... plumbing"): the training script's whole
    control flow -- argument parsing, config dump / re-parse, VecEnv construction, PPO2(CustomLSTMPolicy) with the reference's
    hyper-parameters, rollouts, GAE, 10 epochs of BPTT, logging -- with 64 envs on CPU, the env class bound to the oracle-backed test
    double (the RaiSim path is closed source and absent; the product package itself has no CPU path).  A 10-step rollout instead of
    750 keeps it to seconds; tools/cpu_config1.py times the full-size version (profiles/r02_config1_cpu_*.json)."""
    import os, sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import cpu_config1
    out = cpu_config1.plumbing(2, 2, max_time=0.02)
    assert out["iterations"] == 2 and out["n_steps"] == 10 and len(out["log"]) == 2
    assert [r["nupdates"] for r in out["log"]] == [1, 2]
    import math
    assert all(math.isfinite(r["value_loss"]) for r in out["log"]) and out["samples_per_sec"] > 0
