"""Measurement harness on the MI355X: the device action generator against its numpy twin, the kernels' diagnostic counters,
and `bench.py --gpus 2` launching two ranks by itself (both on cuda:0 over gloo: IRRL_BENCH_ONE_DEVICE / IRRL_BENCH_BACKEND),
with the PPO leg's gradient all-reduce in the loop."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import ROOT, load_env_cfg

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_device_action_stream_matches_the_numpy_twin():
    import torch
    from bench_actions import bench_actions
    from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
    lib = _lib.load()
    out = torch.empty(7, 300, 12, device="cuda")
    _lib.check(lib.irrl_bench_actions(1, 4096, 300, 5, 7, 0.3, C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    want = bench_actions(1, 4096, 300, 5, 7, 0.3)
    np.testing.assert_allclose(out.cpu().numpy(), want, atol=2e-6)


def test_step_rows_equals_the_same_steps_one_by_one():
    """irrl_env_step_rows (K back-to-back launches from one C call, action row = launch argument, rows wrap around) leaves the
    pool and the outputs bit-identical to K irrl_env_step calls -- for a plain pool and for a Crutial one (its own step kernel)"""
    import torch
    from hip_env import HipVecEnv
    n, K, rows = 96, 37, 16
    for over in ({}, {"Crutial": True}):
        a, b = HipVecEnv(load_env_cfg("default_cfg.yaml", num_envs=n, **over)), HipVecEnv(load_env_cfg("default_cfg.yaml", num_envs=n, **over))
        g = torch.Generator(device="cuda").manual_seed(5)
        table = (0.3 * torch.randn(rows, n, 12, device="cuda", generator=g)).clamp(-1, 1)
        outs = []
        for env in (a, b):
            outs.append((torch.zeros(n, 35, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, dtype=torch.bool, device="cuda"), torch.zeros(n, 6, device="cuda")))
        a.impl.step_rows(K, table, 11, *outs[0])
        for k in range(K):
            b.impl.step(table[(11 + k) % rows], *outs[1])
        torch.cuda.synchronize()
        for x, y in zip(outs[0], outs[1]):
            assert torch.equal(x, y)
        np.testing.assert_array_equal(a.get_state(), b.get_state())
        with pytest.raises(RuntimeError, match="step_rows"):
            a.impl.step_rows(-1, table, 0, *outs[0])


@pytest.mark.parametrize("n,cfg_name,over", [(4096, "bp5_imitation.yaml", {}), (330, "default_cfg.yaml", {}), (8192, "bp5_terrain.yaml", {}),
                                             (96, "default_cfg.yaml", {"Crutial": True}), (96, "default_cfg.yaml", {"ContactSolver": 1}),
                                             (96, "default_cfg.yaml", {"ContactIterations": 5}), (96, "bp5_imitation.yaml", {"ContactTolerance": 0.0})])
def test_persistent_multi_step_launch_equals_back_to_back_launches(n, cfg_name, over):
    """irrl_env_step_rows_persistent (ONE launch: every wave walks its own robots through all K steps, no grid-wide boundary between
    steps) leaves the pool and the outputs bit-identical to irrl_env_step_rows (K launches): 400 steps with in-step resets, in the
    16-lane layout (4096 and a ragged 330), the 4-lane layout (8192 envs, rough ground), a Crutial pool and pools whose solver settings are not the
    ones compiled into the multi-step kernel (Gauss-Seidel order, another sweep cap, no tolerance): the launcher's fallback to one launch per step."""
    import torch
    from hip_env import HipVecEnv
    K, rows = 400, 64
    a, b = HipVecEnv(load_env_cfg(cfg_name, num_envs=n, **over)), HipVecEnv(load_env_cfg(cfg_name, num_envs=n, **over))
    assert a.impl.lanes_per_robot == (4 if n > 4096 else 16)
    g = torch.Generator(device="cuda").manual_seed(5)
    table = (0.5 * torch.randn(rows, n, 12, device="cuda", generator=g)).clamp(-1, 1)
    outs = [(torch.zeros(n, 35, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, dtype=torch.bool, device="cuda"), torch.zeros(n, 6, device="cuda"))
            for _ in range(2)]
    c0 = a.impl.counters()
    a.impl.step_rows(K, table, 11, *outs[0], persistent=True)
    b.impl.step_rows(K, table, 11, *outs[1])
    torch.cuda.synchronize()
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)
    np.testing.assert_array_equal(a.get_state(), b.get_state())
    assert a.impl.counters()[0] - c0[0] > 0          # episodes ended and restarted inside the launch
    with pytest.raises(RuntimeError, match="step_rows"):
        a.impl.step_rows(-1, table, 0, *outs[0], persistent=True)


@pytest.mark.parametrize("n,cfg_name,over", [(4096, "bp5_imitation.yaml", {}), (330, "default_cfg.yaml", {}), (8192, "bp5_terrain.yaml", {}),
                                             (96, "default_cfg.yaml", {"Crutial": True})])
def test_multi_step_entry_points_return_every_steps_outputs(n, cfg_name, over):
    """irrl_env_step_rows_out / irrl_env_step_rows_persistent_out: row k of ob [K,N,35], reward [K,N], done [K,N], extraInfo [K,N,6] is
    what the k-th of K irrl_env_step calls returns (VEC:268-278 fills them on EVERY control step; RaisimGymVecEnv.py:26-52) -- bit for
    bit, in both lane layouts, with in-step resets, for one launch per step and for the ONE persistent launch; and the pool ends equal."""
    import torch
    from hip_env import HipVecEnv
    K, rows = 120, 32
    envs = [HipVecEnv(load_env_cfg(cfg_name, num_envs=n, **over)) for _ in range(3)]
    g = torch.Generator(device="cuda").manual_seed(9)
    table = (0.6 * torch.randn(rows, n, 12, device="cuda", generator=g)).clamp(-1, 1)
    for env in envs:                                      # five robots start below the termination height: in-step resets at step 0 for certain
        st = env.get_state()
        st[:5, 2] = 0.1
        env.set_state(st)
    def outs(lead):
        return (torch.full(lead + (n, 35), float("nan"), device="cuda"), torch.full(lead + (n,), float("nan"), device="cuda"),
                torch.zeros(lead + (n,), dtype=torch.bool, device="cuda"), torch.full(lead + (n, 6), float("nan"), device="cuda"))
    pers, back = outs((K,)), outs((K,))
    envs[0].impl.step_rows(K, table, 5, *pers, persistent=True)
    envs[1].impl.step_rows(K, table, 5, *back)
    one = outs(())
    want = outs((K,))
    for k in range(K):
        envs[2].impl.step(table[(5 + k) % rows], *one)
        for w, o in zip(want, one):
            w[k].copy_(o)
    torch.cuda.synchronize()
    for name, p, b, w in zip(("ob", "reward", "done", "extraInfo"), pers, back, want):
        assert torch.equal(p, w), "persistent launch: %s rows differ from K step() calls" % name
        assert torch.equal(b, w), "one launch per step: %s rows differ from K step() calls" % name
    assert want[2][0, :5].all(), "the forced terminations of step 0 are missing from row 0"
    np.testing.assert_array_equal(envs[0].get_state(), envs[2].get_state())
    np.testing.assert_array_equal(envs[1].get_state(), envs[2].get_state())
    with pytest.raises(TypeError):                      # a [K, N, .] request with one array of the wrong leading size
        envs[0].impl.step_rows(K, table, 0, pers[0], pers[1][:-1].contiguous(), pers[2], pers[3], persistent=True)


def test_counters_see_landing_and_resets():
    from hip_env import HipVecEnv
    n = 64
    env = HipVecEnv(load_env_cfg("bp5_imitation.yaml", num_envs=n))
    ep0, cc0, fr0 = env.impl.counters()
    assert (ep0, cc0, fr0) == (n, 0, n)                      # one episode per env, nothing has touched the ground, frame_idx = 1
    a = np.zeros((n, 12), np.float32)
    for _ in range(30):                                      # 4 cm of free fall take ~46 control steps
        env.step(a)
    assert env.impl.counters()[1] == 0
    for _ in range(70):
        env.step(a)
    ep1, cc1, fr1 = env.impl.counters()
    inc = env.get_state()[:, 147:151]
    assert cc1 > 0 and cc1 <= 4 * 8 * n * 70 and inc.sum() > 0
    # one more step: the counter grows by at least the feet that stay down for all 8 substeps and at most by 8 per toe
    env.step(a)
    cc2 = env.impl.counters()[1]
    assert 0 < cc2 - cc1 <= 8 * 4 * n
    st = env.get_state()
    st[:5, 2] = 0.1                                           # below the termination height: five in-step resets
    env.set_state(st)
    env.step(a)
    assert env.impl.counters()[0] == ep1 + 5


def _run_bench(extra, env_over):
    env = dict(os.environ)
    env.update(env_over)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_single_gpu_line_is_steady_state():
    out = _run_bench(["--steps", "20", "--warmup", "5", "--ppo-iters", "0", "--cpu-seconds", "0", "--check-steps", "200"], {})
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["warmup"] == 5
    assert out["contact_fraction_in_timed_region"] > 0.1 and out["resets_in_timed_region"] > 0
    assert out["config"]["preroll"] >= 200
    # the launch mode is FIXED by the flag (default: the persistent launch with every step's outputs kept); nothing is selected inside the run
    assert out["config"]["launch"] == "persistent" and "every one of the 20 steps stores" in out["config"]["outputs"]
    # the 20-step timed region and the 200-step check window measure the same regime
    r, chk = out["roofline"], out["steady_state_check"]
    assert r["bound"] == "valu_fp32" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["steps_per_launch"] == 20 and abs(r["avg_launch_us"] - 20 * r["avg_step_us"]) < 1e-6 * r["avg_launch_us"]
    assert r["algorithmic_bytes_per_launch"] == 1521.0 * 4096 * 20 and r["algorithmic_flops_per_launch"] == 1.18e5 * 4096 * 20
    assert abs(r["achieved"] - 1.18e5 * 4096 / (r["avg_step_us"] * 1e-6) / 1e12) < 1e-6 * r["achieved"]
    assert out["roofline_hbm"]["bound"] == "hbm"
    assert abs(r["avg_step_us"] - chk["persistent_us_per_step"]) < 0.15 * chk["persistent_us_per_step"]
    # the other ways of issuing the same steps are extras, all four + round 4's last-step-only form
    lm = out["launch_modes"]
    assert {"persistent", "rows", "graph", "python", "persistent_last_step_outputs_only"} <= set(lm)
    assert all(lm[k]["env_steps_per_sec"] > 5e7 for k in ("persistent", "rows", "graph", "python"))


def test_bench_every_launch_mode_gives_a_line():
    for mode in ("persistent", "rows", "graph", "python"):
        out = _run_bench(["--steps", "20", "--warmup", "5", "--ppo-iters", "0", "--cpu-seconds", "0", "--check-steps", "0", "--launch", mode, "--no-mode-extras"],
                         {"IRRL_BENCH_NATIVE": "0"})
        assert out["config"]["launch"] == mode and out["launch_modes"] is None
        assert out["roofline"]["steps_per_launch"] == (20 if mode == "persistent" else 1) and out["value"] > 5e7, (mode, out["value"])


def test_bench_gpus_2_runs_two_ranks_with_the_ppo_collectives():
    out = _run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--envs", "512", "--cpu-seconds", "0", "--check-steps", "0",
                      "--ppo-iters", "1", "--ppo-steps", "32", "--ppo-epochs", "2"],
                     {"IRRL_BENCH_BACKEND": "gloo", "IRRL_BENCH_ONE_DEVICE": "1"})
    assert out["n_gpus"] == 2 and out["rccl_ranks_seen"] == 2 and out["config"]["global_envs"] == 1024
    assert out["ppo"]["world"] == 2 and out["ppo"]["global_envs"] == 1024 and out["ppo"]["ppo_iters_per_sec"] > 0
