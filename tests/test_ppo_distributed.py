"""N > 1 path on CPU: world_size-2 gloo processes (the driver covers RCCL on the 8-GPU node).
Checks SURVEY 8e: with equal env shards and nminibatches = 1 the data-parallel update (flat-gradient
all-reduce + 3-float advantage-moment all-reduce, clip after averaging) equals the single-process update on
the concatenated batch, and every rank ends with identical parameters."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_batch(policy_kind, T, N, seed):
    g = torch.Generator().manual_seed(seed)
    b = dict(obs=torch.randn(T, N, 35, generator=g), returns=torch.randn(T, N, generator=g), masks=torch.rand(T, N, generator=g) < 0.1,
             actions=torch.randn(T, N, 12, generator=g) * 0.3, values=torch.randn(T, N, generator=g),
             neglogpacs=torch.rand(T, N, generator=g) * 3 + 8)
    b["states"] = torch.randn(N, 384, generator=g) * 0.1 if policy_kind == "lstm" else None
    return b


def _slice(batch, lo, hi):
    out = {k: (v[:, lo:hi] if v is not None and k != "states" else v) for k, v in batch.items()}
    if batch["states"] is not None:
        out["states"] = batch["states"][lo:hi]
    return out


def _new_model(policy_kind, nminibatches=1):
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy, MlpPolicy
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2
    pol = (CustomLSTMPolicy if policy_kind == "lstm" else MlpPolicy)
    return PPO2(policy=pol, env=None, n_steps=6, nminibatches=nminibatches, noptepochs=2, learning_rate=1e-3, cliprange=0.2, ent_coef=0.0,
                vf_coef=0.5, max_grad_norm=0.5, seed=3, device="cpu")


def _worker(rank, world, port, policy_kind, out_dir, nminibatches=1):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    T, N = 6, 8
    full = _make_batch(policy_kind, T, N, seed=11)
    model = _new_model(policy_kind, nminibatches)
    model.n_envs = N // world
    shard = _slice(full, rank * N // world, (rank + 1) * N // world)
    # nminibatches > 1: ONE permutation over all ranks' samples (resp. envs), every rank keeps the members it owns (`_global_minibatches`)
    losses = model.update(shard, 1e-3, 0.2)
    params = np.concatenate([p.reshape(-1) for p in model.get_parameter_list()])
    np.save(os.path.join(out_dir, "params_%d.npy" % rank), params)
    np.save(os.path.join(out_dir, "loss_%d.npy" % rank), losses.numpy())
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("policy_kind", ["mlp", "lstm"])
def test_two_rank_update_equals_single_process(tmp_path, policy_kind):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, policy_kind, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "params_0.npy"), np.load(tmp_path / "params_1.npy")
    np.testing.assert_array_equal(p0, p1)  # replicas stay bit-identical
    sys.path.insert(0, ROOT)
    single = _new_model(policy_kind)
    single.n_envs = 8
    single.update(_make_batch(policy_kind, 6, 8, seed=11), 1e-3, 0.2)
    ps = np.concatenate([p.reshape(-1) for p in single.get_parameter_list()])
    # same gradient up to summation order (mean over shards of shard-means == full mean for equal shards)
    np.testing.assert_allclose(p0, ps, atol=2e-6)
    assert np.abs(ps - np.concatenate([p.reshape(-1) for p in _new_model(policy_kind).get_parameter_list()])).max() > 1e-4


@pytest.mark.parametrize("policy_kind,nminibatches", [("mlp", 4), ("lstm", 2), ("lstm", 4)])
def test_two_rank_update_with_several_minibatches_equals_single_process(tmp_path, policy_kind, nminibatches):
    """ppo2.py:364-380 / 387-402: the reference shuffles ALL samples (MlpPolicy: the shipped config 2 uses 4 minibatches) resp. ALL envs
    (recurrent) once per epoch and cuts the order into minibatches.  Two ranks draw that one global order, each keeps the members it
    owns (unequal shares, weighted m_r * world / m before the all-reduce; advantage moments all-reduced per minibatch): same parameters
    as the single process on the concatenated batch up to summation order, replicas bit-identical, the logged loss means equal too.
    (("lstm", 4): 8 envs in 4 minibatches of 2 -- some minibatches live entirely on one rank: the other joins the collectives empty.)"""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, policy_kind, str(tmp_path), nminibatches), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "params_0.npy"), np.load(tmp_path / "params_1.npy")
    np.testing.assert_array_equal(p0, p1)
    sys.path.insert(0, ROOT)
    single = _new_model(policy_kind, nminibatches)
    single.n_envs = 8
    ls = single.update(_make_batch(policy_kind, 6, 8, seed=11), 1e-3, 0.2)
    ps = np.concatenate([p.reshape(-1) for p in single.get_parameter_list()])
    np.testing.assert_allclose(p0, ps, atol=3e-6)
    np.testing.assert_allclose(np.load(tmp_path / "loss_0.npy"), ls.numpy(), rtol=2e-4, atol=2e-6)
    np.testing.assert_array_equal(np.load(tmp_path / "loss_0.npy"), np.load(tmp_path / "loss_1.npy"))


def _learn_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from conftest import load_env_cfg
    from oracle_torch_env import OracleTorchEnv
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2
    env = OracleTorchEnv(load_env_cfg("default_cfg.yaml", num_envs=4, EnvIdOffset=4 * rank))   # rank r owns the global env ids 4 r .. 4 r + 3
    model = PPO2(policy=CustomLSTMPolicy, env=env, n_steps=10, nminibatches=1, noptepochs=2, gamma=0.99, lam=0.998, ent_coef=0.0,
                 learning_rate=1e-3, vf_coef=0.5, max_grad_norm=0.5, cliprange=0.2, verbose=1, seed=5)
    model.learn(total_timesteps=2 * 10 * 4 * world, eval_every_n=0)
    params = np.concatenate([p.reshape(-1) for p in model.get_parameter_list()])
    np.save(os.path.join(out_dir, "learn_%d.npy" % rank), params)
    np.save(os.path.join(out_dir, "log_%d.npy" % rank), np.array([r["policy_loss"] for r in model.log]))
    torch.distributed.destroy_process_group()


def test_two_rank_learn_loop_keeps_replicas_in_sync(tmp_path):
    port = _free_port()
    mp.spawn(_learn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "learn_0.npy"), np.load(tmp_path / "learn_1.npy")
    np.testing.assert_array_equal(p0, p1)
    assert len(np.load(tmp_path / "log_0.npy")) == 2 and np.isfinite(p0).all()


def _replica_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from conftest import load_env_cfg
    from oracle_torch_env import OracleTorchEnv
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2
    env = OracleTorchEnv(load_env_cfg("default_cfg.yaml", num_envs=4, EnvIdOffset=4 * rank))
    hp = dict(n_steps=10, nminibatches=1, noptepochs=2, gamma=0.99, lam=0.998, ent_coef=0.0, learning_rate=1e-3, vf_coef=0.5, max_grad_norm=0.5,
              cliprange=0.2, verbose=0)
    notes = {}
    # (0) ranks that disagree on the seed are refused (they would cut different global minibatches)
    try:
        PPO2(policy=CustomLSTMPolicy, env=env, seed=5 + rank, **hp)
        notes["seed_refused"] = np.array(0)
    except ValueError as exc:
        notes["seed_refused"] = np.array(int("different seeds" in str(exc)))
    # (1) DIFFERENT initial weights per rank (a policy object built from a rank-local generator state): construction broadcasts rank 0's
    torch.manual_seed(1000 + rank)
    model = PPO2(policy=CustomLSTMPolicy(n_lstm=[48, 48]), env=env, seed=5, **hp)
    notes["start"] = np.concatenate([p.reshape(-1) for p in model.get_parameter_list()])
    assert model.check_replicas()
    # (2) a rank-local change of the parameters is DETECTED, on every rank
    if rank == 1:
        with torch.no_grad():
            model.policy.pi.w.add_(1e-3)
    try:
        model.check_replicas()
        notes["detected"] = np.array(0)
    except RuntimeError as exc:
        notes["detected"] = np.array(1)
        assert "diverged" in str(exc)
    # (3) run_bp_v5.py:245-248 under a launcher with the checkpoint on rank 0 ONLY: rank 0 reads, everyone receives
    ckpt = os.path.join(out_dir, "only_rank0_sees_this.pkl")
    torch.distributed.barrier()
    if rank == 0:
        rng = np.random.RandomState(3)
        import pickle
        params = [a + 0.01 * rng.standard_normal(a.shape).astype(np.float32) for a in model.get_parameter_list()]
        with open(ckpt, "wb") as f:
            pickle.dump((model._data(), params), f)
    torch.distributed.barrier()
    loaded = PPO2.load(ckpt if rank == 0 else os.path.join(out_dir, "no_such_file_on_rank_%d.pkl" % rank), env=env, verbose=0)
    loaded.seed = 5
    assert loaded.check_replicas()
    notes["loaded"] = np.concatenate([p.reshape(-1) for p in loaded.get_parameter_list()])
    # (4) ... and trains in sync from there (learn's own periodic check included: REPLICA_CHECK_EVERY = 1 here)
    loaded.REPLICA_CHECK_EVERY = 1
    loaded.learn(total_timesteps=2 * 10 * 4 * world, eval_every_n=0)
    notes["trained"] = np.concatenate([p.reshape(-1) for p in loaded.get_parameter_list()])
    np.savez(os.path.join(out_dir, "replica_%d.npz" % rank), **notes)
    torch.distributed.destroy_process_group()


def test_replicas_are_broadcast_at_construction_and_after_load_and_divergence_is_detected(tmp_path):
    """Verdict r5 item 6: the N-rank learner no longer RELIES on identical seeds / identical files.  Rank 0's parameters (and Adam moments)
    are broadcast at construction and after `load_parameters`; `PPO2.load` under a process group reads the file on rank 0 only; a 64-bit
    checksum is compared across the ranks (`check_replicas`, every REPLICA_CHECK_EVERY updates of `learn`) and a mismatch raises."""
    port = _free_port()
    mp.spawn(_replica_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = dict(np.load(tmp_path / "replica_0.npz")), dict(np.load(tmp_path / "replica_1.npz"))
    assert int(r0["seed_refused"]) == 1 and int(r1["seed_refused"]) == 1
    np.testing.assert_array_equal(r0["start"], r1["start"])           # different initial weights, same start: rank 0's
    assert int(r0["detected"]) == 1 and int(r1["detected"]) == 1      # both ranks saw the divergence
    np.testing.assert_array_equal(r0["loaded"], r1["loaded"])         # the file only rank 0 could read
    assert np.abs(r0["loaded"] - r0["start"]).max() > 1e-3            # ... and it was really loaded
    np.testing.assert_array_equal(r0["trained"], r1["trained"])
    assert np.abs(r0["trained"] - r0["loaded"]).max() > 1e-5 and np.isfinite(r0["trained"]).all()


def _run_workers(tmp_path, device, world, envs, steps, policy, cfg="default_cfg.yaml", nminibatches=1, backend="gloo"):
    """start `world` processes of tests/two_rank_ppo_worker.py (gloo on 127.0.0.1; backend "nccl": RCCL, one rank only on a 1-GPU box) and
    return their saved dicts"""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_rank_ppo_worker.py"), "--device", device, "--envs", str(envs),
                                       "--steps", str(steps), "--policy", policy, "--cfg", cfg, "--out", str(tmp_path), "--backend", backend,
                                       "--nminibatches", str(nminibatches)], env=env))
    for p in procs:
        assert p.wait(timeout=900) == 0
    return [dict(np.load(os.path.join(str(tmp_path), "rank%dof%d.npz" % (r, world)))) for r in range(world)]


def check_two_ranks_equal_single_process(tmp_path, device, envs, steps, policy, cfg="default_cfg.yaml", nminibatches=1):
    """N-rank job == the single-process job on the concatenated pool: the ranks' rollouts ARE the halves of the big rollout (bit for
    bit: global env ids address every random draw -- env RNG and sampling noise), and the data-parallel update gives the same
    parameters up to the summation order of the gradient.  (Exact equality of the rollouts is asserted on the GPU path.)"""
    two = _run_workers(tmp_path, device, 2, envs, steps, policy, cfg, nminibatches)
    one = _run_workers(tmp_path, device, 1, 2 * envs, steps, policy, cfg, nminibatches)[0]
    for key in ("obs", "actions", "values", "neglogpacs", "returns", "masks"):
        a = np.concatenate([two[0][key], two[1][key]], axis=1)      # [T, N, ...]
        assert a.shape == one[key].shape, key
        err = np.abs(a.astype(np.float64) - one[key].astype(np.float64)).max()
        if device == "cuda" or key == "masks":
            # the HIP kernels compute every env row on its own (an MFMA row's accumulation order does not depend on its neighbours)
            assert np.array_equal(a, one[key]), (key, err)
        else:
            # CPU BLAS blocks a [3, k] and a [6, k] product differently: last-bit differences in the policy outputs, same samples
            assert err < 2e-4 * (1.0 + np.abs(one[key]).max()), (key, err)
    np.testing.assert_array_equal(two[0]["params"], two[1]["params"])                 # replicas bit-identical
    np.testing.assert_allclose(two[0]["params"], one["params"], atol=3e-6)              # same update up to summation order
    return two, one


@pytest.mark.parametrize("policy,nminibatches", [("lstm", 1), ("mlp", 1), ("mlp", 4)])
def test_two_rank_iteration_equals_single_process_on_the_concatenated_pool(tmp_path, policy, nminibatches):
    check_two_ranks_equal_single_process(tmp_path, "cpu", 3, 12, policy, nminibatches=nminibatches)


def test_single_process_learn_with_mlp_and_lstm():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from conftest import load_env_cfg
    from oracle_torch_env import OracleTorchEnv
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy, MlpPolicy
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2
    for pol, nmb in ((MlpPolicy, 4), (CustomLSTMPolicy, 2)):
        env = OracleTorchEnv(load_env_cfg("default_cfg.yaml", num_envs=4))
        model = PPO2(policy=pol, env=env, n_steps=12, nminibatches=nmb, noptepochs=2, gamma=0.99, lam=0.998, ent_coef=0.0,
                     learning_rate=1e-3, verbose=0, seed=1)
        before = np.concatenate([p.reshape(-1) for p in model.get_parameter_list()])
        model.learn(total_timesteps=2 * 12 * 4, eval_every_n=0)
        after = np.concatenate([p.reshape(-1) for p in model.get_parameter_list()])
        assert np.isfinite(after).all() and np.abs(after - before).max() > 1e-5
        assert model.num_timesteps == 2 * 12 * 4


def _overlap_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from conftest import load_env_cfg
    from oracle_torch_env import OracleTorchEnv
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import MlpPolicy
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2
    msg = "ok"
    try:
        # a launcher that forgot EnvIdOffset: both pools claim the global env ids 0 .. 3
        PPO2(policy=MlpPolicy, env=OracleTorchEnv(load_env_cfg("default_cfg.yaml", num_envs=4)), n_steps=4, nminibatches=1, seed=5)
    except ValueError as e:
        msg = str(e)
    # the same through set_env(): a model built without an env gets its ids when the env arrives
    model = PPO2(policy=MlpPolicy, env=None, n_steps=4, nminibatches=1, seed=5, device="cpu")
    model.set_env(OracleTorchEnv(load_env_cfg("default_cfg.yaml", num_envs=4, EnvIdOffset=4 * rank)))
    with open(os.path.join(out_dir, "overlap_%d.txt" % rank), "w") as f:
        f.write("%s\n%d\n" % (msg, model.env_id_offset))
    torch.distributed.destroy_process_group()


def test_ranks_with_overlapping_env_ids_are_refused(tmp_path):
    """seeds are rank-independent, so what makes ranks differ is ONLY the global env ids they own.  Two pools that claim
    the same ids (EnvIdOffset forgotten) must not train silently on two copies of the same data; PPO2(env=None) + set_env() must
    pick the offset up."""
    port = _free_port()
    mp.spawn(_overlap_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        msg, off = open(tmp_path / ("overlap_%d.txt" % r)).read().strip().split("\n")
        assert "overlapping global env ids" in msg, msg
        assert int(off) == 4 * r
