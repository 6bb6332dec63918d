"""Learner-side device ops on an MI355X: the HIP GAE scan against its C oracle (bit-exact: same f32 operation
order), and the on-device PPO2 loop (rollout buffers, policy step, env.step on device tensors, update)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle as O
from conftest import load_env_cfg


def _env(n, cfg="default_cfg.yaml"):
    import yaml
    import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
    from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
    from high_speed_quadrupedal_locomotion_by_irrl_amd.vec_env import TorchVecEnv
    return TorchVecEnv(FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(load_env_cfg(cfg, num_envs=n))))


@pytest.mark.parametrize("T,N", [(750, 4096), (1, 7), (13, 1), (64, 100)])
def test_gae_kernel_matches_oracle_bit_exact(T, N):
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import gae
    rng = np.random.RandomState(T * 1000 + N)
    r = rng.uniform(-1, 1, (T, N)).astype(np.float32)
    v = rng.uniform(-3, 3, (T, N)).astype(np.float32)
    d = rng.uniform(size=(T, N)) < 0.05
    lv = rng.uniform(-3, 3, N).astype(np.float32)
    ld = rng.uniform(size=N) < 0.1
    adv_o, ret_o = O.gae(r, v, d, lv, ld, 0.99, 0.998)
    dev = torch.device("cuda")
    adv, ret = gae(torch.from_numpy(r).to(dev), torch.from_numpy(v).to(dev), torch.from_numpy(d).to(dev),
                   torch.from_numpy(lv).to(dev), torch.from_numpy(ld).to(dev), 0.99, 0.998)
    # same f32 operation order, FMA contraction off on both sides -> bit-exact
    assert np.array_equal(adv.cpu().numpy(), adv_o)
    assert np.array_equal(ret.cpu().numpy(), ret_o)


@pytest.mark.parametrize("kind", ["mlp", "lstm"])
def test_on_device_ppo_iteration(kind):
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy, MlpPolicy
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2
    env = _env(256)
    pol = CustomLSTMPolicy if kind == "lstm" else MlpPolicy
    model = PPO2(policy=pol, env=env, n_steps=40, nminibatches=1 if kind == "lstm" else 4, noptepochs=2, gamma=0.99, lam=0.998,
                 ent_coef=0.0, learning_rate=1e-3, vf_coef=0.5, max_grad_norm=0.5, cliprange=0.2, verbose=1, seed=2)
    before = np.concatenate([p.reshape(-1) for p in model.get_parameter_list()])
    model.learn(total_timesteps=2 * 40 * 256, eval_every_n=0)
    after = np.concatenate([p.reshape(-1) for p in model.get_parameter_list()])
    assert np.isfinite(after).all() and np.abs(after - before).max() > 1e-5
    assert len(model.log) == 2 and np.isfinite(model.log[-1]["policy_loss"])
    assert next(model.policy.parameters()).is_cuda


@pytest.mark.parametrize("T,N,n_in,hid,prec", [(7, 16, 35, 48, "f32"), (33, 40, 48, 48, "f32"), (5, 3, 35, 32, "f32"), (12, 64, 20, 64, "f32"),
                                                (7, 16, 35, 48, "bf16x6"), (33, 40, 48, 48, "bf16x6"), (34, 48, 35, 48, "bf16x6"), (1, 16, 20, 48, "bf16x6"),
                                                (7, 16, 35, 48, "bf16x3"), (33, 40, 48, 48, "bf16x3"), (34, 48, 35, 48, "bf16x3"),
                                                # around the forward kernel's loader wave (8 steps of rows in flight): sequences shorter than, equal to and just past its depth
                                                (2, 16, 35, 48, "bf16x3"), (8, 32, 48, 48, "bf16x3"), (9, 16, 35, 48, "bf16x3"), (10, 16, 48, 48, "bf16x6")])
def test_fused_lstm_sequence_matches_eager_definition(T, N, n_in, hid, prec, monkeypatch):
    """Persistent MFMA LSTM kernels (forward + BPTT) against the eager stable-baselines definition (SBLstm.sequence),
    same f32 inputs: outputs, final state and every gradient -- the exact-f32 kernels ("f32") and the bf16 matrix-core kernels with
    compensated operand splits (csrc/lstm_bf16.hpp): "bf16x6" (3 planes, 6 products: the f32 level, same 2e-5) and "bf16x3" (2 planes,
    3 products: ~2^-16 per product, 1e-4)."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd import lstm_fused
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import SBLstm
    monkeypatch.setattr(lstm_fused, "PRECISION", prec)
    tol = 1e-4 if prec == "bf16x3" else 2e-5
    torch.manual_seed(T * 100 + N)
    dev = torch.device("cuda")
    layer = SBLstm(n_in, hid).to(dev)
    with torch.no_grad():
        layer.b.copy_(torch.randn(4 * hid, device=dev) * 0.1)
    x = torch.randn(T, N, n_in, device=dev, requires_grad=True)
    state = torch.randn(N, 2 * hid, device=dev) * 0.5
    masks = (torch.rand(T, N, device=dev) < 0.15).float()
    wgt = torch.randn(T, N, hid, device=dev)
    outs = {}
    for fused in (False, True):
        SBLstm.use_fused = fused
        for p in layer.parameters():
            p.grad = None
        x.grad = None
        h, s = layer.sequence(x, state, masks)
        (h * wgt).sum().backward()
        outs[fused] = [h.detach().clone(), s.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
    SBLstm.use_fused = True
    names = ["h_seq", "state", "dx", "dwx", "dwh", "db"]
    for name, a, b in zip(names, outs[False], outs[True]):
        scale = float(a.abs().max()) + 1e-6
        assert float((a - b).abs().max()) / scale < tol, (name, prec, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("prec", ["bf16x3", "bf16x6", "f32"])
def test_lstm_sequence_kernels_hold_their_error_bounds_at_the_training_shape(prec, monkeypatch):
    """BPTT error grows with T, so the bound is asserted where the learner runs: T = 750 steps x N = 4096 envs, one SBLstm layer
    (48 -> 48), forward + full-length BPTT, against FLOAT64 autograd of the eager stable-baselines definition (run_bp_v5.py:143-176),
    per tensor relative to its largest entry.  Bounds = what profiles/r04_lstm_precision_error_and_time.log measured, with headroom:
      bf16x3 (the learner's default: two bf16 planes, ~2^-16 per product)   h, state, dx <= 2e-5;  dwx, dwh, db <= 1e-5
      bf16x6 / f32 (the f32 level)                                          h, state, dx <= 2e-6;  dwx, dwh, db <= 3e-6
    (round 6, verdict r5 item 1c: the weight-gradient bound is per arithmetic -- rounds 4-5 held all three to 1e-5, 6x what the f32 kernels
    measure -- and bf16x6, whose 750 x 6 matrix-core additions per accumulator measured 4.1e-6 / 5.6e-6, accumulates in two levels now,
    csrc/lstm_bf16.hpp `lbf_flush_weight_grads`.)
    For scale: PyTorch's eager f32 graph itself reaches 1.6e-5 on dwx at this shape (its reduction order)."""
    import copy
    from high_speed_quadrupedal_locomotion_by_irrl_amd import lstm_fused
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import SBLstm
    monkeypatch.setattr(lstm_fused, "PRECISION", prec)
    T, N, n_in, hid = 750, 4096, 48, 48
    dev = torch.device("cuda")
    torch.manual_seed(T + N)
    layer = SBLstm(n_in, hid).to(dev)
    with torch.no_grad():
        layer.b.copy_(torch.randn(4 * hid, device=dev) * 0.1)
    x = torch.randn(T, N, n_in, device=dev)
    state = torch.randn(N, 2 * hid, device=dev) * 0.5
    masks = (torch.rand(T, N, device=dev) < 0.01).float()
    wgt = torch.randn(T, N, hid, device=dev) / (T * N) ** 0.5

    def run(lay, xx, st, mk, wg):
        xx = xx.clone().requires_grad_(True)
        for p in lay.parameters():
            p.grad = None
        h, s = lay.sequence(xx, st, mk)
        (h * wg).sum().backward()
        return [h.detach(), s.detach(), xx.grad] + [p.grad for p in lay.parameters()]

    try:
        SBLstm.use_fused = False
        ref = run(copy.deepcopy(layer).double(), x.double(), state.double(), masks.double(), wgt.double())
        SBLstm.use_fused = True
        out = run(layer, x, state, masks, wgt)
    finally:
        SBLstm.use_fused = True
    act_tol = 2e-5 if prec == "bf16x3" else 2e-6
    errs = {}
    for name, a, b in zip(["h_seq", "state", "dx", "dwx", "dwh", "db"], out, ref):
        errs[name] = float((a.double() - b).abs().max()) / (float(b.abs().max()) + 1e-30)
    print("\n[lstm kernels %s, T 750 x N 4096] max |kernel - float64| / max |float64|: %s" % (prec, {k: "%.2e" % v for k, v in errs.items()}))
    wgrad_tol = 1e-5 if prec == "bf16x3" else 3e-6
    for name, e in errs.items():
        assert e <= (act_tol if name in ("h_seq", "state", "dx") else wgrad_tol), (prec, name, e)


@pytest.mark.parametrize("prec", ["bf16x3", "bf16x6"])
def test_lstm_sequence_inference_form_equals_the_training_form(prec, monkeypatch):
    """Under torch.no_grad() the bf16 forward sequence kernel runs in its inference form (no gates / c rows stored: the critic pass behind an
    actor-only rollout, ppo2.Runner._critic_pass): h and the final state are bit-identical to the training form's."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd import lstm_fused
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import SBLstm
    monkeypatch.setattr(lstm_fused, "PRECISION", prec)
    dev = torch.device("cuda")
    torch.manual_seed(5)
    T, N, n_in, hid = 40, 80, 35, 48
    layer = SBLstm(n_in, hid).to(dev)
    x = torch.randn(T, N, n_in, device=dev)
    state = torch.randn(N, 2 * hid, device=dev) * 0.5
    masks = (torch.rand(T, N, device=dev) < 0.1).float()
    SBLstm.use_fused = True
    h_train, s_train = layer.sequence(x, state, masks)
    with torch.no_grad():
        h_inf, s_inf = layer.sequence(x, state, masks)
    assert h_train.requires_grad and not h_inf.requires_grad
    assert torch.equal(h_train.detach(), h_inf) and torch.equal(s_train.detach(), s_inf)


def test_fused_lstm_policy_full_size_agrees_with_eager():
    """CustomLSTMPolicy.evaluate at the training shape (T=750 is covered by the PPO bench; here T=96 x 4096 envs)."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy, SBLstm
    torch.manual_seed(0)
    dev = torch.device("cuda")
    pol = CustomLSTMPolicy().to(dev)
    T, N = 96, 4096
    obs = torch.randn(T, N, 35, device=dev)
    st = torch.randn(N, 384, device=dev) * 0.3
    masks = (torch.rand(T, N, device=dev) < 0.01).float()
    act = torch.randn(T, N, 12, device=dev) * 0.3
    res = {}
    for fused in (False, True):
        SBLstm.use_fused = fused
        pol.zero_grad(set_to_none=True)
        nlp, val, ent = pol.evaluate(obs, st, masks, act)
        (nlp.mean() + val.mean()).backward()
        res[fused] = (nlp.detach().clone(), val.detach().clone(), torch.cat([p.grad.reshape(-1) for p in pol.sb_parameters() if p.grad is not None]))
    SBLstm.use_fused = True
    for a, b in zip(res[False], res[True]):
        assert float((a - b).abs().max()) / (float(a.abs().max()) + 1e-6) < 1e-4


@pytest.mark.parametrize("policy,cfg", [("lstm", "default_cfg.yaml"), ("mlp", "default_cfg.yaml"), ("lstm", "bp5_terrain.yaml")])
def test_two_rank_ppo_iteration_on_the_hip_engine_equals_the_single_process_one(tmp_path, policy, cfg):
    """The only multi-GPU proof available without a node: two ranks (gloo, both on cuda:0), 64 envs each with
    EnvIdOffset = rank * 64, one PPO iteration through the product's Runner (fused policy-step + env-step launches, in-kernel
    sampling noise addressed by the global env id) and PPO2.update (gradient + advantage-moment all-reduce) == the single-process
    iteration on the 128-env pool: rollout buffers bit-identical, parameters equal up to the gradient's summation order.  With
    `bp5_terrain.yaml` this is BASELINE config 4 / 5 in miniature (LSTM policy, env shards, gradient all-reduce; shared height field,
    per-episode friction / mass / COM randomisation, command process) on the one GPU a test box has."""
    import sys, os
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_ppo_distributed import check_two_ranks_equal_single_process
    two, one = check_two_ranks_equal_single_process(tmp_path, "cuda", 64, 24, policy, cfg)
    assert int(one["fused_rollout"]) == 1 and int(two[0]["fused_rollout"]) == 1


def test_two_rank_mlp_iteration_with_four_minibatches_on_the_hip_engine_equals_the_single_process_one(tmp_path):
    """The shipped MlpPolicy configuration (BASELINE config 2's learner: 4 minibatches, ppo2.py:364-380) on two ranks: ONE global
    permutation of all ranks' samples per epoch (the Feistel kernel over world * n ids), every rank keeps the ids of its shard and
    weights its mean gradient by its share before the all-reduce -- through the in-place gradient kernels (`irrl_mlp_ppo_grads_bf16` reading
    the minibatch's rows through the index).  Rollouts bit-identical, parameters equal up to summation order."""
    import sys, os
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_ppo_distributed import check_two_ranks_equal_single_process
    two, one = check_two_ranks_equal_single_process(tmp_path, "cuda", 64, 24, "mlp", "bp5_imitation.yaml", nminibatches=4)
    assert int(one["fused_rollout"]) == 1 and int(two[0]["collective"]) == 1 and int(one["collective"]) == 0


@pytest.mark.parametrize("policy", ["lstm", "mlp"])
def test_rccl_one_rank_communicator_runs_the_ppo_collectives_on_device_tensors(tmp_path, policy):
    """RCCL itself (backend "nccl"), as far as a 1-GPU box allows: a FRESH child process creates a one-rank communicator
    (init_process_group("nccl", world_size=1, device_id=cuda:0)) before anything else touches the GPU, and runs one PPO iteration whose
    update goes through the `collective` branches -- the env-id all_gather, the 3-float advantage-moment all-reduce and the
    flat-gradient all-reduce, all on DEVICE tensors through librccl.  With one rank every reduction is an identity: the result must
    equal the same iteration without a process group up to nothing at all (same kernels, same order)."""
    import sys, os
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_ppo_distributed import _run_workers
    (tmp_path / "nccl").mkdir(); (tmp_path / "plain").mkdir()
    with_rccl = _run_workers(tmp_path / "nccl", "cuda", 1, 64, 24, policy, backend="nccl")[0]
    plain = _run_workers(tmp_path / "plain", "cuda", 1, 64, 24, policy)[0]
    assert str(with_rccl["backend"]) == "nccl" and int(with_rccl["collective"]) == 1 and int(plain["collective"]) == 0
    for key in ("obs", "actions", "values", "neglogpacs", "returns", "masks", "params", "losses"):
        assert np.array_equal(with_rccl[key], plain[key]), key


def test_bench_under_a_one_rank_launcher_uses_rccl():
    """bench.py the way the driver starts it for N > 1 (RANK / WORLD_SIZE / MASTER_* in the environment), with ONE rank: the process group
    is created with the default backend (nccl = RCCL, device_id = its GPU), the barrier around the timed bracket, the max-over-ranks
    and the PPO legs' collectives run on device tensors.  What a 1-GPU box can show of the N-GPU launch path."""
    import json, os, socket, subprocess, sys
    from conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
               IRRL_BENCH_NATIVE="0", IRRL_BENCH_F32_LEVEL="0")
    env.pop("IRRL_BENCH_BACKEND", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--envs", "512", "--cpu-seconds", "0",
                        "--check-steps", "0", "--no-mode-extras", "--ppo-iters", "1", "--ppo-steps", "32", "--ppo-epochs", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["backend"] == "nccl" and out["rccl_ranks_seen"] == 1 and out["n_gpus"] == 1
    assert out["ppo"]["collectives_per_optimizer_step"] and out["ppo"]["ppo_iters_per_sec"] > 0
    assert out["ppo_mlp"]["collectives_per_optimizer_step"] and out["ppo_mlp"]["ppo_iters_per_sec"] > 0


@pytest.mark.parametrize("N,prec", [(1, "f32"), (37, "f32"), (37, "bf16x6"), (37, "bf16x3")])
def test_hip_lstm_kernels_reproduce_the_reference_actor_known_answers(N, prec, monkeypatch):
    """The reference's own known answers on the MI355X kernels (CustomerLstmNN.py:112-175 `predict` on the trained bp5_155 weights,
    tests/golden/lstm_bp5_155.json from tools/gen_golden.py): the eight actor tensors (tests/golden/actor_bp5_155.npz -- the weights
    must travel: /root/reference does not exist on the GPU box) are loaded into the policy, then
      (a) `lstm_policy_step_kernel` (one launch per control step, state carried from step to step) and
      (b) `lstm_seq_fwd_x_kernel` (the train-graph forward over the whole sequence, one launch per layer)
    must give the pickle's action means to 2e-5 and the CSV twin's clipped actions to 5e-5 (its '%.6f' rounding)."""
    import json, os
    from conftest import GOLDEN
    from high_speed_quadrupedal_locomotion_by_irrl_amd import lstm_fused
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy, SBLstm
    monkeypatch.setattr(lstm_fused, "PRECISION", prec)      # arithmetic of the sequence kernels (b); the policy step (a) is always exact f32
    g = json.load(open(os.path.join(GOLDEN, "lstm_bp5_155.json")))
    z = np.load(os.path.join(GOLDEN, "actor_bp5_155.npz"))
    dev = torch.device("cuda")
    torch.manual_seed(3)
    pol = CustomLSTMPolicy().to(dev)
    with torch.no_grad():
        for i, l in enumerate(pol.lstm_pi):
            l.wx.copy_(torch.from_numpy(z["wx%d" % i])); l.wh.copy_(torch.from_numpy(z["wh%d" % i])); l.b.copy_(torch.from_numpy(z["b%d" % i]))
        pol.pi.w.copy_(torch.from_numpy(z["pi_w"])); pol.pi.b.copy_(torch.from_numpy(z["pi_b"]))
    pol.prepare()
    assert SBLstm.use_fused
    obs_seq = torch.tensor(g["obs_seq"], dtype=torch.float32, device=dev)                      # [6, 35]
    want = np.asarray(g["actor_mean_pkl"], np.float64)
    want_csv = np.asarray(g["actor_clipped_csv"], np.float64)
    T = obs_seq.shape[0]
    # (a) the rollout's single-launch policy step
    st = pol.initial_state(N, dev)
    dones = torch.zeros(N, dtype=torch.bool, device=dev)
    assert pol.fused_step_supported(obs_seq[0].expand(N, 35).contiguous())
    for t in range(T):
        act, clipped, _, _, st = pol.fused_step(obs_seq[t].expand(N, 35).contiguous(), st, dones, noise=None)
        a = act.double().cpu().numpy()
        assert np.abs(a - want[t]).max() < 2e-5, (t, np.abs(a - want[t]).max())
        assert np.abs(clipped.double().cpu().numpy() - want_csv[t]).max() < 5e-5
    # (b) the persistent sequence kernel of the train graph
    mean, _ = pol.evaluate_raw(obs_seq.unsqueeze(1).expand(T, N, 35).contiguous(), pol.initial_state(N, dev), torch.zeros(T, N, device=dev))
    m = mean.detach().double().cpu().numpy()                                                    # [T, N, 12]
    tol = 1e-4 if prec == "bf16x3" else 2e-5          # two bf16 planes: ~2^-16 per product (measured 1e-5 of the largest activation)
    assert np.abs(m - want[:, None, :]).max() < tol, (prec, np.abs(m - want[:, None, :]).max())


@pytest.mark.parametrize("N,hid,deterministic", [(4096, 48, False), (48, 48, True), (16, 32, False), (160, 64, False), (200, 48, False), (7, 48, False)])
def test_fused_policy_step_matches_eager_step(N, hid, deterministic):
    """The single-launch rollout step (both LSTM stacks, heads, sample, neglogp, clip, buffer rows) against the eager
    CustomLSTMPolicy.step built from torch ops, same weights / state / noise."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy, SBLstm, diag_gaussian_neglogp
    torch.manual_seed(N + hid)
    dev = torch.device("cuda")
    pol = CustomLSTMPolicy(n_lstm=(hid, hid)).to(dev)
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    obs = torch.randn(N, 35, device=dev)
    st = torch.randn(N, 8 * hid, device=dev) * 0.5
    dones = torch.rand(N, device=dev) < 0.2
    noise = None if deterministic else torch.randn(N, 12, device=dev)
    assert pol.fused_step_supported(obs)
    T = 3
    prev_rew = torch.randn(N, device=dev)
    mb_rew = torch.zeros(T, N, device=dev)
    mb = [torch.zeros(T, N, 35, device=dev), torch.zeros(T, N, 12, device=dev), torch.zeros(T, N, device=dev), torch.zeros(T, N, device=dev),
          torch.zeros(T, N, dtype=torch.bool, device=dev)]
    act, clipped, val, nlp, snew = pol.fused_step(obs, st, dones, noise=noise, rollout=dict(
        row=1, mb_obs=mb[0], mb_actions=mb[1], mb_values=mb[2], mb_neglogpacs=mb[3], mb_dones=mb[4], mb_rewards=mb_rew, prev_reward=prev_rew))
    # eager definition
    SBLstm.use_fused = False
    try:
        mean, v_ref, s_ref = pol._run(obs.unsqueeze(0), st, dones.float().unsqueeze(0))
    finally:
        SBLstm.use_fused = True
    mean, v_ref = mean[0].detach(), v_ref[0].detach()
    a_ref = mean if deterministic else mean + torch.exp(pol.logstd.detach()) * noise
    nlp_ref = diag_gaussian_neglogp(a_ref, mean, pol.logstd.detach())
    for name, a, b, tol in (("action", a_ref, act, 2e-5), ("value", v_ref, val, 2e-5), ("state", s_ref.detach(), snew, 2e-5), ("neglogp", nlp_ref, nlp, 1e-4)):
        assert float((a - b).abs().max()) <= tol * (1.0 + float(a.abs().max())), (name, float((a - b).abs().max()))
    assert torch.equal(clipped, act.clamp(-1.0, 1.0))
    # row t_idx of the rollout buffers, other rows untouched
    assert torch.equal(mb[0][1], obs) and torch.equal(mb[1][1], act) and torch.equal(mb[2][1], val) and torch.equal(mb[3][1], nlp) and torch.equal(mb[4][1], dones)
    for b in mb:
        assert not b[0].any() and not b[2].any()
    # the previous step's reward lands in row t-1
    assert torch.equal(mb_rew[0], prev_rew) and not mb_rew[1:].any()
    # in-place state update gives the same result
    st2 = st.clone()
    pol.fused_step(obs, st2, dones, noise=noise, states_out=st2)
    assert torch.equal(st2, snew)


def test_fused_runner_matches_stepwise_runner():
    """Runner with the single-launch policy step + raw env step (hipGraph replayed) against the generic runner path
    (policy.step, env.step with per-step bookkeeping), same seeds: identical rollout buffers and episode statistics."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2, Runner
    out = {}
    for fused in (True, False):
        env = _env(64)
        model = PPO2(policy=CustomLSTMPolicy, env=env, n_steps=30, nminibatches=1, noptepochs=1, seed=5)
        runner = Runner(env, model, 30, 0.99, 0.998)
        assert runner._fused
        runner._fused = fused
        runner._raw_env = fused           # the stepwise runner keeps the per-step episode bookkeeping of TorchVecEnv.step
        runner.noise_source = "torch"     # same sampling noise on both paths
        b1 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in runner.run().items()}
        b2 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in runner.run().items()}   # second rollout: replayed graph, carried states / dones
        out[fused] = (b1, b2, env.pop_episode_stats())
    for i in (0, 1):
        for k in ("obs", "actions", "values", "neglogpacs", "masks", "true_reward", "returns", "states"):
            a, b = out[True][i][k], out[False][i][k]
            if a.dtype == torch.bool:
                assert torch.equal(a, b), k
            else:
                assert float((a - b).abs().max()) <= 1e-4 * (1.0 + float(a.abs().max())), (i, k, float((a - b).abs().max()))
    sa, sb = out[True][2], out[False][2]
    assert sa[2] == sb[2] and abs(sa[0] - sb[0]) < 1e-3 * (1 + abs(sb[0])) and abs(sa[1] - sb[1]) < 1e-3


def test_fused_policy_step_kernel_noise_is_the_counter_rng():
    """Sampling noise drawn inside the policy kernel: Philox4x32-10 keyed like the env engine (oracle.rng_u01 is the
    independent C statement), Box-Muller on (u0,u1), (u2,u3); reproducible per (seed, env, step), fresh per step."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy
    torch.manual_seed(3)
    dev = torch.device("cuda")
    pol = CustomLSTMPolicy().to(dev)
    N = 4096
    obs = torch.randn(N, 35, device=dev)
    st = torch.zeros(N, 384, device=dev)
    dones = torch.zeros(N, dtype=torch.bool, device=dev)
    seed, step = 1234567, (5 << 32) + 42
    mean = pol.fused_step(obs, st, dones)[0]                      # deterministic
    act = pol.fused_step(obs, st, dones, rng=(seed, step))[0]
    z = ((act - mean) / torch.exp(pol.logstd.detach())).cpu().numpy().astype(np.float64)
    # exact reproduction for a few envs
    for env in (0, 1, 17, 4095):
        ref = []
        for q in range(3):
            u = O.rng_u01(seed, env, step >> 32, step & 0xFFFFFFFF, 0x50 + q)
            for a, b in ((u[0], u[1]), (u[2], u[3])):
                r = np.sqrt(-2.0 * np.log(1.0 - a))
                ref += [r * np.cos(2 * np.pi * b), r * np.sin(2 * np.pi * b)]
        assert np.abs(z[env] - np.array(ref)).max() < 2e-3, (env, z[env], ref)
    # distribution: 49152 samples
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02 and abs((z ** 4).mean() - 3.0) < 0.15
    assert np.abs(np.corrcoef(z.T) - np.eye(12)).max() < 0.06
    # same (seed, step) -> same sample; next step -> independent sample
    act2 = pol.fused_step(obs, st, dones, rng=(seed, step))[0]
    base = torch.tensor([1], device=dev, dtype=torch.long)
    act3 = pol.fused_step(obs, st, dones, rng=(seed, step, base))[0]           # step + *base
    assert torch.equal(act3, pol.fused_step(obs, st, dones, rng=(seed, step + 1))[0])
    assert torch.equal(act, act2)
    z3 = ((act3 - mean) / torch.exp(pol.logstd.detach())).cpu().numpy()
    assert abs(np.corrcoef(z.reshape(-1), z3.reshape(-1))[0, 1]) < 0.02


@pytest.mark.parametrize("N,deterministic", [(4096, False), (200, False), (5, True)])
def test_fused_mlp_policy_step_matches_eager_step(N, deterministic):
    """mlp_policy_step_kernel (both tanh nets on MFMA, heads, sample, neglogp, clip, buffer rows) against MlpPolicy._run."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import MlpPolicy, diag_gaussian_neglogp
    torch.manual_seed(N)
    dev = torch.device("cuda")
    pol = MlpPolicy().to(dev)
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    obs = torch.randn(N, 35, device=dev)
    dones = torch.rand(N, device=dev) < 0.2
    noise = None if deterministic else torch.randn(N, 12, device=dev)
    assert pol.fused_step_supported(obs)
    T = 3
    mb = dict(row=2, mb_obs=torch.zeros(T, N, 35, device=dev), mb_actions=torch.zeros(T, N, 12, device=dev), mb_values=torch.zeros(T, N, device=dev),
              mb_neglogpacs=torch.zeros(T, N, device=dev), mb_dones=torch.zeros(T, N, dtype=torch.bool, device=dev),
              mb_rewards=torch.zeros(T, N, device=dev), prev_reward=torch.randn(N, device=dev))
    act, clipped, val, nlp, _ = pol.fused_step(obs, None, dones, noise=noise, rollout=mb)
    with torch.no_grad():
        mean, v_ref = pol._run(obs)
        a_ref = mean if deterministic else mean + torch.exp(pol.logstd) * noise
        nlp_ref = diag_gaussian_neglogp(a_ref, mean, pol.logstd)
    for name, a, b, tol in (("action", a_ref, act, 2e-5), ("value", v_ref, val, 2e-5), ("neglogp", nlp_ref, nlp, 1e-4)):
        assert float((a - b).abs().max()) <= tol * (1.0 + float(a.abs().max())), (name, float((a - b).abs().max()))
    assert torch.equal(clipped, act.clamp(-1.0, 1.0))
    assert torch.equal(mb["mb_obs"][2], obs) and torch.equal(mb["mb_actions"][2], act) and torch.equal(mb["mb_values"][2], val)
    assert torch.equal(mb["mb_dones"][2], dones) and torch.equal(mb["mb_rewards"][1], mb["prev_reward"]) and not mb["mb_rewards"][2].any()


def test_run_bp_v5_test_branch_headless(tmp_path):
    """`run_bp_v5.py --test` (headless): Manual-mode env, command injected into obs[0:3], delay line + filters, numpy LSTM actor
    from a saved checkpoint, records written to an .npz (run_bp_v5.py:300-470 without gamepad / plots)."""
    import importlib.util, os, sys
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2
    env = _env(16)
    model = PPO2(policy=CustomLSTMPolicy, env=env, n_steps=8, nminibatches=1, noptepochs=1, seed=3)
    ckpt = model.save(str(tmp_path / "rand_policy"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_bp_v5_script", os.path.join(root, "scripts", "run_bp_v5.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = str(tmp_path / "rec.npz")
    mod.main(["--test", "--model", ckpt, "--cmd", "0.8", "--steps", "120", "--delay", "2", "--vel_filter_freq", "50", "--act_filter_freq", "30", "--out", out])
    rec = np.load(out)
    assert rec["oss"].shape == (120, 37) and rec["joint"].shape == (120, 12) and rec["contact"].shape == (120, 4)
    assert np.isfinite(rec["oss"]).all() and np.isfinite(rec["act"]).all()
    assert rec["cmd"][0, 0] < 0.05 and 0.3 < rec["cmd"][-1, 0] <= 0.8 and np.all(np.diff(rec["cmd"][:, 0]) >= 0)   # 1 Hz command ramp
    # Manual start pose: origin, nominal height, then the robot moves
    assert abs(rec["oss"][0, 0]) < 1e-6 and abs(rec["oss"][0, 2] - 0.35) < 0.02


@pytest.mark.parametrize("kind", ["lstm", "mlp"])
def test_fused_ppo_loss_matches_eager_graph(kind):
    """`irrl_ppo_loss` (loss forward + backward in one launch, advantages normalised in the kernel) against the eager
    ppo_loss / DiagGaussian graph: same statistics and the same parameters after two optimizer steps."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy, MlpPolicy
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2, Runner
    res = {}
    for fused in (True, False):
        env = _env(64)
        model = PPO2(policy=CustomLSTMPolicy if kind == "lstm" else MlpPolicy, env=env, n_steps=48, nminibatches=1 if kind == "lstm" else 2,
                     noptepochs=2, gamma=0.99, lam=0.95, ent_coef=0.01, learning_rate=1e-3, cliprange=0.2, seed=11)
        with torch.no_grad():
            model.policy.logstd.add_(torch.linspace(-0.3, 0.2, 12, device=model.device).reshape(1, 12))
        model.fused_loss = fused
        runner = Runner(env, model, 48, 0.99, 0.95)
        runner.noise_source = "torch"
        batch = runner.run()
        # make the clipped branches matter: perturb the stored old values / neglogps
        g = torch.Generator(device=model.device); g.manual_seed(1)
        batch["values"] = batch["values"] + 0.5 * torch.randn(batch["values"].shape, device=model.device, generator=g)
        batch["neglogpacs"] = batch["neglogpacs"] + 0.3 * torch.randn(batch["neglogpacs"].shape, device=model.device, generator=g)
        stats = model.update(batch, 1e-3, 0.2)
        res[fused] = (stats.detach().cpu().numpy(), np.concatenate([p.reshape(-1) for p in model.get_parameter_list()]))
    sa, sb = res[True][0], res[False][0]
    assert np.allclose(sa, sb, rtol=2e-4, atol=2e-5), (sa, sb)
    pa, pb = res[True][1], res[False][1]
    assert np.abs(pa - pb).max() < 2e-5, np.abs(pa - pb).max()


def test_training_is_bitwise_reproducible_and_logs_progress(tmp_path):
    """Counter RNG in the env and in the policy kernel, no atomics in the gradient path: two runs from the same seed give
    bit-identical parameters; the per-update table goes to <tensorboard_log>/progress.csv."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2
    outs = []
    for k in range(2):
        env = _env(64)
        model = PPO2(policy=CustomLSTMPolicy, env=env, n_steps=32, nminibatches=1, noptepochs=3, seed=21, verbose=1,
                     tensorboard_log=str(tmp_path / ("run%d" % k)))
        model.learn(total_timesteps=3 * 32 * 64, eval_every_n=0)
        outs.append(np.concatenate([p.reshape(-1) for p in model.get_parameter_list()]))
        rows = open(tmp_path / ("run%d" % k) / "progress.csv").read().strip().split("\n")
        assert rows[0].startswith("serial_timesteps,nupdates,total_timesteps") and len(rows) == 4
    assert np.array_equal(outs[0], outs[1])


def test_kernels_follow_the_optimizer_and_checkpoints_reproduce_the_live_policy(tmp_path):
    """Regression: the fused Adam step does not bump a parameter's `_version`, so the [unit][gate] weight copies the LSTM
    kernels read must be refreshed explicitly after every optimizer step.  After training, (a) the copies equal the
    parameters, (b) the eager definition (reads the parameters) and the fused kernels (read the copies) agree on the same
    input, (c) a saved + loaded checkpoint reproduces the live policy's actions."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd import lstm_fused
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy, SBLstm
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2
    env = _env(64)
    model = PPO2(policy=CustomLSTMPolicy, env=env, n_steps=32, nminibatches=1, noptepochs=3, learning_rate=3e-3, seed=4)
    w0 = model.policy.lstm_pi[0].wx.detach().clone()
    model.learn(total_timesteps=3 * 32 * 64, eval_every_n=0)
    pol = model.policy
    assert float((pol.lstm_pi[0].wx - w0).abs().max()) > 1e-3                      # the LSTM weights did move
    for l in list(pol.lstm_pi) + list(pol.lstm_v):
        perm = lstm_fused._perm(l.n_hidden, l.wx.device)[0]
        wx_p, wh_p, b_p = lstm_fused._permuted_weights(l.wx, l.wh, l.b, perm)
        assert torch.equal(wx_p, l.wx[:, perm]) and torch.equal(wh_p, l.wh[:, perm]) and torch.equal(b_p, l.b[perm])
    dev = model.device
    torch.manual_seed(0)
    obs = torch.randn(64, 35, device=dev)
    st = torch.randn(64, 384, device=dev) * 0.3
    dones = torch.zeros(64, dtype=torch.bool, device=dev)
    a_fused = pol.step(obs, st, dones, deterministic=True)[0]
    SBLstm.use_fused = False
    try:
        a_eager = pol.step(obs, st, dones, deterministic=True)[0]
    finally:
        SBLstm.use_fused = True
    assert float((a_fused - a_eager).abs().max()) < 2e-5
    ck = model.save(str(tmp_path / "m"))
    model2 = PPO2.load(ck, env=env)
    a_loaded = model2.policy.step(obs, st, dones, deterministic=True)[0]
    assert float((a_loaded - a_fused).abs().max()) < 1e-6


@pytest.mark.parametrize("M", [64 * 5 + 37, 4096 * 8])
def test_fused_heads_and_loss_gradients_match_autograd(M):
    """`irrl_ppo_heads_loss` (heads forward, loss, and every gradient in one launch) against torch autograd of the eager heads +
    ppo_loss on random latents: loss, statistics, d h_pi / d h_v rows, the head weight / bias gradients and d logstd; M not a
    multiple of the 64-row tile exercises the ragged last tile."""
    import math
    from high_speed_quadrupedal_locomotion_by_irrl_amd import ppo2 as P2
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import diag_gaussian_neglogp, diag_gaussian_entropy
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(5)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    h_pi, h_v = rn(M, 48).requires_grad_(), rn(M, 48).requires_grad_()
    pi_w, pi_b = (0.2 * rn(48, 12)).requires_grad_(), (0.1 * rn(12)).requires_grad_()
    vf_w, vf_b = (0.2 * rn(48, 1)).requires_grad_(), (0.1 * rn(1)).requires_grad_()
    logstd = (0.2 * rn(1, 12)).requires_grad_()
    actions, returns, old_v = rn(M, 12), rn(M), rn(M)
    with torch.no_grad():
        old_nlp = diag_gaussian_neglogp(actions, h_pi @ pi_w + pi_b, logstd) + 0.3 * rn(M)
    advs = returns - old_v
    stats_t = torch.stack([advs.mean(), advs.std(unbiased=False)]).to(torch.float32)
    # eager reference
    mean = h_pi @ pi_w + pi_b
    v = (h_v @ vf_w + vf_b).squeeze(-1)
    nadv = (advs - stats_t[0]) / (stats_t[1] + 1e-8)
    loss_e, pg, vf, ent, kl, cf = P2.ppo_loss(diag_gaussian_neglogp(actions, mean, logstd), v, diag_gaussian_entropy(logstd, mean), actions, nadv,
                                              returns, old_nlp, old_v, 0.2, 0.01, 0.5)
    ge = torch.autograd.grad(loss_e, [h_pi, h_v, pi_w, pi_b, vf_w, vf_b, logstd])
    loss_f, st = P2._FusedHeadsLoss.apply(h_pi, h_v, pi_w, pi_b, vf_w, vf_b, logstd, actions, returns, old_v, old_nlp, stats_t, 0.2, 0.01, 0.5)
    gf = torch.autograd.grad(loss_f, [h_pi, h_v, pi_w, pi_b, vf_w, vf_b, logstd])
    assert abs(float(loss_f) - float(loss_e)) < 2e-5 * max(1.0, abs(float(loss_e)))
    np.testing.assert_allclose(st.cpu().numpy(), torch.stack([pg, vf, ent, kl, cf]).detach().cpu().numpy(), rtol=2e-4, atol=2e-5)
    for name, a, b in zip(("d_hpi", "d_hv", "d_wpi", "d_bpi", "d_wv", "d_bv", "d_logstd"), gf, ge):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) < 2e-4 * scale, (name, float((a - b).abs().max()), scale)
    # a caller that scales the loss (loss / world, gradient accumulation) gets scaled gradients: nothing is assumed about the
    # upstream gradient unless the caller says `unit_grad=True` (PPO2._train_step does: it calls loss.backward() itself)
    loss_s, _ = P2._FusedHeadsLoss.apply(h_pi, h_v, pi_w, pi_b, vf_w, vf_b, logstd, actions, returns, old_v, old_nlp, stats_t, 0.2, 0.01, 0.5)
    gs = torch.autograd.grad(0.25 * loss_s, [h_pi, h_v, pi_w, pi_b, vf_w, vf_b, logstd])
    for name, a, b in zip(("d_hpi", "d_hv", "d_wpi", "d_bpi", "d_wv", "d_bv", "d_logstd"), gs, ge):
        scale = 0.25 * float(b.abs().max()) + 1e-12
        assert float((a - 0.25 * b).abs().max()) < 2e-4 * scale, (name, "scaled")


@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("bf16x3", 1e-4)])
@pytest.mark.parametrize("n,indexed", [(16 * 7 + 5, False), (4096 * 6 + 3, True), (200000, True)])
def test_mlp_policy_gradient_kernels_match_autograd(n, indexed, prec, tol, monkeypatch):
    """`irrl_mlp_ppo_grads` / `irrl_mlp_ppo_grads_bf16` (MlpPolicy forward, PPO2 loss and every parameter gradient, one launch per
    network, minibatch rows read through the index) against torch autograd of the eager policy + ppo_loss in float64 on the same rows:
    loss, statistics and all 13 gradients; n not a multiple of the 16-sample tile exercises the ragged last tile, the indexed cases a
    shuffled minibatch of a larger rollout.  "f32": exact-f32 matrix products, 2e-5 of each gradient's largest entry; "bf16x3" (the
    learner's default): every product as three bf16 plane products, 1e-4 (measured 7e-6 at these sizes, tools/mlp_grad_error.py)."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd import ppo2 as P2
    monkeypatch.setattr(P2, "MLP_PRECISION", prec)
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import MlpPolicy, diag_gaussian_neglogp, diag_gaussian_entropy
    dev = torch.device("cuda")
    torch.manual_seed(3)
    pol = MlpPolicy().to(dev)
    g = torch.Generator(device=dev); g.manual_seed(11)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    with torch.no_grad():
        pol.pi.w.mul_(30.0); pol.pi.b.add_(0.1 * rn(12)); pol.logstd.add_(0.2 * rn(1, 12))     # away from the symmetric initial point
        for l in (*pol.pi_fc, *pol.vf_fc):
            l.b.add_(0.1 * rn(*l.b.shape))
    rows = n if not indexed else 2 * n + 77
    obs, actions, returns, old_v = rn(rows, 35), 0.5 * rn(rows, 12), rn(rows), rn(rows)
    index = torch.randperm(rows, device=dev, generator=g)[:n].contiguous() if indexed else None
    sel = (lambda t: t[index]) if indexed else (lambda t: t)
    with torch.no_grad():
        old_nlp = diag_gaussian_neglogp(actions, pol._run(obs)[0], pol.logstd) + 0.3 * rn(rows)
    # reference: the eager graph in float64 (the f32 eager graph itself is up to 3e-3 of a gradient's scale off at these sizes:
    # tools/mlp_grad_error.py)
    import copy
    p64 = copy.deepcopy(pol).double()
    d = lambda t: sel(t).double()
    advs = d(returns) - d(old_v)
    stats64 = torch.stack([advs.mean(), advs.std(unbiased=False)])
    stats_t = stats64.to(torch.float32)
    mean, v = p64._run(d(obs))
    nadv = (advs - stats_t[0].double()) / (stats_t[1].double() + 1e-8)
    loss_e, pg, vf, ent, kl, cf = P2.ppo_loss(diag_gaussian_neglogp(d(actions), mean, p64.logstd), v, diag_gaussian_entropy(p64.logstd, mean),
                                              None, nadv, d(returns), d(old_nlp), d(old_v), 0.2, 0.01, 0.5)
    skip = lambda pp: [q for q in pp.sb_parameters() if q is not pp.q.w and q is not pp.q.b]
    params = skip(pol)
    ge = torch.autograd.grad(loss_e, skip(p64))
    assert P2.mlp_ppo_grads_supported(pol, obs)
    loss_f, st, grads = P2.mlp_ppo_grads(pol, obs, actions, returns, old_v, old_nlp, stats_t, 0.2, 0.01, 0.5, index=index)
    assert abs(float(loss_f) - float(loss_e.detach())) < tol * max(1.0, abs(float(loss_e.detach())))
    np.testing.assert_allclose(st.cpu().numpy(), torch.stack([pg, vf, ent, kl, cf]).detach().cpu().numpy(), rtol=10 * tol, atol=tol)
    assert len(grads) == len(params)
    for q, b in zip(params, ge):
        a = grads[q]
        assert a.shape == b.shape
        scale = float(b.abs().max()) + 1e-12
        assert float((a.double() - b).abs().max()) < tol * scale, (tuple(q.shape), float((a.double() - b).abs().max()), scale)
    # deterministic: the same launch twice gives the same bits
    _l2, _s2, grads2 = P2.mlp_ppo_grads(pol, obs, actions, returns, old_v, old_nlp, stats_t, 0.2, 0.01, 0.5, index=index)
    assert all(torch.equal(grads[q], grads2[q]) for q in params)


@pytest.mark.parametrize("rows,cols,hid", [(256, 48 * 192, 48), (1024, 192, 48), (7, 130, 0), (1, 64, 0), (256, 8356, 0)])
def test_row_sum_kernel_matches_float64_sum_and_undoes_the_gate_permutation(rows, cols, hid):
    """`irrl_sum_rows` (the fixed-order sum of the gradient kernels' per-workgroup partial rows) against a float64 sum; with hid > 0 the
    [unit][gate] column order of the LSTM kernels comes back as the reference's [gate][unit]."""
    import ctypes as C
    from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(rows * 7 + cols)
    part = torch.randn(rows, cols, device=dev, generator=g)
    out = torch.full((cols,), float("nan"), device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    assert lib.irrl_sum_rows(C.c_void_p(part.data_ptr()), rows, cols, hid, C.c_void_p(out.data_ptr()), st) == 0
    want = part.double().sum(0)
    if hid:
        j = torch.arange(4 * hid, device=dev)
        src = 4 * (j % hid) + j // hid
        want = want.view(-1, 4 * hid)[:, src].reshape(-1)
    assert float((out.double() - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))
    out2 = torch.empty_like(out)
    assert lib.irrl_sum_rows(C.c_void_p(part.data_ptr()), rows, cols, hid, C.c_void_p(out2.data_ptr()), st) == 0
    assert torch.equal(out, out2)
    assert lib.irrl_sum_rows(C.c_void_p(part.data_ptr()), rows, 100, 48, C.c_void_p(out2.data_ptr()), st) == 1   # not a multiple of 4 hid


@pytest.mark.parametrize("n,indexed", [(1, False), (1000, True), (768000, True)])
def test_advantage_moments_kernel_matches_float64_sums(n, indexed):
    """`irrl_adv_moments` (sum and sum of squares of returns - values over the minibatch's rows, accumulated in double) against torch."""
    import ctypes as C
    from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(n)
    rows = 3 * n + 5
    ret, val = torch.randn(rows, device=dev, generator=g) + 0.3, torch.randn(rows, device=dev, generator=g)
    idx = torch.randperm(rows, device=dev, generator=g)[:n].contiguous() if indexed else None
    scratch = torch.empty(2 * 256 + 3, device=dev, dtype=torch.float64)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    stats = torch.empty(2, device=dev)
    assert lib.irrl_adv_moments(n, p(idx), p(ret), p(val), p(scratch), 256, p(scratch[512:]), p(stats), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)) == 0
    a = ((ret[idx] - val[idx]) if indexed else (ret[:n] - val[:n])).double()
    want = torch.stack([a.sum(), (a * a).sum(), torch.tensor(float(n), device=dev, dtype=torch.float64)])
    np.testing.assert_allclose(scratch[512:].cpu().numpy(), want.cpu().numpy(), rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(stats.cpu().numpy(), [float(a.mean()), float(a.std(unbiased=False))], rtol=2e-6, atol=1e-7)


def test_mlp_ppo_update_with_gradient_kernels_follows_the_eager_update():
    """One PPO2 update of the MlpPolicy learner (4 minibatches x 2 epochs) through the gradient kernels against the eager graph
    from the same rollout, generator and initial weights: the parameters after 8 Adam steps agree to rounding."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import MlpPolicy
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2, Runner
    after = {}
    for fused in (True, False):
        env = _env(64)
        model = PPO2(policy=MlpPolicy, env=env, n_steps=32, nminibatches=4, noptepochs=2, seed=4)
        model.fused_mlp = fused
        model.fused_loss = fused
        runner = Runner(env, model, 32, 0.99, 0.95)
        batch = runner.run()
        stats = model.update(batch, 3e-4, 0.2)
        after[fused] = ([q.detach().clone() for q in model.policy.parameters()], stats)
    for a, b in zip(after[True][0], after[False][0]):
        assert float((a - b).abs().max()) < 5e-5, float((a - b).abs().max())
    np.testing.assert_allclose(after[True][1].cpu().numpy(), after[False][1].cpu().numpy(), rtol=2e-3, atol=2e-4)


def test_graph_capture_warmup_leaves_no_trace_and_setters_invalidate_the_graph():
    """Ways to issue the fused rollout -- "persistent" (ONE launch for the whole rollout), "one_launch" (one per step), "direct" (the default: 2 x T launches from one C call, irrl_lstm_rollout), "graph"
    (one hipGraph of 2 x T kernel nodes) and "eager" (one Python call per launch) -- give the same rollouts bit for bit:
    (a) the three warm-up steps in front of the hipGraph capture are undone (device snapshot of the env pool + the runner's
    tensors), i.e. the FIRST rollout of a graph runner starts from env.reset();
    (b) a setter that changes a by-value kernel argument after the capture (setSeed) is not silently ignored: the graph runner
    re-captures; the direct runner reads the parameters at launch time anyway.
    "persistent_actor" (round 5: the persistent launch with the critic OFF the per-step path -- the actor stack alone per step, the values of
    the whole rollout from the critic's sequence kernels afterwards): everything the actor and the env produce -- observations, actions,
    neglogp, rewards, dones -- is bit-identical to the other modes; values and returns agree at the f32 level (2e-5 of their scale).  It comes
    in two forms: the actor as each WAVE's own work for its four robots (v_mfma_f32_4x4x1 chains, LSTM state in LDS / registers for the whole
    rollout: the default since the second half of round 5) and the workgroup-wide actor step ("persistent_actor_wg": IRRL_ACTOR_WAVES=0)."""
    import os
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2, Runner
    out = {}
    for mode in ("direct", "one_launch", "persistent", "persistent_actor", "persistent_actor_wg", "graph", "eager"):
        os.environ.pop("IRRL_ACTOR_WAVES", None)
        if mode == "persistent_actor_wg":
            os.environ["IRRL_ACTOR_WAVES"] = "0"
        env = _env(64 if mode != "persistent" else 64)
        model = PPO2(policy=CustomLSTMPolicy, env=env, n_steps=20, nminibatches=1, noptepochs=1, seed=9)
        runner = Runner(env, model, 20, 0.99, 0.998, use_graph=(mode != "eager"))
        assert runner.rollout_launch == "direct"      # what a Runner picks by itself for the LSTM policy on the HIP engine
        runner.rollout_launch = "direct" if mode in ("direct", "one_launch", "persistent", "persistent_actor", "persistent_actor_wg") else "graph"
        # 1: env.step k + policy step k + 1 in one kernel; 2: the whole rollout as ONE persistent launch (a workgroup loops over all
        # steps for its 16 robots, irrl_rollout_persistent_kernel_l16)
        runner.rollout_one_launch_per_step = {"one_launch": 1, "persistent": 2, "persistent_actor": 3, "persistent_actor_wg": 3}.get(mode, 0)
        b1 = {k: v.clone() for k, v in runner.run().items() if torch.is_tensor(v)}
        assert (runner._graph is not None) == (mode == "graph")
        assert runner._actor_only_supported() or mode not in ("persistent_actor", "persistent_actor_wg")   # asked through the C-ABI, not parsed from an error text
        env.wrapper.setSeed(77)                       # new noise / command streams from the next reset on
        b2 = {k: v.clone() for k, v in runner.run().items() if torch.is_tensor(v)}
        b3 = {k: v.clone() for k, v in runner.run().items() if torch.is_tensor(v)}
        out[mode] = (b1, b2, b3)
    os.environ.pop("IRRL_ACTOR_WAVES", None)
    for mode in ("direct", "one_launch", "persistent", "graph"):
        for i in range(3):
            for k in ("obs", "actions", "values", "true_reward", "masks", "neglogpacs", "returns"):
                assert torch.equal(out[mode][i][k], out["eager"][i][k]), (mode, i, k)
    for pa in ("persistent_actor", "persistent_actor_wg"):
        for i in range(3):
            for k in ("obs", "actions", "true_reward", "masks", "neglogpacs", "states"):
                assert torch.equal(out[pa][i][k], out["eager"][i][k]) or k == "states", (pa, i, k)
            # the carried LSTM state: the actor's half bit-identical, the critic's half (from the sequence kernels) at the f32 level
            sa, se = out[pa][i]["states"], out["eager"][i]["states"]
            assert torch.equal(sa[:, :192], se[:, :192]), (pa, i)
            assert float((sa[:, 192:] - se[:, 192:]).abs().max()) < 2e-5 * (1.0 + float(se[:, 192:].abs().max()))
            for k in ("values", "returns"):
                a, b = out[pa][i][k], out["eager"][i][k]
                assert float((a - b).abs().max()) < 2e-5 * (1.0 + float(b.abs().max())), (pa, i, k, float((a - b).abs().max()))
    # the two forms of the actor-only rollout among themselves: everything, the critic pass included, bit for bit
    for i in range(3):
        for k in out["persistent_actor"][i]:
            assert torch.equal(out["persistent_actor"][i][k], out["persistent_actor_wg"][i][k]), ("the two actor-only forms", i, k)
    assert not torch.equal(out["graph"][1]["obs"], out["graph"][0]["obs"])


@pytest.mark.parametrize("cfg", ["default_cfg.yaml", "bp5_terrain.yaml"])
def test_mlp_rollout_modes_give_the_same_rollouts_bit_for_bit(monkeypatch, cfg):
    """MlpPolicy rollouts (config 2's learner): "persistent" (the default: ONE launch for the whole rollout, a workgroup keeps its 16 robots
    and the policy's weights in LDS for all steps -- irrl_rollout_persistent_mlp_kernel_l16), "direct" (2 x T launches from one C call,
    irrl_mlp_rollout), "graph" (one hipGraph of 2 x T nodes) and "eager" (one Python call per launch) agree bit for bit over three
    rollouts of 160 steps with a reseed in between (in-step resets included when robots fall) -- on the training configuration and on
    bp5_terrain.yaml, whose in-step resets re-draw every robot's masses / centres of mass / friction (RandomizePerEpisode: the persistent
    kernels' robots once left the step kernel's by an ulp there, csrc/env_core.hpp model_randomize)."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd import lstm_fused
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import MlpPolicy
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2, Runner
    assert lstm_fused.MLP_ROLLOUT == "persistent"
    out = {}
    for mode in ("persistent", "direct", "graph", "eager"):
        monkeypatch.setattr(lstm_fused, "MLP_ROLLOUT", "direct" if mode == "direct" else "persistent")
        env = _env(96, cfg)
        model = PPO2(policy=MlpPolicy, env=env, n_steps=160, nminibatches=1, noptepochs=1, seed=9)
        runner = Runner(env, model, 160, 0.99, 0.998, use_graph=(mode != "eager"))
        assert runner.rollout_launch == "direct"      # what a Runner picks by itself on the HIP engine: the policy's own rollout call
        runner.rollout_launch = "direct" if mode in ("direct", "persistent") else "graph"
        b1 = {k: v.clone() for k, v in runner.run().items() if torch.is_tensor(v)}
        env.wrapper.setSeed(77)
        b2 = {k: v.clone() for k, v in runner.run().items() if torch.is_tensor(v)}
        b3 = {k: v.clone() for k, v in runner.run().items() if torch.is_tensor(v)}
        out[mode] = (b1, b2, b3)
    print("[mlp rollout modes] episodes ended inside the three rollouts:", [int(out["eager"][i]["masks"].sum()) for i in range(3)])
    for mode in ("persistent", "direct", "graph"):
        for i in range(3):
            for k in ("obs", "actions", "values", "true_reward", "masks", "neglogpacs", "returns"):
                assert torch.equal(out[mode][i][k], out["eager"][i][k]), (mode, i, k)


# ------------------------------------------------------------------------------------------------------------------------
# Round 4: the optimizer step on flat buffers (ppo2.FlatParams, csrc/ppo_optim.hpp)
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,max_norm,world", [(70741, 0.5, 1), (13 * 64 + 3, 0.5, 2), (4096, 0.0, 1), (70741, 1e9, 8)])
def test_clip_adam_kernel_matches_tf_clip_and_adam(n, max_norm, world):
    """`irrl_clip_adam` (global-norm clip + Adam over flat buffers in one launch; ppo2.py:182-197) against tf.clip_by_global_norm +
    tf.train.AdamOptimizer(eps 1e-5) -- `ppo2.clip_by_global_norm_` + `ppo2.TFAdam`, themselves pinned to a float64 transcription of the
    TensorFlow formulas in tests/test_ppo_math.py -- over 12 steps with changing gradients and learning rates, including a gradient far
    above / below the clip threshold and the 1 / world scale of the all-reduced sum; the same launch twice gives the same bits.  The first
    step also shows that this is TensorFlow's epsilon placement, not torch.optim.Adam's."""
    import ctypes as C
    from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(5)
    theta0 = torch.randn(n, device=dev, generator=g)
    pad = (-n) % 4
    bufs = [torch.zeros(n + pad + 4, device=dev) for _ in range(8)]       # two sets of (theta, grad, m, v), 16-byte aligned
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import TFAdam, clip_by_global_norm_
    ref = torch.nn.Parameter(theta0.clone())
    opt = TFAdam([ref], lr=1e-3, eps=1e-5, betas=(0.9, 0.999))
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for k in (0, 4):
        bufs[k][:n].copy_(theta0)
    norm_out = torch.zeros(1, device=dev)
    for step in range(1, 13):
        scale = [1.0, 300.0, 1e-3, 5.0][step % 4]
        grad_sum = scale * torch.randn(n, device=dev, generator=g) * world          # what the all-reduce leaves: the SUM over ranks
        lr = 1e-3 * (1.0 - step / 20.0)
        ref.grad = (grad_sum / world).clone()
        total = float(ref.grad.norm())
        if max_norm > 0:
            clip_by_global_norm_([ref], max_norm)
        for gr in opt.param_groups:
            gr["lr"] = lr
        opt.step()
        for k in (0, 4):
            bufs[k + 1][:n].copy_(grad_sum)
            _lib.check(lib.irrl_clip_adam(n, p(bufs[k]), p(bufs[k + 1]), p(bufs[k + 2]), p(bufs[k + 3]), 1.0 / world, max_norm, lr, 0.9, 0.999, 1e-5,
                                          step, p(norm_out), stream))
        assert torch.equal(bufs[0], bufs[4])                                          # deterministic
        assert abs(float(norm_out) - total) < 2e-5 * total
        err = float((bufs[0][:n] - ref.detach()).abs().max())
        assert err < 3e-6, (step, err)
    assert float((bufs[0][:n] - theta0).abs().max()) > 5e-3                           # the parameters did move
    assert float(bufs[0][n:].abs().max()) == 0.0                                       # nothing written behind the n parameters


@pytest.mark.parametrize("n,indexed", [(16 * 7 + 5, False), (4096 * 6 + 3, True), (200000, True)])
def test_packed_sample_records_give_the_same_gradients_bit_for_bit(n, indexed, monkeypatch):
    """Round 5: the update's samples as ONE 256-byte record each (`irrl_mlp_pack_records`: observation | action | return | old value |
    old neglogp | advantage; two 128-byte lines instead of the seven or eight a shuffled row of the five arrays costs).  The record holds
    the same values, the REC instantiation of the bf16 gradient kernels does the same arithmetic on them: every gradient, the statistics
    row and the advantage moments equal the five-array path BIT FOR BIT (ragged last tile, shuffled index into a larger rollout)."""
    import ctypes as C
    from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib, ppo2 as P2
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import MlpPolicy, diag_gaussian_neglogp
    monkeypatch.setattr(P2, "MLP_PRECISION", "bf16x3")
    dev = torch.device("cuda")
    torch.manual_seed(4)
    pol = MlpPolicy().to(dev)
    flat = P2.FlatParams(pol)
    g = torch.Generator(device=dev); g.manual_seed(12)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    rows = n if not indexed else 2 * n + 77
    obs, actions, returns, old_v = rn(rows, 35), 0.5 * rn(rows, 12), rn(rows), rn(rows)
    with torch.no_grad():
        old_nlp = diag_gaussian_neglogp(actions, pol._run(obs)[0], pol.logstd) + 0.3 * rn(rows)
    index = torch.randperm(rows, device=dev, generator=g)[:n].contiguous() if indexed else None
    rec = P2.mlp_pack_records(obs, actions, returns, old_v, old_nlp)
    assert rec.shape == (rows, 64) and rec.data_ptr() % 256 == 0
    r = rec.cpu().numpy()
    np.testing.assert_array_equal(r[:, :35], obs.cpu().numpy()); np.testing.assert_array_equal(r[:, 36:48], actions.cpu().numpy())
    np.testing.assert_array_equal(r[:, 48], returns.cpu().numpy()); np.testing.assert_array_equal(r[:, 49], old_v.cpu().numpy())
    np.testing.assert_array_equal(r[:, 50], old_nlp.cpu().numpy()); np.testing.assert_array_equal(r[:, 51], (returns - old_v).cpu().numpy())
    assert float(np.abs(r[:, 35]).max()) == 0.0 and float(np.abs(r[:, 52:]).max()) == 0.0
    # advantage moments: records against the arrays
    lib = _lib.load()
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    sums, stats = [], []
    for use_rec in (False, True):
        scratch = torch.zeros(2 * 256 + 3, device=dev, dtype=torch.float64)
        st = torch.zeros(2, device=dev)
        if use_rec:
            _lib.check(lib.irrl_adv_moments_rec(n, p(index), p(rec), p(scratch), 256, p(scratch[512:]), p(st), stream))
        else:
            _lib.check(lib.irrl_adv_moments(n, p(index), p(returns), p(old_v), p(scratch), 256, p(scratch[512:]), p(st), stream))
        sums.append(scratch[512:].clone()); stats.append(st.clone())
    assert torch.equal(sums[0], sums[1]) and torch.equal(stats[0], stats[1])
    out = []
    for use_rec in (False, True):
        flat.grad.zero_()
        row = P2.mlp_ppo_grads_flat(pol, flat, obs, actions, returns, old_v, old_nlp, stats[0], 0.2, 0.01, 0.5, index, rec=rec if use_rec else None)
        out.append((flat.grad.clone(), row.clone()))
    assert float(out[0][0].abs().max()) > 0
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize("n,indexed,n_blocks", [(16 * 7 + 5, False, 256), (16 * 1024 * 3 + 16 * 5 + 3, True, 256), (200000, True, 256), (50000, True, 7)])
def test_wave_pair_mlp_gradient_kernels_equal_the_four_wave_kernels_bit_for_bit(n, indexed, n_blocks, monkeypatch):
    """Round 5: the bf16 MlpPolicy gradient kernels with two waves per SIMD (csrc/mlp_bf16_pc.hpp: producer waves leave the tensors of a tile in
    LDS images, consumer waves run the weight-gradient products a tile behind) against the one-wave-per-SIMD kernels of mlp_bf16.hpp
    (`IRRL_MLP_WAVES=4`): the same products in the same order on the same accumulators -- every per-workgroup partial row, hence every gradient
    and statistic, BIT FOR BIT; sample counts that leave the last tile ragged, pairs that run out of tiles before others (their barriers
    still run), workgroups without any tile, and the packed-record instantiations."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd import ppo2 as P2
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import MlpPolicy, diag_gaussian_neglogp
    monkeypatch.setattr(P2, "MLP_PRECISION", "bf16x3")
    dev = torch.device("cuda")
    torch.manual_seed(5)
    pol = MlpPolicy().to(dev)
    flat = P2.FlatParams(pol)
    g = torch.Generator(device=dev); g.manual_seed(21)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    rows = n if not indexed else 2 * n + 77
    obs, actions, returns, old_v = rn(rows, 35), 0.5 * rn(rows, 12), rn(rows), rn(rows)
    with torch.no_grad():
        old_nlp = diag_gaussian_neglogp(actions, pol._run(obs)[0], pol.logstd) + 0.3 * rn(rows)
    index = torch.randperm(rows, device=dev, generator=g)[:n].contiguous() if indexed else None
    rec = P2.mlp_pack_records(obs, actions, returns, old_v, old_nlp)
    stats = torch.tensor([0.1, 0.9], device=dev)
    out = {}
    for waves in ("4", "8"):
        monkeypatch.setenv("IRRL_MLP_WAVES", waves)
        for use_rec in (False, True):
            flat.grad.zero_()
            row = P2.mlp_ppo_grads_flat(pol, flat, obs, actions, returns, old_v, old_nlp, stats, 0.2, 0.01, 0.5, index, n_blocks=n_blocks,
                                        rec=rec if use_rec else None)
            torch.cuda.synchronize()
            out[waves, use_rec] = (flat.grad.clone(), row.clone())
    assert float(out["4", False][0].abs().max()) > 0
    for use_rec in (False, True):
        assert torch.equal(out["4", use_rec][0], out["8", use_rec][0]), use_rec
        assert torch.equal(out["4", use_rec][1], out["8", use_rec][1]), use_rec


def test_mlp_update_through_packed_records_equals_the_update_through_the_arrays(monkeypatch):
    """PPO2.update of the shipped MlpPolicy configuration (4 minibatches x 3 epochs here) with the packed records (off by default; IRRL_MLP_RECORDS=1 / ppo2.MLP_RECORDS enables them) and without
    (`ppo2.MLP_RECORDS = False`): the same parameters and statistics, bit for bit."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd import ppo2 as P2
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import MlpPolicy
    after = {}
    for use_rec in (True, False):
        monkeypatch.setattr(P2, "MLP_RECORDS", use_rec)
        env = _env(256)
        model = P2.PPO2(policy=MlpPolicy, env=env, n_steps=48, nminibatches=4, noptepochs=3, learning_rate=1e-3, seed=8)
        runner = P2.Runner(env, model, 48, 0.99, 0.95)
        batch = runner.run()
        stats = model.update(batch, 1e-3, 0.2)
        after[use_rec] = (model.flat.theta.clone(), stats.clone())
        assert model._records is None
    assert torch.equal(after[True][0], after[False][0]) and torch.equal(after[True][1], after[False][1])


@pytest.mark.parametrize("kind", ["lstm", "mlp"])
def test_flat_optimizer_step_follows_torch_adam_on_the_same_views(kind):
    """One PPO2 update with the optimizer step on flat buffers (one gather launch, `irrl_clip_adam`) against the same update with
    `clip_by_global_norm_` + `TFAdam` on the same parameter views: same parameters to rounding; every parameter IS a view of the flat
    buffer; the MlpPolicy gradient kernels' scatter into the flat gradient equals `mlp_ppo_grads` bit for bit."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd import ppo2 as P2
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy, MlpPolicy
    after = {}
    for flat_optim in (True, False):
        env = _env(64)
        model = P2.PPO2(policy=CustomLSTMPolicy if kind == "lstm" else MlpPolicy, env=env, n_steps=32, nminibatches=1 if kind == "lstm" else 4,
                        noptepochs=3, learning_rate=1e-3, seed=8)
        model.flat_optim = flat_optim
        fl = model.flat
        for prm, off in zip(fl.params, fl.offsets):
            assert prm.data_ptr() == fl.theta.data_ptr() + 4 * off and off % 64 == 0
        runner = P2.Runner(env, model, 32, 0.99, 0.95)
        batch = runner.run()
        stats = model.update(batch, 1e-3, 0.2)
        after[flat_optim] = (fl.theta.clone(), stats)
        assert float(fl.theta.abs().max()) > 0 and model.flat.step == (3 if kind == "lstm" else 12) * int(flat_optim)
    assert float((after[True][0] - after[False][0]).abs().max()) < 3e-6
    np.testing.assert_allclose(after[True][1].cpu().numpy(), after[False][1].cpu().numpy(), rtol=1e-4, atol=1e-6)
    if kind == "mlp":
        dev = model.device
        g = torch.Generator(device=dev); g.manual_seed(2)
        rows = 5000
        obs, actions = torch.randn(rows, 35, device=dev, generator=g), 0.5 * torch.randn(rows, 12, device=dev, generator=g)
        ret, ov, onlp = (torch.randn(rows, device=dev, generator=g) for _ in range(3))
        onlp = onlp + 9.0
        index = torch.randperm(rows, device=dev, generator=g)[:3000].contiguous()
        st = torch.tensor([0.1, 1.3], device=dev)
        _loss, stats, grads = P2.mlp_ppo_grads(model.policy, obs, actions, ret, ov, onlp, st, 0.2, 0.01, 0.5, index=index)
        model.ent_coef = 0.01
        row = P2.mlp_ppo_grads_flat(model.policy, model.flat, obs, actions, ret, ov, onlp, st, 0.2, 0.01, 0.5, index)
        for prm, gv in zip(model.flat.params, model.flat.grad_views):
            if prm in grads:
                assert torch.equal(gv, grads[prm].reshape(gv.shape)), tuple(prm.shape)
        np.testing.assert_allclose(P2.mlp_stats_rows([row], 3000.0, 12)[0].cpu().numpy(), stats.cpu().numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("n", [1, 7, 4096, 100003, 4096 * 750])
def test_random_permutation_kernel_is_a_permutation_and_equals_its_numpy_twin(n):
    """`irrl_random_permutation` (the epoch's shuffled sample order without the radix sort of torch.randperm): a bijection of [0, n), the same
    bits as ppo2.feistel_permutation (what the CPU path uses), a different order for a different counter."""
    import ctypes as C
    from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import feistel_permutation
    lib = _lib.load()
    dev = torch.device("cuda")
    out = torch.empty(n, dtype=torch.int64, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.irrl_random_permutation(n, 987654321, 5, C.c_void_p(out.data_ptr()), stream))
    got = out.cpu().numpy()
    assert np.array_equal(np.sort(got), np.arange(n))
    assert np.array_equal(got, feistel_permutation(n, 987654321, 5))
    if n > 100:
        _lib.check(lib.irrl_random_permutation(n, 987654321, 6, C.c_void_p(out.data_ptr()), stream))
        assert float((out.cpu().numpy() != got).mean()) > 0.99
        assert abs(float(np.corrcoef(np.arange(n), got)[0, 1])) < 0.05


def test_actor_only_rollout_is_chosen_by_a_capability_query_not_by_an_error_text(monkeypatch):
    """the runner asks `irrl_lstm_rollout_supports(pool, hid, 3)` before it launches.  A 16-lane pool has the actor-only persistent
    kernel, a 4-lane pool has not: there the default mode (3) silently becomes 2 -- two launches per step inside the C call -- with the same
    rollout as the eager reference path, and an explicit fuse = 3 call is still refused by the C-ABI with rc != 0."""
    import torch
    from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2, Runner
    lib = _lib.load()
    env16 = _env(64)
    assert env16.wrapper.lanes_per_robot == 16
    assert [lib.irrl_lstm_rollout_supports(env16.wrapper._h, 48, f) for f in (0, 1, 2, 3, 4)] == [1, 1, 1, 1, 0]
    assert lib.irrl_lstm_rollout_supports(env16.wrapper._h, 64, 3) == 0            # the combined kernels are instantiated for 48 units
    monkeypatch.setenv("IRRL_LANES_PER_ROBOT", "4")
    outs = {}
    for mode in ("default", "eager"):
        env4 = _env(64)
        assert env4.wrapper.lanes_per_robot == 4
        assert lib.irrl_lstm_rollout_supports(env4.wrapper._h, 48, 3) == 0 and lib.irrl_lstm_rollout_supports(env4.wrapper._h, 48, 0) == 1
        model = PPO2(policy=CustomLSTMPolicy, env=env4, n_steps=12, nminibatches=1, noptepochs=1, seed=4)
        runner = Runner(env4, model, 12, 0.99, 0.998, use_graph=False)
        if mode == "eager":
            runner.rollout_launch = "graph"          # with use_graph False: the per-step Python loop
        else:
            assert runner.rollout_one_launch_per_step == 3 and not runner._actor_only_supported()
        outs[mode] = {k: v.clone() for k, v in runner.run().items() if torch.is_tensor(v)}
    for k in ("obs", "actions", "values", "true_reward", "masks", "neglogpacs", "returns"):
        assert torch.equal(outs["default"][k], outs["eager"][k]), k


@pytest.mark.parametrize("T,N,n_in", [(750, 4096, 48), (37, 48, 35), (1, 16, 35), (2, 32, 48), (64, 100, 35)])
def test_recomputing_backward_kernel_equals_the_gate_loading_one_bit_for_bit(T, N, n_in, monkeypatch):
    """Round 6 (verdict r5 item 2): with bf16x3 the update's forward kernel keeps c and h only and the backward kernel recomputes the gates from the
    h / x tiles it stages for the weight gradients (`lstm_seq_bwd_bf16_rc_kernel`) -- the forward kernel's own products in its own order, so EVERY
    output (h, final state, dx, dwx, dwh, db) equals the gate-storing / gate-loading pair's bit for bit.  Training shape, ragged shapes (N not a
    multiple of 16: padded; n_in 35: ragged rows), T = 1 and 2 (prologue / first-step edges), episode boundaries inside the sequence."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd import lstm_fused
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import SBLstm
    monkeypatch.setattr(lstm_fused, "PRECISION", "bf16x3")
    dev = torch.device("cuda")
    torch.manual_seed(T * 7 + N)
    layer = SBLstm(n_in, 48).to(dev)
    with torch.no_grad():
        layer.b.copy_(torch.randn(192, device=dev) * 0.1)
    x = torch.randn(T, N, n_in, device=dev)
    state = torch.randn(N, 96, device=dev) * 0.5
    masks = (torch.rand(T, N, device=dev) < 0.05).float()
    wgt = torch.randn(T, N, 48, device=dev) / (T * N) ** 0.5

    def run(recompute):
        monkeypatch.setattr(lstm_fused, "RECOMPUTE_GATES", recompute)
        xx = x.clone().requires_grad_(True)
        for p in layer.parameters():
            p.grad = None
        h, s = layer.sequence(xx, state, masks)
        (h * wgt).sum().backward()
        return [h.detach().clone(), s.detach().clone(), xx.grad.clone()] + [p.grad.clone() for p in layer.parameters()]

    stored, recomputed = run(False), run(True)
    for name, a, b in zip(["h_seq", "state", "dx", "dwx", "dwh", "db"], stored, recomputed):
        assert torch.equal(a, b), (name, float((a - b).abs().max()))
    assert float(recomputed[3].abs().max()) > 0 and float(recomputed[2].abs().max()) > 0
