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


@pytest.mark.parametrize("T,N,n_in,hid", [(7, 16, 35, 48), (33, 40, 48, 48), (5, 3, 35, 32), (12, 64, 20, 64)])
def test_fused_lstm_sequence_matches_eager_definition(T, N, n_in, hid):
    """Persistent MFMA LSTM kernels (forward + BPTT) against the eager stable-baselines definition (SBLstm.sequence),
    same f32 inputs: outputs, final state and every gradient."""
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import SBLstm
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
        assert float((a - b).abs().max()) / scale < 2e-5, (name, float((a - b).abs().max()), scale)


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
