"""Actor-critic policies of the reference, in PyTorch (runs on the env's MI355X; no host round-trip).

  CustomLSTMPolicy  run_bp_v5.py:117-193: actor LSTM(35->48)->LSTM(48->48)->linear 12 (mean), state-independent
                    logstd[1,12]; critic LSTM(35->48)->LSTM(48->48)->linear 1; an unused `q` head on the critic
                    latent.  The LSTM is stable-baselines 2.8 `a2c.utils.lstm`: wx[n_in,4h], wh[h,4h], b[4h],
                    gate order i,f,o,g, state tensor [c,h], state*(1-mask) before every step (pinned in-repo by
                    script/utils/CustomerLstmNN.py:116-130).  states [N,384] = pi-L0(96) pi-L1(96) v-L0(96) v-L1(96).
  MlpPolicy         flex_gym/archi/policies.py:395-462,524-541: separate pi / vf MLPs [64,64] tanh, pi head
                    init_scale 0.01.

Parameter order of CustomLSTMPolicy.sb_parameters() == the 19 tensors of the reference checkpoint
(script/pkl/bp5_155.pkl): pi-L0 (wx,wh,b) pi-L1 v-L0 v-L1, vf (w,b), pi (w,b), logstd, q (w,b).
"""
import math

import torch
import torch.nn as nn

LOG_2PI = math.log(2.0 * math.pi)


def _ortho(shape, scale=1.0):
    """stable-baselines `ortho_init` (a2c/utils.py): orthogonal matrix from the SVD of a gaussian."""
    a = torch.randn(shape[0], shape[1])
    u, _, vt = torch.linalg.svd(a, full_matrices=False)
    q = u if u.shape == tuple(shape) else vt
    return (scale * q[: shape[0], : shape[1]]).contiguous()


class SBLstm(nn.Module):
    """One stable-baselines lstm layer (see module docstring)."""

    use_fused = True   # class-wide switch (tests flip it to compare the kernels with this eager definition)
    precision = None   # arithmetic of this layer's sequence kernels; None = lstm_fused.PRECISION (CustomLSTMPolicy sets it per stack)

    def __init__(self, n_in, n_hidden):
        super().__init__()
        self.n_hidden = n_hidden
        self.wx = nn.Parameter(_ortho((n_in, 4 * n_hidden)))
        self.wh = nn.Parameter(_ortho((n_hidden, 4 * n_hidden)))
        self.b = nn.Parameter(torch.zeros(4 * n_hidden))

    def prepare(self):
        """Refresh the fused kernels' permuted weight copies from the parameters, unconditionally: the learner calls this
        after every optimizer step, the runner before every rollout (a captured graph cannot call back into Python)."""
        if self.wx.is_cuda and self.use_fused:
            from . import lstm_fused
            lstm_fused.refresh_weights(self.wx, self.wh, self.b)

    def cell(self, zx, c, h, mask):
        """zx = x @ wx + b precomputed; mask [N,1] = done flag before this step."""
        keep = 1.0 - mask
        c = c * keep
        h = h * keep
        z = zx + h @ self.wh
        n = self.n_hidden
        i, f, o, g = z[:, :n], z[:, n:2 * n], z[:, 2 * n:3 * n], z[:, 3 * n:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        return c, h

    def sequence(self, x, state, masks, precision=None):
        """x [T,N,n_in], state [N,2h] = [c,h], masks [T,N] -> outputs [T,N,h], final state [N,2h].
        The input projection of the whole sequence is one GEMM; only h @ wh stays sequential.
        precision: arithmetic of the fused kernels for this call (overrides the layer's / the module's default)."""
        T, N, _ = x.shape
        n = self.n_hidden
        if x.is_cuda and self.use_fused:
            from . import lstm_fused
            if lstm_fused.supported(x, n):
                # persistent MFMA kernel: the whole sequence in one launch (csrc/lstm_kernels.hip)
                return lstm_fused.lstm_sequence(x, self.wx, self.wh, self.b, state, masks,
                                                precision=precision if precision is not None else self.precision)
        zx = (x.reshape(T * N, -1) @ self.wx + self.b).reshape(T, N, 4 * n)
        c, h = state[:, :n], state[:, n:]
        m = masks.unsqueeze(-1)
        outs = []
        for t in range(T):
            c, h = self.cell(zx[t], c, h, m[t])
            outs.append(h)
        return torch.stack(outs, 0), torch.cat([c, h], 1)


class SBLinear(nn.Module):
    """stable-baselines `linear`: x @ w + b, w orthogonal * init_scale, b constant."""

    def __init__(self, n_in, n_out, init_scale=1.0, init_bias=0.0):
        super().__init__()
        self.w = nn.Parameter(_ortho((n_in, n_out), init_scale))
        self.b = nn.Parameter(torch.full((n_out,), float(init_bias)))

    def forward(self, x):
        if x.is_cuda and torch.is_grad_enabled() and self.w.requires_grad and x.numel() // x.shape[-1] >= (1 << 16) and SBLstm.use_fused:
            from . import lstm_fused
            return lstm_fused.tall_linear(x, self.w, self.b)      # split-K weight gradient (see _TallLinearFn)
        return x @ self.w + self.b


def diag_gaussian_neglogp(actions, mean, logstd):
    """DiagGaussianProbabilityDistribution.neglogp (SB distributions.py)."""
    std = torch.exp(logstd)
    return 0.5 * (((actions - mean) / std) ** 2).sum(-1) + 0.5 * LOG_2PI * actions.shape[-1] + logstd.sum(-1)


def diag_gaussian_entropy(logstd, like):
    return (logstd + 0.5 * (LOG_2PI + 1.0)).sum(-1).expand(like.shape[:-1])


class ActorCriticPolicy(nn.Module):
    """Common surface used by PPO2 / Runner (step, value, evaluate)."""
    recurrent = False
    state_dim = 0

    def initial_state(self, n_env, device):
        return torch.zeros(n_env, max(self.state_dim, 1), device=device)


class CustomLSTMPolicy(ActorCriticPolicy):
    recurrent = True

    def __init__(self, ob_dim=35, act_dim=12, n_lstm=(48, 48)):
        super().__init__()
        self.n_lstm = list(n_lstm)
        self.act_dim = act_dim
        dims = [ob_dim] + self.n_lstm
        self.lstm_pi = nn.ModuleList([SBLstm(dims[i], dims[i + 1]) for i in range(len(self.n_lstm))])
        self.lstm_v = nn.ModuleList([SBLstm(dims[i], dims[i + 1]) for i in range(len(self.n_lstm))])
        self.vf = SBLinear(dims[-1], 1)
        self.pi = SBLinear(dims[-1], act_dim, init_scale=1.0, init_bias=0.0)
        self.logstd = nn.Parameter(torch.zeros(1, act_dim))
        self.q = SBLinear(dims[-1], act_dim)  # created by proba_distribution_from_latent, never used
        self.state_dim = sum(self.n_lstm) * 2 * 2  # run_bp_v5.py:136-137
        # per-stack arithmetic of the update's sequence kernels (A/B runs that separate the actor's from the critic's): IRRL_LSTM_PRECISION_PI /
        # IRRL_LSTM_PRECISION_V; unset = lstm_fused.PRECISION for both
        import os
        for stack, key in ((self.lstm_pi, "IRRL_LSTM_PRECISION_PI"), (self.lstm_v, "IRRL_LSTM_PRECISION_V")):
            if os.environ.get(key):
                from . import lstm_fused
                for l in stack:
                    l.precision = lstm_fused.check_precision(os.environ[key], key)

    two_streams = True
    _streams = {}

    def _side_stream(self, device):
        key = str(device)
        if key not in CustomLSTMPolicy._streams:
            CustomLSTMPolicy._streams[key] = torch.cuda.Stream(device=device)
        return CustomLSTMPolicy._streams[key]

    def sb_parameters(self):
        out = []
        for stack in (self.lstm_pi, self.lstm_v):
            for l in stack:
                out += [l.wx, l.wh, l.b]
        out += [self.vf.w, self.vf.b, self.pi.w, self.pi.b, self.logstd, self.q.w, self.q.b]
        return out

    def prepare(self):
        for l in list(self.lstm_pi) + list(self.lstm_v):
            l.prepare()

    def _split(self, states):
        sizes = [2 * k for k in (self.n_lstm + self.n_lstm)]  # run_bp_v5.py:139-140
        return list(torch.split(states, sizes, dim=1))

    def _stack(self, layers, x, parts, masks_seq, precision=None):
        new = []
        for l, st in zip(layers, parts):
            x, s = l.sequence(x, st, masks_seq, precision=precision)
            new.append(s)
        return x, new

    def _run(self, obs_seq, states, masks_seq, latents_only=False):
        parts = self._split(states)
        k = len(self.n_lstm)
        if obs_seq.is_cuda and self.two_streams and obs_seq.shape[0] > 1 and not torch.cuda.is_current_stream_capturing():
            # actor and critic stacks are independent until the heads: run the critic on a second HIP stream so the two
            # persistent sequence kernels (768 waves each) share the chip; autograd replays the same stream assignment
            cur = torch.cuda.current_stream(obs_seq.device)
            side = self._side_stream(obs_seq.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                latent_v, new_v = self._stack(self.lstm_v, obs_seq, parts[k:], masks_seq)
            latent_pi, new_pi = self._stack(self.lstm_pi, obs_seq, parts[:k], masks_seq)
            cur.wait_stream(side)
            for t in [latent_v] + new_v:
                t.record_stream(cur)
        else:
            latent_pi, new_pi = self._stack(self.lstm_pi, obs_seq, parts[:k], masks_seq)
            latent_v, new_v = self._stack(self.lstm_v, obs_seq, parts[k:], masks_seq)
        new = new_pi + new_v
        if latents_only:
            return latent_pi, latent_v, torch.cat(new, 1)
        mean = self.pi(latent_pi)
        value = self.vf(latent_v).squeeze(-1)
        return mean, value, torch.cat(new, 1)

    def fused_step_supported(self, obs):
        if not (obs.is_cuda and SBLstm.use_fused):
            return False
        from . import lstm_fused
        return lstm_fused.policy_step_supported(self, obs)

    @torch.no_grad()
    def fused_step(self, obs, states, dones, noise=None, rng=None, states_out=None, rollout=None, out=None):
        """The whole step in ONE kernel launch (both stacks, heads, sample, neglogp, clip, rollout-buffer rows):
        -> action, clipped action, value, neglogp, new states (written to `states_out`, which may be `states`).
        See lstm_fused.policy_step for `noise` / `rng` / `rollout` / `out`."""
        from . import lstm_fused
        d = dones if dones.element_size() == 1 else (dones != 0)
        return lstm_fused.policy_step(self, obs, states, d.contiguous(), noise=noise, rng=rng, states_out=states_out, rollout=rollout, out=out)

    @torch.no_grad()
    def fused_rollout(self, env_impl, steps, obs, states, dones, rng, rollout, out, env_reward, env_extra, noise_all=None, fused=False):
        """`steps` fused_step + env.step pairs issued by one C call (lstm_fused.policy_rollout)"""
        from . import lstm_fused
        lstm_fused.policy_rollout(self, env_impl, steps, obs, states, dones, rng, rollout, out, env_reward, env_extra, noise_all=noise_all, fused=fused)

    @torch.no_grad()
    def step(self, obs, states, masks, deterministic=False, generator=None, noise=None):
        """run_bp_v5.py:178-185: -> action (unclipped sample), value, new states, neglogp.  `noise` [N,act] overrides the
        generator draw (pre-drawn standard normals)."""
        if self.fused_step_supported(obs):
            if deterministic:
                noise = None
            elif noise is None:
                noise = torch.randn((obs.shape[0], self.act_dim), device=obs.device, dtype=obs.dtype, generator=generator)
            action, _, value, neglogp, snew = self.fused_step(obs, states, masks, noise=noise)
            return action, value, snew, neglogp
        mean, value, snew = self._run(obs.unsqueeze(0), states, masks.to(obs.dtype).unsqueeze(0))
        mean, value = mean[0], value[0]
        if deterministic:
            action = mean
        else:
            if noise is None:
                noise = torch.randn(mean.shape, device=mean.device, dtype=mean.dtype, generator=generator)
            action = mean + torch.exp(self.logstd) * noise
        return action, value, snew, diag_gaussian_neglogp(action, mean, self.logstd)

    @torch.no_grad()
    def value(self, obs, states, masks):
        return self._run(obs.unsqueeze(0), states, masks.to(obs.dtype).unsqueeze(0))[1][0]

    def evaluate_raw(self, obs_seq, states, masks_seq):
        """Train-graph forward up to the distribution parameters: -> mean [T,N,act], value [T,N] (the fused loss kernel
        computes neglogp / entropy / the clipped objectives and their gradients from these)."""
        mean, value, _ = self._run(obs_seq, states, masks_seq.to(obs_seq.dtype))
        return mean, value

    def latents(self, obs_seq, states, masks_seq):
        """Train-graph forward up to the heads' inputs: -> (latent_pi, latent_v) [T,N,H]; the heads, the loss and their gradients
        then run in one launch (ppo2._FusedHeadsLoss, csrc `irrl_ppo_heads_loss`)."""
        lp, lv, _ = self._run(obs_seq, states, masks_seq.to(obs_seq.dtype), latents_only=True)
        return lp, lv

    def fused_heads_supported(self, obs):
        return bool(obs.is_cuda and SBLstm.use_fused and self.n_lstm[-1] == 48 and self.act_dim == 12)

    def evaluate(self, obs_seq, states, masks_seq, actions_seq):
        """Train-graph forward: obs [T,N,35], states [N,384] at the rollout start, masks [T,N], actions [T,N,12]
        -> neglogp [T,N], value [T,N], entropy [T,N] (full 750-step BPTT, ppo2.py:132-134)."""
        mean, value, _ = self._run(obs_seq, states, masks_seq.to(obs_seq.dtype))
        return diag_gaussian_neglogp(actions_seq, mean, self.logstd), value, diag_gaussian_entropy(self.logstd, mean)


class MlpPolicy(ActorCriticPolicy):
    """FeedForwardPolicy with net_arch [dict(vf=[64,64], pi=[64,64])], tanh (policies.py:430-446)."""

    def __init__(self, ob_dim=35, act_dim=12, layers=(64, 64)):
        super().__init__()
        dims = [ob_dim] + list(layers)
        sq2 = math.sqrt(2.0)
        self.pi_fc = nn.ModuleList([SBLinear(dims[i], dims[i + 1], init_scale=sq2) for i in range(len(layers))])
        self.vf_fc = nn.ModuleList([SBLinear(dims[i], dims[i + 1], init_scale=sq2) for i in range(len(layers))])
        self.vf = SBLinear(dims[-1], 1)
        self.pi = SBLinear(dims[-1], act_dim, init_scale=0.01, init_bias=0.0)  # policies.py:443-444
        self.logstd = nn.Parameter(torch.zeros(1, act_dim))
        self.q = SBLinear(dims[-1], act_dim, init_scale=0.01)
        self.act_dim = act_dim

    def sb_parameters(self):
        """stable-baselines variable order of a FeedForwardPolicy with net_arch [dict(vf=[64,64], pi=[64,64])] (archi/policies.py:79-88:
        pi_fc{i} then vf_fc{i} per layer; then vf, pi, logstd, q): lets reference MLP pickles load and this build's load there."""
        out = []
        for lp, lv in zip(self.pi_fc, self.vf_fc):
            out += [lp.w, lp.b, lv.w, lv.b]
        out += [self.vf.w, self.vf.b, self.pi.w, self.pi.b, self.logstd, self.q.w, self.q.b]
        return out

    def _run(self, obs):
        p = obs
        for l in self.pi_fc:
            p = torch.tanh(l(p))
        v = obs
        for l in self.vf_fc:
            v = torch.tanh(l(v))
        return self.pi(p), self.vf(v).squeeze(-1)

    def fused_step_supported(self, obs):
        if not (obs.is_cuda and SBLstm.use_fused):
            return False
        from . import lstm_fused
        return lstm_fused.mlp_policy_step_supported(self, obs)

    @torch.no_grad()
    def fused_step(self, obs, states, dones, noise=None, rng=None, states_out=None, rollout=None, out=None):
        """Whole step in one launch (csrc/lstm_kernels.hip: mlp_policy_step_kernel); same surface as the LSTM policy's."""
        from . import lstm_fused
        d = dones if dones.element_size() == 1 else (dones != 0)
        action, clipped, value, neglogp = lstm_fused.mlp_policy_step(self, obs, d.contiguous(), noise=noise, rng=rng, rollout=rollout, out=out)
        return action, clipped, value, neglogp, states

    @torch.no_grad()
    def fused_rollout(self, env_impl, steps, obs, states, dones, rng, rollout, out, env_reward, env_extra, noise_all=None, fused=False):
        """`steps` fused_step + env.step pairs issued by one C call (lstm_fused.mlp_policy_rollout): ONE persistent launch for the whole
        rollout, or 2 x steps launches -- by lstm_fused.MLP_ROLLOUT / IRRL_MLP_ROLLOUT ("persistent" by default; `fused` is the LSTM
        policy's switch and not looked at here)."""
        from . import lstm_fused
        lstm_fused.mlp_policy_rollout(self, env_impl, steps, obs, dones, rng, rollout, out, env_reward, env_extra, noise_all=noise_all,
                                      fused=None)

    @torch.no_grad()
    def step(self, obs, states=None, masks=None, deterministic=False, generator=None, noise=None):
        mean, value = self._run(obs)
        if deterministic:
            action = mean
        else:
            if noise is None:
                noise = torch.randn(mean.shape, device=mean.device, dtype=mean.dtype, generator=generator)
            action = mean + torch.exp(self.logstd) * noise
        return action, value, states, diag_gaussian_neglogp(action, mean, self.logstd)

    @torch.no_grad()
    def value(self, obs, states=None, masks=None):
        return self._run(obs)[1]

    def evaluate_raw(self, obs, states=None, masks=None):
        return self._run(obs)

    def evaluate(self, obs, states, masks, actions):
        mean, value = self._run(obs)
        return diag_gaussian_neglogp(actions, mean, self.logstd), value, diag_gaussian_entropy(self.logstd, mean)


# name kept for `from flex_gym.archi.policies import ActorCriticPolicy, LstmPolicy, MlpPolicy` (run_bp_v5.py:11)
LstmPolicy = CustomLSTMPolicy
