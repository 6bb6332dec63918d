"""PPO2 of the reference (flex_gym/algo/ppo2/ppo2.py = stable-baselines 2.8 PPO2 + raisimGym edits) on
PyTorch-ROCm, with the rollout buffer, GAE scan and policy/value update resident on the env's MI355X.

Reference semantics reproduced (file:line = ppo2.py unless noted):
  Runner.run 494-582      n_steps x [policy step (stochastic) -> clip to [-1,1] for the env only (529-531; the
                          buffer keeps the UNCLIPPED action, 523) -> env.step]; dones stored BEFORE the step (526);
                          last_values bootstrap (552); GAE (554-568); env-major flatten (572-574, 603-611);
                          global env reset after every rollout (577) WITHOUT resetting the LSTM state / dones.
  loss 136-175            ratio = exp(nlp_old - nlp); pg = mean(max(-A r, -A clip(r, 1 +- eps)));
                          vf = .5 mean(max((v - R)^2, (v_old + clip(v - v_old, +-eps) - R)^2));
                          loss = pg - ent_coef * H + vf_coef * vf; approxkl, clipfrac as logged there.
  _train_step 262-263     advantages normalised per minibatch: (A - mean) / (std + 1e-8)  (population std).
  update 190-197          tf.clip_by_global_norm(max_grad_norm) then Adam(lr, eps 1e-5, beta .9/.999).
  minibatching 364-404    non-recurrent: shuffled flat indices; recurrent: shuffled ENV indices, whole sequences,
                          LSTM state at the rollout start, done-masks inside the unroll.
  learn 325-435           nupdates = total_timesteps // (n_envs * n_steps); fps = n_batch / (rollout + update time);
                          logged keys of 419-435; checkpoint every `eval_every_n` updates when update % n == 1 (331-341).

Multi-GPU: one process per GPU, env shards of n_envs per rank, replicated parameters; per optimizer
step ONE flat all-reduce of the gradient (sum / world, clip after averaging) and one 3-float all-reduce of the
advantage moments so that normalisation equals the single-process full batch.  `torch.distributed` with backend
"nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.

The GAE scan runs in the HIP kernel `irrl_gae` (include/irrl_env.h) for CUDA tensors; CPU tensors (only the
gloo / unit tests create those) take the literal torch transcription below.
"""
import ctypes as C
import math
import os
import pickle
import time

import numpy as np
import torch

from .policies import ActorCriticPolicy, CustomLSTMPolicy, MlpPolicy  # noqa: F401


def gae_reference(rewards, values, dones, last_values, last_dones, gamma, lam):
    """Literal transcription of ppo2.py:554-568 on [T,N] tensors."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards)
    last = torch.zeros_like(last_values)
    for t in reversed(range(T)):
        if t == T - 1:
            nonterm = 1.0 - last_dones.to(rewards.dtype)
            nextv = last_values
        else:
            nonterm = 1.0 - dones[t + 1].to(rewards.dtype)
            nextv = values[t + 1]
        delta = rewards[t] + gamma * nextv * nonterm - values[t]
        last = delta + gamma * lam * nonterm * last
        adv[t] = last
    return adv, adv + values


def gae(rewards, values, dones, last_values, last_dones, gamma, lam):
    """[T,N] f32 rewards/values, [T,N] bool dones (flag before step t), [N] last_values/last_dones."""
    if rewards.is_cuda:
        from . import _lib
        lib = _lib.load()
        rewards, values = rewards.contiguous(), values.contiguous()
        d8 = dones.to(torch.uint8).contiguous()
        ld8 = last_dones.to(torch.uint8).contiguous()
        lv = last_values.contiguous()
        adv, ret = torch.empty_like(rewards), torch.empty_like(rewards)
        T, N = rewards.shape
        stream = torch.cuda.current_stream(rewards.device).cuda_stream
        _lib.check(lib.irrl_gae(T, N, C.c_void_p(rewards.data_ptr()), C.c_void_p(values.data_ptr()), C.c_void_p(d8.data_ptr()),
                                C.c_void_p(lv.data_ptr()), C.c_void_p(ld8.data_ptr()), float(gamma), float(lam),
                                C.c_void_p(adv.data_ptr()), C.c_void_p(ret.data_ptr()), C.c_void_p(stream)))
        return adv, ret
    return gae_reference(rewards, values, dones, last_values, last_dones, gamma, lam)


def ppo_loss(neglogpac, vpred, entropy, actions_unused, advs, returns, old_neglogpac, old_vpred, cliprange, ent_coef, vf_coef):
    """ppo2.py:152-175 on flat tensors; returns (loss, pg_loss, vf_loss, entropy, approxkl, clipfrac)."""
    ent = entropy.mean()
    vpredclipped = old_vpred + torch.clamp(vpred - old_vpred, -cliprange, cliprange)
    vf_loss = 0.5 * torch.maximum((vpred - returns) ** 2, (vpredclipped - returns) ** 2).mean()
    ratio = torch.exp(old_neglogpac - neglogpac)
    pg_loss = torch.maximum(-advs * ratio, -advs * torch.clamp(ratio, 1.0 - cliprange, 1.0 + cliprange)).mean()
    approxkl = 0.5 * ((neglogpac - old_neglogpac) ** 2).mean()
    clipfrac = (torch.abs(ratio - 1.0) > cliprange).to(ratio.dtype).mean()
    loss = pg_loss - ent * ent_coef + vf_loss * vf_coef
    return loss, pg_loss, vf_loss, ent, approxkl, clipfrac


class _FusedPPOLoss(torch.autograd.Function):
    """ppo_loss + DiagGaussian neglogp / entropy with forward and backward in ONE kernel launch (`irrl_ppo_loss`): the eager
    graph is ~60 elementwise / reduction kernels over [T*N] and [T*N, 12] tensors per optimizer step."""
    N_BLOCKS = 2048

    @staticmethod
    def forward(ctx, mean, vpred, logstd, actions, returns, old_values, old_neglogp, adv_stats, cliprange, ent_coef, vf_coef):
        from . import _lib
        lib = _lib.load()
        A = mean.shape[-1]
        M = vpred.numel()
        mean_c, v_c = mean.contiguous(), vpred.contiguous()
        d_mean, d_v = torch.empty_like(mean_c), torch.empty_like(v_c)
        partials = torch.empty(_FusedPPOLoss.N_BLOCKS, 4 + A, device=mean.device, dtype=torch.float32)
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(lib.irrl_ppo_loss(M, A, p(mean_c), p(logstd.contiguous()), p(v_c), p(actions.contiguous()), p(returns.contiguous()),
                                     p(old_values.contiguous()), p(old_neglogp.contiguous()), p(adv_stats), float(cliprange), float(vf_coef),
                                     p(d_mean), p(d_v), p(partials), _FusedPPOLoss.N_BLOCKS,
                                     C.c_void_p(torch.cuda.current_stream(mean.device).cuda_stream)))
        sums = partials.sum(0)
        pg, vf, kl, cf = sums[0] / M, sums[1] / M, sums[2] / M, sums[3] / M
        ent = (logstd + 0.5 * (math.log(2.0 * math.pi) + 1.0)).sum()
        loss = pg - ent * ent_coef + vf * vf_coef
        ctx.save_for_backward(d_mean, d_v, (sums[4:] - ent_coef).reshape(logstd.shape))
        stats = torch.stack([pg, vf, ent, kl, cf])
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        d_mean, d_v, d_logstd = ctx.saved_tensors
        return d_mean.mul_(g_loss), d_v.mul_(g_loss), d_logstd * g_loss, None, None, None, None, None, None, None, None


class _FusedHeadsLoss(torch.autograd.Function):
    """The policy / value heads (mean = h_pi W_pi + b_pi, v = h_v w_v + b_v), the loss of `_FusedPPOLoss` and ALL their
    gradients in one launch (`irrl_ppo_heads_loss`): replaces the heads' forward GEMMs, the [M,12] x [12,48] dx GEMM that alone
    took 2.45 ms per epoch at 4096 x 750 (it writes 590 MB), the two tall weight-gradient reductions and the loss kernel.
    backward() scales every returned gradient by the upstream gradient of `loss` (a caller may scale the loss: loss / world,
    gradient accumulation, loss scaling) -- two more passes over [M,48], 0.3-0.5 ms per epoch at 4096 x 750.  A caller that
    backpropagates the loss ITSELF, unscaled, says so per call (`unit_grad=True`, the last argument: PPO2._train_step calls
    loss.backward() on it directly) and the row gradients are returned as they are; nothing is assumed by default."""
    N_BLOCKS = 1024

    @staticmethod
    def forward(ctx, h_pi, h_v, pi_w, pi_b, vf_w, vf_b, logstd, actions, returns, old_values, old_neglogp, adv_stats, cliprange, ent_coef, vf_coef,
                unit_grad=False):
        from . import _lib
        lib = _lib.load()
        ctx.unit_grad = bool(unit_grad)
        H, A = h_pi.shape[-1], pi_w.shape[1]
        M = h_pi.numel() // H
        hp, hv = h_pi.contiguous(), h_v.contiguous()
        d_hp, d_hv = torch.empty_like(hp), torch.empty_like(hv)
        P = 4 + A + A + 1 + H + H * A
        partials = torch.empty(_FusedHeadsLoss.N_BLOCKS, P, device=hp.device, dtype=torch.float32)
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(lib.irrl_ppo_heads_loss(M, A, H, p(hp), p(hv), p(pi_w.contiguous()), p(pi_b.contiguous()), p(vf_w.contiguous()), p(vf_b.contiguous()),
                                           p(logstd.contiguous()), p(actions.contiguous()), p(returns.contiguous()), p(old_values.contiguous()),
                                           p(old_neglogp.contiguous()), p(adv_stats), float(cliprange), float(vf_coef), p(d_hp), p(d_hv), None, None,
                                           p(partials), _FusedHeadsLoss.N_BLOCKS, C.c_void_p(torch.cuda.current_stream(hp.device).cuda_stream)))
        sums = partials.sum(0)
        pg, vf, kl, cf = sums[0] / M, sums[1] / M, sums[2] / M, sums[3] / M
        ent = (logstd + 0.5 * (math.log(2.0 * math.pi) + 1.0)).sum()
        loss = pg - ent * ent_coef + vf * vf_coef
        o = 4
        d_logstd = (sums[o:o + A] - ent_coef).reshape(logstd.shape); o += A
        d_bpi = sums[o:o + A]; o += A
        d_bv = sums[o:o + 1]; o += 1
        d_wv = sums[o:o + H].reshape(vf_w.shape); o += H
        d_wpi = sums[o:o + H * A].reshape(H, A)
        ctx.save_for_backward(d_hp.view_as(h_pi), d_hv.view_as(h_v), d_wpi, d_bpi, d_wv, d_bv, d_logstd)
        stats = torch.stack([pg, vf, ent, kl, cf])
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        d_hp, d_hv, d_wpi, d_bpi, d_wv, d_bv, d_logstd = ctx.saved_tensors
        if not ctx.unit_grad:
            d_hp, d_hv = d_hp * g_loss, d_hv * g_loss
            d_wpi, d_bpi, d_wv, d_bv, d_logstd = d_wpi * g_loss, d_bpi * g_loss, d_wv * g_loss, d_bv * g_loss, d_logstd * g_loss
        return d_hp, d_hv, d_wpi, d_bpi, d_wv, d_bv, d_logstd, None, None, None, None, None, None, None, None, None


# arithmetic of the critic's sequence kernels when the rollout's values are computed after the rollout (Runner._critic_pass): the f32 level
# ("bf16x6": three bf16 planes per operand, ~2^-24 per product; "f32": the exact-f32 MFMA kernels), whatever the update itself uses
CRITIC_PASS_PRECISION = os.environ.get("IRRL_CRITIC_PASS_PRECISION", "bf16x6")
if CRITIC_PASS_PRECISION not in ("bf16x6", "f32"):
    raise ValueError("IRRL_CRITIC_PASS_PRECISION is 'bf16x6' or 'f32' (the f32 level), not %r" % (CRITIC_PASS_PRECISION,))

# arithmetic of the MlpPolicy gradient kernels: "bf16x3" = every product as three bf16 plane products on the matrix cores (two planes per
# operand, f32 accumulation, ~2^-16 relative per product; csrc/mlp_bf16.hpp), "f32" = v_mfma_f32_16x16x4_f32 (csrc/mlp_update.hpp)
MLP_PRECISION = os.environ.get("IRRL_MLP_PRECISION", "bf16x3")
# the update's samples as packed 256-byte records (csrc/mlp_update.hpp IRRL_MLP_REC; bf16x3 kernels): OFF by default (the five arrays are read as they
# lie); IRRL_MLP_RECORDS=1 enables the records -- bit-identical gradients, same time
MLP_RECORDS = os.environ.get("IRRL_MLP_RECORDS", "0") != "0"


def _mlp_grads_entry(lib):
    if MLP_PRECISION == "bf16x3":
        return lib.irrl_mlp_ppo_grads_bf16
    if MLP_PRECISION == "f32":
        return lib.irrl_mlp_ppo_grads
    raise ValueError("IRRL_MLP_PRECISION / ppo2.MLP_PRECISION is 'bf16x3' or 'f32', not %r" % (MLP_PRECISION,))


def mlp_ppo_grads_supported(policy, obs):
    """The single-launch-per-network gradient kernels (`irrl_mlp_ppo_grads`) cover the reference's MlpPolicy as configured:
    35 observations, [64, 64] tanh stacks, 12 actions."""
    fc = getattr(policy, "pi_fc", None)
    return bool(obs.is_cuda and fc is not None and len(fc) == 2 and tuple(fc[0].w.shape) == (35, 64) and tuple(fc[1].w.shape) == (64, 64)
                and tuple(policy.pi.w.shape) == (64, 12) and tuple(policy.vf.w.shape) == (64, 1) and obs.dtype == torch.float32)


def mlp_ppo_grads(policy, obs, actions, returns, old_values, old_neglogp, adv_stats, cliprange, ent_coef, vf_coef, index=None, n_blocks=256,
                  want_loss=True):
    """loss = pg - ent_coef * entropy + vf_coef * vf of MlpPolicy on one minibatch, and its gradient with respect to every parameter,
    in two launches (policy network, value network; csrc/mlp_update.hpp) -- the whole of ppo2.py:243-298's graph evaluation.
    obs / actions / returns / old_values / old_neglogp are the FLAT rollout arrays; index (int64 device vector) picks the
    minibatch's rows in place (None: all rows).  -> (loss, stats[pg, vf, entropy, approxkl, clipfrac], {parameter: gradient})."""
    from . import _lib
    lib = _lib.load()
    dev = obs.device
    n = int(index.numel()) if index is not None else int(returns.numel())
    P = lib.irrl_mlp_ppo_partial_len()
    partials = torch.empty(2, n_blocks, P, device=dev, dtype=torch.float32)
    p = lambda t: C.c_void_p(t.data_ptr())
    c = lambda t: t if t.is_contiguous() else t.contiguous()
    obs, actions, returns, old_values, old_neglogp = c(obs), c(actions), c(returns), c(old_values), c(old_neglogp)
    ip = p(index) if index is not None else None
    if index is not None:
        assert index.dtype == torch.int64 and index.is_contiguous()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    nets = ((0, policy.pi_fc, policy.pi), (1, policy.vf_fc, policy.vf))
    for kind, fc, head in nets:
        _lib.check(_mlp_grads_entry(lib)(kind, n, ip, obs.shape[-1], fc[0].w.shape[1], actions.shape[-1], p(obs), p(actions), p(returns),
                                          p(old_values), p(old_neglogp), p(c(fc[0].w)), p(c(fc[0].b)), p(c(fc[1].w)), p(c(fc[1].b)),
                                          p(c(head.w)), p(c(head.b)), p(c(policy.logstd)), p(adv_stats), float(cliprange), float(vf_coef),
                                          p(partials[kind]), n_blocks, stream))
    sums = torch.empty(2, P, device=dev, dtype=torch.float32)   # workgroups added in one fixed order
    for kind in (0, 1):
        if lib.irrl_sum_rows(p(partials[kind]), n_blocks, P, 0, p(sums[kind]), stream) != 0:
            raise RuntimeError("irrl_sum_rows failed")
    A = actions.shape[-1]
    sc = sums[:, :4] / n                                        # per-sample means: [pg, kl, clipfrac, -] and [vf, -, -, -]
    pg, kl, cf, vf = sc[0, 0], sc[0, 1], sc[0, 2], sc[1, 0]
    ent = policy.logstd.detach().sum() + 0.5 * (math.log(2.0 * math.pi) + 1.0) * A
    loss = (pg - ent * ent_coef + vf * vf_coef) if want_loss else None
    grads = {policy.logstd: (sums[0, 4:4 + A] - ent_coef).reshape(policy.logstd.shape)}
    o1, o2, o3, w1, w2, w3 = 20, 84, 148, 164, 164 + 48 * 64, 164 + 48 * 64 + 64 * 64
    for kind, fc, head in nets:
        r = sums[kind]
        out = head.w.shape[1]
        grads[fc[0].b], grads[fc[1].b], grads[head.b] = r[o1:o1 + 64], r[o2:o2 + 64], r[o3:o3 + out]
        grads[fc[0].w] = r[w1:w1 + 48 * 64].view(48, 64)[:35]
        grads[fc[1].w] = r[w2:w2 + 64 * 64].view(64, 64)
        grads[head.w] = r[w3:w3 + 64 * 16].view(64, 16)[:, :out]
    return loss, torch.stack([pg, vf, ent, kl, cf]), grads


def clip_by_global_norm_(params, max_norm):
    """tf.clip_by_global_norm (ppo2.py:192) on the .grad tensors, in place: g *= max_norm / max(|g|_global, max_norm).  -> |g|_global"""
    grads = [p.grad for p in params if p.grad is not None]
    norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = max_norm / torch.clamp(norm, min=max_norm)
    torch._foreach_mul_(grads, coef)
    return norm


class TFAdam(torch.optim.Optimizer):
    """tf.train.AdamOptimizer(learning_rate, epsilon) as the reference builds it (ppo2.py:195), TensorFlow 1 semantics:
        lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  theta -= lr_t * m / (sqrt(v) + eps)
    i.e. epsilon sits OUTSIDE the bias correction ("epsilon hat" of Kingma & Ba) -- an effective epsilon of eps / sqrt(1 - beta2^t),
    3e-4 at t = 1 for eps 1e-5, where torch.optim.Adam (sqrt(v) / sqrt(1 - beta2^t) + eps) uses 1e-5: the early updates of parameters with
    small gradients differ.  The kernel `irrl_clip_adam` (csrc/ppo_optim.hpp) implements the same formula on the flat buffers."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-5):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            b1, b2 = group["betas"]
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"], st["m"], st["v"] = 0, torch.zeros_like(p), torch.zeros_like(p)
                st["step"] += 1
            t = self.state[ps[0]]["step"]
            lr_t = group["lr"] * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
            gs, ms, vs = [p.grad for p in ps], [self.state[p]["m"] for p in ps], [self.state[p]["v"] for p in ps]
            torch._foreach_lerp_(ms, gs, 1.0 - b1)
            torch._foreach_mul_(vs, b2)
            torch._foreach_addcmul_(vs, gs, gs, value=1.0 - b2)
            den = torch._foreach_sqrt(vs)
            torch._foreach_add_(den, group["eps"])
            torch._foreach_addcdiv_(ps, ms, den, value=-lr_t)


class FlatParams(object):
    """Every parameter of a policy as a VIEW of one persistent flat buffer, and the gradient and the Adam moments likewise.
    The data-parallel exchange of an optimizer step (one all-reduce of the flat gradient) is then ONE collective on
    `grad[:n]` with nothing packed or unpacked around it, and on the GPU clip_by_global_norm + Adam (ppo2.py:182-197) is one launch
    over the four buffers (`irrl_clip_adam`, csrc/ppo_optim.hpp).  Parameters start at multiples of 64 floats (the kernels that take
    a parameter pointer may load 16-byte vectors); the padding carries zero gradients and never moves.  `grad` has a tail of TAIL
    floats behind the parameters' slots for per-step statistics that the gradient kernels write beside the gradients."""
    ALIGN = 64
    TAIL = 64

    def __init__(self, policy):
        self.params = list(policy.parameters())
        dev = self.params[0].device
        self.offsets = []
        o = 0
        for p in self.params:
            self.offsets.append(o)
            o += -(-p.numel() // self.ALIGN) * self.ALIGN
        self.n = o
        self.theta = torch.zeros(o, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(o + self.TAIL, device=dev, dtype=torch.float32)
        self.m = torch.zeros(o, device=dev, dtype=torch.float32)
        self.v = torch.zeros(o, device=dev, dtype=torch.float32)
        self.step = 0
        self.grad_views = []
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.theta[off:off + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
                self.grad_views.append(self.grad[off:off + p.numel()].view_as(p))
        self._dirty = [False] * len(self.params)      # slot holds a gradient some earlier step wrote
        self.offset_of = {id(p): off for p, off in zip(self.params, self.offsets)}

    def gather(self):
        """autograd left the gradients in separate tensors (p.grad): ONE multi-tensor copy into the flat buffer; slots of parameters
        without a gradient are (kept) zero"""
        src, dst, stale = [], [], []
        for i, (p, gv) in enumerate(zip(self.params, self.grad_views)):
            if p.grad is not None:
                if p.grad.data_ptr() != gv.data_ptr():
                    src.append(p.grad)
                    dst.append(gv)
                self._dirty[i] = True
            elif self._dirty[i]:
                stale.append(gv)
                self._dirty[i] = False
        if src:
            torch._foreach_copy_(dst, src)
        if stale:
            torch._foreach_zero_(stale)

    def point_grads_at_views(self):
        for p, gv in zip(self.params, self.grad_views):
            p.grad = gv


_MLP_MAPS = {}


def _mlp_scatter_map(policy, flat, ent_coef):
    """column of a partial-sum row of the MlpPolicy gradient kernels (csrc/mlp_update.hpp: scalars[4] | d logstd[16] | d b1[64] | d b2[64] |
    d b3[16] | d W1[48][64] | d W2[64][64] | d W3[64][16]) -> slot of the flat gradient buffer; the four scalars of network k go to the
    tail slots n + 4 k ..; `add` carries -ent_coef for d logstd.  Built once per (policy, ent_coef)."""
    from . import _lib
    key = (id(flat), float(ent_coef))
    hit = _MLP_MAPS.get(key)
    if hit is not None:
        return hit
    P = _lib.load().irrl_mlp_ppo_partial_len()
    A = policy.act_dim
    mp = np.full((2, P), -1, np.int32)
    add = np.zeros((2, P), np.float32)
    o1, o2, o3, w1, w2, w3 = 20, 84, 148, 164, 164 + 48 * 64, 164 + 48 * 64 + 64 * 64
    off = flat.offset_of
    for kind, fc, head in ((0, policy.pi_fc, policy.pi), (1, policy.vf_fc, policy.vf)):
        out = head.w.shape[1]
        mp[kind, 0:4] = flat.n + 4 * kind + np.arange(4)
        if kind == 0:
            mp[0, 4:4 + A] = off[id(policy.logstd)] + np.arange(A)
            add[0, 4:4 + A] = -float(ent_coef)
        mp[kind, o1:o1 + 64] = off[id(fc[0].b)] + np.arange(64)
        mp[kind, o2:o2 + 64] = off[id(fc[1].b)] + np.arange(64)
        mp[kind, o3:o3 + out] = off[id(head.b)] + np.arange(out)
        k35 = np.arange(35)[:, None] * 64 + np.arange(64)[None, :]
        mp[kind, w1:w1 + 35 * 64] = (off[id(fc[0].w)] + k35).reshape(-1)
        mp[kind, w2:w2 + 64 * 64] = off[id(fc[1].w)] + np.arange(64 * 64)
        cols = (w3 + np.arange(64)[:, None] * 16 + np.arange(out)[None, :]).reshape(-1)
        mp[kind, cols] = (off[id(head.w)] + np.arange(64)[:, None] * out + np.arange(out)[None, :]).reshape(-1)
    dev = flat.grad.device
    hit = (torch.from_numpy(mp).to(dev), torch.from_numpy(add).to(dev), P, {})
    _MLP_MAPS.clear()          # one live learner per process is the rule; do not keep dead ones' buffers
    _MLP_MAPS[key] = hit
    return hit


def mlp_pack_records(obs, actions, returns, old_values, old_neglogp):
    """The five per-sample arrays of the flat rollout as ONE 256-byte record per sample (csrc/mlp_update.hpp IRRL_MLP_REC; built once per
    update): a shuffled minibatch row then costs two 128-byte lines instead of seven or eight (`irrl_mlp_ppo_grads_bf16_rec`)."""
    from . import _lib
    lib = _lib.load()
    n = int(returns.numel())
    rec = torch.empty(n, lib.irrl_mlp_record_floats(), device=obs.device, dtype=torch.float32)
    assert rec.data_ptr() % 256 == 0
    p = lambda t: C.c_void_p(t.data_ptr())
    c = lambda t: t if t.is_contiguous() else t.contiguous()
    _lib.check(lib.irrl_mlp_pack_records(n, p(c(obs)), p(c(actions)), p(c(returns)), p(c(old_values)), p(c(old_neglogp)), p(rec),
                                         C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)))
    return rec


def mlp_ppo_grads_flat(policy, flat, obs, actions, returns, old_values, old_neglogp, adv_stats, cliprange, ent_coef, vf_coef, index, n_blocks=256, rec=None):
    """`mlp_ppo_grads` with the gradients summed over the workgroups STRAIGHT INTO the flat gradient buffer (one
    `irrl_sum_rows_scatter` launch for both networks instead of two row sums + one copy per parameter).  -> the step's raw
    statistics row [8 sums | logstd 12] (one small launch; `mlp_stats_rows` turns the rows of an update into the logged means).
    rec: the samples as packed records (`mlp_pack_records`; bf16x3 kernels only) -- same values, bit-identical gradients."""
    from . import _lib
    lib = _lib.load()
    dev = obs.device
    n = int(index.numel()) if index is not None else int(returns.numel())
    mp, add, P, cache = _mlp_scatter_map(policy, flat, ent_coef)
    partials = cache.get(n_blocks)
    if partials is None:
        partials = cache[n_blocks] = torch.empty(2, n_blocks, P, device=dev, dtype=torch.float32)
    p = lambda t: C.c_void_p(t.data_ptr())
    c = lambda t: t if t.is_contiguous() else t.contiguous()
    obs, actions, returns, old_values, old_neglogp = c(obs), c(actions), c(returns), c(old_values), c(old_neglogp)
    ip = p(index) if index is not None else None
    if index is not None:
        assert index.dtype == torch.int64 and index.is_contiguous()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for kind, fc, head in ((0, policy.pi_fc, policy.pi), (1, policy.vf_fc, policy.vf)):
        if rec is not None and MLP_PRECISION == "bf16x3":
            _lib.check(lib.irrl_mlp_ppo_grads_bf16_rec(kind, n, ip, p(rec), p(fc[0].w), p(fc[0].b), p(fc[1].w), p(fc[1].b), p(head.w), p(head.b),
                                                       p(policy.logstd), p(adv_stats), float(cliprange), float(vf_coef), p(partials[kind]), n_blocks, stream))
            continue
        _lib.check(_mlp_grads_entry(lib)(kind, n, ip, obs.shape[-1], fc[0].w.shape[1], actions.shape[-1], p(obs), p(actions), p(returns),
                                          p(old_values), p(old_neglogp), p(fc[0].w), p(fc[0].b), p(fc[1].w), p(fc[1].b),
                                          p(head.w), p(head.b), p(policy.logstd), p(adv_stats), float(cliprange), float(vf_coef),
                                          p(partials[kind]), n_blocks, stream))
    _lib.check(lib.irrl_sum_rows_scatter(p(partials), 2, n_blocks, P, p(mp), p(add), p(flat.grad), stream))
    for i in range(len(flat._dirty)):
        flat._dirty[i] = True
    return torch.cat([flat.grad[flat.n:flat.n + 8], policy.logstd.detach().reshape(-1)])


def mlp_stats_rows(rows, n, act_dim):
    """[steps, 8 + A] raw rows of `mlp_ppo_grads_flat` -> [steps, 5] (pg, vf, entropy, approxkl, clipfrac)"""
    r = torch.stack(rows)
    ent = r[:, 8:8 + act_dim].sum(1) + 0.5 * (math.log(2.0 * math.pi) + 1.0) * act_dim
    return torch.stack([r[:, 0] / n, r[:, 4] / n, ent, r[:, 1] / n, r[:, 2] / n], 1)


def _perm_hash(x):
    x = x ^ (x >> np.uint32(16)); x = x * np.uint32(0x7feb352d); x = x ^ (x >> np.uint32(15)); x = x * np.uint32(0x846ca68b); x = x ^ (x >> np.uint32(16))
    return x


def feistel_permutation(n, seed, counter):
    """numpy twin of `irrl_random_permutation` (csrc/ppo_optim.hpp): the keyed bijection E of [0, n) -- 4-round Feistel network on the bits of
    n - 1, cycle walking -- as an int64 vector out[i] = E(i).  Same bits as the kernel."""
    bits = 1
    while (1 << bits) < n:
        bits += 1
    hb = max(1, (bits + 1) // 2)
    mask = np.uint32((1 << hb) - 1)
    keys = []
    with np.errstate(over="ignore"):
        for r in range(4):
            a = _perm_hash(np.uint32((seed + 0x9E3779B9 * (r + 1)) & 0xFFFFFFFF))
            b = _perm_hash(np.uint32((counter + 0x85EBCA6B * (r + 1)) & 0xFFFFFFFF))
            keys.append(np.uint32(a ^ b))

        def enc(v):
            L, R = v >> np.uint32(hb), v & mask
            for k in keys:
                F = _perm_hash(R ^ k) & mask
                L, R = R, L ^ F
            return (L << np.uint32(hb)) | R

        v = enc(np.arange(n, dtype=np.uint32))
        while True:
            out_of_range = v >= np.uint32(n)
            if not out_of_range.any():
                break
            v[out_of_range] = enc(v[out_of_range])
    return v.astype(np.int64)


def fused_ppo_loss_supported(policy, obs):
    return bool(obs.is_cuda and hasattr(policy, "evaluate_raw") and getattr(policy, "act_dim", 0) == 12)


class Runner(object):
    """ppo2.py:479-582 with every buffer a device tensor of shape [T, N, ...]."""

    def __init__(self, env, model, n_steps, gamma, lam, use_graph=None):
        self.env, self.model, self.n_steps, self.gamma, self.lam = env, model, n_steps, gamma, lam
        dev = model.device
        n = env.num_envs
        self.obs = env.reset().clone()                               # AbstractEnvRunner.__init__: obs = env.reset()
        self.states = model.policy.initial_state(n, dev)             # zeros [N, 384]
        self.dones = torch.zeros(n, dtype=torch.bool, device=dev)    # [False] * n_envs
        T = n_steps
        self.mb_obs = torch.zeros(T, n, env.num_obs, device=dev)
        self.mb_actions = torch.zeros(T, n, env.num_acts, device=dev)
        self.mb_values = torch.zeros(T, n, device=dev)
        self.mb_neglogpacs = torch.zeros(T, n, device=dev)
        self.mb_dones = torch.zeros(T, n, dtype=torch.bool, device=dev)
        self.mb_rewards = torch.zeros(T, n, device=dev)
        self.t_idx = torch.zeros(1, dtype=torch.long, device=dev)     # device row counter of the generic (per-step graph) path
        self.use_graph = (dev.type == "cuda") if use_graph is None else bool(use_graph)
        self._graph = None
        self._graph_epoch = 0
        # sampling noise: the model's seeded generator; under graph capture it is registered with the graph
        self._gen = model.generator
        # single-launch policy step + raw env step when the policy / env pair supports it (LSTM policy on the GPU)
        self._fused = bool(hasattr(model.policy, "fused_step_supported") and hasattr(env, "step_into")
                           and model.policy.fused_step_supported(self.obs))
        self.rew = torch.zeros(n, device=dev)
        # fused path: "direct" = the 2 x T launches go out back to back from one C call (irrl_lstm_rollout; policies that have
        # `fused_rollout`), "graph" = one hipGraph of 2 x T kernel nodes.  Direct launches need no capture, no warm-up and no
        # re-capture after a setter, start sooner and run ~0.5 us per kernel shorter than graph nodes (ROCm 7.2, MI355X)
        self.rollout_launch = "direct" if (self._fused and hasattr(model.policy, "fused_rollout") and hasattr(getattr(env, "wrapper", None), "_h")
                                           and hasattr(env, "extra")) else "graph"
        # 0 = two launches per step, 1 = env.step k + policy step k + 1 in one kernel (an experiment, slower), 2 (default until round 5) = the whole rollout as
        # ONE persistent launch (a workgroup loops over all steps for its 16 robots: nothing waits for the slowest wave of a step; falls back
        # to 0 where the kernel is not instantiated).  Same bits.  (MlpPolicy: lstm_fused.MLP_ROLLOUT, persistent by default as well.)
        # 3 (round 5, the default) = 2 with the CRITIC OFF THE PER-STEP PATH: V(s_t) depends on the observation history only and nothing in the rollout
        # needs it before GAE (ppo2.py:519-568), so the persistent launch runs the actor stack alone -- all of its operands resident in LDS -- and
        # `_critic_pass` evaluates the critic stack over the recorded observations afterwards with the sequence kernels (two launches for the
        # whole rollout).  Actor-side buffers (obs, actions, neglogp, rewards, dones, actor states) bit-identical to the other modes; values at
        # the f32 level of the sequence kernels (CRITIC_PASS_PRECISION), i.e. within ~1e-6 of the step-by-step ones.  Falls back to 2 where
        # the kernel is not instantiated.
        self.rollout_one_launch_per_step = int(os.environ.get("IRRL_ROLLOUT_FUSED", "3"))
        self._raw_env = hasattr(env, "step_into") and hasattr(env, "account_rollout") and dev.type == "cuda"
        # sampling noise: "kernel" = the engine's counter RNG inside the fused policy kernel (fused path only; the generic
        # path draws from the model's generator per step), "torch" = standard normals for the whole rollout drawn up front
        # from the model's generator and used by either path (tests compare the two paths with it)
        self.noise_source = "kernel"
        self.noise_all = None
        self.rng_base = torch.zeros(1, dtype=torch.long, device=dev)   # policy steps of all earlier rollouts (RNG counter)
        self._out = (torch.empty(n, env.num_acts, device=dev), torch.empty(n, env.num_acts, device=dev), torch.empty(n, device=dev),
                     torch.empty(n, device=dev))

    def _draw_noise(self, shape, env_axis):
        """standard normals for this rank's envs out of the draw for ALL ranks' envs (identically seeded generators): rank r takes
        rows r * n .. of the env axis, so the union over ranks is the single-process draw"""
        world, rank = self.model.world, self.model.rank
        if world == 1:
            return torch.randn(shape, device=self.obs.device, dtype=self.obs.dtype, generator=self._gen)
        full = list(shape)
        n = full[env_axis]
        full[env_axis] = n * world
        # every rank materialises ALL ranks' draw and keeps its slice (what makes the union the single-process draw): memory and time
        # grow with the world size.  Fine for the tests that compare the two noise sources; a multi-GPU TRAINING run uses the
        # kernel's counter RNG (noise_source "kernel", the default), which addresses the noise by global env id and draws nothing else.
        if int(np.prod(full)) * 4 > (1 << 29):
            raise RuntimeError("noise_source='torch' would draw %.1f GB per rollout on every rank at world size %d: use the default "
                               "noise_source='kernel' for multi-GPU runs" % (np.prod(full) * 4 / 1e9, world))
        return torch.randn(full, device=self.obs.device, dtype=self.obs.dtype, generator=self._gen).narrow(env_axis, rank * n, n)

    @torch.no_grad()
    def _critic_pass(self, states0):
        """The values of a rollout whose per-step part ran the actor alone (`rollout_one_launch_per_step` = 3): the critic stack over the
        recorded observations [T, N, 35] from its state in front of the rollout, masks = the recorded dones (the state is cleared where an
        episode ended before step t, run_bp_v5.py:143-176) -- the sequence kernels of the update, at CRITIC_PASS_PRECISION -- then the
        value head; writes mb_values and the critic's half of the carried LSTM state."""
        pol = self.model.policy
        k = len(pol.n_lstm)
        parts = pol._split(states0)
        latent_v, new_v = pol._stack(pol.lstm_v, self.mb_obs, parts[k:], self.mb_dones.to(self.mb_obs.dtype), precision=CRITIC_PASS_PRECISION)
        self.mb_values.copy_(pol.vf(latent_v).squeeze(-1))
        off = sum(2 * h for h in pol.n_lstm)
        self.states[:, off:].copy_(torch.cat(new_v, 1))

    def _actor_only_supported(self):
        """irrl_lstm_rollout_supports(pool, hid, fuse = 3), asked once per runner: the persistent actor-only kernel exists for 16 lanes per
        robot, 48 hidden units, Crutial off and the published contact rule; MlpPolicy has no critic stack to take off the path."""
        if not hasattr(self, "_actor_only_ok"):
            pol = self.model.policy
            ok = False
            if hasattr(pol, "lstm_v") and hasattr(getattr(self.env, "wrapper", None), "_h"):
                from . import _lib
                rc = _lib.load().irrl_lstm_rollout_supports(self.env.wrapper._h, int(pol.n_lstm[0]), 3)
                if rc < 0:
                    _lib.check(1)
                ok = rc == 1
            self._actor_only_ok = ok
        return self._actor_only_ok

    def _fused_step(self, t):
        """Rollout step t as two launches: the whole policy step (sample, clip, buffer rows incl. the previous reward) and
        the env step, which writes obs / reward / dones straight into the runner's tensors."""
        noise = self.noise_all[t] if self.noise_all is not None else None
        _, clipped, _, _, _ = self.model.policy.fused_step(
            self.obs, self.states, self.dones, noise=noise, rng=(self.model.noise_seed, t, self.rng_base, self.model.env_id_offset), states_out=self.states, out=self._out,
            rollout=dict(row=t, mb_obs=self.mb_obs, mb_actions=self.mb_actions, mb_values=self.mb_values, mb_neglogpacs=self.mb_neglogpacs,
                         mb_dones=self.mb_dones, mb_rewards=self.mb_rewards, prev_reward=self.rew))
        self.env.step_into(clipped, self.obs, self.rew, self.dones)

    def _one_step(self):
        """One rollout step of the generic path with every index on the device, so the same sequence of kernels can be
        replayed from a hipGraph: policy step -> buffer rows [t] -> clip -> env.step -> obs/dones update -> t += 1."""
        pol = self.model.policy
        if self.noise_all is not None:
            actions, values, states, neglogpacs = pol.step(self.obs, self.states, self.dones, noise=self.noise_all.index_select(0, self.t_idx)[0])
        else:
            actions, values, states, neglogpacs = pol.step(self.obs, self.states, self.dones, noise=self._draw_noise((self.obs.shape[0], self.env.num_acts), 0))
        self.mb_obs.index_copy_(0, self.t_idx, self.obs.unsqueeze(0))
        self.mb_actions.index_copy_(0, self.t_idx, actions.unsqueeze(0))
        self.mb_values.index_copy_(0, self.t_idx, values.unsqueeze(0))
        self.mb_neglogpacs.index_copy_(0, self.t_idx, neglogpacs.unsqueeze(0))
        self.mb_dones.index_copy_(0, self.t_idx, self.dones.unsqueeze(0))
        if states is not None:
            self.states.copy_(states)
        clipped = torch.clamp(actions, -1.0, 1.0)
        if self._raw_env:
            # the env kernel writes straight into the runner's tensors; episode statistics are accounted per rollout
            self.env.step_into(clipped, self.obs, self.rew, self.dones)
            self.mb_rewards.index_copy_(0, self.t_idx, self.rew.unsqueeze(0))
        else:
            obs, rewards, dones = self.env.step(clipped)
            self.mb_rewards.index_copy_(0, self.t_idx, rewards.unsqueeze(0))
            self.obs.copy_(obs)
            self.dones.copy_(dones)
        self.t_idx += 1

    def _maybe_capture(self):
        """Capture `_one_step` into a hipGraph (torch.cuda.CUDAGraph) after a short warm-up on a side stream.  The env
        kernel is launched through the C-ABI on torch's current stream, so it is recorded like any torch op."""
        raw = getattr(self.env, "wrapper", None)
        epoch = getattr(raw, "params_epoch", 0)
        if self._graph is not None and epoch != self._graph_epoch:
            # a setter changed a by-value kernel argument (seed, time steps, reference table) after the capture: the recorded
            # launches still carry the old values -> drop the graph and record the rollout again
            self._graph = None
        if self._graph is not None or not self.use_graph:
            return
        dev = self.model.device
        self._graph_epoch = epoch
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            # the warm-up steps are real env steps (the rollout simply continues from there) but must not leak into the
            # episode statistics of the first rollout
            stat_names = ("ep_ret", "ep_len", "finished_ret_sum", "finished_len_sum", "finished_count")
            stats = [(getattr(self.env, k), getattr(self.env, k).clone()) for k in stat_names if hasattr(self.env, k)]
            with torch.cuda.stream(side):
                # the warm-up steps must leave no trace: the first rollout starts from env.reset() like the reference Runner and
                # like the eager (use_graph=False) path.  The env pool is snapshotted on the device and put back, so are the
                # runner's own tensors; the buffer rows the warm-up wrote are overwritten by the rollout.
                can_restore = hasattr(raw, "snapshot")
                if can_restore:
                    raw.snapshot()
                keep = [(t, t.clone()) for t in (self.obs, self.states, self.dones, self.rew)]
                for i in range(3):
                    self._fused_step(min(i, self.n_steps - 1)) if self._fused else self._one_step()
                if can_restore:
                    raw.restore()
                    for live, saved in keep:
                        live.copy_(saved)
                for live, saved in stats:
                    live.copy_(saved)
            torch.cuda.current_stream(dev).wait_stream(side)
            self.t_idx.zero_()
            g = torch.cuda.CUDAGraph()
            if self._fused:
                # the WHOLE rollout as one graph: 2 kernel nodes per step, the row index a launch argument
                with torch.cuda.graph(g):
                    for t in range(self.n_steps):
                        self._fused_step(t)
            else:
                if self._gen is not None and hasattr(g, "register_generator_state"):
                    g.register_generator_state(self._gen)
                with torch.cuda.graph(g):
                    self._one_step()
            self._graph = g
            self.t_idx.zero_()
        except Exception as exc:  # capture is an optimisation only; the eager loop below is the same code
            print("[PPO2] hipGraph capture of the rollout step unavailable (%s); running eagerly" % (str(exc).splitlines()[0],))
            self.use_graph = False
            self._graph = None
            self.t_idx.zero_()

    def run(self):
        pol = self.model.policy
        if hasattr(pol, "prepare"):
            pol.prepare()
        if self.noise_source == "torch":
            shape = (self.n_steps,) + tuple(self.mb_actions.shape[1:])
            if self.noise_all is None:
                self.noise_all = torch.empty(shape, device=self.obs.device, dtype=self.obs.dtype)   # fixed address: graphs read it
            self.noise_all.copy_(self._draw_noise(shape, 1))
        direct = self._fused and self.rollout_launch == "direct"
        if self.use_graph and not direct:
            self._maybe_capture()
        mb_states = self.states.clone()
        self.t_idx.zero_()
        if self._fused:
            if direct:
                d = self.dones if self.dones.element_size() == 1 else None
                assert d is not None
                args = (self.env.wrapper, self.n_steps, self.obs, self.states, d, (self.model.noise_seed, 0, self.rng_base, self.model.env_id_offset),
                        dict(row=0, mb_obs=self.mb_obs, mb_actions=self.mb_actions, mb_values=self.mb_values,
                             mb_neglogpacs=self.mb_neglogpacs, mb_dones=self.mb_dones, mb_rewards=self.mb_rewards),
                        self._out, self.rew, self.env.extra)
                mode = self.rollout_one_launch_per_step
                if mode == 3 and not self._actor_only_supported():
                    mode = 2      # the actor-only kernel is not instantiated for this pool / network (asked through the C-ABI, nothing was launched)
                pol.fused_rollout(*args, noise_all=self.noise_all, fused=mode)
                if mode == 3:
                    self._critic_pass(mb_states)
            elif self._graph is not None:
                self._graph.replay()
            else:
                for t in range(self.n_steps):
                    self._fused_step(t)
            self.rng_base += self.n_steps
            self.mb_rewards[self.n_steps - 1].copy_(self.rew)       # rows 0 .. T-2 were written by the following policy step
        else:
            for _ in range(self.n_steps):
                if self._graph is not None:
                    self._graph.replay()
                else:
                    self._one_step()
        last_values = pol.value(self.obs, self.states, self.dones)
        advs, returns = gae(self.mb_rewards, self.mb_values, self.mb_dones, last_values, self.dones, self.gamma, self.lam)
        # resetting environments (ppo2.py:577); LSTM states and dones deliberately survive
        if self._fused or self._raw_env:
            n_done = self.mb_dones[1:].sum() + self.dones.sum()
            self.env.account_rollout(self.mb_rewards.sum(), float(self.n_steps * self.env.num_envs), n_done + float(self.env.num_envs))
            self.obs.copy_(self.env.reset())
        else:
            self.obs.copy_(self.env.reset_and_update_info())
        return dict(obs=self.mb_obs, returns=returns, masks=self.mb_dones, actions=self.mb_actions, values=self.mb_values,
                    neglogpacs=self.mb_neglogpacs, states=mb_states if pol.recurrent else None, true_reward=self.mb_rewards)


class PPO2(object):
    """PPO2(policy, env, gamma, n_steps, ent_coef, learning_rate, vf_coef, max_grad_norm, lam, nminibatches,
    noptepochs, cliprange, verbose, tensorboard_log, policy_kwargs)   (ppo2.py:45-48, run_bp_v5.py:227-242)."""

    def __init__(self, policy, env, gamma=0.99, n_steps=128, ent_coef=0.01, learning_rate=2.5e-4, vf_coef=0.5,
                 max_grad_norm=0.5, lam=0.95, nminibatches=4, noptepochs=4, cliprange=0.2, verbose=0,
                 tensorboard_log=None, _init_setup_model=True, policy_kwargs=None, full_tensorboard_log=False, seed=None,
                 device=None):
        self.env = env
        self.gamma, self.n_steps, self.ent_coef, self.learning_rate = gamma, n_steps, ent_coef, learning_rate
        self.vf_coef, self.max_grad_norm, self.lam = vf_coef, max_grad_norm, lam
        self.nminibatches, self.noptepochs, self.cliprange, self.verbose = nminibatches, noptepochs, cliprange, verbose
        self.tensorboard_log = tensorboard_log
        self.policy_kwargs = dict(policy_kwargs or {})
        self.policy_class = policy
        self.n_envs = env.num_envs if env is not None else None
        self.device = torch.device(device) if device is not None else (env.device if env is not None and hasattr(env, "device") else torch.device("cpu"))
        self.num_timesteps = 0
        self.world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
        self.rank = torch.distributed.get_rank() if self.world > 1 else 0
        # a job started under a launcher runs its collectives whatever its size: with ONE rank the all-reduces are identities, and the
        # RCCL path (backend "nccl": device tensors, the library loaded, a communicator created) is exercised on a single-GPU box
        self.collective = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.seed = 0 if seed is None else int(seed)
        # identical initial weights and identical generators on every rank; what differs per rank is WHICH robots it owns: rank r
        # holds the global env ids r * n_envs .. (r + 1) * n_envs - 1 of the one big pool (env RNG: the pool's EnvIdOffset; sampling
        # noise: stream = global env id), so an N-rank job collects exactly the samples the single-process job on N * n_envs envs does
        torch.manual_seed(self.seed)
        if isinstance(policy, type):
            self.policy = policy(**self.policy_kwargs)
        else:
            self.policy = policy
        self.policy.to(self.device)
        self.generator = torch.Generator(device=self.device)
        self.generator.manual_seed(self.seed * 1000003 + 1)
        self.noise_seed = (self.seed * 1000003 + 1) & 0xFFFFFFFF   # key of the in-kernel sampling noise
        self.env_id_offset = 0                                     # global id of this rank's env 0: `_bind_env_ids`
        self._bind_env_ids()
        # parameters, gradients and Adam moments as views of flat buffers (FlatParams): one collective per optimizer step, and on
        # the GPU clip + Adam as one launch (`flat_optim`; tests flip it to compare with `TFAdam` on the same views)
        self.flat = FlatParams(self.policy)
        self.flat_optim = self.device.type == "cuda"
        adam_kw = dict(lr=float(learning_rate) if not callable(learning_rate) else 1e-3, eps=1e-5, betas=(0.9, 0.999))
        # TensorFlow's Adam (epsilon outside the bias correction), like the reference's tf.train.AdamOptimizer -- the CPU path and what the
        # tests compare the flat-buffer kernel with
        self.optimizer = TFAdam(self.policy.parameters(), **adam_kw)
        # the LSTM kernels read [unit][gate]-permuted COPIES of the weights; the in-place update does not tell them,
        # so the copies are refreshed explicitly after every step of this optimizer, whoever calls it
        if hasattr(self.policy, "prepare") and hasattr(self.optimizer, "register_step_post_hook"):
            self.optimizer.register_step_post_hook(lambda *_a, **_k: self.policy.prepare())
        self.loss_names = ['policy_loss', 'value_loss', 'policy_entropy', 'approxkl', 'clipfrac']
        # replicas: identical seeds give identical initial weights, but nothing else guarantees it (a rank-local `load`, a different torch
        # build on one node) -- rank 0's parameters and Adam moments are broadcast once here and after every `load_parameters`, and
        # `check_replicas` compares a checksum across the ranks every REPLICA_CHECK_EVERY updates of `learn`
        self.sync_replicas()
        if self.collective:
            # the minibatch permutations and the sampling-noise key derive from `seed`: ranks that disagree on it would cut DIFFERENT global
            # minibatches while their replicas stay bit-identical -- a wrong gradient no checksum can see
            sd = torch.tensor([self.seed, -self.seed], dtype=torch.int64, device=self._collective_device())
            torch.distributed.all_reduce(sd, op=torch.distributed.ReduceOp.MAX)
            if int(sd[0]) != -int(sd[1]):
                raise ValueError("PPO2: the ranks were given different seeds (%d .. %d): the seed must be the same on every rank (the ranks' "
                                 "data differs by EnvIdOffset, not by seed)" % (-int(sd[1]), int(sd[0])))
        self.fused_loss = True   # single-launch loss forward + backward on the GPU (tests flip it to compare with the eager graph)
        self.fused_mlp = True    # MlpPolicy: forward + loss + every gradient in one launch per network (tests flip it likewise)
        self.fused_heads = True  # ... including the policy / value heads and their gradients (LSTM policy, 48-unit latents)
        self.log = []

    # -- one optimizer step on one minibatch (ppo2.py:243-298) --
    def _adv_stats_indexed(self, returns, values, index):
        """adv_stats = (mean, std) of the raw advantages of the minibatch `index` picks from the flat rollout, float32 [2] on the device:
        two launches of `irrl_adv_moments` instead of the gathers, casts and reductions; several ranks all-reduce the sums first."""
        from . import _lib
        lib = _lib.load()
        dev = returns.device
        scratch = torch.empty(2 * 256 + 3, device=dev, dtype=torch.float64)
        stats = torch.empty(2, device=dev, dtype=torch.float32)
        p = lambda t: C.c_void_p(t.data_ptr())
        rec = getattr(self, "_records", None)       # the packed sample records of this update (`update`): the advantage is word 51 of a record
        adv = getattr(self, "_flat_adv", None)      # returns - values of the whole rollout, formed once per update (`update`): one gather, not two
        if rec is not None and rec.shape[0] == returns.numel():
            _lib.check(lib.irrl_adv_moments_rec(int(index.numel()), p(index), p(rec), p(scratch), 256, p(scratch[512:]),
                                                p(stats) if not self.collective else None, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        else:
            if adv is not None and adv.numel() == returns.numel():
                returns, values = adv, None
            _lib.check(lib.irrl_adv_moments(int(index.numel()), p(index), p(returns), p(values) if values is not None else None, p(scratch), 256, p(scratch[512:]),
                                            p(stats) if not self.collective else None, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        if not self.collective:
            return stats
        mean, var = self._adv_moments(None, None, moments=scratch[512:])
        return torch.stack([mean, torch.sqrt(var)]).to(torch.float32)

    def _adv_moments(self, returns, values, moments=None):
        """(mean, var) of the raw advantages over ALL ranks' samples (ppo2.py:263 `advs.mean()/std()`), float64 device scalars.
        moments: this rank's (sum, sum of squares, count) when they are already on the device."""
        if moments is None:
            advs = returns - values
            n_local = torch.tensor([float(advs.numel())], device=advs.device, dtype=torch.float64)
            moments = torch.stack([advs.double().sum(), (advs.double() ** 2).sum(), n_local[0]])
        if self.collective:
            torch.distributed.all_reduce(moments)                     # C2: 3 floats
        mean = moments[0] / moments[2]
        var = torch.clamp(moments[1] / moments[2] - mean * mean, min=0.0)
        return mean, var

    def _train_step(self, lr_now, cliprange_now, obs, returns, masks, actions, values, neglogpacs, states=None, adv_moments=None, index=None,
                    grad_weight=1.0, empty=False):
        """index: the arrays are the FLAT rollout and `index` picks this minibatch's rows (MlpPolicy's gradient kernels read them
        in place); otherwise the arrays are the minibatch.  grad_weight: this rank's share of a GLOBAL minibatch that the ranks hold
        unequal parts of (`_global_minibatches`): its mean gradient is weighted m_r * world / m before the all-reduce, so that the
        averaged sum is the mean over the global minibatch's m samples; empty: this rank holds none of them (it still joins the collectives)."""
        if empty:
            if adv_moments is None:
                self._adv_moments(None, None, moments=torch.zeros(3, device=self.device, dtype=torch.float64))
            self.flat.grad[:self.flat.n].zero_()
            self._apply_gradients(lr_now, gathered=True)
            return None
        if index is not None:
            if adv_moments is not None:
                adv_stats = torch.stack([adv_moments[0], torch.sqrt(adv_moments[1])]).to(torch.float32)
            else:
                adv_stats = self._adv_stats_indexed(returns, values, index)
            row = mlp_ppo_grads_flat(self.policy, self.flat, obs, actions, returns, values, neglogpacs, adv_stats, cliprange_now, self.ent_coef,
                                     self.vf_coef, index, rec=getattr(self, "_records", None))
            self._apply_gradients(lr_now, gathered=True, weight=grad_weight)
            return row      # raw sums: `update` turns the rows of all steps into the logged means at once (mlp_stats_rows)
        mean, var = adv_moments if adv_moments is not None else self._adv_moments(returns, values)
        advs = None if (self.fused_loss and fused_ppo_loss_supported(self.policy, obs)) else returns - values
        stats = None
        if self.fused_loss and fused_ppo_loss_supported(self.policy, obs):
            # forward + backward of the whole loss in one launch; advantages are normalised inside the kernel
            adv_stats = torch.stack([mean, torch.sqrt(var)]).to(torch.float32)
            pol = self.policy
            if self.fused_heads and hasattr(pol, "fused_heads_supported") and pol.fused_heads_supported(obs):
                # ... and the two heads with it: the LSTM stacks hand their last-layer outputs straight to the kernel
                h_pi, h_v = pol.latents(obs, states, masks)
                loss, stats = _FusedHeadsLoss.apply(h_pi, h_v, pol.pi.w, pol.pi.b, pol.vf.w, pol.vf.b, pol.logstd, actions, returns, values,
                                                    neglogpacs, adv_stats, cliprange_now, self.ent_coef, self.vf_coef, True)   # loss.backward() below, unscaled
            else:
                pmean, vpred = pol.evaluate_raw(obs, states, masks)
                loss, stats = _FusedPPOLoss.apply(pmean, vpred, pol.logstd, actions, returns, values, neglogpacs, adv_stats,
                                                  cliprange_now, self.ent_coef, self.vf_coef)
        else:
            advs = (advs - mean.to(advs.dtype)) / (torch.sqrt(var).to(advs.dtype) + 1e-8)
            neglogpac, vpred, entropy = self.policy.evaluate(obs, states, masks, actions)
            loss, pg, vf, ent, kl, cf = ppo_loss(neglogpac, vpred, entropy, actions, advs, returns, neglogpacs, values, cliprange_now,
                                                 self.ent_coef, self.vf_coef)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self._apply_gradients(lr_now, weight=grad_weight)
        if stats is not None:
            return stats.detach()
        return torch.stack([pg.detach(), vf.detach(), ent.detach(), kl.detach(), cf.detach()])

    def _apply_gradients(self, lr_now, gathered=False, weight=1.0):
        """Average over ranks, clip by the global norm, Adam (ppo2.py:182-189, 283-298).  The gradient lives in ONE flat buffer
        (FlatParams): `gathered` says the gradient kernels wrote it there themselves, otherwise autograd's per-parameter tensors are
        copied in by one multi-tensor launch.  Several ranks: ONE all-reduce of that buffer (283 KB for the LSTM
        policy), nothing packed or unpacked around it; the mean over ranks is folded into the clip scale."""
        fl = self.flat
        if not gathered:
            fl.gather()
        if weight != 1.0:
            fl.grad[:fl.n] *= float(weight)      # this rank's share of a global minibatch held in unequal parts (`_global_minibatches`)
        if self.collective:
            torch.distributed.all_reduce(fl.grad[:fl.n])              # C1: the only collective of the step besides the 3 moment floats
        if self.flat_optim and self.device.type == "cuda":
            from . import _lib
            fl.step += 1
            p = lambda t: C.c_void_p(t.data_ptr())
            _lib.check(_lib.load().irrl_clip_adam(fl.n, p(fl.theta), p(fl.grad), p(fl.m), p(fl.v), 1.0 / self.world,
                                                  float(self.max_grad_norm) if self.max_grad_norm is not None else 0.0, float(lr_now), 0.9, 0.999, 1e-5,
                                                  fl.step, None, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
            if hasattr(self.policy, "prepare"):
                self.policy.prepare()      # the LSTM kernels read [unit][gate]-permuted COPIES of the weights
            return
        if self.world > 1:
            fl.grad[:fl.n] /= self.world
        fl.point_grads_at_views()
        if self.max_grad_norm is not None:
            clip_by_global_norm_(fl.params, self.max_grad_norm)  # tf.clip_by_global_norm AFTER averaging (ppo2.py:192)
        for g in self.optimizer.param_groups:
            g['lr'] = lr_now
        self.optimizer.step()       # (its post-step hook refreshes the kernels' permuted weight copies)
        for prm in fl.params:
            prm.grad = None         # the next backward() leaves fresh tensors, gather() copies them into the views

    def _sample_order(self, n):
        """np.random.shuffle(inds) of ppo2.py:366-367 as a keyed bijection of [0, n) (one launch on the GPU instead of the radix sort behind
        torch.randperm; the numpy twin on the CPU gives the same order): depends on (n, seed, number of shuffles so far) only, so it is the
        same on every rank and on either device."""
        self._shuffles = getattr(self, "_shuffles", 0) + 1
        key = (self.seed * 1000003 + 12345) & 0xFFFFFFFF
        if self.device.type == "cuda" and n <= (1 << 30):
            from . import _lib
            out = torch.empty(n, dtype=torch.int64, device=self.device)
            _lib.check(_lib.load().irrl_random_permutation(n, key, self._shuffles, C.c_void_p(out.data_ptr()),
                                                           C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
            return out
        return torch.from_numpy(feistel_permutation(n, key, self._shuffles)).to(self.device)

    def _global_minibatches(self, order, per_row, size):
        """Several ranks, nminibatches > 1: the reference draws ONE permutation over ALL samples (non-recurrent, ppo2.py:364-380) resp. ALL
        envs (recurrent, ppo2.py:387-402) and cuts it into minibatches.  `order` is that permutation of the GLOBAL ids, identical on every
        rank (global sample id = t * (world * N) + global env id; global env id = rank * N + local env id, the pool's EnvIdOffset layout);
        this rank keeps, of every global minibatch of `size` ids, the ones whose env it owns -- an unequal share m_r of the minibatch
        (binomial around size / world).  Yields (local ids, grad_weight = m_r * world / size): the weighted, all-reduced, averaged
        gradients are the mean over the global minibatch, so the N-rank update equals the single-process one up to summation order.
        per_row: ids per time row of the GLOBAL layout (world * N for sample ids, None for env ids)."""
        n_loc = int(self.n_envs)
        lo = self.rank * n_loc
        if per_row is None:
            env_g, t = order, None
        else:
            t = torch.div(order, per_row, rounding_mode="floor")
            env_g = order - t * per_row
        mine = (env_g >= lo) & (env_g < lo + n_loc)
        local = (env_g - lo) if t is None else (t * n_loc + (env_g - lo))
        for start in range(0, int(order.numel()), size):
            part = local[start:start + size][mine[start:start + size]].contiguous()     # (one host sync per minibatch: its length)
            yield part, float(part.numel()) * self.world / float(size)

    def update(self, batch, lr_now, cliprange_now):
        """All epochs / minibatches of one PPO iteration (ppo2.py:362-404)."""
        T, N = batch["values"].shape
        losses = []
        recurrent = batch["states"] is not None
        if recurrent:
            assert N % self.nminibatches == 0, "For recurrent policies, the number of environments run in parallel " \
                                               "should be a multiple of nminibatches."
            envs_per_batch = N // self.nminibatches
            # one minibatch = the whole rollout: the advantage moments are the same in every epoch
            whole = self._adv_moments(batch["returns"], batch["values"]) if self.nminibatches == 1 else None
            split = self.world > 1 and self.nminibatches > 1      # one permutation over ALL ranks' envs, every rank keeps its own (ppo2.py:387-402)
            for _ in range(self.noptepochs):
                if split:
                    perm_g = torch.randperm(N * self.world, device=self.device, generator=self.generator)   # same generator state on every rank
                    for idx, w in self._global_minibatches(perm_g, None, envs_per_batch * self.world):
                        if idx.numel() == 0:
                            self._train_step(lr_now, cliprange_now, None, None, None, None, None, None, empty=True)
                            continue
                        sl = lambda x: x[:, idx]
                        losses.append(w / self.world * self._train_step(lr_now, cliprange_now, sl(batch["obs"]), sl(batch["returns"]), sl(batch["masks"]),
                                                                        sl(batch["actions"]), sl(batch["values"]), sl(batch["neglogpacs"]),
                                                                        states=batch["states"][idx], grad_weight=w))
                    continue
                perm = torch.randperm(N, device=self.device, generator=self.generator)
                for start in range(0, N, envs_per_batch):
                    idx = perm[start:start + envs_per_batch]
                    if self.nminibatches == 1:
                        sl = lambda x: x                               # whole batch: the env order does not matter
                        st = batch["states"]
                    else:
                        sl = lambda x: x[:, idx]
                        st = batch["states"][idx]
                    losses.append(self._train_step(lr_now, cliprange_now, sl(batch["obs"]), sl(batch["returns"]), sl(batch["masks"]),
                                                   sl(batch["actions"]), sl(batch["values"]), sl(batch["neglogpacs"]), states=st,
                                                   adv_moments=whole))
            if split:
                tot = torch.stack(losses).sum(0) / (self.noptepochs * self.nminibatches)    # this rank's share of every global minibatch's mean
                torch.distributed.all_reduce(tot)
                return tot
        else:
            n_batch = T * N
            assert n_batch % self.nminibatches == 0
            bs = n_batch // self.nminibatches
            # ppo2.py:573 swap_and_flatten makes the batch env-major; for a non-recurrent policy the order of the flat batch only names the
            # samples (every epoch draws a uniform shuffle of them, ppo2.py:366-367), so the [T, N, ...] buffers are flattened as they lie:
            # views, no 0.6 GB of copies per update
            flat = {k: batch[k].reshape(n_batch, *batch[k].shape[2:]) for k in ("obs", "returns", "masks", "actions", "values", "neglogpacs")}
            in_place = self.fused_mlp and mlp_ppo_grads_supported(self.policy, flat["obs"])
            # the samples as packed 256-byte records, built once for the update's noptepochs x nminibatches passes (`MLP_RECORDS`; bf16x3 kernels)
            self._records = (mlp_pack_records(flat["obs"], flat["actions"], flat["returns"], flat["values"], flat["neglogpacs"])
                             if in_place and MLP_RECORDS and MLP_PRECISION == "bf16x3" else None)
            self._flat_adv = (flat["returns"] - flat["values"]).contiguous() if in_place and self._records is None else None
            split = self.world > 1 and self.nminibatches > 1      # ONE permutation over ALL ranks' samples (ppo2.py:364-380), every rank keeps its own
            weights = []
            for _ in range(self.noptepochs):
                if split:
                    order = self._sample_order(n_batch * self.world)
                    parts = self._global_minibatches(order, N * self.world, bs * self.world)
                else:
                    inds = self._sample_order(n_batch)
                    parts = ((inds[start:start + bs], 1.0) for start in range(0, n_batch, bs))
                for mb, w in parts:
                    if mb.numel() == 0:
                        self._train_step(lr_now, cliprange_now, None, None, None, None, None, None, empty=True)
                        continue
                    weights.append(w / self.world)
                    if in_place:     # the gradient kernels read the minibatch's rows through the index: nothing is gathered
                        losses.append(self._train_step(lr_now, cliprange_now, flat["obs"], flat["returns"], flat["masks"], flat["actions"],
                                                       flat["values"], flat["neglogpacs"], index=mb.contiguous(), grad_weight=w))
                        continue
                    losses.append(self._train_step(lr_now, cliprange_now, flat["obs"][mb], flat["returns"][mb], flat["masks"][mb],
                                                   flat["actions"][mb], flat["values"][mb], flat["neglogpacs"][mb], grad_weight=w))
            if split:
                # logged means over the GLOBAL minibatches: every rank's rows are sums (kernels) / means (graph) over ITS share
                self._flat_adv = None
                self._records = None
                if in_place:
                    tot = mlp_stats_rows(losses, float(bs * self.world), self.policy.act_dim)
                    tot[:, 2] *= torch.tensor(weights, device=tot.device, dtype=tot.dtype)      # (the entropy column is not a sum over samples)
                    tot = tot.sum(0)
                else:
                    tot = (torch.stack(losses) * torch.tensor(weights, device=self.device, dtype=torch.float32).unsqueeze(1)).sum(0)
                tot = tot / (self.noptepochs * self.nminibatches)
                torch.distributed.all_reduce(tot)
                return tot
            self._flat_adv = None
            self._records = None
            if in_place:
                return mlp_stats_rows(losses, float(bs), self.policy.act_dim).mean(0)
        return torch.stack(losses).mean(0)

    def learn(self, total_timesteps, callback=None, seed=None, log_interval=1, tb_log_name="PPO2", eval_every_n=5,
              reset_num_timesteps=True, record_video=False, log_dir=""):
        lr_fn = self.learning_rate if callable(self.learning_rate) else (lambda _f: float(self.learning_rate))
        clip_fn = self.cliprange if callable(self.cliprange) else (lambda _f: float(self.cliprange))
        runner = Runner(self.env, self, self.n_steps, self.gamma, self.lam)
        n_batch = self.n_envs * self.n_steps
        nupdates = int(total_timesteps) // (n_batch * self.world)       # total_timesteps counts all ranks' samples
        t_first = time.time()
        csv_path = None
        if self.tensorboard_log and self.rank == 0:
            # the reference logs through tensorboard / the SB logger; headless here: the same table as progress.csv
            os.makedirs(self.tensorboard_log, exist_ok=True)
            csv_path = os.path.join(self.tensorboard_log, "progress.csv")
        try:
            self._learn_loop(nupdates, n_batch, runner, lr_fn, clip_fn, callback, log_interval, eval_every_n, log_dir, t_first, csv_path)
        except KeyboardInterrupt:
            # ppo2.py:443-448: keep what has been learned, then leave
            if self.rank == 0 and log_dir:
                print("[PPO2] interrupted: checkpoint", self.save(log_dir + "_interrupted"))
            raise SystemExit(0)
        return self

    def _learn_loop(self, nupdates, n_batch, runner, lr_fn, clip_fn, callback, log_interval, eval_every_n, log_dir, t_first, csv_path):
        for update in range(1, nupdates + 1):
            if eval_every_n and update % eval_every_n == 1 and self.rank == 0 and log_dir:
                # ppo2.py:331-341: visual test rollout (headless here: skipped) + checkpoint
                self.save(log_dir + "_Iteration_{}".format(update - 1))
            t_start = time.time()
            frac = 1.0 - (update - 1.0) / nupdates
            lr_now, clip_now = lr_fn(frac), clip_fn(frac)
            batch = runner.run()
            loss_vals = self.update(batch, lr_now, clip_now)
            self.num_timesteps += n_batch * self.world
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            t_now = time.time()
            fps = int(n_batch * self.world / (t_now - t_start))
            if self.verbose >= 1 and (update % log_interval == 0 or update == 1):
                y, ypred = batch["returns"].reshape(-1), batch["values"].reshape(-1)
                vary = y.var(unbiased=False)
                ev = float("nan") if float(vary) == 0 else float(1 - (y - ypred).var(unbiased=False) / vary)
                ep_r, ep_l, ep_n = self.env.pop_episode_stats() if hasattr(self.env, "pop_episode_stats") else (float("nan"), float("nan"), 0)
                row = {"serial_timesteps": update * self.n_steps, "nupdates": update, "total_timesteps": self.num_timesteps, "fps": fps,
                       "explained_variance": ev, "ep_reward_mean": ep_r, "ep_len_mean": ep_l, "time_elapsed": t_start - t_first,
                       "iters_per_sec": 1.0 / (t_now - t_start)}
                row.update({k: float(v) for k, v in zip(self.loss_names, loss_vals.tolist())})
                self.log.append(row)
                if self.rank == 0:
                    print(" | ".join("%s %s" % (k, ("%.4g" % v) if isinstance(v, float) else v) for k, v in row.items()), flush=True)
                    if csv_path:
                        new_file = not os.path.exists(csv_path)
                        with open(csv_path, "a") as f:
                            if new_file:
                                f.write(",".join(row.keys()) + "\n")
                            f.write(",".join(repr(v) for v in row.values()) + "\n")
            if self.collective and (update % self.REPLICA_CHECK_EVERY == 0 or update == nupdates):
                self.check_replicas()
            if callback is not None and callback(locals(), globals()) is False:
                break

    # -- stable-baselines BaseRLModel surface used around PPO2 (base_class.py: predict / get_env / set_env) --
    @torch.no_grad()
    def predict(self, observation, state=None, mask=None, deterministic=False):
        """-> (actions clipped to the action space, next LSTM states).  numpy or torch observations [n_env, ob_dim]."""
        is_np = isinstance(observation, np.ndarray)
        obs = torch.as_tensor(observation, dtype=torch.float32, device=self.device)
        if obs.dim() == 1:
            obs = obs.unsqueeze(0)
        n = obs.shape[0]
        st = self.policy.initial_state(n, self.device) if state is None else torch.as_tensor(state, dtype=torch.float32, device=self.device)
        mk = torch.zeros(n, dtype=torch.bool, device=self.device) if mask is None else torch.as_tensor(mask, device=self.device).to(torch.bool)
        actions, _, new_state, _ = self.policy.step(obs, st, mk, deterministic=deterministic, generator=self.generator)
        actions = actions.clamp(-1.0, 1.0)
        if is_np:
            return actions.cpu().numpy(), (new_state.cpu().numpy() if new_state is not None else None)
        return actions, new_state

    def get_env(self):
        return self.env

    def set_env(self, env):
        self.env = env
        self.n_envs = env.num_envs if env is not None else None
        self._bind_env_ids()

    def _bind_env_ids(self):
        """Which robots of the one big pool this rank owns.  Every random draw of the path is addressed by the GLOBAL env id: the env
        pool's own draws by its `EnvIdOffset`, the policy's sampling noise by `env_id_offset` -- the two must be the same number, and
        the ranks' ranges must not overlap, or the N-rank job silently trains on N copies of the same data.  The offset is therefore
        READ from the pool when it exposes one (`env.env_id_offset`: TorchVecEnv / RaisimGymVecEnv over the C-ABI pool); an env without
        one gets rank * n_envs and, with several ranks, a warning; with several ranks the ranges are all-gathered (by every rank) and checked."""
        n = int(self.n_envs or 0)
        off = getattr(self.env, "env_id_offset", None) if self.env is not None else None
        if off is None and self.env is not None and self.world > 1:
            import warnings
            warnings.warn("PPO2: %s exposes no `env_id_offset`; assuming this rank's pool was created with EnvIdOffset = rank * num_envs = %d. "
                          "If it was not, every rank draws the same env random streams and the overlap check below cannot see it."
                          % (type(self.env).__name__, self.rank * n))
        self.env_id_offset = int(off) if off is not None else self.rank * n
        if self.collective:
            # entered by EVERY rank (a rank without an env contributes an empty range): ranks that disagree on having an env must not hang
            mine = torch.tensor([self.env_id_offset, n if self.env is not None else 0], dtype=torch.int64)
            backend = torch.distributed.get_backend()
            if backend == "nccl":
                mine = mine.to(self.device)
            got = [torch.zeros_like(mine) for _ in range(self.world)]
            torch.distributed.all_gather(got, mine)
            spans = sorted((int(g[0]), int(g[0]) + int(g[1])) for g in got if int(g[1]) > 0)
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                if b0 < a1:
                    raise ValueError("PPO2: ranks own overlapping global env ids %s -- give rank r's pool EnvIdOffset = r * num_envs "
                                     "(cfg['environment']['EnvIdOffset']), otherwise every rank draws the same random streams" % (spans,))

    # -- replica consistency of the N-rank learner (no counterpart in the single-process reference) --
    REPLICA_CHECK_EVERY = 50

    def _collective_device(self):
        return self.device if torch.distributed.get_backend() == "nccl" else torch.device("cpu")

    def sync_replicas(self, src=0):
        """Rank `src`'s parameters, Adam moments and step counts to every rank (broadcasts of the flat buffers + one small integer vector); the
        LSTM kernels' permuted weight copies are rebuilt.  A no-op outside a process group."""
        if not self.collective:
            return
        fl = self.flat
        cdev = self._collective_device()
        st0 = self.optimizer.state.get(fl.params[0])
        meta = torch.tensor([fl.step, int(st0["step"]) if st0 else 0, 1 if st0 else 0], dtype=torch.int64, device=cdev)
        torch.distributed.broadcast(meta, src)
        fl.step = int(meta[0])

        def bcast(buf):
            tmp = buf if buf.device == cdev else buf.to(cdev)
            torch.distributed.broadcast(tmp, src)
            if tmp is not buf:
                buf.copy_(tmp)

        for buf in (fl.theta, fl.m, fl.v):
            bcast(buf)
        # the eager optimizer (CPU path; what the tests compare the flat-buffer kernel with) keeps its moments per parameter
        if int(meta[2]):
            for key in ("m", "v"):
                flat_state = torch.cat([(self.optimizer.state[p][key] if self.optimizer.state.get(p) else torch.zeros_like(p)).reshape(-1)
                                        for p in fl.params])
                bcast(flat_state)
                o = 0
                for p in fl.params:
                    st = self.optimizer.state[p]
                    if not st:
                        st["step"], st["m"], st["v"] = 0, torch.zeros_like(p), torch.zeros_like(p)
                    st[key].copy_(flat_state[o:o + p.numel()].view_as(p))
                    st["step"] = int(meta[1])
                    o += p.numel()
        else:
            self.optimizer.state.clear()
        if hasattr(self.policy, "prepare"):
            self.policy.prepare()

    def replica_checksum(self):
        """64-bit checksum of this rank's parameters: the float32 words of the flat buffer read as integers, summed (wrapping)."""
        return self.flat.theta.view(torch.int32).to(torch.int64).sum()

    def check_replicas(self):
        """All ranks hold bit-identical parameters, or RuntimeError: ONE all-reduce (MAX) of (checksum, -checksum) -> max and -min over the
        ranks.  Data-parallel PPO keeps the replicas identical by construction (same averaged gradient, same Adam step on every rank); a
        silent divergence -- a rank-local load, a rank that skipped a collective -- otherwise only shows up as a bad policy hours later."""
        if not self.collective:
            return True
        cs = self.replica_checksum().to(self._collective_device())
        pair = torch.stack([cs, -cs])
        torch.distributed.all_reduce(pair, op=torch.distributed.ReduceOp.MAX)
        if int(pair[0]) != -int(pair[1]):
            raise RuntimeError("PPO2: the ranks' policy parameters differ (checksum max %d, min %d; this rank %d of %d: %d) -- replicas have "
                               "diverged; call sync_replicas() after any rank-local change of the parameters"
                               % (int(pair[0]), -int(pair[1]), self.rank, self.world, int(cs)))
        return True

    # -- checkpoints (ppo2.py:452-476): (data dict, parameter list in stable-baselines order) --
    def _data(self):
        return {"gamma": self.gamma, "n_steps": self.n_steps, "vf_coef": self.vf_coef, "ent_coef": self.ent_coef,
                "max_grad_norm": self.max_grad_norm, "learning_rate": self.learning_rate if not callable(self.learning_rate) else None,
                "lam": self.lam, "nminibatches": self.nminibatches, "noptepochs": self.noptepochs,
                "cliprange": self.cliprange if not callable(self.cliprange) else None, "verbose": self.verbose,
                "policy": type(self.policy).__name__, "n_envs": self.n_envs, "policy_kwargs": self.policy_kwargs}

    def get_parameter_list(self):
        ps = self.policy.sb_parameters() if hasattr(self.policy, "sb_parameters") else list(self.policy.parameters())
        return [p.detach().cpu().numpy().copy() for p in ps]

    def save(self, save_path):
        d = os.path.dirname(save_path)
        if d:
            os.makedirs(d, exist_ok=True)
        path = save_path if save_path.endswith(".pkl") else save_path + ".pkl"
        with open(path, "wb") as f:
            pickle.dump((self._data(), self.get_parameter_list()), f)
        return path

    def load_parameters(self, params):
        ps = self.policy.sb_parameters() if hasattr(self.policy, "sb_parameters") else list(self.policy.parameters())
        assert len(ps) == len(params), "expected %d tensors, got %d" % (len(ps), len(params))
        with torch.no_grad():
            for p, a in zip(ps, params):
                a = torch.as_tensor(np.asarray(a), dtype=p.dtype)
                assert tuple(a.shape) == tuple(p.shape), (tuple(a.shape), tuple(p.shape))
                p.copy_(a.to(p.device))
        self.sync_replicas()      # several ranks: rank 0's copy is THE copy (a no-op for identical files, a repair for a rank-local one)

    @classmethod
    def load(cls, load_path, env=None, device=None, **kwargs):
        """Loads this build's checkpoints AND the reference's stable-baselines pickles (e.g. script/pkl/bp5_155.pkl,
        decoded without tensorflow / stable_baselines / gym by a stub unpickler) -- the IRRL stage-2 warm start
        (run_bp_v5.py:244-249, readme.md:66-70)."""
        from .checkpoint import read_checkpoint
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        if dist_on:
            # run_bp_v5.py:245-248 under a launcher: rank 0 reads the file, every rank gets its content (the other ranks need not see the path)
            box = [read_checkpoint(load_path) if torch.distributed.get_rank() == 0 else None]
            kw = {}
            if torch.distributed.get_backend() == "nccl":
                kw["device"] = torch.device("cuda", torch.cuda.current_device())
            torch.distributed.broadcast_object_list(box, 0, **kw)
            data, params = box[0]
        else:
            data, params = read_checkpoint(load_path)
        pk = data.get("policy_kwargs") or {}
        n_lstm = pk.get("n_lstm", [48, 48])
        is_lstm = len(params) == 19 or data.get("policy") in ("CustomLSTMPolicy",)
        if is_lstm and len(params) == 19:
            n_lstm = [int(np.asarray(params[1]).shape[0]), int(np.asarray(params[4]).shape[0])]   # wh of the two actor layers: [h, 4h]
        policy = CustomLSTMPolicy(n_lstm=n_lstm) if is_lstm else MlpPolicy()
        hp = {k: data[k] for k in ("gamma", "n_steps", "ent_coef", "vf_coef", "max_grad_norm", "lam", "nminibatches", "noptepochs")
              if k in data and isinstance(data[k], (int, float))}
        for k in ("learning_rate", "cliprange"):
            hp[k] = data[k] if isinstance(data.get(k), float) else {"learning_rate": 1e-3, "cliprange": 0.2}[k]
        hp.update(kwargs)
        model = cls(policy=policy, env=env, device=device, policy_kwargs={"n_lstm": list(n_lstm)} if is_lstm else None, **hp)
        model.load_parameters(params)
        return model
