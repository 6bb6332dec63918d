"""autograd wrapper of the persistent LSTM sequence kernels (csrc/lstm_kernels.hip).

forward : zx = x @ wx + b as one GEMM over all T*N rows (gate columns permuted to [unit][gate]); the recurrence runs
          in ONE kernel launch per layer (`irrl_lstm_seq_forward`), saving post-activation gates and c for the
          backward pass.
backward: `irrl_lstm_seq_backward` walks the sequence in reverse and emits dz [T,N,H,4]; the weight / input gradients
          are then four large GEMMs (dwx = x^T dz, dwh = (h_prev keep)^T dz, db = sum dz, dx = dz wx^T).
Semantics = stable-baselines `lstm` (policies.SBLstm.sequence is the eager definition the tests compare with).
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib

_PERM_CACHE = {}
FUSE_WEIGHT_GRADIENTS = True   # backward: dx / dwx / dwh / db inside the sequence kernel
FUSE_INPUT_PROJECTION = True   # module switch (benchmarks / tests compare the two forward kernels)
# Arithmetic of the two sequence kernels of the UPDATE (hid 48, n_in <= 48; the rollout's policy step always runs the exact-f32 MFMA):
#   "bf16x3"  bf16 matrix cores, every operand split into 2 bf16 planes, 3 plane products per product (~2^-16 relative), f32 accumulation.
#             DEFAULT since round 4: PPO update 158 -> 109 ms at 4096 x 750; weight gradients within 6e-6 / 8e-6 of float64 autograd
#             (the eager f32 graph: 1.6e-5, the exact-f32 kernels: 2e-6), h and dx within 1e-5 (exact-f32 kernels: 4e-7)
#             -- profiles/r04_lstm_precision_error_and_time.log
#   "bf16x6"  3 planes, 6 plane products (~2^-24): every output within ~2x of the exact-f32 kernels' error, update 147 ms
#   "f32"     v_mfma_f32_16x16x4_f32, bitwise an fmaf chain (rounds 1-3), update 158 ms
# (csrc/lstm_bf16.hpp; IRRL_LSTM_PRECISION overrides the default)
PRECISION = os.environ.get("IRRL_LSTM_PRECISION", "bf16x3")
_NSPLIT = {"bf16x3": 2, "bf16x6": 3, "f32": 0}
# bf16x3, OPT-IN (round 6, verdict r5 item 2 -- built, bit-identical, measured slower): IRRL_LSTM_RECOMPUTE=1 makes the forward kernel of the update keep
# c and h only and the backward kernel RECOMPUTE the gates from the h / x tiles it stages for the weight gradients anyway (csrc/lstm_bf16.hpp
# lstm_seq_bwd_bf16_rc_kernel): 2/3 of the forward kernel's stores and half of the backward kernel's loads gone, forward pairs 3.2 -> 2.1 ms per epoch, the
# four backward launches 5.7 -> ~7.5 ms: update 94.4 -> 106.4 ms on the same box (profiles/r06_ab_lstm_recompute_same_box.log).  Default: the gates are
# stored and loaded as in rounds 4-5.
RECOMPUTE_GATES = os.environ.get("IRRL_LSTM_RECOMPUTE", "0") != "0"


def check_precision(name, what="lstm_fused.PRECISION / IRRL_LSTM_PRECISION"):
    """A typo must not fall through to some other arithmetic silently."""
    if name not in _NSPLIT:
        raise ValueError("%s is one of %s, not %r" % (what, sorted(_NSPLIT), name))
    return name


check_precision(PRECISION)


def _perm(hid, device):
    """permuted column u*4+g  <-  reference column g*hid+u (gate order i,f,o,g)."""
    key = (hid, str(device))
    if key not in _PERM_CACHE:
        u = torch.arange(hid, device=device).repeat_interleave(4)
        g = torch.arange(4, device=device).repeat(hid)
        perm = g * hid + u
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(4 * hid, device=device)
        _PERM_CACHE[key] = (perm, inv)
    return _PERM_CACHE[key]


def _ptr(t):
    return C.c_void_p(t.data_ptr())


_WCACHE = {}


def _permuted_weights(wx, wh, b, perm, force=False):
    """[unit][gate]-ordered copies of the layer's weights.  They are refreshed (in place: captured graphs keep reading the
    same buffers) when `force` is set -- `SBLstm.prepare()`, which the learner calls after EVERY optimizer step and the
    runner before every rollout -- or when a parameter's `_version` changed.  The version alone is NOT enough: the fused
    Adam kernel updates the parameters without bumping it."""
    key = (wx.data_ptr(), wh.data_ptr(), b.data_ptr())
    ver = (wx._version, wh._version, b._version)
    hit = _WCACHE.get(key)
    if hit is not None and hit[2]() is None:
        hit = None   # the parameter these copies were made from is gone: another tensor now lives at its address
    if hit is not None and hit[0] == ver and not force:
        return hit[1]
    with torch.no_grad():
        if hit is not None:
            # refresh IN PLACE: a captured hipGraph of the rollout step keeps reading these very buffers
            out = hit[1]
            torch.index_select(wx, 1, perm, out=out[0])
            torch.index_select(wh, 1, perm, out=out[1])
            torch.index_select(b, 0, perm, out=out[2])
        else:
            out = (wx[:, perm].contiguous(), wh[:, perm].contiguous(), b[perm].contiguous())
    _WCACHE[key] = (ver, out, weakref.ref(wx))
    return out


def refresh_weights(wx, wh, b):
    """Rebuild the cached [unit][gate] copies from the parameters, unconditionally (after an optimizer step, before a rollout)."""
    if wx.is_cuda:
        _permuted_weights(wx, wh, b, _perm(wh.shape[0], wx.device)[0], force=True)


def _tall_gemm_t(a, b, chunks=256):
    """a^T @ b for a [K, m], b [K, n] with K in the millions and m, n <= 192: a batched GEMM over K-chunks followed
    by a small sum exposes enough parallelism (one skinny GEMM with K = 3e6 runs on a handful of CUs)."""
    K = a.shape[0]
    if K % chunks != 0 or K < 64 * chunks:
        return a.t() @ b
    ac = a.reshape(chunks, K // chunks, a.shape[1])
    bc = b.reshape(chunks, K // chunks, b.shape[1])
    return torch.bmm(ac.transpose(1, 2), bc).sum(0)


class _TallLinearFn(torch.autograd.Function):
    """y = x @ w + b for x with millions of rows and a handful of columns (the policy / value heads over a whole
    rollout).  Autograd's dw = x^T dy is one skinny GEMM with K = T*N that rocBLAS runs on a few CUs (5.5 ms for the
    [3.07M, 48]^T [3.07M, 12] of the action head); the split-K batched form takes ~0.1 ms."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x.reshape(-1, x.shape[-1]), w).reshape(*x.shape[:-1], w.shape[1])

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        dx = (dy2 @ w.t()).reshape(x.shape) if ctx.needs_input_grad[0] else None
        return dx, _tall_gemm_t(x2, dy2.contiguous()), dy2.sum(0)


def tall_linear(x, w, b):
    return _TallLinearFn.apply(x, w, b)


class _LstmSeqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wx, wh, b, state0, masks, no_grad=False, precision=None):
        lib = _lib.load()
        T, N, n_in = x.shape
        hid = wh.shape[0]
        perm, inv = _perm(hid, x.device)
        wx_p, wh_p, b_p = _permuted_weights(wx, wh, b, perm)
        x = x.contiguous()
        masks = masks.to(torch.float32).contiguous()
        state0 = state0.contiguous()
        pad = (-N) % 16
        if pad:
            x_k = torch.cat([x, x.new_zeros(T, pad, n_in)], 1)
            masks_k = torch.cat([masks, masks.new_zeros(T, pad)], 1)
            state0_k = torch.cat([state0, state0.new_zeros(pad, 2 * hid)], 0)
        else:
            x_k, masks_k, state0_k = x, masks, state0
        Np = N + pad
        nsplit = _NSPLIT[check_precision(precision if precision is not None else PRECISION)] if (hid == 48 and n_in <= 48 and FUSE_INPUT_PROJECTION and FUSE_WEIGHT_GRADIENTS) else 0
        # no gradient will be asked for (the caller ran under torch.no_grad(): the critic pass behind an actor-only rollout): the bf16 kernels'
        # inference form keeps neither the gates nor the c rows -- a third of the stores.  (`no_grad` comes from the caller: inside forward()
        # autograd has switched grad mode off whatever the caller's was.)
        infer = bool(nsplit) and bool(no_grad)
        recompute = nsplit == 2 and RECOMPUTE_GATES and not infer
        gates = None if (infer or recompute) else torch.empty(T, Np, hid, 4, device=x.device, dtype=torch.float32)
        cseq = None if infer else torch.empty(T, Np, hid, device=x.device, dtype=torch.float32)
        hseq = torch.empty(T, Np, hid, device=x.device, dtype=torch.float32)
        state_out = torch.empty(Np, 2 * hid, device=x.device, dtype=torch.float32)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        ctx.nsplit = nsplit
        if nsplit:
            rc = lib.irrl_lstm_seq_forward_bf16(nsplit, hid, T, Np, n_in, _ptr(x_k), _ptr(wx_p), _ptr(b_p), _ptr(wh_p), _ptr(masks_k), _ptr(state0_k),
                                                None if gates is None else _ptr(gates), None if infer else _ptr(cseq), _ptr(hseq), _ptr(state_out), stream)
        elif n_in <= 48 and FUSE_INPUT_PROJECTION:
            # x wx + b inside the sequence kernel: no [T*N, 4H] zx round trip through HBM
            rc = lib.irrl_lstm_seq_forward_x(hid, T, Np, n_in, _ptr(x_k), _ptr(wx_p), _ptr(b_p), _ptr(wh_p), _ptr(masks_k), _ptr(state0_k),
                                             _ptr(gates), _ptr(cseq), _ptr(hseq), _ptr(state_out), stream)
        else:
            zx = torch.addmm(b_p, x_k.reshape(T * Np, n_in), wx_p)             # [T*Np, 4H] in [unit][gate] order
            rc = lib.irrl_lstm_seq_forward(hid, T, Np, _ptr(zx), _ptr(wh_p), _ptr(masks_k), _ptr(state0_k), _ptr(gates), _ptr(cseq),
                                           _ptr(hseq), _ptr(state_out), stream)
        if rc != 0:
            raise RuntimeError("irrl_lstm_seq_forward failed (rc=%d, hid=%d, T=%d, N=%d)" % (rc, hid, T, Np))
        if not infer:
            ctx.save_for_backward(x_k, wx_p, wh_p, gates, cseq, hseq, masks_k, state0_k, b_p)
        ctx.dims = (T, N, Np, n_in, hid)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(state_out)
        return (hseq[:, :N] if pad else hseq), (state_out[:N] if pad else state_out)

    @staticmethod
    def backward(ctx, dh_seq, _dstate):
        lib = _lib.load()
        x_k, wx_p, wh_p, gates, cseq, hseq, masks_k, state0_k, b_p = ctx.saved_tensors
        T, N, Np, n_in, hid = ctx.dims
        perm, inv = _perm(hid, x_k.device)
        if dh_seq is None:
            dh_seq = torch.zeros(T, N, hid, device=x_k.device)
        dh_seq = dh_seq.contiguous()
        if Np != N:
            dh_seq = torch.cat([dh_seq, dh_seq.new_zeros(T, Np - N, hid)], 1)
        stream = C.c_void_p(torch.cuda.current_stream(x_k.device).cuda_stream)
        if n_in <= 48 and FUSE_WEIGHT_GRADIENTS:
            # one launch: recurrence + dx + per-workgroup dwx / dwh / db partials (no dz tensor, no tall GEMMs)
            nb = Np // 16
            dev = x_k.device
            dx_k = torch.empty(T, Np, n_in, device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
            dwx_part = torch.empty(nb, n_in, 4 * hid, device=dev, dtype=torch.float32)
            dwh_part = torch.empty(nb, hid, 4 * hid, device=dev, dtype=torch.float32)
            db_part = torch.empty(nb * 4, 4 * hid, device=dev, dtype=torch.float32)
            if getattr(ctx, "nsplit", 0):
                rc = lib.irrl_lstm_seq_backward_bf16(ctx.nsplit, hid, T, Np, n_in, _ptr(gates) if gates is not None else None, _ptr(cseq), _ptr(hseq), _ptr(x_k),
                                                     _ptr(masks_k), _ptr(state0_k), _ptr(dh_seq), _ptr(wh_p), _ptr(wx_p), _ptr(b_p),
                                                     _ptr(dx_k) if dx_k is not None else None,
                                                     _ptr(dwx_part), _ptr(dwh_part), _ptr(db_part), stream)
            else:
                rc = lib.irrl_lstm_seq_backward_x(hid, T, Np, n_in, _ptr(gates), _ptr(cseq), _ptr(hseq), _ptr(x_k), _ptr(masks_k), _ptr(state0_k),
                                                  _ptr(dh_seq), _ptr(wh_p), _ptr(wx_p), _ptr(dx_k) if dx_k is not None else None,
                                                  _ptr(dwx_part), _ptr(dwh_part), _ptr(db_part), stream)
            if rc != 0:
                raise RuntimeError("irrl_lstm_seq_backward_x failed (rc=%d)" % rc)
            # workgroup rows added in one fixed order, gate columns back in the reference's order (the library's reduction over the
            # outer dimension of a [256, 9216] matrix takes 250 us; irrl_sum_rows 10)
            dwx = torch.empty(n_in, 4 * hid, device=dev, dtype=torch.float32)
            dwh = torch.empty(hid, 4 * hid, device=dev, dtype=torch.float32)
            db = torch.empty(4 * hid, device=dev, dtype=torch.float32)
            for part, rows, out in ((dwx_part, nb, dwx), (dwh_part, nb, dwh), (db_part, nb * 4, db)):
                if lib.irrl_sum_rows(_ptr(part), rows, out.numel(), hid, _ptr(out), stream) != 0:
                    raise RuntimeError("irrl_sum_rows failed")
            dx = dx_k[:, :N] if dx_k is not None else None
            return dx, dwx, dwh, db, None, None, None, None
        dz = torch.empty(T, Np, hid, 4, device=x_k.device, dtype=torch.float32)
        rc = lib.irrl_lstm_seq_backward(hid, T, Np, _ptr(gates), _ptr(cseq), _ptr(masks_k), _ptr(state0_k), _ptr(dh_seq), _ptr(wh_p),
                                        _ptr(dz), stream)
        if rc != 0:
            raise RuntimeError("irrl_lstm_seq_backward failed (rc=%d)" % rc)
        dzf = dz.reshape(T * Np, 4 * hid)
        keep = (1.0 - masks_k).unsqueeze(-1)
        hprev = torch.cat([state0_k[:, hid:].unsqueeze(0), hseq[:-1]], 0) * keep     # h_{t-1} as it entered step t
        dwh = _tall_gemm_t(hprev.reshape(T * Np, hid), dzf)[:, inv]
        dwx = _tall_gemm_t(x_k.reshape(T * Np, n_in), dzf)[:, inv]
        db = dzf.sum(0)[inv]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = (dzf @ wx_p.t()).reshape(T, Np, n_in)[:, :N]
        return dx, dwx, dwh, db, None, None, None, None


def lstm_sequence(x, wx, wh, b, state0, masks, precision=None):
    """x [T,N,n_in], state0 [N,2H] = [c|h], masks [T,N] -> (h_seq [T,N,H], final state [N,2H]) on the MI355X.
    precision: arithmetic of this call's kernels ("bf16x3" | "bf16x6" | "f32"); None = the module default PRECISION, read at call time."""
    return _LstmSeqFn.apply(x, wx, wh, b, state0, masks, not torch.is_grad_enabled(), precision)


def supported(x, hid):
    return x.is_cuda and x.dtype == torch.float32 and hid in (32, 48, 64)


# ---- the whole policy step of a rollout in one launch (csrc/lstm_kernels.hip: lstm_policy_step_kernel) ----
def policy_step_supported(policy, obs):
    n = policy.n_lstm
    return (obs.is_cuda and obs.dtype == torch.float32 and len(n) == 2 and n[0] == n[1] and n[0] in (32, 48, 64)
            and 16 * policy.act_dim + 16 <= 8 * n[0] and policy.act_dim <= 16)


def policy_step(policy, obs, states, dones, noise=None, rng=None, states_out=None, rollout=None, out=None):
    """obs [N,ob], states [N,8H], dones [N] bool/u8 -> action, clipped, value, neglogp, states_out.
    Sampling: `noise` [N,act] if given; else rng = (seed, step[, base[, env0]]) draws it in the kernel (counter RNG at step
    `step + base`, base an int64 device scalar); else deterministic.
    `rollout` = dict(row=t, mb_obs, mb_actions, mb_values, mb_neglogpacs, mb_dones, and optionally mb_rewards +
    prev_reward): row t of each buffer is written and the reward of the previous step goes to row t-1.
    `out` = preallocated (action, clipped, value, neglogp) (graph capture of many steps without allocations)."""
    lib = _lib.load()
    N, ob_dim = obs.shape
    hid, act = policy.n_lstm[0], policy.act_dim
    dev = obs.device
    perm = _perm(hid, dev)[0]
    ptrs = []
    for l in list(policy.lstm_pi) + list(policy.lstm_v):
        wx_p, wh_p, b_p = _permuted_weights(l.wx, l.wh, l.b, perm)
        ptrs += [wx_p.data_ptr(), wh_p.data_ptr(), b_p.data_ptr()]
    warr = (C.c_void_p * 12)(*ptrs)
    obs = obs.contiguous()
    states = states.contiguous()
    if states_out is None:
        states_out = torch.empty_like(states)
    assert states_out.is_contiguous() and dones.is_contiguous() and dones.element_size() == 1
    if out is None:
        out = (torch.empty(N, act, device=dev), torch.empty(N, act, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev))
    action, clipped, value, neglogp = out
    if noise is not None:
        noise = noise.contiguous()
    rng_on, seed, step, base, env0 = 0, 0, 0, None, 0
    if rng is not None and noise is None:
        rng_on, seed, step = 1, int(rng[0]) & 0xFFFFFFFF, int(rng[1])
        base = _ptr(rng[2]) if len(rng) > 2 and rng[2] is not None else None
        env0 = int(rng[3]) if len(rng) > 3 else 0      # global id of env 0 (multi-GPU shards)
    if rollout is not None:
        opt = lambda k: _ptr(rollout[k]) if rollout.get(k) is not None else None
        row = int(rollout["row"])
        rptr = [_ptr(rollout["mb_obs"]), _ptr(rollout["mb_actions"]), _ptr(rollout["mb_values"]),
                _ptr(rollout["mb_neglogpacs"]), _ptr(rollout["mb_dones"]), opt("mb_rewards"), opt("prev_reward")]
    else:
        row, rptr = -1, [None] * 7
    rc = lib.irrl_lstm_policy_step(hid, ob_dim, act, N, _ptr(obs), _ptr(dones), _ptr(states), _ptr(states_out), warr,
                                   _ptr(policy.pi.w), _ptr(policy.pi.b), _ptr(policy.vf.w), _ptr(policy.vf.b), _ptr(policy.logstd),
                                   _ptr(noise) if noise is not None else None, rng_on, seed, step, base, env0,
                                   _ptr(action), _ptr(clipped), _ptr(value), _ptr(neglogp), row,
                                   *rptr, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError("irrl_lstm_policy_step failed (rc=%d)" % rc)
    return action, clipped, value, neglogp, states_out


def policy_rollout(policy, env_impl, steps, obs, states, dones, rng, rollout, out, env_reward, env_extra, noise_all=None, fused=False):
    """`steps` rollout steps (policy_step + env.step on the clipped action) issued back to back by one C call
    (irrl_lstm_rollout): obs / dones / states are updated in place, rows row .. row + steps - 1 of the rollout buffers are written
    (rewards one row behind, the last one is left in `env_reward`).  `env_impl` is the FlexibleGymEnv that owns the pool.
    fused=2: the whole rollout as ONE persistent launch (irrl_rollout_persistent_kernel_l16: a workgroup loops over all steps for its
    16 robots); fused=1 / True: env.step k and policy step k + 1 as one launch (csrc/env_kernels.hip irrl_step_policy_kernel_l16; same bits,
    measured slower than two launches -- an experiment)."""
    lib = _lib.load()
    N, ob_dim = obs.shape
    hid, act = policy.n_lstm[0], policy.act_dim
    dev = obs.device
    perm = _perm(hid, dev)[0]
    ptrs = []
    for l in list(policy.lstm_pi) + list(policy.lstm_v):
        wx_p, wh_p, b_p = _permuted_weights(l.wx, l.wh, l.b, perm)
        ptrs += [wx_p.data_ptr(), wh_p.data_ptr(), b_p.data_ptr()]
    warr = (C.c_void_p * 12)(*ptrs)
    assert obs.is_contiguous() and states.is_contiguous() and dones.is_contiguous() and dones.element_size() == 1
    assert env_reward.is_contiguous() and env_extra.is_contiguous() and tuple(env_extra.shape) == (N, 6)
    action, clipped, value, neglogp = out
    rng_on, seed, step, base, env0 = 0, 0, 0, None, 0
    if noise_all is not None:
        assert noise_all.is_contiguous() and tuple(noise_all.shape[1:]) == (N, act) and noise_all.shape[0] >= steps
    elif rng is not None:
        rng_on, seed, step = 1, int(rng[0]) & 0xFFFFFFFF, int(rng[1])
        base = _ptr(rng[2]) if len(rng) > 2 and rng[2] is not None else None
        env0 = int(rng[3]) if len(rng) > 3 else 0      # global id of env 0 (multi-GPU shards)
    rc = lib.irrl_lstm_rollout(env_impl._h, int(steps), hid, ob_dim, act, _ptr(obs), _ptr(dones), _ptr(states), _ptr(states), warr,
                               _ptr(policy.pi.w), _ptr(policy.pi.b), _ptr(policy.vf.w), _ptr(policy.vf.b), _ptr(policy.logstd),
                               _ptr(noise_all) if noise_all is not None else None, rng_on, seed, step, base, env0,
                               _ptr(action), _ptr(clipped), _ptr(value), _ptr(neglogp), int(rollout["row"]),
                               _ptr(rollout["mb_obs"]), _ptr(rollout["mb_actions"]), _ptr(rollout["mb_values"]), _ptr(rollout["mb_neglogpacs"]),
                               _ptr(rollout["mb_dones"]), _ptr(rollout["mb_rewards"]), _ptr(env_reward), _ptr(env_extra), int(fused),
                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _lib.check(rc)


def mlp_policy_step_supported(policy, obs):
    return (obs.is_cuda and obs.dtype == torch.float32 and len(policy.pi_fc) == 2 and policy.pi_fc[0].w.shape[1] == 64
            and policy.pi_fc[1].w.shape == (64, 64) and policy.act_dim <= 15)


def mlp_policy_step(policy, obs, dones, noise=None, rng=None, rollout=None, out=None):
    """MlpPolicy counterpart of `policy_step` (no recurrent state): -> action, clipped, value, neglogp."""
    lib = _lib.load()
    N, ob_dim = obs.shape
    act = policy.act_dim
    dev = obs.device
    ws = [policy.pi_fc[0].w, policy.pi_fc[0].b, policy.pi_fc[1].w, policy.pi_fc[1].b,
          policy.vf_fc[0].w, policy.vf_fc[0].b, policy.vf_fc[1].w, policy.vf_fc[1].b]
    warr = (C.c_void_p * 8)(*[t.data_ptr() for t in ws])
    obs = obs.contiguous()
    assert dones.is_contiguous() and dones.element_size() == 1
    if out is None:
        out = (torch.empty(N, act, device=dev), torch.empty(N, act, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev))
    action, clipped, value, neglogp = out
    if noise is not None:
        noise = noise.contiguous()
    rng_on, seed, step, base, env0 = 0, 0, 0, None, 0
    if rng is not None and noise is None:
        rng_on, seed, step = 1, int(rng[0]) & 0xFFFFFFFF, int(rng[1])
        base = _ptr(rng[2]) if len(rng) > 2 and rng[2] is not None else None
        env0 = int(rng[3]) if len(rng) > 3 else 0      # global id of env 0 (multi-GPU shards)
    if rollout is not None:
        opt = lambda k: _ptr(rollout[k]) if rollout.get(k) is not None else None
        row = int(rollout["row"])
        rptr = [_ptr(rollout["mb_obs"]), _ptr(rollout["mb_actions"]), _ptr(rollout["mb_values"]),
                _ptr(rollout["mb_neglogpacs"]), _ptr(rollout["mb_dones"]), opt("mb_rewards"), opt("prev_reward")]
    else:
        row, rptr = -1, [None] * 7
    rc = lib.irrl_mlp_policy_step(64, ob_dim, act, N, _ptr(obs), _ptr(dones), warr, _ptr(policy.pi.w), _ptr(policy.pi.b), _ptr(policy.vf.w),
                                  _ptr(policy.vf.b), _ptr(policy.logstd), _ptr(noise) if noise is not None else None, rng_on, seed, step, base, env0,
                                  _ptr(action), _ptr(clipped), _ptr(value), _ptr(neglogp), row, *rptr,
                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError("irrl_mlp_policy_step failed (rc=%d)" % rc)
    return action, clipped, value, neglogp


# how the MlpPolicy rollout is issued by `mlp_policy_rollout`: "persistent" = ONE launch for the whole rollout (csrc/env_kernels.hip
# irrl_rollout_persistent_mlp_kernel_l16: a workgroup keeps its 16 robots and the policy's weights for all steps), "direct" = 2 x steps
# launches from one C call.  Same bits.
MLP_ROLLOUT = os.environ.get("IRRL_MLP_ROLLOUT", "persistent")


def mlp_policy_rollout(policy, env_impl, steps, obs, dones, rng, rollout, out, env_reward, env_extra, noise_all=None, fused=None):
    """MlpPolicy counterpart of `policy_rollout` (irrl_mlp_rollout): `steps` rollout steps from one C call; obs / dones updated in place,
    rows row .. row + steps - 1 of the rollout buffers written (rewards one row behind, the last one is left in `env_reward`).
    fused: 2 = one persistent launch, 0 = two launches per step; None: by MLP_ROLLOUT."""
    lib = _lib.load()
    N, ob_dim = obs.shape
    act = policy.act_dim
    dev = obs.device
    ws = [policy.pi_fc[0].w, policy.pi_fc[0].b, policy.pi_fc[1].w, policy.pi_fc[1].b,
          policy.vf_fc[0].w, policy.vf_fc[0].b, policy.vf_fc[1].w, policy.vf_fc[1].b]
    assert all(t.is_contiguous() for t in ws)
    warr = (C.c_void_p * 8)(*[t.data_ptr() for t in ws])
    assert obs.is_contiguous() and dones.is_contiguous() and dones.element_size() == 1
    assert env_reward.is_contiguous() and env_extra.is_contiguous() and tuple(env_extra.shape) == (N, 6)
    action, clipped, value, neglogp = out
    rng_on, seed, step, base, env0 = 0, 0, 0, None, 0
    if noise_all is not None:
        assert noise_all.is_contiguous() and tuple(noise_all.shape[1:]) == (N, act) and noise_all.shape[0] >= steps
    elif rng is not None:
        rng_on, seed, step = 1, int(rng[0]) & 0xFFFFFFFF, int(rng[1])
        base = _ptr(rng[2]) if len(rng) > 2 and rng[2] is not None else None
        env0 = int(rng[3]) if len(rng) > 3 else 0      # global id of env 0 (multi-GPU shards)
    if fused is None:
        if MLP_ROLLOUT not in ("persistent", "direct"):
            raise ValueError("IRRL_MLP_ROLLOUT / lstm_fused.MLP_ROLLOUT is 'persistent' or 'direct', not %r" % (MLP_ROLLOUT,))
        fused = 2 if MLP_ROLLOUT == "persistent" else 0
    rc = lib.irrl_mlp_rollout(env_impl._h, int(steps), 64, ob_dim, act, _ptr(obs), _ptr(dones), warr, _ptr(policy.pi.w), _ptr(policy.pi.b),
                              _ptr(policy.vf.w), _ptr(policy.vf.b), _ptr(policy.logstd), _ptr(noise_all) if noise_all is not None else None,
                              rng_on, seed, step, base, env0, _ptr(action), _ptr(clipped), _ptr(value), _ptr(neglogp), int(rollout["row"]),
                              _ptr(rollout["mb_obs"]), _ptr(rollout["mb_actions"]), _ptr(rollout["mb_values"]), _ptr(rollout["mb_neglogpacs"]),
                              _ptr(rollout["mb_dones"]), _ptr(rollout["mb_rewards"]), _ptr(env_reward), _ptr(env_extra), int(fused),
                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _lib.check(rc)
