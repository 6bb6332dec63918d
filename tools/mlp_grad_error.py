"""Error of the MlpPolicy gradient kernels (irrl_mlp_ppo_grads) and of the eager f32 autograd graph against float64 autograd, per
parameter, on a shuffled minibatch of n rows: python tools/mlp_grad_error.py [n]"""
import copy
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from high_speed_quadrupedal_locomotion_by_irrl_amd import ppo2 as P2
from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import MlpPolicy, diag_gaussian_neglogp, diag_gaussian_entropy


def main(n):
    dev = torch.device("cuda")
    torch.manual_seed(3)
    pol = MlpPolicy().to(dev)
    g = torch.Generator(device=dev); g.manual_seed(11)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    with torch.no_grad():
        pol.pi.w.mul_(30.0); pol.pi.b.add_(0.1 * rn(12)); pol.logstd.add_(0.2 * rn(1, 12))
        for l in (*pol.pi_fc, *pol.vf_fc):
            l.b.add_(0.1 * rn(*l.b.shape))
    rows = 2 * n + 77
    obs, actions, returns, old_v = rn(rows, 35), 0.5 * rn(rows, 12), rn(rows), rn(rows)
    index = torch.randperm(rows, device=dev, generator=g)[:n].contiguous()
    with torch.no_grad():
        old_nlp = diag_gaussian_neglogp(actions, pol._run(obs)[0], pol.logstd) + 0.3 * rn(rows)

    def eager(policy, dt):
        c = lambda t: t[index].to(dt)
        advs = c(returns) - c(old_v)
        m, s = advs.mean(), advs.std(unbiased=False)
        mean, v = policy._run(c(obs))
        nadv = (advs - m) / (s + 1e-8)
        loss = P2.ppo_loss(diag_gaussian_neglogp(c(actions), mean, policy.logstd), v, diag_gaussian_entropy(policy.logstd, mean), None, nadv,
                           c(returns), c(old_nlp), c(old_v), 0.2, 0.01, 0.5)[0]
        params = [q for q in policy.sb_parameters() if q is not policy.q.w and q is not policy.q.b]
        return torch.autograd.grad(loss, params), params, torch.stack([m, s]).to(torch.float32)

    g32, params, stats_t = eager(pol, torch.float32)
    g64, _, _ = eager(copy.deepcopy(pol).double(), torch.float64)
    res = {}
    for prec in ("f32", "bf16x3"):
        P2.MLP_PRECISION = prec
        _l, _s, grads = P2.mlp_ppo_grads(pol, obs, actions, returns, old_v, old_nlp, stats_t, 0.2, 0.01, 0.5, index=index)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            P2.mlp_ppo_grads(pol, obs, actions, returns, old_v, old_nlp, stats_t, 0.2, 0.01, 0.5, index=index, want_loss=False)
        ev[1].record()
        torch.cuda.synchronize()
        res[prec] = (grads, 1e3 * ev[0].elapsed_time(ev[1]) / 5)
    print("n = %d; both networks' kernels + row sums: f32 %.1f us, bf16x3 %.1f us" % (n, res["f32"][1], res["bf16x3"][1]))
    worst = {"f32": 0.0, "bf16x3": 0.0, "eager": 0.0}
    for q, a32, a64 in zip(params, g32, g64):
        scale = float(a64.abs().max())
        e = {k: float((res[k][0][q].double() - a64).abs().max()) for k in ("f32", "bf16x3")}
        e["eager"] = float((a32.double() - a64).abs().max())
        for k in worst:
            worst[k] = max(worst[k], e[k] / scale)
        print("%-10s scale %.3e   |kernel - f64|: f32 %.3e  bf16x3 %.3e   eager f32 graph %.3e" % (tuple(q.shape), scale, e["f32"], e["bf16x3"], e["eager"]))
    print("worst error / largest entry of the parameter's gradient: f32 %.2e  bf16x3 %.2e  eager f32 graph %.2e" % (worst["f32"], worst["bf16x3"], worst["eager"]))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200000)
