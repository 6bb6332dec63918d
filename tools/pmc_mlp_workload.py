#!/usr/bin/env python3
"""Workload of the PMC passes of the MlpPolicy gradient kernels (tools/gpu.sh pmcmlp): a rollout-shaped batch (4096 envs x 750 steps = 3.07 M
samples of random data), and for each of the two ways the kernels read their samples -- the five arrays through the shuffled index
(`irrl_mlp_ppo_bf16_kernel<kind, false>`) and the packed 256-byte records (`<kind, true>`, built by `irrl_mlp_pack_kernel`) -- one epoch
= 4 minibatches of 768 k samples of both networks.  Same index, same data: the two differ only in what they fetch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from high_speed_quadrupedal_locomotion_by_irrl_amd import ppo2 as P2  # noqa: E402
from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import MlpPolicy, diag_gaussian_neglogp  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(1)
n = 4096 * 750
pol = MlpPolicy().to(dev)
flat = P2.FlatParams(pol)
g = torch.Generator(device=dev); g.manual_seed(3)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
obs, actions, returns, old_v = rn(n, 35), 0.5 * rn(n, 12), rn(n), rn(n)
with torch.no_grad():
    old_nlp = diag_gaussian_neglogp(actions, pol._run(obs)[0], pol.logstd) + 0.3 * rn(n)
perm = torch.randperm(n, device=dev, generator=g)
stats = torch.tensor([0.0, 1.4], device=dev)
rec = P2.mlp_pack_records(obs, actions, returns, old_v, old_nlp)
bs = n // 4
for use_rec in (False, True):
    for k in range(4):
        idx = perm[k * bs:(k + 1) * bs].contiguous()
        P2.mlp_ppo_grads_flat(pol, flat, obs, actions, returns, old_v, old_nlp, stats, 0.2, 0.0, 0.5, idx, rec=rec if use_rec else None)
torch.cuda.synchronize()
print("ok", float(flat.grad.abs().max()))
