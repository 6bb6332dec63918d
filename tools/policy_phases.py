#!/usr/bin/env python3
"""Phase time stamps of ONE workgroup of the single-launch LSTM policy step (diagnostic build: tools/build_variants.py
pol=-DIRRL_PROFILE_POLICY; the kernel then writes 100 MHz stamps of its phases over neglogp[0:7]).
    IRRL_ENV_LIB=.../libirrl_env_pol.so python tools/policy_phases.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy
dev = torch.device("cuda")
N = 4096
pol = CustomLSTMPolicy().to(dev)
obs = torch.randn(N, 35, device=dev)
st = torch.randn(N, 384, device=dev) * 0.3
dones = torch.zeros(N, dtype=torch.bool, device=dev)
rows = []
for k in range(30):
    out = pol.fused_step(obs, st, dones, rng=(1, k), states_out=st)
    torch.cuda.synchronize()
    rows.append(out[3][N // 2:N // 2 + 7].cpu().numpy() * 0.01)
t = np.median(np.array(rows[5:]), axis=0)
names = ["start", "loads issued", "L0 + recurrent L1 MFMAs (loads landed)", "barrier", "L0 cell + L1 input MFMAs", "L1 cell", "heads / sample / rows"]
for n, a, b in zip(names[1:], t[:-1], t[1:]):
    print("%-45s %6.2f us (at %.2f)" % (n, b - a, b))
