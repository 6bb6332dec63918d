#!/usr/bin/env python3
"""Error of the LSTM sequence kernels (forward + BPTT of one SBLstm layer) against float64 autograd of the eager definition, for each arithmetic of
lstm_fused.PRECISION -- "f32" (exact-f32 MFMA), "bf16x3", "bf16x6" (bf16 matrix cores with compensated operand splits, csrc/lstm_bf16.hpp) -- and
for the eager f32 graph itself; per tensor, relative to the tensor's largest entry.  Also the kernels' time.
    python tools/lstm_grad_error.py [T] [N] [n_in]      (defaults 256 1024 35)"""
import copy
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from high_speed_quadrupedal_locomotion_by_irrl_amd import lstm_fused
from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import SBLstm


def main(T, N, n_in, hid=48):
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

    SBLstm.use_fused = False
    l64 = copy.deepcopy(layer).double()
    ref = run(l64, x.double(), state.double(), masks.double(), wgt.double())
    res = {"eager f32": run(layer, x, state, masks, wgt)}
    SBLstm.use_fused = True
    times = {}
    for prec in ("f32", "bf16x3", "bf16x6"):
        lstm_fused.PRECISION = prec
        run(layer, x, state, masks, wgt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = run(layer, x, state, masks, wgt)
        torch.cuda.synchronize()
        times[prec] = (time.perf_counter() - t0) / 3
        res[prec] = out
    names = ["h_seq", "state", "dx", "dwx", "dwh", "db"]
    print("T %d N %d n_in %d: max |kernel - float64| / max |float64| per tensor" % (T, N, n_in))
    print("%-10s" % "" + "".join("%12s" % n for n in names) + "   fwd+bwd ms")
    for k, out in res.items():
        errs = [float((a.double() - b).abs().max()) / (float(b.abs().max()) + 1e-30) for a, b in zip(out, ref)]
        print("%-10s" % k + "".join("%12.2e" % e for e in errs) + ("   %8.2f" % (1e3 * times[k]) if k in times else ""))


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    main(*(a + [256, 1024, 35][len(a):]))
