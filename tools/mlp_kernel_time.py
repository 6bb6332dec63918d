"""Time of the MlpPolicy gradient kernels alone (irrl_mlp_ppo_grads, one launch per network) on a shuffled minibatch:
python tools/mlp_kernel_time.py [n] [rows]      (IRRL_ENV_LIB=<path> selects an A/B build; default: 768000 of 3072000 rows; HIP events on the launch stream)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import MlpPolicy


def main(n, rows):
    lib = _lib.load()      # IRRL_ENV_LIB=<path> selects an A/B build of the same ABI
    dev = torch.device("cuda")
    torch.manual_seed(0)
    pol = MlpPolicy().to(dev)
    obs, act = torch.randn(rows, 35, device=dev), torch.randn(rows, 12, device=dev)
    ret, val, nlp = torch.randn(rows, device=dev), torch.randn(rows, device=dev), torch.randn(rows, device=dev) + 12.0
    idx = torch.randperm(rows, device=dev)[:n].contiguous()
    stats = torch.tensor([0.0, 1.0], device=dev)
    P = lib.irrl_mlp_ppo_partial_len()
    part = torch.empty(2, 256, P, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    nets = ((0, pol.pi_fc, pol.pi), (1, pol.vf_fc, pol.vf))

    def launch(kind, fc, head, index):
        rc = entry(kind, n, p(index) if index is not None else None, 35, 64, 12, p(obs), p(act), p(ret), p(val), p(nlp), p(fc[0].w), p(fc[0].b),
                                    p(fc[1].w), p(fc[1].b), p(head.w), p(head.b), p(pol.logstd), p(stats), 0.2, 0.5, p(part[kind]), 256, st)
        assert rc == 0
    for prec, label, index in [(pr, lb, ix) for pr in ("f32", "bf16x3") for lb, ix in (("indexed", idx), ("rows 0..n-1", None))]:
        entry = lib.irrl_mlp_ppo_grads if prec == "f32" else lib.irrl_mlp_ppo_grads_bf16
        for kind, fc, head in nets:
            for _ in range(3):
                launch(kind, fc, head, index)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                launch(kind, fc, head, index)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100.0
            tiles = (n + 15) // 16 / 1024.0
            print("%-7s %-12s kind %d: %.1f us per launch, %.2f us per 16-sample tile per wave" % (prec, label, kind, us, us / tiles))


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 768000, int(a[1]) if len(a) > 1 else 3072000)
