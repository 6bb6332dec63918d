#!/usr/bin/env python3
"""Time of the bf16 forward sequence kernel of one LSTM layer alone and of two launches side by side on two streams (the actor's and the critic's
layer of the update):
    python tools/lstm_fwd_time.py [T] [N] [n_in]      (defaults 750 4096 48)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
T, N, n_in = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 750), (2, 4096), (3, 48)))
hid = 48
dev = torch.device("cuda")
lib = _lib.load()
p = lambda t: C.c_void_p(t.data_ptr())
def tensors():
    return dict(x=torch.randn(T, N, n_in, device=dev), wx=torch.randn(n_in, hid, 4, device=dev) * 0.1, b=torch.randn(hid, 4, device=dev) * 0.1, wh=torch.randn(hid, hid, 4, device=dev) * 0.1,
                masks=(torch.rand(T, N, device=dev) < 0.01).float(), s0=torch.randn(N, 2 * hid, device=dev) * 0.5, gates=torch.empty(T, N, hid, 4, device=dev),
                c=torch.empty(T, N, hid, device=dev), h=torch.empty(T, N, hid, device=dev), so=torch.empty(N, 2 * hid, device=dev))
A, B = tensors(), tensors()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def launch(t, flag, stream):
    rc = lib.irrl_lstm_seq_forward_bf16(2 | flag, hid, T, N, n_in, p(t["x"]), p(t["wx"]), p(t["b"]), p(t["wh"]), p(t["masks"]), p(t["s0"]), p(t["gates"]), p(t["c"]), p(t["h"]), p(t["so"]),
                                        C.c_void_p(stream.cuda_stream))
    assert rc == 0
for name, flag in (("lstm_seq_fwd_bf16_kernel<2>", 0), ("lstm_seq_fwd_bf16_kernel<2>", 0)):
    for pair in (False, True):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(3):
            if rep == 1:
                torch.cuda.synchronize(); e0.record(sa)
            launch(A, flag, sa)
            if pair:
                launch(B, flag, sb)
        torch.cuda.synchronize()
        # wall clock of the two timed repetitions through a device-wide event pair
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0.record()
        for rep in range(4):
            launch(A, flag, sa)
            if pair:
                launch(B, flag, sb)
        sa.synchronize(); sb.synchronize(); t1.record(); torch.cuda.synchronize()
        import time
        w0 = time.perf_counter()
        for rep in range(4):
            launch(A, flag, sa)
            if pair:
                launch(B, flag, sb)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - w0) / 4
        print("%s %-22s %8.1f us per %s (T %d, N %d, n_in %d)" % (name, "two side by side:" if pair else "one alone:", 1e6 * wall, "pair" if pair else "launch", T, N, n_in), flush=True)
