#!/usr/bin/env python3
"""Where a step of the forward sequence kernel goes (diagnostic build: tools/build_variants.py fwd=-DIRRL_PROFILE_FWD): 100 MHz stamps
of wave 0 of the middle workgroup, summed over the T steps, and the helper wave's time at the barrier vs. at work; one kernel alone
on the chip and two side by side on two streams (what the update does with the actor's and the critic's stacks).
    IRRL_ENV_LIB=.../libirrl_env_fwd.so python tools/lstm_fwd_phases.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
lib = _lib.load()
dev = torch.device("cuda")
T, N, hid = 750, 4096, 48
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
names = ["prefetch issue + helper's tile + own share of x wx", "recurrent MFMAs", "cell, stores, h to LDS", "barrier"]
for n_in in (35, 48):
    g = torch.Generator(device="cuda").manual_seed(1)
    r = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.3
    def bufs():
        return dict(x=r(T, N, n_in), wx=r(n_in, hid, 4), b=r(hid, 4), wh=r(hid, hid, 4), masks=(torch.rand(T, N, device=dev, generator=g) < 0.002).float(),
                    s0=r(N, 2 * hid), gates=torch.empty(T, N, hid, 4, device=dev), c=torch.empty(T, N, hid, device=dev), h=torch.empty(T, N, hid, device=dev),
                    so=torch.zeros(N + 1, 2 * hid, device=dev))
    A, B = bufs(), bufs()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    def launch(d, stream):
        return lib.irrl_lstm_seq_forward_x(hid, T, N, n_in, p(d["x"]), p(d["wx"]), p(d["b"]), p(d["wh"]), p(d["masks"]), p(d["s0"]), p(d["gates"]), p(d["c"]),
                                           p(d["h"]), p(d["so"]), C.c_void_p(stream.cuda_stream))
    for mode in ("alone", "two side by side"):
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            assert launch(A, sa) == 0
            if mode != "alone":
                assert launch(B, sb) == 0
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        v = A["so"][N, :10].cpu().numpy() * 0.01 / T
        print("n_in %d, %s: %.2f ms = %.2f us per step" % (n_in, mode, dt * 1e3, dt * 1e6 / T))
        for nme, val in zip(names, v[:4]):
            print("   %-52s %6.3f us" % (nme, val))
        print("   sum %.3f us;   helper wave: %.3f us at the barrier, %.3f us at work" % (v[:4].sum(), v[8], v[9]))
