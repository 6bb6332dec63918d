#!/usr/bin/env python3
"""Where a step of the backward sequence kernel goes (diagnostic build: tools/build_variants.py bwd=-DIRRL_PROFILE_BWD): 100 MHz
stamps of wave 0 of the middle workgroup, summed over the T steps, and the helper wave's time at the barrier vs. at work.
    IRRL_ENV_LIB=.../libirrl_env_bwd.so python tools/lstm_bwd_phases.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
lib = _lib.load()
dev = torch.device("cuda")
T, N, hid = 750, 4096, 48
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for n_in, need_dx in ((48, True), (35, False)):
    g = torch.Generator(device="cuda").manual_seed(1)
    r = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.3
    gates = torch.sigmoid(r(T, N, hid, 4)); cseq = r(T, N, hid); hseq = r(T, N, hid); x = r(T, N, n_in)
    masks = (torch.rand(T, N, device=dev, generator=g) < 0.002).float(); state0 = r(N, 2 * hid); dh = r(T, N, hid)
    wh = r(hid, hid, 4); wx = r(n_in, hid, 4)
    nb = N // 16
    dx = torch.empty(T, N, n_in, device=dev) if need_dx else None
    dwx = torch.empty(nb, n_in, 4 * hid, device=dev); dwh = torch.empty(nb, hid, 4 * hid, device=dev)
    db = torch.zeros(nb * 4 + 1, 4 * hid, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = lib.irrl_lstm_seq_backward_x(hid, T, N, n_in, p(gates), p(cseq), p(hseq), p(x), p(masks), p(state0), p(dh), p(wh), p(wx), p(dx), p(dwx), p(dwh), p(db), s)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert rc == 0
    v = db[nb * 4, :10].cpu().numpy() * 0.01 / T      # us per step
    names = ["stage + prefetch issue + gate arithmetic + dz tile", "recurrence (+dx) MFMAs, partials to LDS", "barrier", "partial sums -> dh_prev (dx rows)", "weight-gradient MFMAs"]
    print("n_in %d dx %s: kernel %.2f ms = %.2f us per step" % (n_in, need_dx, dt * 1e3, dt * 1e6 / T))
    for nme, val in zip(names, v[:5]):
        print("   %-52s %6.3f us" % (nme, val))
    print("   sum %.3f us;   helper wave: %.3f us at the barrier, %.3f us at work" % (v[:5].sum(), v[8], v[9]))
