#!/usr/bin/env python3
"""Where does the multi-step kernel (lane context carried in registers) leave the per-step kernel?  Pool state after K steps both ways, K = 1, 2, 3, ...:
the first K at which they differ, which state words, by how much."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import load_env_cfg
from hip_env import HipVecEnv
names = dict(GC=0, GV=19, PTL=37, TQL=49, TQ=61, JR=73, JRL=85, JDR=97, EER=109, CMD=121, CMDF=124, T0=127, FRAME=128, EP=129, UPH=130, CONTACT=131, LAMW=135, INC=147,
             MAT=151, MASS=154, COM=167, DZ=206, OB=207, OBL=242, SPH=277, END=286)
keys = sorted(names.items(), key=lambda x: x[1])
def field(i):
    for (k, v), (k2, v2) in zip(keys, keys[1:]):
        if v <= i < v2:
            return "%s[%d]" % (k, i - v)
    return str(i)
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "bp5_imitation.yaml"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pre = int(sys.argv[3]) if len(sys.argv) > 3 else 60
g = torch.Generator(device="cuda").manual_seed(5)
table = (0.5 * torch.randn(64, n, 12, device="cuda", generator=g)).clamp(-1, 1)
for K in (1, 2, 3, 5, 10):
    a, b = HipVecEnv(load_env_cfg(cfg_name, num_envs=n)), HipVecEnv(load_env_cfg(cfg_name, num_envs=n))
    outs = [(torch.zeros(n, 35, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, dtype=torch.bool, device="cuda"), torch.zeros(n, 6, device="cuda")) for _ in range(2)]
    for env, o in ((a, outs[0]), (b, outs[1])):      # same pre-roll both ways (per-step launches): robots on the ground
        env.impl.step_rows(pre, table, 0, *o)
    a.impl.step_rows(K, table, pre, *outs[0], persistent=True)
    b.impl.step_rows(K, table, pre, *outs[1])
    torch.cuda.synchronize()
    sa, sb = a.get_state(), b.get_state()
    d = np.argwhere(sa != sb)
    print("K = %d: %d differing words in %d envs" % (K, len(d), len(set(d[:, 0])) if len(d) else 0))
    seen = {}
    for e, i in d:
        f = field(int(i)).split("[")[0]
        seen.setdefault(f, []).append(abs(sa[e, i] - sb[e, i]))
    for f, v in seen.items():
        print("    %-8s %5d words, max |diff| %.3e" % (f, len(v), max(v)))
    ob_d = (outs[0][0] != outs[1][0])
    print("    last step's outputs: ob %d words differ (columns %s), reward %d, extra %d (columns %s)" % (
        ob_d.sum().item(), sorted(set(ob_d.nonzero()[:, 1].tolist())), (outs[0][1] != outs[1][1]).sum().item(),
        (outs[0][3] != outs[1][3]).sum().item(), sorted(set((outs[0][3] != outs[1][3]).nonzero()[:, 1].tolist()))))
