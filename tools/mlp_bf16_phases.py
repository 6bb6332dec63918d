"""Where a 16-sample tile of the bf16 MlpPolicy gradient kernel goes (csrc/mlp_bf16.hpp), per wave, averaged over the tiles.
Needs the diagnostic build (tools/build_variants.py mbprof=-DIRRL_MB_PROFILE):
    IRRL_ENV_LIB=.../libirrl_env_mbprof.so python tools/mlp_bf16_phases.py [n]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import MlpPolicy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 768000
rows = 4 * n
lib = _lib.load()
dev = torch.device("cuda")
torch.manual_seed(0)
pol = MlpPolicy().to(dev)
obs, act = torch.randn(rows, 35, device=dev), torch.randn(rows, 12, device=dev)
ret, val, nlp = torch.randn(rows, device=dev), torch.randn(rows, device=dev), torch.randn(rows, device=dev) + 12.0
idx = torch.randperm(rows, device=dev)[:n].contiguous()
stats = torch.tensor([0.0, 1.0], device=dev)
P = lib.irrl_mlp_ppo_partial_len()
part = torch.zeros(2, 256, P, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
names = ["observations + layer 1", "layer 2", "head + loss", "d W3, d h2, d z2", "d W2, d h1, d z1", "d W1 (+ next tile's rows arriving)"]
if os.environ.get("IRRL_MLP_WAVES", "8") != "4":      # the producer waves of csrc/mlp_bf16_pc.hpp
    names = ["observations + layer 1", "waiting at B", "x, h1 images + layer 2", "head + loss", "d h2, d z2", "d h1, d z1", "waiting at A", "next tile's rows arriving"]
for kind, fc, head in ((0, pol.pi_fc, pol.pi), (1, pol.vf_fc, pol.vf)):
    for _ in range(2):
        rc = lib.irrl_mlp_ppo_grads_bf16(kind, n, p(idx), 35, 64, 12, p(obs), p(act), p(ret), p(val), p(nlp), p(fc[0].w), p(fc[0].b), p(fc[1].w), p(fc[1].b),
                                         p(head.w), p(head.b), p(pol.logstd), p(stats), 0.2, 0.5, p(part[kind]), 256, st)
        assert rc == 0
    torch.cuda.synchronize()
    tiles_per_wave = (n + 15) // 16 / 1024.0
    ph = part[kind][:, 4:4 + len(names)].double().sum(0).cpu().numpy() * 0.01 / 1024.0 / tiles_per_wave     # us per tile per wave
    print("kind %d: %.2f tiles per wave; us per tile: %s; sum %.2f" % (kind, tiles_per_wave, ", ".join("%s %.2f" % (nm, x) for nm, x in zip(names, ph)), ph.sum()))
