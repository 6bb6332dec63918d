"""Extract the eight actor tensors of a PPO2 checkpoint of this build into an .npz fixture (same layout as
tests/golden/actor_bp5_155.npz).   usage: python tools/export_actor_fixture.py ckpt.pkl out.npz"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from high_speed_quadrupedal_locomotion_by_irrl_amd.checkpoint import read_checkpoint

_, params = read_checkpoint(sys.argv[1])
params = [np.asarray(p, np.float32) for p in params]
out = {n: params[i] for i, n in enumerate(["wx0", "wh0", "b0", "wx1", "wh1", "b1"])}
out["pi_w"], out["pi_b"] = params[14], params[15]
np.savez_compressed(sys.argv[2], **out)
print("wrote", sys.argv[2], {k: v.shape for k, v in out.items()})
