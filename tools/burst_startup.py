#!/usr/bin/env python3
"""Why is a 20-step timed window (the driver's --steps 20) slower per step than a 3000-step one?  Replays the same 20-step
hipGraph ten times, each bracketed by torch.cuda.synchronize() like bench.py's timed region, and prints wall-clock and
HIP-event microseconds per step of every burst; then the same with the synchronisation replaced by back-to-back replays."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml
import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib
from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
n, K = 4096, 20
cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, "bp5_imitation.yaml")))["environment"]
cfg["num_envs"] = n
dev = torch.device("cuda", 0)
lib = _lib.load()
env = FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg)); env.init()
rows = 400 + K
actions = torch.empty(rows, n, 12, device=dev)
_lib.check(lib.irrl_bench_actions(1, 0, n, 0, rows, 0.3, C.c_void_p(actions.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
ob, rew = torch.zeros(n, 35, device=dev), torch.zeros(n, device=dev)
done, extra = torch.zeros(n, dtype=torch.bool, device=dev), torch.zeros(n, 6, device=dev)
for k in range(400):
    env.step(actions[k], ob, rew, done, extra)
torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream(dev))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    for k in range(K):
        env.step(actions[400 + k], ob, rew, done, extra)
torch.cuda.synchronize()
out = []
for rep in range(10):
    torch.cuda.synchronize()
    time.sleep(0.002 if rep % 2 else 0.0)          # odd bursts: the GPU idles 2 ms first
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out.append((round(1e6 * dt / K, 2), round(1e3 * e0.elapsed_time(e1) / K, 2)))
print("synchronised bursts of %d steps (wall us/step, event us/step); odd ones after 2 ms of idle:" % K, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for rep in range(10):
    g.replay()
e1.record(); torch.cuda.synchronize()
print("ten bursts back to back: %.2f us/step" % (1e3 * e0.elapsed_time(e1) / (10 * K)))
