#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (tools/gpu.sh pmc): 1 calibration copy of 256 MiB with dword-per-lane accesses (known byte
count) followed by N env steps at 4096 envs.  Run once per counter group (TCC slots do not fit FETCH_SIZE and
WRITE_SIZE in one pass).  Writes the measured library's irrl_version() to gpurun_out/pmc_env_version.txt so that
the summary (tools/pmc_summarize.py) can be tied to the binary it was measured on."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import yaml  # noqa: E402
import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg  # noqa: E402
from high_speed_quadrupedal_locomotion_by_irrl_amd import _lib  # noqa: E402
from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
lib = _lib.load()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "pmc_env_version.txt"), "w") as f:
    f.write(_lib.version() + "\n")
n = 64 * 1024 * 1024  # 256 MiB of floats
src = torch.ones(n, device=dev)
dst = torch.empty(n, device=dev)
torch.cuda.synchronize()
lib.irrl_calib_copy_dword(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(n), C.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, "bp5_imitation.yaml")))["environment"]
cfg["num_envs"] = envs
env = FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg))
env.init()
# bench.py's action stream (Philox, seed 1); the first 100 steps are the landing pre-roll: the per-launch MEDIANS that
# tools/pmc_summarize.py takes over all launches are steady-state values
rows = steps + 100
actions = torch.empty(rows, envs, 12, device=dev)
_lib.check(lib.irrl_bench_actions(1, 0, envs, 0, rows, 0.3, C.c_void_p(actions.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
ob = torch.zeros(envs, 35, device=dev)
rew = torch.zeros(envs, device=dev)
done = torch.zeros(envs, dtype=torch.bool, device=dev)
extra = torch.zeros(envs, 6, device=dev)
for k in range(rows):
    env.step(actions[k], ob, rew, done, extra)
# the multi-step persistent kernel as bench.py times it (--launch persistent: EVERY step's outputs kept in [K, N, .] tables): five launches
# of PERSIST_STEPS steps each from the same steady state
PERSIST_STEPS = 100
ob_rows = torch.zeros(PERSIST_STEPS, envs, 35, device=dev)
rew_rows = torch.zeros(PERSIST_STEPS, envs, device=dev)
done_rows = torch.zeros(PERSIST_STEPS, envs, dtype=torch.bool, device=dev)
extra_rows = torch.zeros(PERSIST_STEPS, envs, 6, device=dev)
for i in range(5):
    env.step_rows(PERSIST_STEPS, actions, (i * PERSIST_STEPS) % rows, ob_rows, rew_rows, done_rows, extra_rows, persistent=True)
torch.cuda.synchronize()
print("ok", float(rew.mean()))
