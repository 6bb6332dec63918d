"""PCIe-inclusive rate of the numpy (reference-style, host-buffer) boundary: RaisimGymVecEnv.step on numpy arrays at 4096 envs
(H2D of the actions + kernel + one packed D2H + copies).  Never the headline `value`."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, yaml, torch  # noqa: F401
import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
from high_speed_quadrupedal_locomotion_by_irrl_amd.vec_env import RaisimGymVecEnv
cfg = yaml.safe_load(open(os.path.join(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, "bp5_imitation.yaml")))["environment"]
env = RaisimGymVecEnv(FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg)))
a = np.clip(0.3 * np.random.RandomState(0).normal(size=(4096, 12)), -1, 1).astype(np.float32)
for _ in range(1000): env.step(a)
t = time.perf_counter()
for _ in range(1000): env.step(a)
dt = time.perf_counter() - t
print("host numpy path (RaisimGymVecEnv.step, H2D+kernel+D2H+copies): %.1f us/step, %.2f M env-steps/s" % (dt * 1e3, 4096 * 1000 / dt / 1e6))
