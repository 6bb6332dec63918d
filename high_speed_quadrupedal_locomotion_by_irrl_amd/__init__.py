"""MI355X-native engine for the env.step() hot path of High_Speed_Quadrupedal_Locomotion_by_IRRL.

Only what the path needs lives here:
  csrc/            hand-written HIP (gfx950) kernels + the C-ABI (include/irrl_env.h)
  flexible_robot   `FlexibleGymEnv`   -- mirror of the reference's pybind class (raisim_gym.cpp:14-46)
  vec_env          `RaisimGymVecEnv`  -- mirror of flex_gym/env/RaisimGymVecEnv.py
  ppo2 / policies  PyTorch-ROCm PPO2 + MLP / LSTM policies on the same device (ppo2.py, run_bp_v5.py:117-193)
  rsc/             resource directory (configs)

The env kernels have no CPU / PyTorch fallback: importing is cheap, but creating an env without the built
``libirrl_env.so`` or without an MI355X raises.
"""
import os

__BLACKPANTHER_V55_RESOURCE_DIRECTORY__ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rsc")
__version__ = "0.1.0"
