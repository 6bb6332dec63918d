"""Drop-in for the reference's compiled module `_flexible_robot` (raisim_gym.cpp:14: PYBIND11_MODULE):
`from _flexible_robot import FlexibleGymEnv` (run_bp_v5.py:13) resolves to the MI355X engine."""
from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv  # noqa: F401
