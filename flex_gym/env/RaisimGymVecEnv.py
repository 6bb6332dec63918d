from high_speed_quadrupedal_locomotion_by_irrl_amd.vec_env import RaisimGymVecEnv, TorchVecEnv  # noqa: F401
