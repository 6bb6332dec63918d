from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2, Runner  # noqa: F401
