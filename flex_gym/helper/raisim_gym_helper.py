from high_speed_quadrupedal_locomotion_by_irrl_amd.helper import ConfigurationSaver, TensorboardLauncher  # noqa: F401
