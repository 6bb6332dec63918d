"""flex_gym/helper/raisim_gym_helper.py counterparts (run_bp_v5.py:12,212-216,250-252)."""
import datetime
import os
import shutil


class ConfigurationSaver(object):
    """Timestamped run directory + copies of the listed files (raisim_gym_helper.py:6-14)."""

    def __init__(self, log_dir, save_items):
        self._data_dir = os.path.join(log_dir, datetime.datetime.now().strftime('%Y-%m-%d-%H-%M-%S'))
        os.makedirs(self._data_dir, exist_ok=True)
        for item in save_items or []:
            if item and os.path.exists(item):
                shutil.copy2(item, os.path.join(self._data_dir, os.path.basename(item)))

    @property
    def data_dir(self):
        return self._data_dir


def TensorboardLauncher(directory_path):
    """The reference spawns tensorboard + a browser (raisim_gym_helper.py:17-29); the headless engine only
    reports where the logs are."""
    print("[IRRL] logs under", directory_path)


class DelayTool(object):
    """Observation / command delay line of the evaluation script (IRRL/script/utils/DelayTool.py:5-21): the first sample
    fills the line, afterwards `input_output(s)` returns the sample from int(delay_time / dt) calls ago.
    With a zero-length line the reference blocks forever on an empty queue; here the sample passes straight through."""

    def __init__(self, dt, delay_time):
        import collections
        self.num = int(delay_time / dt)
        self.Q = collections.deque()
        self.flag_first = True

    def input_output(self, s0):
        if self.num <= 0:
            return s0
        if self.flag_first:
            self.flag_first = False
            for _ in range(self.num):
                self.Q.append(s0)
        res = self.Q.popleft()
        self.Q.append(s0)
        return res


def obs_normalisation(env_cfg):
    """(obs_mean[35], obs_std[35], action_mean[12], action_std[12]) of the environment (Environment.hpp:371-393; the script-side
    copy is IRRL/script/bp5_config.py:19-55): obs = [cmd 3 | phase 2 | joint 12 | joint rate 12 | body z-axis 3 | body omega 3]."""
    import numpy as np
    abad = float(env_cfg["abad"])
    nominal = np.array([-abad, -0.78, 1.57, abad, -0.78, 1.57, -abad, -0.78, 1.57, abad, -0.78, 1.57])
    vx, vy, om = float(env_cfg["Vx"]), float(env_cfg["Vy"]), float(env_cfg["Omega"])
    mean = np.concatenate([[(vx + 0.0) / 2.0, (vy - vy) / 2.0, (om - om) / 2.0], [0.0, 0.0], nominal, np.zeros(12), [0.0, 0.0, 1.0], np.zeros(3)])
    std = np.concatenate([np.ones(3), np.ones(2), np.ones(12), np.tile([5.0, 35.0, 40.0], 4), np.full(3, 0.7), np.full(3, 3.0)])
    return mean, std, nominal.copy(), np.ones(12)
