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
