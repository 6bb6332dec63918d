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
