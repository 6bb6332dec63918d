"""pytest configuration: registers the ``gpu`` marker and puts the repo root / oracle on sys.path.

The oracle (oracle/) is test infrastructure: tests may import it, the product package never does.
"""
import os
import sys

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

RSC = os.path.join(ROOT, "high_speed_quadrupedal_locomotion_by_irrl_amd", "rsc")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_env_cfg(name="bp5_imitation.yaml", **overrides):
    with open(os.path.join(RSC, name)) as f:
        cfg = yaml.safe_load(f)["environment"]
    cfg.update(overrides)
    return cfg


@pytest.fixture
def imitation_cfg():
    return load_env_cfg("bp5_imitation.yaml", num_envs=8)


@pytest.fixture
def train_cfg():
    return load_env_cfg("default_cfg.yaml", num_envs=8)
