"""CPU-side checks of the KERNEL SOURCE (csrc/env_core.hpp) through the host lane emulation
(tests/host_emulation): the same Schur-complement dynamics, quad reductions and masked reset the GPU
runs, compared with the f64 oracle.  The real parity tests (HIP through the C-ABI) are in
test_gpu_parity.py and need an MI355X; these run in the build container."""
import numpy as np
import pytest

import oracle as O
import parity_lib as PL
from conftest import load_env_cfg
from host_emulation import emu as E


def _pair(cfg):
    return O.OracleVecEnv(cfg), E.EmuVecEnv(cfg)


def test_init_matches_oracle_train_cfg():
    PL.check_init(*_pair(load_env_cfg("default_cfg.yaml", num_envs=24)))


def test_init_matches_oracle_imitation_cfg():
    PL.check_init(*_pair(load_env_cfg("bp5_imitation.yaml", num_envs=8)))


def test_dynamics_probe_matches_oracle():
    orc, cand = _pair(load_env_cfg("default_cfg.yaml", num_envs=12))
    rng = np.random.RandomState(0)
    for _ in range(5):
        a = PL.random_actions(rng, 12)
        orc.step(a)
    cand.set_state(PL.f32_round_state(orc.get_state()))
    orc.set_state(PL.f32_round_state(orc.get_state()))
    PL.check_probe(orc, cand)


def test_teacher_forced_train_cfg_with_noise_randomisation_and_resets():
    orc, cand = _pair(load_env_cfg("default_cfg.yaml", num_envs=8))
    worst, n_done = PL.check_teacher_forced(orc, cand, steps=120, force_terminal_every=7)
    assert n_done >= 10  # the masked in-step reset path was exercised


def test_teacher_forced_eval_style_cfg():
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=4, GaitType=0, WILDCAT=False, HeightVariable=True,
                       MotorCriticalSpeed=14.2, MotorMaxSpeed=40, stand_height=0.30, ObsFilter=True, Filter=True,
                       TimeBasedContact=True, ActionNoise=0.1, SharedNoiseScalar=False)
    orc, cand = _pair(cfg)
    PL.check_teacher_forced(orc, cand, steps=60, seed=2, force_terminal_every=9)


def test_free_running_horizons():
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=8)
    PL.check_free_running(O.OracleVecEnv, E.EmuVecEnv, cfg)


def test_invariants_small():
    cand = E.EmuVecEnv(load_env_cfg("default_cfg.yaml", num_envs=16))
    PL.check_invariants(cand, steps=30)


def test_config_parser_matches_pyyaml():
    import yaml
    cfg = load_env_cfg("default_cfg.yaml")
    text = yaml.safe_dump(cfg)
    cand = E.EmuVecEnv(dict(cfg, num_envs=2))
    assert cand.n == 2
    # missing mandatory key -> error naming the key (reference: RSFATAL "Node ... doesn't exist")
    bad = dict(cfg)
    del bad["Stiffness"]
    with pytest.raises(RuntimeError, match="Stiffness"):
        E.EmuVecEnv(bad)
    for key in ("Crutial", "Terrain"):
        with pytest.raises(RuntimeError, match=key):
            E.EmuVecEnv(dict(cfg, **{key: True}))
    assert "seedd" in text
