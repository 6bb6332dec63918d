"""Global env ids (VEC:273: envs never interact): every random draw of the path is addressed by (seed, GLOBAL env id,
episode, step, purpose), and a pool's `EnvIdOffset` says which global ids it owns.  So the shards of an N-GPU job (rank r: offset
r * n) ARE the robots r * n .. (r + 1) * n - 1 of the one big pool, bit for bit -- checked here on the oracle and on the kernel
source (emulated lanes); tests/test_gpu_parity.py has the same test on the HIP kernels, tests/test_ppo_distributed.py the PPO-level
one."""
import numpy as np
import pytest

import oracle as O
import parity_lib as PL
from conftest import load_env_cfg
from host_emulation import emu as E


def _run(make, cfg, actions):
    env = make(cfg)
    outs = [env.observe()]
    for a in actions:
        ob, r, d, x = env.step(a)
        outs += [ob.copy(), r.copy(), d.copy(), x.copy()]
    return outs, env.get_state()


@pytest.mark.parametrize("make", [O.OracleVecEnv, E.EmuVecEnv16, E.EmuVecEnv])
@pytest.mark.parametrize("name", ["default_cfg.yaml", "bp5_terrain.yaml"])
def test_two_shards_with_offsets_are_the_big_pool_bit_for_bit(make, name):
    n, split, steps = 8, 3, 40
    rng = np.random.RandomState(2)
    acts = [PL.random_actions(rng, n, 0.5) for _ in range(steps)]
    big, st_big = _run(make, load_env_cfg(name, num_envs=n), acts)
    lo, st_lo = _run(make, load_env_cfg(name, num_envs=split, EnvIdOffset=0), [a[:split].copy() for a in acts])
    hi, st_hi = _run(make, load_env_cfg(name, num_envs=n - split, EnvIdOffset=split), [a[split:].copy() for a in acts])
    for b, l, h in zip(big, lo, hi):
        assert np.array_equal(b[:split], l) and np.array_equal(b[split:], h)
    assert np.array_equal(st_big[:split], st_lo) and np.array_equal(st_big[split:], st_hi)
    # and an offset does change the robots
    other, _ = _run(make, load_env_cfg(name, num_envs=split, EnvIdOffset=100), [a[:split].copy() for a in acts[:2]])
    assert not np.array_equal(other[0], lo[0])


def test_unsupported_build_defined_contact_key_is_an_error_in_both_engines():
    """a contact-solver knob that only one implementation knows must not be ignored silently (the two engines would solve
    different iterations and the parity tooling would report a physics mismatch instead of a configuration error)"""
    cfg = load_env_cfg("default_cfg.yaml", num_envs=2, ContactRelax=0.8)
    with pytest.raises(KeyError, match="ContactRelax"):
        O.OracleVecEnv(cfg)
    with pytest.raises(RuntimeError, match="ContactRelax"):
        E.EmuVecEnv16(cfg)
    with pytest.raises(RuntimeError, match="ContactSolver"):
        E.EmuVecEnv16(load_env_cfg("default_cfg.yaml", num_envs=2, ContactSolver=7))
