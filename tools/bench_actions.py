"""numpy twin of the benchmark's synthetic action stream (csrc/irrl_env_abi.hip `irrl_bench_actions_kernel`):
a = clip(sigma * N(0,1), -1, 1) with N(0,1) by Box-Muller on Philox4x32-10 uniforms, key (seed, 'ACT1'), counter
(global env id, step, block j, 0); block j yields actions 4j .. 4j+3.  Used by bench.py's cpu_baseline leg (the CPU
oracle is driven by the same actions as the GPU) and by the tests that pin the kernel to it."""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """vectorised Philox4x32-10; all arguments uint32 arrays (broadcastable) -> four uint32 arrays"""
    c0, c1, c2, c3 = (np.asarray(c, np.uint32) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * _M0
            p1 = c2.astype(np.uint64) * _M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
            n0 = hi1 ^ c1 ^ k0
            n2 = hi0 ^ c3 ^ k1
            c0, c1, c2, c3 = n0, lo1, n2, lo0
            k0 = np.uint32(k0 + _W0)
            k1 = np.uint32(k1 + _W1)
    return c0, c1, c2, c3


def bench_actions(seed, env0, n_envs, step0, n_steps, sigma=0.3):
    """-> float32 [n_steps, n_envs, 12]; same values as the device kernel up to the libm rounding of log / sqrt / sincos"""
    s = np.arange(step0, step0 + n_steps, dtype=np.int64).astype(np.uint32)[:, None, None]
    e = (np.arange(env0, env0 + n_envs, dtype=np.int64).astype(np.uint32))[None, :, None]
    j = np.arange(3, dtype=np.uint32)[None, None, :]
    r = philox4x32_10(e, s, j, np.uint32(0), seed, 0x41435431)
    sc = np.float32(1.0 / 16777216.0)
    u = [(x >> np.uint32(8)).astype(np.float32) * sc for x in r]
    ra = np.sqrt(np.float32(-2.0) * np.log(np.float32(1.0) - u[0]))
    rb = np.sqrt(np.float32(-2.0) * np.log(np.float32(1.0) - u[2]))
    aa = np.float32(6.283185307179586) * u[1]
    ab = np.float32(6.283185307179586) * u[3]
    z = np.stack([ra * np.cos(aa), ra * np.sin(aa), rb * np.cos(ab), rb * np.sin(ab)], axis=-1)   # [S, E, 3, 4]
    a = np.clip(np.float32(sigma) * z, -1.0, 1.0).astype(np.float32)
    return a.reshape(n_steps, n_envs, 12)
