"""ctypes wrapper of the TEST-ONLY host emulation of the env kernels (see emu_main.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np
import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "..", "..", "high_speed_quadrupedal_locomotion_by_irrl_amd", "csrc")
_LIB = {}
STATE_DIM = 288


def build(width=4):
    out = os.path.join(_HERE, "_build", "libirrl_emu%d.so" % width)
    srcs = [os.path.join(_HERE, f) for f in ("emu_main.cpp", "lanes_cpu.hpp")] + \
           [os.path.join(_CSRC, f) for f in ("env_core.hpp", "env_params.h", "irrl_config.hpp", "irrl_state_pool.hpp", "irrl_terrain.hpp", "irrl_csv.hpp")]
    if (not os.path.exists(out)) or os.path.getmtime(out) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DIRRL_EMU_W=%d" % width,
                               "-I" + _CSRC, "-o", out, os.path.join(_HERE, "emu_main.cpp")])
    return out


def lib(width=4):
    if width not in _LIB:
        l = C.CDLL(build(width))
        l.emu_create.restype = C.c_void_p
        l.emu_create.argtypes = [C.c_char_p]
        l.emu_last_error.restype = C.c_char_p
        vp, fp, dp, u8 = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
        for n, a in dict(emu_destroy=[vp], emu_init=[vp], emu_reset=[vp, fp], emu_observe=[vp, fp],
                         emu_step=[vp, fp, fp, fp, u8, fp], emu_probe=[vp, fp, fp], emu_get_state=[vp, dp],
                         emu_set_state=[vp, dp]).items():
            getattr(l, n).restype = None
            getattr(l, n).argtypes = a
        l.emu_steps_carried.restype = None
        l.emu_steps_carried.argtypes = [vp, C.c_int, fp, fp, fp, u8, fp, C.c_int]
        l.emu_num_envs.restype = C.c_int
        l.emu_num_envs.argtypes = [vp]
        _LIB[width] = l
    return _LIB[width]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class EmuVecEnv(object):
    WIDTH = 4   # lanes per robot: 4 = one leg per lane, 16 = four sub-lanes per leg

    def __init__(self, env_cfg, width=None):
        self.l = lib(self.WIDTH if width is None else width)
        env_cfg = dict(env_cfg)
        ref = env_cfg.pop("_ref_table", None)     # reference-trajectory table of a ManualTraj: False config (test plumbing)
        text = yaml.safe_dump(env_cfg, default_flow_style=False)
        self.h = self.l.emu_create(text.encode())
        if not self.h:
            raise RuntimeError(self.l.emu_last_error().decode())
        self.n = self.l.emu_num_envs(self.h)
        if ref is not None:
            ref = np.ascontiguousarray(ref, np.float32)
            self.l.emu_set_ref.restype = C.c_int
            self.l.emu_set_ref.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int]
            assert self.l.emu_set_ref(self.h, _fp(ref), ref.shape[0], ref.shape[1]) == 0
        self.l.emu_init(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.l.emu_destroy(self.h)
            self.h = None

    def reset(self):
        ob = np.zeros((self.n, 35), np.float32)
        self.l.emu_reset(self.h, _fp(ob))
        return ob

    def observe(self):
        ob = np.zeros((self.n, 35), np.float32)
        self.l.emu_observe(self.h, _fp(ob))
        return ob

    def step(self, action):
        action = np.ascontiguousarray(action, np.float32)
        ob = np.zeros((self.n, 35), np.float32)
        rew = np.zeros(self.n, np.float32)
        done = np.zeros(self.n, np.uint8)
        extra = np.zeros((self.n, 6), np.float32)
        self.l.emu_step(self.h, _fp(action), _fp(ob), _fp(rew), done.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(extra))
        return ob, rew, done.astype(bool), extra

    def steps_carried(self, action_rows, poison=False):
        """`len(action_rows)` steps with the lane context carried in registers from step to step (the step loop of the multi-step kernels):
        -> per-step outputs [K, N, .]"""
        action_rows = np.ascontiguousarray(action_rows, np.float32)
        k = action_rows.shape[0]
        ob = np.zeros((k, self.n, 35), np.float32)
        rew = np.zeros((k, self.n), np.float32)
        done = np.zeros((k, self.n), np.uint8)
        extra = np.zeros((k, self.n, 6), np.float32)
        self.l.emu_steps_carried(self.h, k, _fp(action_rows), _fp(ob), _fp(rew), done.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(extra), int(poison))
        return ob, rew, done.astype(bool), extra

    def probe(self):
        minv = np.zeros((self.n, 324), np.float32)
        nl = np.zeros((self.n, 18), np.float32)
        self.l.emu_probe(self.h, _fp(minv), _fp(nl))
        return minv, nl

    def heightfield(self):
        self.l.emu_heightfield.restype = C.c_int
        self.l.emu_heightfield.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        if not self.l.emu_heightfield(self.h, None):
            return None
        out = np.zeros((5000, 500), np.float32)
        self.l.emu_heightfield(self.h, _fp(out))
        return out

    def get_state(self):
        out = np.zeros((self.n, STATE_DIM), np.float64)
        self.l.emu_get_state(self.h, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def set_state(self, st):
        st = np.ascontiguousarray(st, np.float64)
        self.l.emu_set_state(self.h, st.ctypes.data_as(C.POINTER(C.c_double)))


class EmuVecEnv16(EmuVecEnv):
    """The 16-lanes-per-robot layout of the same kernel source."""
    WIDTH = 16
