"""ctypes front-end of the CPU oracle (oracle/irrl_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py.  Nothing under ``high_speed_quadrupedal_locomotion_by_irrl_amd``
imports this module; the product path is the HIP library and fails loudly without it.

Parity status: task math pinned by tests/golden (generated from the reference's importable Python
twins); rigid-body physics + contact is the build's own formulation -> "parity unpinned" against
RaiSim (closed source, absent), pinned against first-principles invariants instead.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
STATE_DIM = 288  # == ORC_STATE_DIM (irrl_oracle.h)


class OrcCfg(C.Structure):
    """Mirror of ``struct orc_cfg`` (irrl_oracle.h)."""
    _fields_ = [
        ("num_envs", C.c_int32), ("num_threads", C.c_int32),
        ("simulation_dt", C.c_double), ("control_dt", C.c_double), ("seedd", C.c_int32),
        ("abad", C.c_double), ("period", C.c_double), ("lam", C.c_double), ("stand_height", C.c_double),
        ("up_height", C.c_double), ("down_height", C.c_double), ("gait_step", C.c_double),
        ("Vx", C.c_double), ("Vy", C.c_double), ("Omega", C.c_double), ("LeanFront", C.c_double),
        ("LeanHind", C.c_double),
        ("Terrain", C.c_int32), ("Manual", C.c_int32), ("Crutial", C.c_int32), ("Filter", C.c_int32),
        ("Camera", C.c_int32), ("StochasticDynamics", C.c_int32), ("HeightVariable", C.c_int32),
        ("TimeBasedContact", C.c_int32), ("ManualTraj", C.c_int32), ("MotorDynamics", C.c_int32),
        ("ObsFilter", C.c_int32), ("WILDCAT", C.c_int32), ("ForceDisturbance", C.c_int32),
        ("Convert2Torque", C.c_int32),
        ("terminalRewardCoeff", C.c_double), ("EndEffectorRewardCoeff", C.c_double),
        ("BodyPosRewardCoeff", C.c_double), ("BodyAttitudeRewardCoeff", C.c_double),
        ("JointRewardCoeff", C.c_double), ("VelRewardCoeff", C.c_double), ("TorqueCoeff", C.c_double),
        ("ContactCoeff", C.c_double),
        ("Stiffness", C.c_double), ("Stiffness_Low", C.c_double), ("AbadRatio", C.c_double),
        ("Damping", C.c_double), ("Freq", C.c_double), ("max_time", C.c_double), ("CubeNum", C.c_int32),
        ("FPS", C.c_double), ("ActionNoise", C.c_double), ("ObsNoise", C.c_double), ("GaitType", C.c_int32),
        ("MotorMaxTorque", C.c_double), ("MotorCriticalSpeed", C.c_double), ("MotorMaxSpeed", C.c_double),
        ("ContactIterations", C.c_int32), ("SharedNoiseScalar", C.c_int32), ("RandomizePerEpisode", C.c_int32), ("EnvIdOffset", C.c_int32),
        ("ContactTolerance", C.c_double), ("ContactSolver", C.c_int32), ("ContactExit", C.c_int32),
    ]


_EXT_DEFAULTS = {"ContactIterations": 6, "SharedNoiseScalar": 1, "RandomizePerEpisode": 0, "EnvIdOffset": 0, "ContactTolerance": 0.0, "ContactSolver": 3, "ContactExit": 1}


def cfg_from_dict(env_cfg):
    """Fill an OrcCfg from the ``environment:`` mapping of a config (every reference key mandatory,
    ENV:1594-1659 / BASE:41-42)."""
    c = OrcCfg()
    for k in env_cfg:   # the build-defined contact keys are a closed set (same check as csrc/irrl_config.hpp)
        if str(k).startswith("Contact") and k not in ("ContactCoeff", "ContactIterations", "ContactTolerance", "ContactSolver", "ContactExit"):
            raise KeyError("unsupported build-defined key cfg[%r]" % k)
    for name, ctype in OrcCfg._fields_:
        if name in env_cfg:
            v = env_cfg[name]
        elif name in _EXT_DEFAULTS:
            v = _EXT_DEFAULTS[name]
        else:
            raise KeyError("Node cfg[%r] doesn't exist" % name)
        setattr(c, name, int(v) if ctype is C.c_int32 else float(v))
    return c


def build(force=False):
    """Compile liborc_f64.so / liborc_f32.so with the committed Makefile."""
    libs = [os.path.join(_HERE, "liborc_f64.so"), os.path.join(_HERE, "liborc_f32.so")]
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("irrl_oracle.c", "irrl_oracle.h", "Makefile"))
    stale = force or any((not os.path.exists(p)) or os.path.getmtime(p) < newest for p in libs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "all"])
    return libs


_LIBS = {}


def _lib(precision="f64"):
    if precision in _LIBS:
        return _LIBS[precision]
    build()
    lib = C.CDLL(os.path.join(_HERE, "liborc_%s.so" % precision))
    fp = C.POINTER(C.c_float)
    dp = C.POINTER(C.c_double)
    u8 = C.POINTER(C.c_uint8)
    vp = C.c_void_p
    lib.orc_create.restype = vp
    lib.orc_create.argtypes = [C.POINTER(OrcCfg)]
    for name, args in {
        "orc_destroy": [vp], "orc_init": [vp], "orc_reset": [vp, fp], "orc_observe": [vp, fp],
        "orc_step": [vp, fp, fp, fp, u8, fp], "orc_is_terminal": [vp, u8], "orc_set_seed": [vp, C.c_int],
        "orc_origin_state": [vp, fp], "orc_reference_state": [vp, fp], "orc_joint_effort": [vp, fp],
        "orc_generalized_force": [vp, fp], "orc_inverse_mass_matrix": [vp, fp], "orc_nonlinear": [vp, fp],
        "orc_set_contact_coeff": [vp, fp], "orc_get_state": [vp, dp], "orc_set_state": [vp, dp],
    }.items():
        getattr(lib, name).restype = None
        getattr(lib, name).argtypes = args
    lib.orc_set_control_dt.restype = None
    lib.orc_set_control_dt.argtypes = [vp, C.c_double]
    lib.orc_num_envs.restype = C.c_int
    lib.orc_num_envs.argtypes = [vp]
    lib.orc_real_bytes.restype = C.c_int
    lib.orc_mean_contact_sweeps.restype = C.c_double
    lib.orc_mean_contact_sweeps.argtypes = [vp]
    d3 = C.c_double * 3
    lib.orc_cubic_bezier.argtypes = [d3, d3, C.c_double, d3]
    lib.orc_solve_contact.argtypes = [C.c_int, C.c_double * 9, d3, d3, C.c_double, C.c_double, d3]
    lib.orc_bezier2.argtypes = [d3, d3, C.c_double, C.c_double, d3]
    lib.orc_gauss.restype = C.c_double
    lib.orc_gauss.argtypes = [C.c_double] * 3
    for n in ("orc_smooth_function", "orc_smooth_function2"):
        getattr(lib, n).restype = C.c_double
        getattr(lib, n).argtypes = [C.c_double] * 3
    lib.orc_sampling_reshape.restype = C.c_double
    lib.orc_sampling_reshape.argtypes = [C.c_double]
    lib.orc_inverse_kinematics.restype = C.c_int
    lib.orc_inverse_kinematics.argtypes = [C.c_double] * 7 + [C.c_int, d3]
    lib.orc_torque_clamp.argtypes = [dp, dp, C.c_double, C.c_double, C.c_double, dp, dp, dp]
    lib.orc_gait_reference.argtypes = [C.POINTER(OrcCfg), d3, C.c_double, C.c_int, dp, dp, dp, dp, dp]
    lib.orc_obs_scaling.argtypes = [C.POINTER(OrcCfg), dp, dp]
    lib.orc_gae.argtypes = [C.c_int, C.c_int, fp, fp, u8, fp, u8, C.c_float, C.c_float, fp, fp]
    lib.orc_mass_matrix_world.argtypes = [dp, dp]
    lib.orc_nonlinear_world.argtypes = [dp, dp, dp]
    lib.orc_toe_kinematics.argtypes = [dp, dp, dp, dp]
    lib.orc_rng_u01.argtypes = [C.c_uint32] * 5 + [dp]
    _LIBS[precision] = lib
    return lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class OracleVecEnv(object):
    """Vector env with the call surface of VEC:127-382, running the CPU oracle."""

    OB_DIM, ACT_DIM, EXTRA_DIM = 35, 12, 6
    EXTRA_NAMES = ["EndEffectorReward(0.15)", "Height_Keep_Reward(0.1)", "base height",
                   "Balance_Keep_Reward(0.1)", "JointReward(0.65)", "VelocityReward(0.2)"]

    def __init__(self, env_cfg, precision="f64"):
        self.lib = _lib(precision)
        env_cfg = dict(env_cfg)
        ref = env_cfg.pop("_ref_table", None)     # reference-trajectory table of a ManualTraj: False config
        self.cfg = cfg_from_dict(env_cfg)
        self.n = self.cfg.num_envs
        self.h = self.lib.orc_create(C.byref(self.cfg))
        if not self.h:
            raise RuntimeError("orc_create refused this configuration")
        if ref is not None:
            ref = np.ascontiguousarray(ref, np.float32)
            self.lib.orc_set_ref.restype = C.c_int
            self.lib.orc_set_ref.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int]
            if self.lib.orc_set_ref(self.h, _fp(ref), ref.shape[0], ref.shape[1]) != 0:
                raise RuntimeError("orc_set_ref refused the table")
        self.lib.orc_init(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_destroy(self.h)
            self.h = None

    def heightfield(self):
        nx, ny = C.c_int(0), C.c_int(0)
        self.lib.orc_heightfield.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        self.lib.orc_heightfield.restype = C.c_int
        if not self.lib.orc_heightfield(self.h, None, C.byref(nx), C.byref(ny)):
            return None
        out = np.zeros((nx.value, ny.value), np.float32)
        self.lib.orc_heightfield(self.h, _fp(out), C.byref(nx), C.byref(ny))
        return out

    def terrain_sample(self, x, y):
        out = (C.c_double * 4)()
        self.lib.orc_terrain_sample.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double * 4]
        self.lib.orc_terrain_sample.restype = None
        self.lib.orc_terrain_sample(self.h, float(x), float(y), out)
        return np.array(out[:])

    def contact_probe(self, env_id):
        """switch the contact-problem capture to env `env_id` (None: read the last captured problem back)"""
        self.lib.orc_set_probe.argtypes = [C.c_void_p, C.c_int]
        self.lib.orc_set_probe.restype = None
        self.lib.orc_set_probe(self.h, int(env_id))

    def contact_problem(self):
        G, cf, n, vs, lam = np.zeros((12, 12)), np.zeros((4, 3)), np.zeros((4, 3)), np.zeros(4), np.zeros((4, 3))
        act = np.zeros(4, np.int32)
        self.lib.orc_get_probe.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 5 + [C.POINTER(C.c_int)]
        self.lib.orc_get_probe.restype = None
        self.lib.orc_get_probe(self.h, _dp(G), _dp(cf), _dp(n), _dp(vs), _dp(lam), act.ctypes.data_as(C.POINTER(C.c_int)))
        return dict(G=G, cfree=cf, n=n, vstar=vs, lam=lam, active=act.astype(bool))

    def sphere_info(self):
        """GetSphereInfo (ENV:1423-1436) for every env: [N,4] = meteorite centre + radius"""
        out = np.zeros((self.n, 4), np.float32)
        self.lib.orc_sphere_info.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.orc_sphere_info(self.h, out.ctypes.data)
        return out

    def sphere_hits(self):
        self.lib.orc_sphere_hits.restype = C.c_long
        self.lib.orc_sphere_hits.argtypes = [C.c_void_p]
        return int(self.lib.orc_sphere_hits(self.h))

    def box_hits(self):
        self.lib.orc_box_hits.restype = C.c_long
        self.lib.orc_box_hits.argtypes = [C.c_void_p]
        return int(self.lib.orc_box_hits(self.h))

    def mean_contact_sweeps(self):
        return self.lib.orc_mean_contact_sweeps(self.h)

    def reset(self):
        ob = np.zeros((self.n, 35), np.float32)
        self.lib.orc_reset(self.h, _fp(ob))
        return ob

    def observe(self):
        ob = np.zeros((self.n, 35), np.float32)
        self.lib.orc_observe(self.h, _fp(ob))
        return ob

    def step(self, action):
        action = np.ascontiguousarray(action, np.float32)
        assert action.shape == (self.n, 12)
        ob = np.zeros((self.n, 35), np.float32)
        rew = np.zeros(self.n, np.float32)
        done = np.zeros(self.n, np.uint8)
        extra = np.zeros((self.n, 6), np.float32)
        self.lib.orc_step(self.h, _fp(action), _fp(ob), _fp(rew), _u8(done), _fp(extra))
        return ob, rew, done.astype(bool), extra

    def set_control_dt(self, dt):
        self.lib.orc_set_control_dt(self.h, float(dt))

    def is_terminal(self):
        done = np.zeros(self.n, np.uint8)
        self.lib.orc_is_terminal(self.h, _u8(done))
        return done.astype(bool)

    def _getter(self, fn, width):
        out = np.zeros((self.n, width), np.float32)
        getattr(self.lib, fn)(self.h, _fp(out))
        return out

    def origin_state(self):
        return self._getter("orc_origin_state", 41)

    def reference_state(self):
        return self._getter("orc_reference_state", 24)

    def joint_effort(self):
        return self._getter("orc_joint_effort", 12)

    def generalized_force(self):
        return self._getter("orc_generalized_force", 18)

    def inverse_mass_matrix(self):
        return self._getter("orc_inverse_mass_matrix", 324)

    def nonlinear(self):
        return self._getter("orc_nonlinear", 18)

    def set_contact_coeff(self, coeff):
        coeff = np.ascontiguousarray(coeff, np.float32)
        self.lib.orc_set_contact_coeff(self.h, _fp(coeff))

    def get_state(self):
        out = np.zeros((self.n, STATE_DIM), np.float64)
        self.lib.orc_get_state(self.h, _dp(out))
        return out

    def set_state(self, state):
        state = np.ascontiguousarray(state, np.float64)
        assert state.shape == (self.n, STATE_DIM)
        self.lib.orc_set_state(self.h, _dp(state))


# ---- flat-state field offsets (irrl_oracle.c "S_*" enum == include/irrl_env.h) ----
S = dict(GC=0, GV=19, PTL=37, TQL=49, TQ=61, JR=73, JRL=85, JDR=97, EER=109, CMD=121, CMDF=124, T0=127,
         FRAME=128, EPISODE=129, UPH=130, CONTACT=131, LAMW=135, INCONTACT=147, MATERIAL=151, MASS=154,
         COM=167, THIGH=206, OB=207, OBLAST=242, SPHERE=277, END=286)


# ---- unit probes ----
def solve_contact_md(G, c, n, vstar, mu, precision="f64", rule=1):
    """the single-contact solve of the oracle (rule 1: the published maximum-dissipation rule; 0: the build's first rule)"""
    out = (C.c_double * 3)()
    _lib(precision).orc_solve_contact(int(rule), (C.c_double * 9)(*np.asarray(G, float).ravel()), (C.c_double * 3)(*c), (C.c_double * 3)(*n),
                                      float(vstar), float(mu), out)
    return np.array(out[:])


def cubic_bezier(p0, pf, s, precision="f64"):
    out = (C.c_double * 3)()
    _lib(precision).orc_cubic_bezier((C.c_double * 3)(*p0), (C.c_double * 3)(*pf), s, out)
    return np.array(out[:])


def bezier2(p0, pf, s, h, precision="f64"):
    out = (C.c_double * 3)()
    _lib(precision).orc_bezier2((C.c_double * 3)(*p0), (C.c_double * 3)(*pf), s, h, out)
    return np.array(out[:])


def gauss(x, w, h, precision="f64"):
    return _lib(precision).orc_gauss(x, w, h)


def smooth_function(p, s, lam, precision="f64"):
    return _lib(precision).orc_smooth_function(p, s, lam)


def smooth_function2(p, s, lam, precision="f64"):
    return _lib(precision).orc_smooth_function2(p, s, lam)


def sampling_reshape(r, precision="f64"):
    return _lib(precision).orc_sampling_reshape(r)


def inverse_kinematics(x, y, z, is_right, theta0=(0.0, 0.0, 0.0), l_hip=0.085, l_thigh=0.209, l_calf=0.2175,
                       precision="f64"):
    max_len = float(np.sqrt(l_hip * l_hip + (l_thigh + l_calf) ** 2))
    th = (C.c_double * 3)(*theta0)
    err = _lib(precision).orc_inverse_kinematics(x, y, z, l_hip, l_thigh, l_calf, max_len, int(is_right), th)
    return np.array(th[:]), err


def torque_clamp(tau, qd, tau_max, w_crit, w_max, precision="f64"):
    tau = np.ascontiguousarray(tau, np.float64)
    qd = np.ascontiguousarray(qd, np.float64)
    out, up, lo = np.zeros(12), np.zeros(12), np.zeros(12)
    _lib(precision).orc_torque_clamp(_dp(tau), _dp(qd), tau_max, w_crit, w_max, _dp(out), _dp(up), _dp(lo))
    return out, up, lo


def gait_reference(env_cfg, cmd_f, t, is_first, joint_ref_last=None, joint_ref=None, up_height=None,
                   precision="f64"):
    cfg = cfg_from_dict(env_cfg)
    jrl = np.zeros(12) if joint_ref_last is None else np.array(joint_ref_last, np.float64)
    jr = np.zeros(12) if joint_ref is None else np.array(joint_ref, np.float64)
    jdr, ee = np.zeros(12), np.zeros(12)
    uh = C.c_double(env_cfg["up_height"] if up_height is None else up_height)
    _lib(precision).orc_gait_reference(C.byref(cfg), (C.c_double * 3)(*cmd_f), t, int(is_first), _dp(jrl),
                                       _dp(jr), _dp(jdr), _dp(ee), C.byref(uh))
    return dict(jointRefLast=jrl, jointRef=jr, jointDotRef=jdr, eeRef=ee, up_height=uh.value)


def obs_scaling(env_cfg, precision="f64"):
    cfg = cfg_from_dict(env_cfg)
    mean, std = np.zeros(35), np.zeros(35)
    _lib(precision).orc_obs_scaling(C.byref(cfg), _dp(mean), _dp(std))
    return mean, std


def gae(rewards, values, dones, last_values, last_dones, gamma, lam):
    T, N = rewards.shape
    rewards = np.ascontiguousarray(rewards, np.float32)
    values = np.ascontiguousarray(values, np.float32)
    dones = np.ascontiguousarray(dones, np.uint8)
    last_values = np.ascontiguousarray(last_values, np.float32)
    last_dones = np.ascontiguousarray(last_dones, np.uint8)
    adv = np.zeros((T, N), np.float32)
    ret = np.zeros((T, N), np.float32)
    _lib("f64").orc_gae(T, N, _fp(rewards), _fp(values), _u8(dones), _fp(last_values), _u8(last_dones),
                        gamma, lam, _fp(adv), _fp(ret))
    return adv, ret


def mass_matrix_world(gc, precision="f64"):
    gc = np.ascontiguousarray(gc, np.float64)
    M = np.zeros(324)
    _lib(precision).orc_mass_matrix_world(_dp(gc), _dp(M))
    return M.reshape(18, 18)


def nonlinear_world(gc, gv, precision="f64"):
    gc = np.ascontiguousarray(gc, np.float64)
    gv = np.ascontiguousarray(gv, np.float64)
    h = np.zeros(18)
    _lib(precision).orc_nonlinear_world(_dp(gc), _dp(gv), _dp(h))
    return h


def toe_kinematics(gc, gv, precision="f64"):
    gc = np.ascontiguousarray(gc, np.float64)
    gv = np.ascontiguousarray(gv, np.float64)
    pos, vel = np.zeros(12), np.zeros(12)
    _lib(precision).orc_toe_kinematics(_dp(gc), _dp(gv), _dp(pos), _dp(vel))
    return pos.reshape(4, 3), vel.reshape(4, 3)


def rng_u01(seed, env, episode, step, purpose):
    out = np.zeros(4)
    _lib("f64").orc_rng_u01(seed, env, episode, step, purpose, _dp(out))
    return out
