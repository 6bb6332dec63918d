"""`FlexibleGymEnv`: Python face of the C-ABI with the method names, argument meaning and in-place
semantics of the reference's pybind11 class (flex_gym/env/raisim_gym.cpp:14-46, bound to
VectorizedEnvironment<ENVIRONMENT>, VectorizedEnvironment.hpp:127-382).

Arguments may be
  * C-contiguous numpy arrays (float32 / bool) -- the reference's contract (Eigen::Ref<RowMajor>, no copy,
    filled in place); goes through the pinned-staging `_host` entry points, or
  * torch CUDA tensors on the env's device -- zero-copy device path, stream-ordered on torch's current
    stream (this is what the on-device PPO rollout uses).
A wrong dtype / layout raises TypeError like pybind11 does.
"""
import ctypes as C

import numpy as np

from . import _lib

_fp = C.POINTER(C.c_float)
_u8 = C.POINTER(C.c_uint8)


def _is_torch_cuda(x):
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


def _np_f32(a, shape, name):
    if not isinstance(a, np.ndarray) or a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"] or tuple(a.shape) != tuple(shape):
        raise TypeError("%s must be a C-contiguous float32 numpy array of shape %s" % (name, tuple(shape)))
    return a.ctypes.data_as(_fp)


def _np_bool(a, shape, name):
    if not isinstance(a, np.ndarray) or a.dtype not in (np.bool_, np.uint8) or not a.flags["C_CONTIGUOUS"] or tuple(a.shape) != tuple(shape):
        raise TypeError("%s must be a C-contiguous bool numpy array of shape %s" % (name, tuple(shape)))
    return a.ctypes.data_as(_u8)


def _dev_ptr(t, shape, dtypes, name):
    import torch
    if tuple(t.shape) != tuple(shape) or not t.is_contiguous() or t.dtype not in dtypes:
        raise TypeError("%s must be a contiguous CUDA tensor of shape %s and dtype in %s" % (name, tuple(shape), dtypes))
    return C.c_void_p(t.data_ptr())


class FlexibleGymEnv(object):
    """FlexibleGymEnv(resource_dir: str, cfg_yaml: str) -- raisim_gym.cpp:16."""

    def __init__(self, resource_dir, cfg, device=None):
        self._lib = _lib.load()
        if device is None:
            device = 0
            try:
                import torch
                if torch.cuda.is_available():
                    device = torch.cuda.current_device()
            except Exception:
                pass
        self._device = int(device)
        self._h = self._lib.irrl_env_create(str(resource_dir).encode(), str(cfg).encode(), self._device)
        if not self._h:
            raise RuntimeError("FlexibleGymEnv: " + _lib.last_error())
        self._n = self._lib.irrl_env_num_envs(self._h)
        # bumped by every setter whose value is a BY-VALUE kernel argument (seed, time steps, reference table): a captured
        # hipGraph of step launches has those arguments frozen, its owner compares epochs and re-captures (ppo2.Runner)
        self.params_epoch = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.irrl_env_destroy(h)

    # -- helpers --
    def _sync_stream(self):
        import torch
        self._lib.irrl_env_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self._device).cuda_stream))

    @property
    def device_index(self):
        return self._device

    @property
    def lanes_per_robot(self):
        return self._lib.irrl_env_lanes_per_robot(self._h)

    @property
    def waves_per_simd(self):
        return self._lib.irrl_env_waves_per_simd(self._h)

    # -- raisim_gym.cpp:17-46 --
    def init(self):
        _lib.check(self._lib.irrl_env_init(self._h))

    def getExtraInfoNames(self):
        return [self._lib.irrl_env_extra_name(self._h, j).decode() for j in range(self._lib.irrl_env_extra_dim(self._h))]

    def reset(self, ob):
        if _is_torch_cuda(ob):
            import torch
            self._sync_stream()
            _lib.check(self._lib.irrl_env_reset(self._h, _dev_ptr(ob, (self._n, 35), (torch.float32,), "ob")))
        else:
            _lib.check(self._lib.irrl_env_reset_host(self._h, _np_f32(ob, (self._n, 35), "ob")))

    def observe(self, ob):
        if _is_torch_cuda(ob):
            import torch
            self._sync_stream()
            _lib.check(self._lib.irrl_env_observe(self._h, _dev_ptr(ob, (self._n, 35), (torch.float32,), "ob")))
        else:
            _lib.check(self._lib.irrl_env_observe_host(self._h, _np_f32(ob, (self._n, 35), "ob")))

    def step(self, action, ob, reward, done, extraInfo):
        n = self._n
        if _is_torch_cuda(action):
            import torch
            self._sync_stream()
            f32 = (torch.float32,)
            _lib.check(self._lib.irrl_env_step(
                self._h, _dev_ptr(action, (n, 12), f32, "action"), _dev_ptr(ob, (n, 35), f32, "ob"),
                _dev_ptr(reward, (n,), f32, "reward"), _dev_ptr(done, (n,), (torch.bool, torch.uint8), "done"),
                _dev_ptr(extraInfo, (n, 6), f32, "extraInfo")))
        else:
            _lib.check(self._lib.irrl_env_step_host(
                self._h, _np_f32(action, (n, 12), "action"), _np_f32(ob, (n, 35), "ob"), _np_f32(reward, (n,), "reward"),
                _np_bool(done, (n,), "done"), _np_f32(extraInfo, (n, 6), "extraInfo")))

    def _step_rows_args(self, count, action_rows, first_row, ob, reward, done, extraInfo, persistent):
        """-> (C function, argument tuple).  Outputs of shape [N, .] are overwritten by every step (the last step's survive); outputs of
        shape [count, N, .] select the `_out` entry points: step k fills row k, the trajectory `count` step() calls would have returned"""
        import torch
        n, count = self._n, int(count)
        rows = int(action_rows.shape[0])
        self._sync_stream()
        f32 = (torch.float32,)
        per_step = ob.dim() == 3
        lead = (count,) if per_step else ()
        args = (self._h, count, _dev_ptr(action_rows, (rows, n, 12), f32, "action_rows"), rows, int(first_row),
                _dev_ptr(ob, lead + (n, 35), f32, "ob"), _dev_ptr(reward, lead + (n,), f32, "reward"),
                _dev_ptr(done, lead + (n,), (torch.bool, torch.uint8), "done"), _dev_ptr(extraInfo, lead + (n, 6), f32, "extraInfo"))
        name = "irrl_env_step_rows" + ("_persistent" if persistent else "") + ("_out" if per_step else "")
        return getattr(self._lib, name), args

    def step_rows(self, count, action_rows, first_row, ob, reward, done, extraInfo, persistent=False):
        """build-defined: `count` consecutive steps from a device-resident action table [rows, N, 12] (step k takes row
        (first_row + k) % rows) in one call -- the launches go out back to back from C (irrl_env_step_rows), or, with persistent=True,
        as ONE launch in which every wave walks its own robots through all the steps (irrl_env_step_rows_persistent; same bits).
        ob / reward / done / extraInfo of shape [count, N, .] receive EVERY step's outputs (row k = step k: irrl_env_step_rows[_persistent]_out,
        what `count` step() calls return one after the other); of shape [N, .] only the last step's survive."""
        fn, args = self._step_rows_args(count, action_rows, first_row, ob, reward, done, extraInfo, persistent)
        _lib.check(fn(*args))

    def step_rows_call(self, count, action_rows, first_row, ob, reward, done, extraInfo, persistent=False):
        """the same call with its arguments checked and marshalled NOW: returns a zero-argument callable that only issues the
        launches (for callers that time them: bench.py's 20-step bracket is 0.8 ms long and the checks above cost ~15 us)"""
        fn, args = self._step_rows_args(count, action_rows, first_row, ob, reward, done, extraInfo, persistent)
        return lambda: _lib.check(fn(*args))

    def testStep(self, action, ob, reward, done, extraInfo):
        n = self._n
        _lib.check(self._lib.irrl_env_test_step_host(
            self._h, _np_f32(action, (n, 12), "action"), _np_f32(ob, (n, 35), "ob"), _np_f32(reward, (n,), "reward"),
            _np_bool(done, (n,), "done"), _np_f32(extraInfo, (n, 6), "extraInfo")))

    def setSeed(self, seed):
        _lib.check(self._lib.irrl_env_set_seed(self._h, int(seed)))
        self.params_epoch += 1

    def close(self):
        _lib.check(self._lib.irrl_env_close(self._h))

    def isTerminalState(self, done):
        if _is_torch_cuda(done):
            import torch
            self._sync_stream()
            _lib.check(self._lib.irrl_env_is_terminal(self._h, _dev_ptr(done, (self._n,), (torch.bool, torch.uint8), "done")))
        else:
            _lib.check(self._lib.irrl_env_is_terminal_host(self._h, _np_bool(done, (self._n,), "done")))

    def setSimulationTimeStep(self, dt):
        _lib.check(self._lib.irrl_env_set_simulation_dt(self._h, float(dt)))
        self.params_epoch += 1

    def setControlTimeStep(self, dt):
        _lib.check(self._lib.irrl_env_set_control_dt(self._h, float(dt)))
        self.params_epoch += 1

    def getObDim(self):
        return self._lib.irrl_env_ob_dim(self._h)

    def getActionDim(self):
        return self._lib.irrl_env_action_dim(self._h)

    def getExtraInfoDim(self):
        return self._lib.irrl_env_extra_dim(self._h)

    def getNumOfEnvs(self):
        return self._n

    # rendering entry points: the engine is headless (SURVEY section 2, rows 15-16 out of scope)
    def startRecordingVideo(self, file_name):
        pass

    def stopRecordingVideo(self):
        pass

    def showWindow(self):
        pass

    def hideWindow(self):
        pass

    def curriculumUpdate(self):
        _lib.check(self._lib.irrl_env_curriculum_update(self._h))

    def OriginState(self, out):
        _lib.check(self._lib.irrl_env_origin_state_host(self._h, _np_f32(out, (self._n, 41), "origin_state")))

    def GetOriginStateDim(self):
        return 41

    def ReferenceState(self, out):
        # VectorizedEnvironment.hpp:223-226 dispatches ReferenceState to OriginState (a reference bug): the
        # caller's [N,24] buffer receives the first 24 origin-state entries.  Reproduced deliberately; the
        # real (jointRef, jointDotRef) rows are available through `reference_state()`.
        tmp = np.zeros((self._n, 41), np.float32)
        self.OriginState(tmp)
        if not isinstance(out, np.ndarray) or out.dtype != np.float32 or out.shape != (self._n, 24):
            raise TypeError("refer_state must be a float32 numpy array of shape (%d, 24)" % self._n)
        out[:] = tmp[:, :24]

    def reference_state(self):
        out = np.zeros((self._n, 24), np.float32)
        _lib.check(self._lib.irrl_env_reference_state_host(self._h, _np_f32(out, (self._n, 24), "out")))
        return out

    def GetJointEffort(self, out):
        _lib.check(self._lib.irrl_env_joint_effort_host(self._h, _np_f32(out, (self._n, 12), "joint_effort")))

    def GetGeneralizedForce(self, out):
        _lib.check(self._lib.irrl_env_generalized_force_host(self._h, _np_f32(out, (self._n, 18), "generalized_force")))

    def GetInverseMassMatrix(self, out):
        _lib.check(self._lib.irrl_env_inverse_mass_matrix_host(self._h, _np_f32(out, (self._n, 324), "inverse_mass")))

    def GetNonlinear(self, out):
        _lib.check(self._lib.irrl_env_nonlinear_host(self._h, _np_f32(out, (self._n, 18), "nonlinear")))

    def SetContactCoefficient(self, coeff):
        _lib.check(self._lib.irrl_env_set_contact_coeff_host(self._h, _np_f32(coeff, (self._n, 3), "contact_coeff")))

    def GetSphereInfo(self, out):
        _lib.check(self._lib.irrl_env_sphere_info_host(self._h, _np_f32(out, (self._n, 4), "sphere_info")))

    # -- build-defined extras (checkpoint / parity) --
    def get_state(self):
        out = np.zeros((self._n, 288), np.float64)
        _lib.check(self._lib.irrl_env_get_state_host(self._h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def set_state(self, state):
        state = np.ascontiguousarray(state, np.float64)
        if state.shape != (self._n, 288):
            raise TypeError("state must have shape (%d, 288)" % self._n)
        _lib.check(self._lib.irrl_env_set_state_host(self._h, state.ctypes.data_as(C.POINTER(C.c_double))))

    def set_ref(self, table):
        """Reference-trajectory table of a `ManualTraj: False` pool (Environment.hpp:1895 `set_ref`): [rows, >= 30] float32,
        one row per control step = theta 12 | theta_dot 12 | z | phase 2 | cmd 3.  Call before init() when cfg["RefTraj"]
        does not name a readable CSV."""
        table = np.ascontiguousarray(table, dtype=np.float32)
        assert table.ndim == 2
        _lib.check(self._lib.irrl_env_set_ref_host(self._h, table.ctypes.data_as(_fp), table.shape[0], table.shape[1]))
        self.params_epoch += 1

    def heightfield(self):
        """[5000, 500] float32 height field of a Terrain: True pool (None on flat ground)."""
        nx, ny = C.c_int(0), C.c_int(0)
        if self._lib.irrl_env_heightfield_host(self._h, None, C.byref(nx), C.byref(ny)) != 0:
            return None
        out = np.zeros((nx.value, ny.value), np.float32)
        _lib.check(self._lib.irrl_env_heightfield_host(self._h, out.ctypes.data_as(_fp), C.byref(nx), C.byref(ny)))
        return out

    def snapshot(self):
        """device copy of the whole state pool (stream-ordered on torch's current stream); `restore()` puts it back"""
        self._sync_stream()
        _lib.check(self._lib.irrl_env_snapshot(self._h))

    def restore(self):
        self._sync_stream()
        _lib.check(self._lib.irrl_env_restore(self._h))

    def counters(self):
        """(episodes started, toe-substeps in contact, sum of frame_idx) summed over the pool -- diagnostic, synchronises."""
        out = (C.c_ulonglong * 3)()
        _lib.check(self._lib.irrl_env_counters_host(self._h, out))
        return int(out[0]), int(out[1]), int(out[2])

    def counters_into(self, out):
        """the same three sums written into `out` (CUDA int64 tensor of 3 elements), stream-ordered, no synchronisation"""
        import torch
        self._sync_stream()
        _lib.check(self._lib.irrl_env_counters(self._h, _dev_ptr(out, (3,), (torch.int64,), "out")))

    def cfg_value(self, key):
        return self._lib.irrl_env_cfg_value(self._h, key.encode())
