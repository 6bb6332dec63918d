"""VecEnv adapters with the call surface of flex_gym/env/RaisimGymVecEnv.py (reference lines cited per
method).  Two flavours over the same `FlexibleGymEnv` handle:

  RaisimGymVecEnv   numpy in / numpy out, exactly the reference's surface (step -> copies + info list).
                    The reference's per-env Python bookkeeping loop (RaisimGymVecEnv.py:42-50, O(N) dict
                    work per step) is vectorised with numpy; `info[i]['episode']` is still delivered for
                    every env whose episode ended.
  TorchVecEnv       the same env stepped with torch CUDA tensors, nothing leaves the device: this is what
                    the on-device PPO2 rollout drives (no host round-trip per step).
"""
import numpy as np


class Box(object):
    """Minimal stand-in for gym.spaces.Box (gym is not a dependency of this engine)."""

    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

    def __repr__(self):
        return "Box(%s, %s, %s)" % (self.low.min(), self.high.max(), self.shape)


class RaisimGymVecEnv(object):
    """RaisimGymVecEnv.py:6-189."""

    def __init__(self, impl, legacy_info=False):
        self.wrapper = impl
        self.wrapper.init()                                             # RaisimGymVecEnv.py:10
        self.num_obs = self.wrapper.getObDim()
        self.num_acts = self.wrapper.getActionDim()
        self._observation_space = Box(np.ones(self.num_obs) * -np.inf, np.ones(self.num_obs) * np.inf)
        self._action_space = Box(np.ones(self.num_acts) * -1.0, np.ones(self.num_acts) * 1.0)
        self._observation = np.zeros([self.num_envs, self.num_obs], dtype=np.float32)
        self._reward = np.zeros(self.num_envs, dtype=np.float32)
        self._done = np.zeros(self.num_envs, dtype=np.bool_)
        self._extraInfoNames = self.wrapper.getExtraInfoNames()
        self._extraInfo = np.zeros([self.num_envs, len(self._extraInfoNames)], dtype=np.float32)
        # vectorised replacement of `self.rewards = [[] for _ in range(num_envs)]`
        self._ep_ret = np.zeros(self.num_envs, dtype=np.float64)
        self._ep_len = np.zeros(self.num_envs, dtype=np.int64)
        self._legacy_info = legacy_info
        self._empty = {}

    def seed(self, seed=None):
        self.wrapper.setSeed(0 if seed is None else int(seed))

    @property
    def env_id_offset(self):
        """global id of this pool's env 0 (cfg `EnvIdOffset`; build-defined, see TorchVecEnv.env_id_offset): what PPO2 checks across ranks"""
        v = self.wrapper.cfg_value("EnvIdOffset") if hasattr(self.wrapper, "cfg_value") else float("nan")
        return 0 if v != v else int(v)

    def step(self, action, visualize=False):                            # RaisimGymVecEnv.py:26-52
        action = np.ascontiguousarray(action, dtype=np.float32)
        if not visualize:
            self.wrapper.step(action, self._observation, self._reward, self._done, self._extraInfo)
        else:
            self.wrapper.testStep(action, self._observation, self._reward, self._done, self._extraInfo)
        if self._legacy_info and len(self._extraInfoNames) != 0:
            info = [{'extra_info': {self._extraInfoNames[j]: self._extraInfo[i, j]}}
                    for j in range(len(self._extraInfoNames)) for i in range(self.num_envs)]
        else:
            info = [self._empty] * self.num_envs
        self._ep_ret += self._reward
        self._ep_len += 1
        for i in np.flatnonzero(self._done):
            d = dict(info[i]) if info[i] else {}
            d['episode'] = {"r": float(self._ep_ret[i]), "l": int(self._ep_len[i])}
            info[i] = d
            self._ep_ret[i] = 0.0
            self._ep_len[i] = 0
        return self._observation.copy(), self._reward.copy(), self._done.copy(), info

    @property
    def extra_info(self):
        """[N, 6] float32 extraInfo of the last step (columns = extra_info_names)."""
        return self._extraInfo

    def _getter(self, fn, width):
        temp = np.zeros([self.num_envs, width], dtype=np.float32)
        fn(temp)
        return temp

    def OriginState(self):                                              # RaisimGymVecEnv.py:54-57
        return self._getter(self.wrapper.OriginState, self.wrapper.GetOriginStateDim())

    def ReferenceState(self):                                           # RaisimGymVecEnv.py:59-62
        return self._getter(self.wrapper.ReferenceState, self.num_acts * 2)

    def GetJointEffort(self):                                           # RaisimGymVecEnv.py:64-67
        return self._getter(self.wrapper.GetJointEffort, self.num_acts)

    def GetGeneralizedForce(self):                                      # RaisimGymVecEnv.py:69-74
        return self._getter(self.wrapper.GetGeneralizedForce, self.num_acts + 6)

    def GetInverseMassMatrix(self):                                     # RaisimGymVecEnv.py:76-79
        return self._getter(self.wrapper.GetInverseMassMatrix, (self.num_acts + 6) * (self.num_acts + 6))

    def GetNonlinear(self):                                             # RaisimGymVecEnv.py:81-84
        return self._getter(self.wrapper.GetNonlinear, self.num_acts + 6)

    def GetSphereInfo(self):                                            # RaisimGymVecEnv.py:86-89
        return self._getter(self.wrapper.GetSphereInfo, 4)

    def SetContactCoefficient(self, contact_coeff):                     # RaisimGymVecEnv.py:91-93
        self.wrapper.SetContactCoefficient(np.ascontiguousarray(contact_coeff, dtype=np.float32))

    def reset(self):                                                    # RaisimGymVecEnv.py:95-98
        self._reward = np.zeros(self.num_envs, dtype=np.float32)
        self.wrapper.reset(self._observation)
        return self._observation.copy()

    def reset_and_update_info(self):                                    # RaisimGymVecEnv.py:100-101
        return self.reset(), self._update_epi_info()

    def _update_epi_info(self):                                         # RaisimGymVecEnv.py:103-113
        info = [{'episode': {"r": float(r), "l": int(l)}} for r, l in zip(self._ep_ret, self._ep_len)]
        self._ep_ret[:] = 0.0
        self._ep_len[:] = 0
        return info

    def render(self, mode='human'):
        raise RuntimeError('This method is not implemented')

    def close(self):
        self.wrapper.close()

    def start_recording_video(self, file_name):
        self.wrapper.startRecordingVideo(file_name)

    def stop_recording_video(self):
        self.wrapper.stopRecordingVideo()

    def curriculum_callback(self):
        self.wrapper.curriculumUpdate()

    def step_async(self):
        raise RuntimeError('This method is not implemented')

    def step_wait(self):
        raise RuntimeError('This method is not implemented')

    def get_attr(self, attr_name, indices=None):
        raise RuntimeError('This method is not implemented')

    def set_attr(self, attr_name, value, indices=None):
        raise RuntimeError('This method is not implemented')

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        raise RuntimeError('This method is not implemented')

    def show_window(self):
        self.wrapper.showWindow()

    def hide_window(self):
        self.wrapper.hideWindow()

    @property
    def num_envs(self):
        return self.wrapper.getNumOfEnvs()

    @property
    def observation_space(self):
        return self._observation_space

    @property
    def action_space(self):
        return self._action_space

    @property
    def extra_info_names(self):
        return self._extraInfoNames


class TorchVecEnv(object):
    """Device-resident stepping of the same env: torch CUDA tensors in and out, episode statistics kept on
    the device, no synchronisation in step().  The returned tensors are the env's own buffers and are
    overwritten by the next step (the rollout buffer copies them)."""

    def __init__(self, impl, init=True):
        import torch
        self.torch = torch
        self.wrapper = impl
        if init:
            self.wrapper.init()
        self.num_envs = self.wrapper.getNumOfEnvs()
        self.num_obs = self.wrapper.getObDim()
        self.num_acts = self.wrapper.getActionDim()
        self.device = torch.device("cuda", self.wrapper.device_index)
        n = self.num_envs
        self.obs = torch.zeros(n, self.num_obs, device=self.device)
        self.reward = torch.zeros(n, device=self.device)
        self.done = torch.zeros(n, dtype=torch.bool, device=self.device)
        self.extra = torch.zeros(n, 6, device=self.device)
        self.extra_info_names = self.wrapper.getExtraInfoNames()
        self.observation_space = Box(np.ones(self.num_obs) * -np.inf, np.ones(self.num_obs) * np.inf)
        self.action_space = Box(np.ones(self.num_acts) * -1.0, np.ones(self.num_acts) * 1.0)
        self.ep_ret = torch.zeros(n, device=self.device)
        self.ep_len = torch.zeros(n, device=self.device)
        # running sums over finished episodes (read by the learner's log once per iteration)
        self.finished_ret_sum = torch.zeros((), device=self.device)
        self.finished_len_sum = torch.zeros((), device=self.device)
        self.finished_count = torch.zeros((), device=self.device)

    def step(self, action):
        action = action.contiguous()
        self.wrapper.step(action, self.obs, self.reward, self.done, self.extra)
        self.ep_ret += self.reward
        self.ep_len += 1.0
        d = self.done.to(self.reward.dtype)
        self.finished_ret_sum += (self.ep_ret * d).sum()
        self.finished_len_sum += (self.ep_len * d).sum()
        self.finished_count += d.sum()
        self.ep_ret *= (1.0 - d)
        self.ep_len *= (1.0 - d)
        return self.obs, self.reward, self.done

    @property
    def env_id_offset(self):
        """global id of this pool's env 0 (cfg `EnvIdOffset`; rank r of an N-GPU job owns r * num_envs ..): PPO2 addresses the policy's
        sampling noise by the same ids and checks that the ranks' ranges do not overlap"""
        v = self.wrapper.cfg_value("EnvIdOffset") if hasattr(self.wrapper, "cfg_value") else float("nan")
        return 0 if v != v else int(v)

    def step_into(self, action, obs, reward, done):
        """step() writing straight into the caller's tensors and WITHOUT the per-step episode bookkeeping (one launch);
        the caller accounts whole rollouts with `account_rollout`."""
        self.wrapper.step(action, obs, reward, done, self.extra)

    def account_rollout(self, reward_sum, n_steps, n_episodes):
        """Episode statistics of a whole rollout that ends with a global reset: every episode is counted exactly once
        (finished ones plus the N cut off by the reset), so the sums are just the rollout's totals."""
        self.finished_ret_sum += reward_sum
        self.finished_len_sum += n_steps
        self.finished_count += n_episodes

    def reset(self):
        self.wrapper.reset(self.obs)
        return self.obs

    def reset_and_update_info(self):
        """ppo2.py:577: global reset after every rollout; unfinished episodes are counted like the
        reference's _update_epi_info does (RaisimGymVecEnv.py:103-113)."""
        self.finished_ret_sum += self.ep_ret.sum()
        self.finished_len_sum += self.ep_len.sum()
        self.finished_count += float(self.num_envs)
        self.ep_ret.zero_()
        self.ep_len.zero_()
        return self.reset()

    def pop_episode_stats(self):
        """(mean return, mean length, count) of the episodes finished since the last call; one host sync."""
        s = self.torch.stack([self.finished_ret_sum, self.finished_len_sum, self.finished_count]).tolist()
        self.finished_ret_sum.zero_()
        self.finished_len_sum.zero_()
        self.finished_count.zero_()
        c = max(s[2], 1.0)
        return s[0] / c, s[1] / c, int(s[2])

    def close(self):
        self.wrapper.close()
