"""Test double: the CPU oracle behind the TorchVecEnv interface, so the PPO2 learner (host logic) can be
exercised on CPU tensors -- single process and gloo world_size 2 -- in the GPU-less container."""
import numpy as np
import torch

import oracle as O


class OracleTorchEnv(object):
    def __init__(self, env_cfg):
        self.env = O.OracleVecEnv(env_cfg)
        self.num_envs, self.num_obs, self.num_acts = self.env.n, 35, 12
        self.device = torch.device("cpu")
        self.env_id_offset = int(env_cfg.get("EnvIdOffset", 0))     # what TorchVecEnv reads from the C-ABI pool's configuration
        self._ret = np.zeros(self.num_envs)
        self._len = np.zeros(self.num_envs)
        self._fin = [0.0, 0.0, 0]

    def step(self, action):
        ob, rew, done, _ = self.env.step(action.detach().cpu().numpy().astype(np.float32))
        self._ret += rew
        self._len += 1
        for i in np.flatnonzero(done):
            self._fin[0] += self._ret[i]; self._fin[1] += self._len[i]; self._fin[2] += 1
            self._ret[i] = 0; self._len[i] = 0
        return torch.from_numpy(ob), torch.from_numpy(rew), torch.from_numpy(done)

    def reset(self):
        return torch.from_numpy(self.env.reset())

    def reset_and_update_info(self):
        self._fin[0] += self._ret.sum(); self._fin[1] += self._len.sum(); self._fin[2] += self.num_envs
        self._ret[:] = 0; self._len[:] = 0
        return self.reset()

    def pop_episode_stats(self):
        r, l, c = self._fin
        self._fin = [0.0, 0.0, 0]
        return r / max(c, 1), l / max(c, 1), c
