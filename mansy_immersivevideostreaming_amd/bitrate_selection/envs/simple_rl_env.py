"""Device-resident counterpart of the A2C baseline's environment (bitrate_selection/envs/simple_rl_env.py:12-203).

SimpleRLEnv runs the same simulator, tile-rate allocation and QoE model as MANSYEnv and differs in what it shows the agent
(five keys) and in the reward rule (train: qoe / sum(w); valid / test: raw qoe).  `SimpleRLVecEnv` therefore steps a
`MANSYVecEnv` (one kernel launch for N environments) and derives the five-key observation rows with `mansy_a2c_obs`;
`SimpleRLEnv` is the single-environment wrapper with the reference's constructor signature and 4-tuple gym API.
"""
import ctypes
import os

import numpy as np
import torch

from ..._lib import check, lib, ptr, stream_ptr
from ..models.simple_rl import LD, obs_to_dict
from .mansy_env import EnvTables, MANSYVecEnv


class SimpleRLVecEnv:
    def __init__(self, tables, n_env, seed=0, index_offset=0, worker_num=None, **kw):
        self.inner = MANSYVecEnv(tables, n_env, seed=seed, index_offset=index_offset, worker_num=worker_num, **kw)
        self.tables, self.n_env, self.device = tables, self.inner.n_env, tables.device
        self._rates = (ctypes.c_int * 5)(*tables.video_rates)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.obs = torch.zeros(self.n_env, LD, **f32)
        self.obs_next = torch.zeros(self.n_env, LD, **f32)

    def _rows(self, src, actions, fresh, out):
        check(lib().mansy_a2c_obs(ptr(src), ptr(self.inner.qoe_parts) if actions is not None else None, ptr(actions), ptr(fresh), self.n_env,
                                  self._rates, ptr(out), stream_ptr(self.device)), 'mansy_a2c_obs')
        return out

    def reset(self):
        return self._rows(self.inner.reset(), None, None, self.obs)

    def step(self, actions, auto_reset=True, obs_out=None, obs_next_out=None, reward_out=None, done_out=None):
        """actions int32 [N] on the device -> (obs [N,416], reward, done, {}): obs shows the next episode's first observation
        for finished environments when auto_reset (vector-env semantics), obs_next the post-action one."""
        cur, rew, done, _ = self.inner.step(actions, auto_reset=auto_reset, reward_out=reward_out, done_out=done_out)
        nxt = self._rows(self.inner.obs_next, actions, None, obs_next_out if obs_next_out is not None else self.obs_next)
        if not auto_reset:
            return nxt, rew, done, {}
        return self._rows(cur, actions, done, obs_out if obs_out is not None else self.obs), rew, done, {}

    def pop_episode_log(self):
        return self.inner.pop_episode_log()


class SimpleRLEnv:
    """Drop-in single environment (reference constructor signature, simple_rl_env.py:15-16)."""

    def __init__(self, config, dataset, network_dataset, qoe_weights, log_path, startup_download, mode='train', seed=0, worker_num=1, device='cuda'):
        assert mode in ['train', 'valid', 'test']
        self.config, self.qoe_weights, self.log_path, self.mode, self.worker_num = config, qoe_weights, log_path, mode, worker_num
        dev = device if str(device).startswith('cuda') else 'cuda'
        # reward = qoe / sum(w) in train mode, raw qoe otherwise (simple_rl_env.py:133-136)
        self.tables = EnvTables.from_dataset(config, dataset, network_dataset, mode, qoe_weights, dev, seed=seed, use_identifier=(mode == 'train'))
        self.samples = self.tables.host['samples']
        self._venv = SimpleRLVecEnv(self.tables, 1, seed=seed, worker_num=worker_num)
        self._act = torch.zeros(1, dtype=torch.int32, device=self.tables.device)
        self.state = None
        self.current_video = self.current_user = self.current_trace = None

    def seed(self, seed):
        np.random.seed(seed)
        self._venv = SimpleRLVecEnv(self.tables, 1, seed=seed, worker_num=self.worker_num)

    def render(self, mode='human'):
        """gym API (simple_rl_env.py opens an empty classic-control window): nothing to draw on the device path."""
        return None

    def close(self):
        """gym API (simple_rl_env.py)."""
        return None

    def sample_count(self):
        return len(self.samples)

    def reset(self, seed=None, options=None):
        self.state = obs_to_dict(self._venv.reset()[0].cpu().numpy())
        return self.state

    def step(self, action):
        self._act[0] = int(action)
        obs, rew, done, _ = self._venv.step(self._act, auto_reset=False)
        over = bool(done[0].item())
        self.state = obs_to_dict(obs[0].cpu().numpy())
        if over:
            from ..models.mansy_trainer import write_episode_log
            write_episode_log(self.log_path, self.tables, self.qoe_weights, self._venv.pop_episode_log())
        return self.state, np.float32(rew[0].item()), over, {}
