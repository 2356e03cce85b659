"""Device-resident counterpart of the reference's MPC expert (bitrate_selection/envs/expert_env.py:12-423).

* `ExpertCache`  -- the per-(video, user, chunk, action) profile of viewport quality / intra-viewport variance / chunk size
                    for the ground-truth and the predicted viewport (expert_env.py:126-181), computed by one kernel launch
                    over the HBM-resident tables; `to_reference()` returns the six nested dicts the reference pickles.
* `ExpertVecEnv` -- N environments (MANSYVecEnv, reward = raw QoE) + `choose_action()`: the exhaustive search over the
                    15^horizon bitrate plans (expert_env.py:358-422) for all N environments in three launches.
* `ExpertEnv`    -- single-environment wrapper with the reference's constructor signature and methods
                    (`reset`, `choose_action`, `step`, `sample_count`), cache pickle and CSV log included.
"""
import ctypes
import os
import pickle

import numpy as np
import torch

from ..._lib import MansyError, check, lib, ptr, stream_ptr
from .mansy_env import EnvTables, MANSYVecEnv, obs_to_dict

ACTION2RATES = [(1, 0), (2, 0), (3, 0), (4, 0), (2, 1), (3, 1), (4, 1), (3, 2), (4, 2), (4, 3), (0, 0), (1, 1), (2, 2), (3, 3), (4, 4)]
MAX_HORIZON = 6
CACHE_KEYS = ('gt_quality', 'pred_quality', 'gt_var', 'pred_var', 'gt_size', 'pred_size')


def action2rates(action):
    """utils/common.py:101-119."""
    return ACTION2RATES[action] if 0 <= action < 15 else (0, 0)


def rates2action(rate_in, rate_out):
    """utils/common.py:122-139."""
    return ACTION2RATES.index((rate_in, rate_out)) if (rate_in, rate_out) in ACTION2RATES else 0


class ExpertCache:
    def __init__(self, tables):
        self.tables = tables
        dev = tables.device
        smp = tables.host['samples']
        n_vp, nvc = tables.host['vp_gt'].shape[:2]
        vp_video = np.full(n_vp, -1, np.int32)
        for video, vp in {(int(s[0]), int(s[1])) for s in smp}:
            if vp_video[vp] not in (-1, video):
                raise MansyError(f'viewport slot {vp} is paired with two videos ({vp_video[vp]} and {video})')
            vp_video[vp] = video
        vp_video[vp_video < 0] = 0       # slots no episode uses still get a (meaningless) profile
        self.vp_video = torch.from_numpy(vp_video).to(dev)
        self.t = {k: torch.zeros(n_vp, nvc, 15, dtype=torch.int32 if 'size' in k else torch.float32, device=dev) for k in CACHE_KEYS}
        check(lib().mansy_expert_profile(ctypes.byref(tables.c), ptr(self.vp_video), n_vp, *[ptr(self.t[k]) for k in CACHE_KEYS],
                                         stream_ptr(dev)), 'mansy_expert_profile')

    def to_reference(self):
        """The six dicts of expert_env.py:21-26 / :100-102: {(video, user): {chunk: {(rate_in, rate_out): value}}}."""
        if self.tables.ids is None:
            raise MansyError('to_reference() needs tables built by EnvTables.from_dataset (video / user ids)')
        host = {k: v.cpu().numpy() for k, v in self.t.items()}
        vstart, vend = self.tables.host['vp_start'], self.tables.host['vp_end']
        vlen, vp_video = self.tables.host['video_len'], self.vp_video.cpu().numpy()
        out = []
        for k in CACHE_KEYS:
            d = {}
            for i, pair in enumerate(self.tables.ids[1]):
                end = min(int(vend[i]), int(vlen[vp_video[i]]) - 1)
                d[pair] = {c: {ACTION2RATES[a]: (int(host[k][i, c - vstart[i], a]) if 'size' in k else host[k][i, c - vstart[i], a])
                               for a in range(15)} for c in range(self.tables.startup_download + 1, end + 1)}
            out.append(d)
        # reference order: gt quality, pred quality, gt variance, pred variance, gt size, pred size
        return out


def dataset_cache(config, dataset, network_dataset, qoe_weights, device):
    """The `<dataset>_cache.pkl` content of expert_env.py:92-111: the profile of EVERY (video, user) pair of the train, valid
    and test splits (the simulator is built with trace 0 there: 'a random trace is just fine')."""
    videos, users = [], []
    for split in ('train', 'valid', 'test'):
        videos += config.video_split[dataset][split]
        users += config.user_split[dataset][split]
    videos, users = list(set(videos)), list(set(users))
    trace = config.network_split[network_dataset]['train'][:1]
    samples = [(i, j, 0, 0) for i in range(len(videos)) for j in range(len(users))]
    tables = EnvTables.from_dataset(config, dataset, network_dataset, 'train', qoe_weights, device, samples=samples, lists=(videos, users, trace))
    return ExpertCache(tables).to_reference()


class ExpertVecEnv(MANSYVecEnv):
    def __init__(self, tables, n_env, horizon, seed=0, index_offset=0, worker_num=None, cache=None, **kw):
        if not 1 <= int(horizon) <= MAX_HORIZON:
            raise MansyError(f'horizon must be in [1, {MAX_HORIZON}]')
        if tables.c.train_identifier_reward:
            raise MansyError('the expert environment rewards raw QoE: build the tables with train_identifier_reward=False')
        super().__init__(tables, n_env, seed=seed, index_offset=index_offset, worker_num=worker_num, **kw)
        self.horizon = int(horizon)
        self.cache = cache if cache is not None else ExpertCache(tables)
        self._keys = torch.zeros(self.n_env, dtype=torch.int64, device=self.device)
        self.actions = torch.zeros(self.n_env, dtype=torch.int32, device=self.device)
        self.best_value = torch.zeros(self.n_env, dtype=torch.float32, device=self.device)
        self.best_index = torch.zeros(self.n_env, dtype=torch.int64, device=self.device)

    def choose_action(self):
        """int32 device tensor [N]: first action of the best plan of every environment (also fills best_value / best_index)."""
        c = self.cache.t
        check(lib().mansy_expert_choose_action(ctypes.byref(self.tables.c), ptr(self.state), self.n_env, self.horizon, ptr(c['pred_quality']),
                                               ptr(c['pred_var']), ptr(c['pred_size']), ptr(self._keys), ptr(self.actions),
                                               ptr(self.best_value), ptr(self.best_index), stream_ptr(self.device)),
              'mansy_expert_choose_action')
        return self.actions


class ExpertEnv:
    """Drop-in single environment (reference constructor signature, expert_env.py:29-30)."""

    def __init__(self, config, dataset, network_dataset, qoe_weights, samples, demos_dir, cache_path, log_path, startup_download, horizon,
                 refresh_cache=True, mode='train', seed=0, device='cuda'):
        self.config, self.dataset, self.network_dataset = config, dataset, network_dataset
        self.qoe_weights, self.samples, self.demos_dir, self.log_path = qoe_weights, samples, demos_dir, log_path
        self.horizon, self.mode, self.random_seed = horizon, mode, seed
        dev = device if str(device).startswith('cuda') else 'cuda'
        self.tables = EnvTables.from_dataset(config, dataset, network_dataset, mode, qoe_weights, dev, seed=seed, samples=samples)
        self._venv = ExpertVecEnv(self.tables, 1, horizon, seed=0, worker_num=1)       # walks `samples` in order (expert_env.py:185)
        if cache_path and (refresh_cache or not os.path.exists(cache_path)):
            pickle.dump(dataset_cache(config, dataset, network_dataset, qoe_weights, dev), open(cache_path, 'wb'))
            print('Save expert cache at', cache_path)
        self._act = torch.zeros(1, dtype=torch.int32, device=self.tables.device)
        self.sample_id = -1
        self.current_video = self.current_user = self.current_trace = self.current_qoe_weight = None
        self.state = None

    def seed(self, seed):
        """expert_env.py:318-321: seeds numpy; the expert walks its sample list in order whatever the seed."""
        np.random.seed(seed)
        self.random_seed = seed

    def render(self, mode='human'):
        """gym API (expert_env.py:323-330 opens an empty classic-control window): nothing to draw on the device path."""
        return None

    def close(self):
        """gym API (expert_env.py:332-335)."""
        return None

    def sample_count(self):
        return len(self.samples)

    def reset(self):
        self.sample_id = (self.sample_id + 1) % len(self.samples)
        obs = self._venv.reset()
        self.current_video, self.current_user, self.current_trace = self.tables.ids[3][self.sample_id]
        self.current_qoe_weight = np.array(self.qoe_weights[self.samples[self.sample_id][3]], dtype=np.float32)
        self.state = obs_to_dict(obs[0].cpu().numpy())
        return self.state

    def choose_action(self):
        return int(self._venv.choose_action()[0].item())

    def step(self, action):
        self._act[0] = int(action)
        obs, rew, done, _ = self._venv.step(self._act, auto_reset=False)
        over = bool(done[0].item())
        self.state = obs_to_dict(obs[0].cpu().numpy())
        if over:       # expert_env.py:338-356 (same columns and rounding as MANSYEnv._log)
            from ..models.mansy_trainer import write_episode_log
            write_episode_log(self.log_path, self.tables, self.qoe_weights, self._venv.pop_episode_log())
        return self.state, np.float32(rew[0].item()), over, {}
