"""Vectorised, device-resident counterpart of the reference gym environment
(bitrate_selection/envs/mansy_env.py:16-290 + simulators/*.py + utils/qoe.py + utils/common.py:40-193).

* `EnvTables`   -- manifests / viewport maps / network traces / episode catalogue as HBM-resident tensors
                   (built from the dataset files exactly as Simulator.__init__ reads them, or synthetic).
* `MANSYVecEnv` -- N environments stepped by ONE kernel launch (csrc/env.hip), observations as rows of a
                   [N, 780] float32 tensor whose column ranges are the reference's dict keys (`OBS_SLICES`).
* `MANSYEnv`    -- single-environment wrapper with the reference's constructor signature and the old gym 4-tuple
                   API (`reset() -> dict`, `step(a) -> (dict, reward, over, {})`), CSV episode log included.
"""
import ctypes
import json
import math
import os
import pickle

import numpy as np
import torch

from ..._lib import EnvTables as _CTables, EpisodeLog, MansyError, check, lib, ptr, stream_ptr

OBS_DIM, OBS_LD = 779, 780
OBS_SLICES = {
    'throughput': (0, 8, (1, 8)), 'next_chunk_size': (8, 328, (5, 64)), 'next_chunk_quality': (328, 648, (5, 64)),
    'pred_viewport': (648, 712, (1, 64)), 'viewport_acc': (712, 720, (1, 8)), 'past_viewport_qualities': (720, 728, (1, 8)),
    'past_quality_variances': (728, 736, (1, 8)), 'past_rebuffering': (736, 744, (1, 8)), 'buffer': (744, 745, (1,)),
    'qoe_weight': (745, 748, (3,)), 'action_one_hot': (748, 763, (15,)), 'rates_inside': (763, 771, (1, 8)),
    'rates_outside': (771, 779, (1, 8)),
}


def generate_environment_samples(video_list, user_list, trace_list, qoe_list, seed=0):
    """utils/common.py:60-84."""
    nv, nu, nt, nq = len(video_list), len(user_list), len(trace_list), len(qoe_list)
    max_len = max(nv, nu, nt, nq)
    total = max(max_len, nv * nq * math.ceil(max_len / (nv * nq)))
    return [(i % nv, i % nu, i % nt, i % nq) for i in range(total)]


def generate_environment_test_samples(video_list, user_list, trace_list, qoe_list):
    """utils/common.py:87-98."""
    return [(i, j, k, l) for i in range(len(video_list)) for j in range(len(user_list)) for k in range(len(trace_list))
            for l in range(len(qoe_list))]


class EnvTables:
    """Device copies of the tables + the C struct handed to the kernels."""

    FIELDS = ('size', 'quality', 'video_len', 'vp_gt', 'vp_pred', 'vp_acc', 'vp_start', 'vp_end', 'trace_bw', 'trace_len', 'samples')
    DTYPES = dict(size=np.int32, quality=np.float32, video_len=np.int32, vp_gt=np.uint8, vp_pred=np.uint8, vp_acc=np.float64,
                  vp_start=np.int32, vp_end=np.int32, trace_bw=np.float64, trace_len=np.int32, samples=np.int32)

    def __init__(self, arrays, qoe_weights, device, video_rates=(1, 5, 8, 16, 35), startup_download=5, chunk_length=1,
                 max_size=500000, max_throughput=5000000, train_identifier_reward=False, ids=None):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise MansyError('EnvTables live in HBM: a cuda (ROCm) device is required')
        self.host = {k: np.ascontiguousarray(arrays[k], dtype=self.DTYPES[k]) for k in self.FIELDS}
        self.host['qoe_w'] = np.ascontiguousarray(qoe_weights, dtype=np.float32).reshape(-1, 3)
        # simulate_download (simulators/network.py) walks the trace bins until the chunk is through: a trace without one positive bin
        # (or with a negative / non-finite one) never gets there -- an endless loop in the reference, a hung queue on the device
        bw, tl = self.host['trace_bw'], self.host['trace_len']
        if bw.ndim != 2 or len(tl) != bw.shape[0] or (tl < 1).any() or (tl > bw.shape[1]).any():
            raise MansyError('EnvTables: trace_len must lie in [1, trace_bw.shape[1]] for every trace')
        live = np.arange(bw.shape[1])[None, :] < tl[:, None]
        if not np.isfinite(bw[live]).all() or (bw[live] < 0).any() or not ((bw * live) > 0).any(1).all():
            raise MansyError('EnvTables: every network trace needs finite, non-negative bandwidth bins and at least one positive bin')
        if (self.host['samples'] < 0).any():
            bad = np.nonzero((self.host['samples'] < 0).any(1))[0]
            self.unvisitable = set(int(b) for b in bad)
        else:
            self.unvisitable = set()
        self.t = {k: torch.from_numpy(v).to(self.device) for k, v in self.host.items()}
        self.ids = ids            # optional (videos, vp_pairs, traces) id lists for logging
        c = _CTables()
        for k in self.FIELDS + ('qoe_w',):
            setattr(c, k, self.t[k].data_ptr())
        c.n_chunk_max = self.host['size'].shape[1]
        c.n_vpchunk_max = self.host['vp_gt'].shape[1]
        c.trace_len_max = self.host['trace_bw'].shape[1]
        c.n_sample = self.host['samples'].shape[0]
        for i, r in enumerate(video_rates):
            c.video_rates[i] = int(r)
        c.startup_download, c.chunk_length = int(startup_download), int(chunk_length)
        c.max_size, c.max_throughput = float(max_size), float(max_throughput)
        c.train_identifier_reward = int(train_identifier_reward)
        self.c = c
        self.video_rates = tuple(int(r) for r in video_rates)
        self.startup_download = int(startup_download)

    @property
    def n_sample(self):
        return self.c.n_sample

    # ---- builders ------------------------------------------------------------------------------
    @staticmethod
    def arrays_from_dataset(config, dataset, network_dataset, mode, qoe_weights, seed=0, samples=None, lists=None):
        """Host half of `from_dataset`: reads the files Simulator.__init__ reads (simulator.py:30-45: prediction pickles, manifests,
        traces) for the episode catalogue MANSYEnv.__init__ enumerates (mansy_env.py:44-52) and returns (arrays, ids) -- numpy only, no
        device.  `samples`: explicit episode catalogue of (video, user, trace, qoe) list positions (ExpertEnv takes one);
        `lists`: explicit (videos, users, traces) id lists instead of the split of `mode`."""
        videos, users, traces = lists if lists is not None else (
            config.video_split[dataset][mode], config.user_split[dataset][mode], config.network_split[network_dataset][mode])
        if samples is not None:
            samples = [tuple(int(x) for x in s) for s in samples]
        elif mode != 'test':
            samples = generate_environment_samples(videos, users, traces, qoe_weights, seed=seed)
        else:
            samples = generate_environment_test_samples(videos, users, traces, qoe_weights)
        used_v = sorted({videos[s[0]] for s in samples})
        used_vp = sorted({(videos[s[0]], users[s[1]]) for s in samples})
        used_t = sorted({traces[s[2]] for s in samples})
        n_chunk = 0
        manifests = {}
        for v in used_v:
            m = json.load(open(os.path.join(config.video_datasets_dir[dataset], f'video{v}.json'), 'r', encoding='utf-8'))
            manifests[v] = m
            n_chunk = max(n_chunk, max(int(c) for c in m['Chunks']) + 1)
        size = np.zeros((len(used_v), n_chunk, 5, 64), np.int32)
        qual = np.zeros((len(used_v), n_chunk, 5, 64), np.float32)
        vlen = np.zeros(len(used_v), np.int32)
        for i, v in enumerate(used_v):
            vlen[i] = manifests[v]['Video_Time']
            for c, info in manifests[v]['Chunks'].items():
                size[i, int(c)] = np.array(info['size'], np.int32)
                qual[i, int(c)] = np.array(info['quality'], np.float32)
        pks = []
        for v, u in used_vp:
            pks.append(pickle.load(open(os.path.join(config.viewport_datasets_dir[dataset], 'prediction', f'video{v}', f'user{u}.pkl'), 'rb')))
        nvc = max(len(p) for p in pks)
        gt = np.zeros((len(pks), nvc, 64), np.uint8)
        pr = np.zeros((len(pks), nvc, 64), np.uint8)
        acc = np.zeros((len(pks), nvc), np.float64)
        vstart = np.zeros(len(pks), np.int32)
        vend = np.zeros(len(pks), np.int32)
        for i, pk in enumerate(pks):
            vstart[i], vend[i] = pk[0][0], pk[-1][0]
            for j, p in enumerate(pk):
                gt[i, j], pr[i, j], acc[i, j] = p[1], p[2], p[3]
        trs = []
        for t in used_t:
            tr = pickle.load(open(os.path.join(config.network_datasets_dir[network_dataset], config.network_info[network_dataset][t]), 'rb'))
            trs.append(np.array([x[1] for x in tr], np.float64))
        tmax = max(len(t) for t in trs)
        bw = np.zeros((len(trs), tmax), np.float64)
        tl = np.zeros(len(trs), np.int32)
        for i, t in enumerate(trs):
            bw[i, :len(t)], tl[i] = t, len(t)
        smp = np.array([(used_v.index(videos[a]), used_vp.index((videos[a], users[b])), used_t.index(traces[c]), d)
                        for a, b, c, d in samples], np.int32)
        # The device code indexes the viewport tables with `chunk - vp_start` and the manifest with `chunk`, unchecked; the
        # reference fails loudly on the same inputs (simulator.py:45 `assert startup_download + 1 >= start_chunk`, list / dict
        # lookups raising on chunks a prediction pickle or a manifest does not hold).  Check every catalogue entry here.
        first = int(config.startup_download) + 1
        for i, (v, u) in enumerate(used_vp):
            vi = used_v.index(v)
            end = min(int(vend[i]), int(vlen[vi]) - 1)
            if first < int(vstart[i]):
                raise MansyError(f'video{v}/user{u}.pkl starts at chunk {int(vstart[i])} but the session starts at chunk {first} '
                                 f'(startup_download + 1): re-export the predictions with a smaller --trim-head (simulator.py:45)')
            if len(pks[i]) != int(vend[i]) - int(vstart[i]) + 1 or any(int(p[0]) != int(vstart[i]) + j for j, p in enumerate(pks[i])):
                raise MansyError(f'video{v}/user{u}.pkl does not hold consecutive chunks {int(vstart[i])}..{int(vend[i])}')
            missing = [c for c in range(first, end + 1) if str(c) not in manifests[v]['Chunks']]
            if missing:
                raise MansyError(f'video{v}.json has no chunk {missing[0]} (needed up to chunk {end}: Video_Time '
                                 f'{int(vlen[vi])}, predictions end at {int(vend[i])})')
        arrays = dict(size=size, quality=qual, video_len=vlen, vp_gt=gt, vp_pred=pr, vp_acc=acc, vp_start=vstart, vp_end=vend,
                      trace_bw=bw, trace_len=tl, samples=smp)
        return arrays, (used_v, used_vp, used_t, [(videos[a], users[b], traces[c]) for a, b, c, _ in samples])

    @classmethod
    def from_dataset(cls, config, dataset, network_dataset, mode, qoe_weights, device, seed=0, use_identifier=False, samples=None,
                     lists=None):
        """The tables of one split read from the dataset tree (`arrays_from_dataset`), uploaded."""
        arrays, ids = cls.arrays_from_dataset(config, dataset, network_dataset, mode, qoe_weights, seed=seed, samples=samples, lists=lists)
        return cls(arrays, qoe_weights, device, video_rates=config.video_rates, startup_download=config.startup_download,
                   chunk_length=config.chunk_length, max_size=config.max_size, max_throughput=config.max_throughput,
                   train_identifier_reward=(mode == 'train' and use_identifier), ids=ids)

    @classmethod
    def from_file(cls, path, split, device, qoe_weights=None, use_identifier=False):
        """The tables of one split from a packed table file (`.npz` with `<split>/<field>` arrays, `<split>/qoe_w`, `<split>/ids_*` and
        `const/*`: the layout tools/gen_golden_tables_full.py writes for the reference's Jin2022 x 4G splits) -- one file read instead of
        the split's ~100 pickles / JSON manifests.  bench.py's real-table PPO leg and the shipped-run tests load the reference's tables this way."""
        z = np.load(path)
        arrays = {k: z[f'{split}/{k}'] for k in cls.FIELDS}
        qw = z[f'{split}/qoe_w'] if qoe_weights is None else qoe_weights
        misc = z['const/misc']
        ids = ([int(v) for v in z[f'{split}/ids_v']], [tuple(int(x) for x in r) for r in z[f'{split}/ids_vp']],
               [int(t) for t in z[f'{split}/ids_t']], [tuple(int(x) for x in r) for r in z[f'{split}/ids_samples']])
        return cls(arrays, qw, device, video_rates=tuple(int(r) for r in z['const/video_rates']), startup_download=int(misc[0]),
                   chunk_length=int(misc[1]), max_size=float(misc[2]), max_throughput=float(misc[3]),
                   train_identifier_reward=(split == 'train' and use_identifier), ids=ids)

    @classmethod
    def synthetic(cls, device, n_video=27, n_user=60, n_trace=40, n_chunk=60, seed=5, qoe_weights=((7, 1, 1), (1, 7, 1), (1, 1, 7), (3, 3, 3)),
                  train_identifier_reward=True, n_sample=None):
        """Same-shape synthetic tables: sizes ~ LogUniform(4e3, 1.8e5), quality = bitrate value,
        traces of 600 bins ~ U(0, 1.4e7) B/s, viewports = 3x3..4x4 wrapped tile blobs."""
        rs = np.random.RandomState(seed)
        rates = np.array([1, 5, 8, 16, 35], np.float32)
        base = np.exp(rs.uniform(np.log(4e3), np.log(3.6e4), size=(n_video, n_chunk, 1, 64)))
        scale = np.array([1.0, 1.8, 2.4, 3.4, 5.0]).reshape(1, 1, 5, 1)
        size = np.minimum(base * scale, 1.8e5).astype(np.int32)
        qual = np.broadcast_to(rates.reshape(1, 1, 5, 1), size.shape).astype(np.float32).copy()
        vlen = np.full(n_video, n_chunk, np.int32)
        n_vp = n_video * n_user
        nvc = n_chunk - 6
        gt = np.zeros((n_vp, nvc, 64), np.uint8)
        pr = np.zeros((n_vp, nvc, 64), np.uint8)

        def blob(r0, c0, h, w):
            m = np.zeros((8, 8), np.uint8)
            for dr in range(h):
                for dc in range(w):
                    m[(r0 + dr) % 8, (c0 + dc) % 8] = 1
            return m.reshape(-1)
        for i in range(n_vp):
            r0, c0 = rs.randint(0, 8), rs.randint(0, 8)
            for j in range(nvc):
                r0 = (r0 + rs.randint(-1, 2)) % 8
                c0 = (c0 + rs.randint(-1, 2)) % 8
                gt[i, j] = blob(r0, c0, rs.randint(3, 5), rs.randint(3, 5))
                pr[i, j] = blob((r0 + rs.randint(-1, 2)) % 8, (c0 + rs.randint(-1, 2)) % 8, rs.randint(3, 5), rs.randint(3, 5))
        inter = (gt & pr).sum(-1).astype(np.float64)
        union = np.maximum((gt | pr).sum(-1), 1).astype(np.float64)
        acc = inter / union
        vstart = np.full(n_vp, 3, np.int32)
        vend = np.full(n_vp, 3 + nvc - 1, np.int32)
        bw = rs.randint(0, int(1.4e7), size=(n_trace, 600)).astype(np.float64)
        tl = np.full(n_trace, 600, np.int32)
        if n_sample is None:
            n_sample = max(n_video, n_user, n_trace, len(qoe_weights)) * 4
        smp = np.array([(i % n_video, (i % n_video) * n_user + (i * 7) % n_user, i % n_trace, i % len(qoe_weights)) for i in range(n_sample)], np.int32)
        arrays = dict(size=size, quality=qual, video_len=vlen, vp_gt=gt, vp_pred=pr, vp_acc=acc, vp_start=vstart, vp_end=vend,
                      trace_bw=bw, trace_len=tl, samples=smp)
        return cls(arrays, qoe_weights, device, train_identifier_reward=train_identifier_reward)


class MANSYVecEnv:
    """N environments, one kernel launch per step.  Environment i (global index `index_offset + i` of `worker_num`)
    walks the episode catalogue exactly like the reference's worker scheme (mansy_env.py:55-56,100-101)."""

    def __init__(self, tables, n_env, seed=0, index_offset=0, worker_num=None, episode_log_capacity=65536):
        self.tables = tables
        self.n_env = int(n_env)
        self.device = tables.device
        worker_num = int(worker_num) if worker_num is not None else self.n_env
        L = lib()
        self.state = torch.zeros(self.n_env * L.mansy_env_state_bytes(), dtype=torch.uint8, device=self.device)
        check(L.mansy_env_init(ptr(self.state), self.n_env, int(index_offset), worker_num, int(seed), stream_ptr(self.device)), 'mansy_env_init')
        f32 = dict(dtype=torch.float32, device=self.device)
        self.obs = torch.zeros(self.n_env, OBS_LD, **f32)          # what the policy sees (auto-reset applied)
        self.obs_next = torch.zeros(self.n_env, OBS_LD, **f32)     # post-action observation (terminal one when done)
        self.reward = torch.zeros(self.n_env, **f32)
        self.done = torch.zeros(self.n_env, dtype=torch.uint8, device=self.device)
        self.qoe_parts = torch.zeros(self.n_env, 4, **f32)
        self.elog_records = torch.zeros(episode_log_capacity, 8, dtype=torch.float64, device=self.device)
        self.elog_count = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._elog = EpisodeLog(self.elog_records.data_ptr(), self.elog_count.data_ptr(), episode_log_capacity)

    def reset(self):
        check(lib().mansy_env_reset(ctypes.byref(self.tables.c), ptr(self.state), self.n_env, ptr(self.obs), stream_ptr(self.device)),
              'mansy_env_reset')
        return self.obs

    def step(self, actions, auto_reset=True, obs_out=None, obs_next_out=None, reward_out=None, done_out=None):
        """actions: int32 tensor [N] on the device.  Returns (obs, reward, done, {}), all device tensors; the buffers are
        reused between calls unless output tensors are passed (the rollout writes straight into its slabs)."""
        if actions.dtype != torch.int32 or not actions.is_cuda:
            raise MansyError('actions must be an int32 cuda tensor')
        obs = obs_out if obs_out is not None else self.obs
        obs_next = obs_next_out if obs_next_out is not None else self.obs_next
        rew = reward_out if reward_out is not None else self.reward
        done = done_out if done_out is not None else self.done
        check(lib().mansy_env_step(ctypes.byref(self.tables.c), ptr(self.state), self.n_env, ptr(actions), ptr(obs_next),
                                   ptr(obs) if auto_reset else None, ptr(rew), ptr(done), ptr(self.qoe_parts), ctypes.byref(self._elog),
                                   stream_ptr(self.device)), 'mansy_env_step')
        return (obs if auto_reset else obs_next), rew, done, {}

    def pop_episode_log(self):
        """Finished-episode records since the last call: array [k, 8] (sample_id, env, n, sum qoe, qoe1, qoe2, qoe3, qoe idx)."""
        n = int(self.elog_count.item())
        rec = self.elog_records[:min(n, self.elog_records.shape[0])].cpu().numpy().copy()
        self.elog_count.zero_()
        return rec


def obs_to_dict(row):
    """One observation row (numpy [780]) -> dict with the reference's keys/shapes (mansy_env.py:136-150)."""
    return {k: np.array(row[a:b], dtype=np.float32).reshape(shape) for k, (a, b, shape) in OBS_SLICES.items()}


class MANSYEnv:
    """Drop-in single environment (reference constructor signature, mansy_env.py:19-20)."""

    def __init__(self, config, dataset, network_dataset, qoe_weights, identifier, lamb, log_path, startup_download, mode='train', seed=0,
                 worker_num=1, device='cuda', use_identifier=False):
        assert mode in ['train', 'valid', 'test']
        self.config, self.dataset, self.network_dataset = config, dataset, network_dataset
        self.qoe_weights, self.identifier, self.lamb, self.log_path = qoe_weights, identifier, lamb, log_path
        self.mode, self.random_seed, self.worker_num, self.use_identifier = mode, seed, worker_num, use_identifier
        print('Use Identifier:', use_identifier)
        dev = device if str(device).startswith('cuda') else 'cuda'
        self.tables = EnvTables.from_dataset(config, dataset, network_dataset, mode, qoe_weights, dev, seed=seed, use_identifier=use_identifier)
        self.samples = self.tables.host['samples']
        self._venv = MANSYVecEnv(self.tables, 1, seed=seed, index_offset=0, worker_num=worker_num)
        self._act = torch.zeros(1, dtype=torch.int32, device=self.tables.device)
        self.action_space_n = config.action_space
        self.state = None

    def seed(self, seed):
        np.random.seed(seed)
        self.random_seed = seed
        self._venv = MANSYVecEnv(self.tables, 1, seed=seed, index_offset=0, worker_num=self.worker_num)


    def render(self, mode='human'):
        """gym API (mansy_env.py:258-264 opens an empty classic-control window): nothing to draw on the device path."""
        return None

    def close(self):
        """gym API (mansy_env.py:266-269)."""
        return None

    def sample_count(self):
        return len(self.samples)

    def reset(self, seed=None, options=None):
        obs = self._venv.reset()
        self.state = obs_to_dict(obs[0].cpu().numpy())
        return self.state

    def step(self, action):
        self._act[0] = int(action)
        obs, rew, done, _ = self._venv.step(self._act, auto_reset=False)
        over = bool(done[0].item())
        self.state = obs_to_dict(obs[0].cpu().numpy())
        reward = np.float32(rew[0].item())
        if over:
            self._log()
        return self.state, reward, over, {}

    def _log(self):
        """mansy_env.py:271-290 from the device-side episode accumulators."""
        rec = self._venv.pop_episode_log()
        if not len(rec):
            return
        if not os.path.exists(self.log_path):
            with open(self.log_path, 'w', encoding='utf-8') as file:
                file.write('video,user,trace,qoe_w1,qoe_w2,qoe_w3,qoe,qoe1,qoe2,qoe3\n')
        with open(self.log_path, 'a', encoding='utf-8') as file:
            for sid, _, n, sq, s1, s2, s3, qi in rec:
                w = np.array(self.qoe_weights[int(qi)], dtype=np.float32)
                video, user, trace = self.tables.ids[3][int(sid)]
                qoe = round(sq / n / sum(w), 5)
                file.write(f'{video},{user},{trace},{w[0]},{w[1]},{w[2]},{qoe},{round(s1 / n, 5)},{round(s2 / n, 5)},{round(s3 / n, 5)}\n')
