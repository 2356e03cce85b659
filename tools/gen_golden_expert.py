#!/usr/bin/env python3
"""Golden vectors of the reference's MPC expert (bitrate_selection/envs/expert_env.py), produced by importing and
running the reference in this container (stubs for gym/munch/prettytable only, config splits narrowed to a few videos /
users / traces so the profile cache is small).  Data only:

  * the expert cache (per (video, user), chunk, action: viewport quality, intra-viewport variance and chunk size for the
    ground-truth and the predicted viewport -- expert_env.py:126-181),
  * whole episodes driven by `choose_action()` (expert_env.py:358-422) at horizons 1..3 and the first decisions of a
    horizon-4 episode: chosen action, reward, done flag, flattened observation after every step, the CSV log line,
  * the tables (manifest rows, viewport maps, traces) those episodes touch, in the layout of oracle/env.c.
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refstubs  # noqa: E402
refstubs.install()
import gen_golden_env as gge  # noqa: E402  (chdirs into the reference tree, imports its config helpers)
from envs.expert_env import ExpertEnv  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
VIDEOS, USERS, TRACES = [1, 2], [22, 27], [26, 24]
ACTIONS = 15
PAIRS = [(1, 0), (2, 0), (3, 0), (4, 0), (2, 1), (3, 1), (4, 1), (3, 2), (4, 2), (4, 3), (0, 0), (1, 1), (2, 2), (3, 3), (4, 4)]


def run(config, samples, qoe_weights, horizon, max_decisions=None):
    log = tempfile.mktemp(suffix='.csv')
    cache = tempfile.mktemp(suffix='.pkl')
    env = ExpertEnv(config, 'Jin2022', '4G', qoe_weights, samples, tempfile.gettempdir(), cache, log, config.startup_download, horizon,
                    refresh_cache=True, mode='train', seed=1)
    env.videos, env.users, env.traces = VIDEOS, USERS, TRACES
    eps = []
    for _ in range(len(samples)):
        st = env.reset()
        rec = dict(sample_id=env.sample_id, video=env.current_video, user=env.current_user, trace=env.current_trace,
                   obs=[gge.flat_obs(st)], act=[], rew=[], done=[])
        over, n = False, 0
        while not over and (max_decisions is None or n < max_decisions):
            a = int(env.choose_action())
            st, r, over, _ = env.step(a)
            rec['act'].append(a)
            rec['rew'].append(np.float32(r))
            rec['done'].append(bool(over))
            rec['obs'].append(gge.flat_obs(st))
            n += 1
        eps.append(rec)
    csv = open(log).read() if os.path.exists(log) else ''
    for p in (log, cache):
        if os.path.exists(p):
            os.remove(p)
    return env, eps, csv


def cache_arrays(env, vps, nvc, vstart):
    """Expert cache as [n_vp, nvc, 15] arrays (chunk index relative to the viewport trace's first chunk)."""
    out = {k: np.zeros((len(vps), nvc, ACTIONS), np.float32 if 'size' not in k else np.int64)
           for k in ('gt_quality', 'pred_quality', 'gt_var', 'pred_var', 'gt_size', 'pred_size')}
    filled = np.zeros((len(vps), nvc), np.bool_)
    src = dict(gt_quality=env.chunk_gt_viewport_qualities, pred_quality=env.chunk_pred_viewport_qualities,
               gt_var=env.chunk_gt_intra_quality_variance, pred_var=env.chunk_pred_intra_quality_variance,
               gt_size=env.chunk_gt_sizes, pred_size=env.chunk_pred_sizes)
    for i, (v, u) in enumerate(vps):
        for chunk in src['gt_size'][v, u]:
            filled[i, chunk - vstart[i]] = True
            for k, d in src.items():
                for a, pair in enumerate(PAIRS):
                    out[k][i, chunk - vstart[i], a] = d[v, u][chunk][pair]
    out['filled'] = filled
    return out


def main():
    config = gge.get_config_from_yml()
    for split in ('train', 'valid', 'test'):
        config.video_split['Jin2022'][split] = VIDEOS
        config.user_split['Jin2022'][split] = USERS
        config.network_split['4G'][split] = TRACES
    qoe = config.qoe_split['train']
    rec = {}
    env = None
    for tag, horizon, samples, maxd in [('h1', 1, [(0, 0, 0, 0), (1, 1, 1, 1)], None), ('h2', 2, [(0, 1, 1, 2), (1, 0, 0, 3)], None),
                                        ('h3', 3, [(1, 1, 0, 0), (0, 0, 1, 1), (1, 0, 1, 2)], None), ('h4', 4, [(0, 1, 0, 1)], 12)]:
        env, eps, csv = run(config, samples, qoe, horizon, maxd)
        # build_tables expects env.samples in the env's own enumeration: give it the narrowed lists
        tb = gge.build_tables(config, env, eps)
        for k, v in tb.items():
            rec[f'{tag}/{k}'] = v
        vids = sorted({e['video'] for e in eps})
        vps = sorted({(e['video'], e['user']) for e in eps})
        rec[f'{tag}/vp_video'] = np.array([vids.index(v) for v, _ in vps], np.int32)
        for k, v in cache_arrays(env, vps, tb['vp_gt'].shape[1], tb['vp_start']).items():
            rec[f'{tag}/cache/{k}'] = v
        rec[f'{tag}/meta'] = np.array([horizon, len(eps)], np.int32)
        rec[f'{tag}/qoe_w'] = np.array(qoe, np.float32)
        rec[f'{tag}/csv'] = np.array(csv)
        for i, e in enumerate(eps):
            rec[f'{tag}/ep{i}/sample_id'] = np.int32(e['sample_id'])
            rec[f'{tag}/ep{i}/ids'] = np.array([e['video'], e['user'], e['trace']], np.int32)
            rec[f'{tag}/ep{i}/obs'] = np.stack(e['obs'])
            rec[f'{tag}/ep{i}/act'] = np.array(e['act'], np.int32)
            rec[f'{tag}/ep{i}/rew'] = np.array(e['rew'], np.float32)
            rec[f'{tag}/ep{i}/done'] = np.array(e['done'], np.bool_)
        print(tag, [len(e['act']) for e in eps], [e['act'][:12] for e in eps])
    rec['const/video_rates'] = np.array(config.video_rates, np.int32)
    rec['const/misc'] = np.array([config.startup_download, config.chunk_length, config.max_size, config.max_throughput], np.float64)
    path = os.path.join(OUT, 'expert_reference.npz')
    np.savez_compressed(path, **rec)
    print('written', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
