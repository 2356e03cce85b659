#!/usr/bin/env python3
"""Golden trajectories of the reference streaming environment (bitrate_selection/envs/mansy_env.py
and everything below it), produced by importing and running the reference in this container with
stubs for gym/munch/prettytable only.  Data only: the tables (manifest rows, viewport maps, network
traces) of the visited episodes + for every step the action, reward, done flag and the full
observation dict flattened into the 779-float layout of oracle/env.c.

Also records the episode enumerations (generate_environment_samples / _test_samples) and
allocate_tile_rates for all 15 actions x a set of predicted viewports.
"""
import json
import os
import pickle
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refstubs  # noqa: E402
refstubs.install()
REF = '/root/reference/bitrate_selection'
sys.path.insert(0, REF)
os.chdir(REF)          # the reference resolves '../config.yml' relative to its own directory
from utils.common import (get_config_from_yml, generate_environment_samples, generate_environment_test_samples,  # noqa: E402
                          allocate_tile_rates, action2rates)
from envs.mansy_env import MANSYEnv  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
KEYS = [('throughput', 8), ('next_chunk_size', 320), ('next_chunk_quality', 320), ('pred_viewport', 64), ('viewport_acc', 8),
        ('past_viewport_qualities', 8), ('past_quality_variances', 8), ('past_rebuffering', 8), ('buffer', 1), ('qoe_weight', 3),
        ('action_one_hot', 15), ('rates_inside', 8), ('rates_outside', 8)]


def flat_obs(state):
    out = []
    for k, n in KEYS:
        a = np.asarray(state[k], dtype=np.float32).reshape(-1)
        assert a.size == n, (k, a.size)
        out.append(a)
    return np.concatenate(out)


def run(config, mode, seed, worker_num, use_identifier, n_episodes, qoe_weights, act_seed):
    log = tempfile.mktemp(suffix='.csv')
    env = MANSYEnv(config, 'Jin2022', '4G', qoe_weights, None, 0.5, log, config.startup_download, mode=mode, seed=seed,
                   worker_num=worker_num, device='cpu', use_identifier=use_identifier)
    env.seed(seed)
    rs = np.random.RandomState(act_seed)
    eps = []
    for _ in range(n_episodes):
        st = env.reset()
        rec = dict(sample_id=env.sample_id, video=env.current_video, user=env.current_user, trace=env.current_trace,
                   obs=[flat_obs(st)], act=[], rew=[], done=[])
        over = False
        while not over:
            a = int(rs.randint(0, 15))
            st, r, over, _ = env.step(a)
            rec['act'].append(a)
            rec['rew'].append(np.float32(r))
            rec['done'].append(bool(over))
            rec['obs'].append(flat_obs(st))
        eps.append(rec)
    csv = open(log).read()
    os.remove(log)
    return env, eps, csv


def build_tables(config, env, eps):
    vids = sorted({e['video'] for e in eps})
    vps = sorted({(e['video'], e['user']) for e in eps})
    trs = sorted({e['trace'] for e in eps})
    n_chunk = 60
    size = np.zeros((len(vids), n_chunk, 5, 64), np.int32)
    qual = np.zeros((len(vids), n_chunk, 5, 64), np.float32)
    vlen = np.zeros(len(vids), np.int32)
    for i, v in enumerate(vids):
        m = json.load(open(os.path.join(config.video_datasets_dir['Jin2022'], f'video{v}.json')))
        vlen[i] = m['Video_Time']
        for c, info in m['Chunks'].items():
            size[i, int(c)] = np.array(info['size'], np.int32)
            qual[i, int(c)] = np.array(info['quality'], np.float32)
    nvc = 64
    gt = np.zeros((len(vps), nvc, 64), np.uint8)
    pr = np.zeros((len(vps), nvc, 64), np.uint8)
    acc = np.zeros((len(vps), nvc), np.float64)
    vstart = np.zeros(len(vps), np.int32)
    vend = np.zeros(len(vps), np.int32)
    for i, (v, u) in enumerate(vps):
        pk = pickle.load(open(os.path.join(config.viewport_datasets_dir['Jin2022'], 'prediction', f'video{v}', f'user{u}.pkl'), 'rb'))
        vstart[i], vend[i] = pk[0][0], pk[-1][0]
        for j, p in enumerate(pk):
            gt[i, j], pr[i, j], acc[i, j] = p[1], p[2], p[3]
    tl = np.zeros(len(trs), np.int32)
    traces = []
    for t in trs:
        tr = pickle.load(open(os.path.join(config.network_datasets_dir['4G'], config.network_info['4G'][t]), 'rb'))
        traces.append(np.array([x[1] for x in tr], np.float64))
    tmax = max(len(t) for t in traces)
    bw = np.zeros((len(trs), tmax), np.float64)
    for i, t in enumerate(traces):
        bw[i, :len(t)] = t
        tl[i] = len(t)
    # catalogue in the env's enumeration order; unvisited samples point at slot -1
    samples = np.full((len(env.samples), 4), -1, np.int32)
    for sid, (vi, ui, ti, qi) in enumerate(env.samples):
        v, u, t = env.videos[vi], env.users[ui], env.traces[ti]
        if v in vids and (v, u) in vps and t in trs:
            samples[sid] = (vids.index(v), vps.index((v, u)), trs.index(t), qi)
    return dict(size=size, quality=qual, video_len=vlen, vp_gt=gt, vp_pred=pr, vp_acc=acc, vp_start=vstart, vp_end=vend,
                trace_bw=bw, trace_len=tl, samples=samples)


def main():
    config = get_config_from_yml()
    qoe_train = config.qoe_split['train']
    rec = {}
    for tag, mode, seed, wn, use_id, n_ep, act_seed in [('train_id', 'train', 5, 1, True, 5, 11), ('valid_w3', 'valid', 4, 3, False, 3, 12),
                                                        ('train_noid', 'train', 2, 1, False, 2, 13)]:
        env, eps, csv = run(config, mode, seed, wn, use_id, n_ep, qoe_train, act_seed)
        tb = build_tables(config, env, eps)
        for k, v in tb.items():
            rec[f'{tag}/{k}'] = v
        rec[f'{tag}/meta'] = np.array([seed, wn, int(use_id), n_ep, int(mode == 'train' and use_id)], np.int32)
        rec[f'{tag}/qoe_w'] = np.array(qoe_train, np.float32)
        rec[f'{tag}/csv'] = np.array(csv)
        for i, e in enumerate(eps):
            rec[f'{tag}/ep{i}/sample_id'] = np.int32(e['sample_id'])
            rec[f'{tag}/ep{i}/ids'] = np.array([e['video'], e['user'], e['trace']], np.int32)
            rec[f'{tag}/ep{i}/obs'] = np.stack(e['obs'])
            rec[f'{tag}/ep{i}/act'] = np.array(e['act'], np.int32)
            rec[f'{tag}/ep{i}/rew'] = np.array(e['rew'], np.float32)
            rec[f'{tag}/ep{i}/done'] = np.array(e['done'], np.bool_)
    rec['const/video_rates'] = np.array(config.video_rates, np.int32)
    rec['const/misc'] = np.array([config.startup_download, config.chunk_length, config.max_size, config.max_throughput], np.float64)
    # episode enumerations
    for mode in ('train', 'valid', 'test'):
        vl, ul, tl_ = config.video_split['Jin2022'][mode], config.user_split['Jin2022'][mode], config.network_split['4G'][mode]
        fn = generate_environment_test_samples if mode == 'test' else generate_environment_samples
        rec[f'enum/{mode}'] = np.array(fn(vl, ul, tl_, config.qoe_split[mode if mode != 'valid' else 'valid']), np.int32)
        rec[f'enum/{mode}_lens'] = np.array([len(vl), len(ul), len(tl_), len(config.qoe_split[mode])], np.int32)
    # allocate_tile_rates known answers
    rs = np.random.RandomState(0)
    vps, outs = [], []
    for k in range(40):
        vp = np.zeros((8, 8), np.float32)
        if k > 0:
            r0, c0, h, w = rs.randint(0, 8), rs.randint(0, 8), rs.randint(1, 5), rs.randint(1, 6)
            for dr in range(h):
                for dc in range(w):
                    vp[(r0 + dr) % 8, (c0 + dc) % 8] = 1
            if k % 5 == 0:
                vp[rs.randint(0, 8), rs.randint(0, 8)] = 1
        vps.append(vp.reshape(-1))
        row = []
        for a in range(15):
            rin, rout = action2rates(a)
            ver, _ = allocate_tile_rates(rin, rout, vp.reshape(-1), config.video_rates, 8, 8)
            row.append(ver.astype(np.int32))
        outs.append(np.stack(row))
    rec['alloc/pred_viewport'] = np.stack(vps)
    rec['alloc/versions'] = np.stack(outs)          # [40,15,64]
    path = os.path.join(OUT, 'env_reference.npz')
    np.savez_compressed(path, **rec)
    print('written', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
