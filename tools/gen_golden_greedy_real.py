#!/usr/bin/env python3
"""Greedy (argmax) closed-loop episodes of the IMPORTED reference on its real Jin2022 x 4G test split with the weights it ships:
reference nets (bitrate_selection/models/mansy.py) loaded from best_policy.pth drive the reference environment (envs/mansy_env.py, mode
'test', --test-on-seen preferences) exactly as run_mansy.py:159-175 does, except that the action is `logits.argmax()` instead of a sample
-- so the whole episode is a deterministic function of (weights, tables) and every decision can be compared bit for bit.

Recorded for 96 catalogue entries (every 15th of the 1440): per step the action, the actor's logits, the reward; per episode the CSV row the
environment wrote.  Data only (tests/golden/greedy_real_reference.npz); build container only."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refstubs  # noqa: E402
refstubs.install()
REF = '/root/reference/bitrate_selection'
sys.path.insert(0, REF)
os.chdir(REF)
from utils.common import get_config_from_yml  # noqa: E402
from envs.mansy_env import MANSYEnv  # noqa: E402
from models.mansy import Actor, FeatureNet  # noqa: E402

BASE = '/root/reference/models/bitrate_selection/mansy/Jin2022_4G/qoe0_1_2_3/' \
       'epochs_1_bs_512_lr_0.0005_gamma_0.95_seed_5_ent_0.02_useid_True_lambda_0.5_ilr_0.0001_iur_2_bc_False/'
STRIDE = 15


def main():
    config = get_config_from_yml()
    qw = config.qoe_split['train']
    pol = torch.load(BASE + 'best_policy.pth', map_location='cpu')
    actor = Actor(FeatureNet(8, 64, 5, 128, device='cpu'), 1280, 128, 15, 'cpu')
    actor.load_state_dict({k[len('actor.'):]: v for k, v in pol.items() if k.startswith('actor.')})
    log = tempfile.mktemp(suffix='.csv')
    env = MANSYEnv(config, 'Jin2022', '4G', qw, None, 0.5, log, config.startup_download, mode='test', seed=5, device='cpu')
    env.seed(5)
    entries = list(range(0, env.sample_count(), STRIDE))
    acts, logits_all, rews = [], [], []
    for e in entries:
        env.worker_id = e                       # jump the catalogue walk to entry e (reset reads worker_id, mansy_env.py:100)
        state = env.reset()
        assert env.sample_id == e
        a_ep, l_ep, r_ep = [], [], []
        done = False
        while not done:
            batch = {k: np.expand_dims(v, 0) for k, v in state.items()}       # run_mansy.py:165-166
            with torch.no_grad():
                lg, _ = actor(batch)
            a = int(lg[0].argmax())
            state, r, done, _ = env.step(a)
            a_ep.append(a)
            l_ep.append(lg[0].numpy().copy())
            r_ep.append(np.float32(r))
        acts.append(a_ep)
        logits_all.append(l_ep)
        rews.append(r_ep)
    csv = open(log).read()
    os.remove(log)
    L = np.array(logits_all, np.float32)
    top2 = np.sort(L, -1)
    gap = top2[..., -1] - top2[..., -2]
    path = os.path.join(ROOT, 'tests', 'golden', 'greedy_real_reference.npz')
    np.savez_compressed(path, entries=np.array(entries, np.int32), act=np.array(acts, np.int8), logits=L, rew=np.array(rews, np.float32), csv=np.array(csv))
    print('written', path, os.path.getsize(path) // 1024, 'KiB;', len(entries), 'episodes x', L.shape[1], 'steps; action histogram',
          np.bincount(np.array(acts).reshape(-1), minlength=15).tolist(), '; smallest top-2 logit gap', float(gap.min()), 'gaps < 1e-4:', int((gap < 1e-4).sum()))


if __name__ == '__main__':
    main()
