#!/usr/bin/env python3
"""Golden vectors of the reference's sliding-window dataset (viewport_prediction/utils/load_dataset.py:6-128) on the shipped
Jin2022 traces, by importing it here: the split sizes of the full dataset, and -- on a six-trace subset that travels with the
fixture (raw [len, 3] arrays: timestamp, x, y) -- every (video, user, timestep) index of each split plus a strided sample of
items (history, current, future).  Data only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refstubs  # noqa: E402
refstubs.install()
REF = '/root/reference/viewport_prediction'
sys.path.insert(0, REF)
os.chdir(REF)
from utils.common import get_config_from_yml  # noqa: E402
from utils.load_dataset import create_dataset  # noqa: E402

INCLUDE = ['train', 'valid', 'test', 'test_seen', 'test_unseen']


def main():
    config = get_config_from_yml()
    rec = {}
    full = create_dataset('Jin2022', config, his_window=10, fut_window=10, frequency=5, sample_step=5, trim_head=15, trim_tail=15)
    rec['full_sizes'] = np.array([len(d) for d in full], np.int64)
    print('full sizes', dict(zip(INCLUDE, rec['full_sizes'])))
    vsplit = {'train': [1, 2], 'valid': [12], 'test': [14]}
    usplit = {'train': [22, 4], 'valid': [22, 27], 'test': [10]}
    for (S, T, step, th, tt), tag in (((10, 10, 5, 15, 15), 'a'), ((5, 15, 3, 8, 20), 'b')):
        sub = create_dataset('Jin2022', config, his_window=S, fut_window=T, frequency=5, sample_step=step, trim_head=th, trim_tail=tt,
                             dataset_video_split=dict(vsplit), dataset_user_split=dict(usplit))
        rec[f'{tag}/params'] = np.array([S, T, step, th, tt], np.int64)
        for name, ds in zip(INCLUDE, sub):
            rec[f'{tag}/{name}/indices'] = np.array(ds.trace_indices, np.int64).reshape(-1, 3)
            pick = list(range(0, len(ds), max(1, len(ds) // 7)))
            rec[f'{tag}/{name}/pick'] = np.array(pick, np.int64)
            for i in pick:
                h, c, f, v, u, t = ds[i]
                rec[f'{tag}/{name}/item{i}/history'], rec[f'{tag}/{name}/item{i}/current'], rec[f'{tag}/{name}/item{i}/future'] = h, c, f
        if tag == 'a':
            for v in ds.total_traces:
                for u in ds.total_traces[v]:
                    raw = np.load(os.path.join(config.viewport_datasets_dir['Jin2022'], f'video{v}', '5Hz', f'simple_5Hz_user{u}.npy'))
                    rec[f'trace/{v}/{u}'] = raw
    rec['vsplit'] = np.array(str(vsplit)); rec['usplit'] = np.array(str(usplit))
    path = os.path.join(ROOT, 'tests', 'golden', 'dataset_reference.npz')
    np.savez_compressed(path, **rec)
    print('written', path, os.path.getsize(path) // 1024, 'KiB', [k for k in rec if k.startswith('trace/')])


if __name__ == '__main__':
    main()
