"""Test helper: the reference's Jin2022 x 4G dataset tree (manifest JSONs, prediction pickles in the HMDTrace format, 4G trace pickles,
config.yml) written back out of tests/golden/env_tables_jin2022_4g.npz -- the DATA the reference's loaders hold for the train / valid / test
splits of its shipped PPO run (tools/gen_golden_tables_full.py).  With it the build's unchanged CLI (`run_mansy --config <tree>/config.yml`)
reads real tables through `EnvTables.from_dataset` on a box that has no /root/reference."""
import json
import os
import pickle

import numpy as np
import yaml

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'env_tables_jin2022_4g.npz')
SPLITS = ('train', 'valid', 'test')


def load():
    return np.load(GOLDEN)


def make_tree(root, G=None):
    G = load() if G is None else G
    ds = os.path.join(root, 'datasets')
    os.makedirs(os.path.join(ds, 'Jin2022', 'video_manifests'), exist_ok=True)
    os.makedirs(os.path.join(ds, 'network', '4G'), exist_ok=True)
    info = {}
    for sp in SPLITS:
        size, qual, vlen = G[f'{sp}/size'], G[f'{sp}/quality'], G[f'{sp}/video_len']
        for i, v in enumerate(G[f'{sp}/ids_v']):
            # chunks the reference's manifest holds: every chunk of the video (Video_Time of them); rows past it stay out of the file
            chunks = {str(c): {'size': size[i, c].tolist(), 'quality': qual[i, c].tolist()} for c in range(int(vlen[i]))}
            json.dump({'Video_Time': int(vlen[i]), 'Chunk_Count': int(vlen[i]), 'Chunk_Time': 1, 'Available_Bitrates': [1, 5, 8, 16, 35], 'Chunks': chunks},
                      open(os.path.join(ds, 'Jin2022', 'video_manifests', f'video{int(v)}.json'), 'w'))
        gt, pr, acc, vs, ve = G[f'{sp}/vp_gt'], G[f'{sp}/vp_pred'], G[f'{sp}/vp_acc'], G[f'{sp}/vp_start'], G[f'{sp}/vp_end']
        for i, (v, u) in enumerate(G[f'{sp}/ids_vp']):
            d = os.path.join(ds, 'Jin2022', 'viewports', 'prediction', f'video{int(v)}')
            os.makedirs(d, exist_ok=True)
            rows = [(int(vs[i]) + j, gt[i, j].copy(), pr[i, j].copy(), np.float64(acc[i, j])) for j in range(int(ve[i]) - int(vs[i]) + 1)]
            pickle.dump(rows, open(os.path.join(d, f'user{int(u)}.pkl'), 'wb'))
        bw, tl = G[f'{sp}/trace_bw'], G[f'{sp}/trace_len']
        for i, t in enumerate(G[f'{sp}/ids_t']):
            info[int(t)] = f'trace_{int(t)}.pkl'
            pickle.dump([(k, float(bw[i, k])) for k in range(int(tl[i]))], open(os.path.join(ds, 'network', '4G', info[int(t)]), 'wb'))
    lists = {kind: {sp: [int(x) for x in G[f'{sp}/list_{kind}']] for sp in SPLITS} for kind in ('videos', 'users', 'traces')}
    misc = G['const/misc']
    qw = G['train/qoe_w'].astype(int).tolist()
    cfg = dict(datasets_base_dir=ds + '/', raw_datasets_dir={'Jin2022': 'raw/'}, raw_network_datasets_dir={'4G': 'rawn/'},
               viewport_datasets_dir={'Jin2022': 'Jin2022/viewports/'}, video_datasets_dir={'Jin2022': 'Jin2022/video_manifests/'},
               network_datasets_dir={'4G': 'network/4G'}, results_base_dir=os.path.join(root, 'results') + '/', vp_results_dir='viewport_prediction',
               bs_results_dir='bitrate_selection', models_base_dir=os.path.join(root, 'models') + '/', vp_models_dir='viewport_prediction',
               bs_models_dir='bitrate_selection', tile_num_width=8, tile_num_height=8, tile_total_num=64, video_width=2560, video_height=1440,
               chunk_length=int(misc[1]), video_rates=[int(r) for r in G['const/video_rates']], network_info={'4G': info},
               network_split={'4G': lists['traces']}, video_split={'Jin2022': lists['videos']}, user_split={'Jin2022': lists['users']},
               qoe_split={'train': qw, 'valid': qw, 'test': [[5, 1, 3], [2, 4, 3], [1, 3, 5], [4, 4, 1]]},
               startup_download=int(misc[0]), max_size=int(misc[2]), max_throughput=int(misc[3]), past_k=8, action_space=15)
    path = os.path.join(root, 'config.yml')
    yaml.safe_dump(cfg, open(path, 'w'))
    return path


def csv_rows(text):
    return [l.split(',') for l in str(text).strip().splitlines()[1:]]
