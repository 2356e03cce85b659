#!/usr/bin/env python3
"""CLI counterpart of bitrate_selection/run_expert.py (flags :160-177, directory naming :128-142, collect_demonstrations
:17-43, create_demonstrations :46-90, test :93-117) on the device-resident MPC expert.

  python -m mansy_immersivevideostreaming_amd.bitrate_selection.run_expert --train-dataset Jin2022 --train --valid --horizon 4
  python -m mansy_immersivevideostreaming_amd.bitrate_selection.run_expert --test-dataset Jin2022 --test --horizon 2 \\
      --qoe-test-ids 3 --test-on-seen

The reference farms the sample list out to `--proc-num` processes, each walking its share sequentially; here every sample
is its own environment and all of them take their look-ahead decision in the same three kernel launches (`--env-num` caps
the number of environments, the rest of the samples follow in strides).  Outputs keep the reference's names:
`<mode>_demonstrations.pkl`, `<mode>_log.csv` / `results.csv`, `<dataset>_cache.pkl`.  A demonstration is stored as a plain
dict {'obs': float32 [len, 780] rows in the OBS_SLICES layout, 'act': int64 [len], 'done': bool [len]} under the
reference's key (video, user, trace, qoe-weight tuple) -- the reference pickles tianshou ReplayBuffer objects, which cannot
be produced without that package; `obs_to_dict` turns a row back into the reference's observation dict."""
import argparse
import math
import os
import pickle
import random
import time

import numpy as np
import torch

from .envs.expert_env import ExpertVecEnv, dataset_cache
from .envs.mansy_env import EnvTables, generate_environment_samples, generate_environment_test_samples
from .models.mansy_trainer import write_episode_log
from .utils.common import get_config_from_yml, read_log_file


def run_samples(tables, horizon, env_num=None, record=True):
    """Every catalogue entry of `tables` played once by the expert.  Environment e plays samples e, e + N, e + 2N, ...
    Returns ({sample_id: dict(obs, act, done)} or None, [episode-log record per sample id in catalogue order], cache)."""
    n = tables.n_sample
    N = n if not env_num else min(int(env_num), n)
    venv = ExpertVecEnv(tables, N, horizon, seed=0, worker_num=N)
    target = np.array([math.ceil((n - e) / N) for e in range(N)])        # episodes environment e owes
    finished = np.zeros(N, np.int64)
    obs = venv.reset()
    obs_steps, act_steps, done_steps = [], [], []
    while (finished < target).any():
        act = venv.choose_action()
        if record:
            obs_steps.append(obs.clone())
            act_steps.append(act.clone())
        obs, _, done, _ = venv.step(act)
        d = done.cpu().numpy().astype(bool)
        done_steps.append(d)
        finished += d
    first = {}
    for r in venv.pop_episode_log():                     # environments that ran ahead may have replayed a sample: keep the first
        first.setdefault(int(r[0]), r)
    records = [first[k] for k in sorted(first) if k < n]
    demos = None
    if record:
        all_obs = torch.stack(obs_steps).cpu().numpy()   # [steps, N, 780]
        all_act = torch.stack(act_steps).cpu().numpy()
        all_done = np.stack(done_steps)
        demos = {}
        for e in range(N):
            ends = np.nonzero(all_done[:, e])[0]
            start = 0
            for k in range(int(target[e])):
                sl = slice(start, int(ends[k]) + 1)
                done = np.zeros(sl.stop - sl.start, bool)
                done[-1] = True
                demos[e + k * N] = dict(obs=all_obs[sl, e].copy(), act=all_act[sl, e].astype(np.int64), done=done)
                start = sl.stop
    return demos, records, venv.cache


def save_cache(args, config, dataset, qoe_weights, cache_path):
    """The search reads the device-resident profile directly; the pickle is written for tools that read the reference's file."""
    if (args.refresh_cache and cache_path not in save_cache.done) or not os.path.exists(cache_path):
        pickle.dump(dataset_cache(config, dataset, args.network_dataset, qoe_weights, args.device), open(cache_path, 'wb'))
        save_cache.done.add(cache_path)
        print('Save expert cache at', cache_path)


save_cache.done = set()


def create_demonstrations(args, config, qoe_weights, models_dir, demos_dir, cache_path, mode='train'):
    log_path = os.path.join(models_dir, f'{mode}_log.csv')
    demo_path = os.path.join(demos_dir, f'{mode}_demonstrations.pkl')
    if os.path.exists(log_path):
        os.remove(log_path)
    videos = config.video_split[args.train_dataset][mode]
    users = config.user_split[args.train_dataset][mode]
    traces = config.network_split[args.network_dataset][mode]
    all_samples = generate_environment_samples(videos, users, traces, qoe_weights, seed=args.seed)
    print('Total samples:', len(all_samples))
    start_time = time.time()
    tables = EnvTables.from_dataset(config, args.train_dataset, args.network_dataset, mode, qoe_weights, args.device, seed=args.seed,
                                    samples=all_samples)
    demos, records, _ = run_samples(tables, args.horizon, args.env_num)
    save_cache(args, config, args.train_dataset, qoe_weights, cache_path)
    total = {}
    for sid, (vi, ui, ti, qi) in enumerate(all_samples):
        key = (videos[vi], users[ui], traces[ti], tuple(int(w) for w in qoe_weights[qi]))
        total[key] = demos[sid]
        print(f'Demonstration of video-{key[0]}, user-{key[1]}, trace-{key[2]}, qoe-weight-{key[3]} done!')
    write_episode_log(log_path, tables, qoe_weights, records)
    pickle.dump(total, open(demo_path, 'wb'))
    print(f'Create {len(all_samples)} demonstrations, saved at {demo_path}, cost {round((time.time() - start_time) / 3600, 2)}h')
    return total


def test(args, config, qoe_weights, results_dir, demos_dir, cache_path):
    log_path = os.path.join(results_dir, 'results.csv')
    if os.path.exists(log_path):
        os.remove(log_path)
    videos = config.video_split[args.test_dataset]['test']
    users = config.user_split[args.test_dataset]['test']
    traces = config.network_split[args.network_dataset]['test']
    all_samples = generate_environment_test_samples(videos, users, traces, qoe_weights)
    tables = EnvTables.from_dataset(config, args.test_dataset, args.network_dataset, 'test', qoe_weights, args.device, seed=args.seed,
                                    samples=all_samples)
    _, records, _ = run_samples(tables, args.horizon, args.env_num, record=False)
    save_cache(args, config, args.test_dataset, qoe_weights, cache_path)
    write_episode_log(log_path, tables, qoe_weights, records)          # catalogue order = the reference's sequential order
    return read_log_file(log_path, verbose=args.verbose_table)


def run(args, config):
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed_all(args.seed)
    random.seed(args.seed)
    if args.qoe_train_ids is None:
        args.qoe_train_ids = list(range(len(config.qoe_split['train'])))
    split = 'train' if args.test_on_seen else 'test'
    if args.qoe_test_ids is None:
        args.qoe_test_ids = list(range(len(config.qoe_split[split])))
    pair = args.train_dataset + '_' + args.network_dataset
    models_dir = os.path.join(config.bs_models_dir, args.model, pair, 'qoe' + '_'.join(map(str, args.qoe_train_ids)))
    demos_dir = models_dir
    seen = 'seen_qoe' if args.test_on_seen else 'unseen_qoe'
    results_dir = os.path.join(config.bs_results_dir, args.model, args.test_dataset + '_' + args.network_dataset,
                               seen + '_'.join(map(str, args.qoe_test_ids)))
    train_cache_path = os.path.join(config.bs_models_dir, args.model, f'{args.train_dataset}_cache.pkl')
    test_cache_path = os.path.join(config.bs_models_dir, args.model, f'{args.test_dataset}_cache.pkl')
    for d in (models_dir, demos_dir, results_dir):
        os.makedirs(d, exist_ok=True)
    if args.train:
        qoe_weights = [config.qoe_split['train'][i] for i in args.qoe_train_ids]
        print('Training QoE weights:', qoe_weights)
        create_demonstrations(args, config, qoe_weights, models_dir, demos_dir, train_cache_path, 'train')
    if args.valid:
        qoe_weights = [config.qoe_split['valid'][i] for i in args.qoe_train_ids]
        print('Validating QoE weights:', qoe_weights)
        create_demonstrations(args, config, qoe_weights, models_dir, demos_dir, train_cache_path, 'valid')
    if args.test:
        qoe_weights = [config.qoe_split[split][i] for i in args.qoe_test_ids]
        print('Testing QoE weights:', qoe_weights)
        test(args, config, qoe_weights, results_dir, demos_dir, test_cache_path)


def build_parser():
    parser = argparse.ArgumentParser(description='MPC expert demonstrations on MI355X')
    parser.add_argument('--logdir', type=str, default='log_tensorboard')
    parser.add_argument('--model', type=str, default='expert')
    for flag in ('--train', '--valid', '--test', '--test-on-seen', '--refresh-cache', '--verbose-table'):
        parser.add_argument(flag, action='store_true')
    parser.add_argument('--train-dataset', type=str, default='Wu2017')
    parser.add_argument('--test-dataset', type=str, default='Wu2017')
    parser.add_argument('--network-dataset', type=str, default='4G')
    parser.add_argument('--qoe-train-ids', type=int, nargs='*')
    parser.add_argument('--qoe-test-ids', type=int, nargs='*')
    parser.add_argument('--proc-num', type=int, help='accepted for compatibility; the search is batched over environments instead')
    parser.add_argument('--horizon', type=int, default=4, help='The horizon for expert to look ahead')
    parser.add_argument('--seed', type=int, default=1)
    # additions of this build
    parser.add_argument('--device', type=str, default='cuda:0')
    parser.add_argument('--config', type=str, default=None)
    parser.add_argument('--env-num', type=int, default=None, help='environments searched per launch (default: one per sample)')
    return parser


def main(argv=None):
    args = build_parser().parse_known_args(argv)[0]
    print(args)
    run(args, get_config_from_yml(args.config))


if __name__ == '__main__':
    main()
