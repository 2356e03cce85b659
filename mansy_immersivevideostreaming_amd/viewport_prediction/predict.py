#!/usr/bin/env python3
"""CLI counterpart of viewport_prediction/predict.py (:15-65, flags :115-143): sample() over every (video, user), per
1-second chunk OR of the tile maps of the first `dataset_frequency` future steps for ground truth and prediction, IoU,
and the `(chunk, gt u8[64], pred u8[64], iou)` pickle + CSV that bitrate_selection/simulators/hmdtrace.py reads.
Tile maps / OR / IoU run on the device; tile size is derived as video_size // tile_num (the reference reads
config.tile_width / tile_height, which config.yml does not define -- predict.py:41-45)."""
import argparse
import os
import pickle
import random

import numpy as np
import torch

from .. import kernels
from .run_models import create_model
from .utils.common import get_config_from_yml
from .utils.load_dataset import DeviceLoader, create_dataset


def chunk_maps(config, merge, freq):
    """merge [n, T, 4] (gt xy, pred xy) on the device -> (gt maps, pred maps, iou) for n samples."""
    first = merge[:, :freq].contiguous()
    g = kernels.tilemap(first[..., 0:2].contiguous(), config.video_width, config.video_height, config.tile_num_width, config.tile_num_height)
    p = kernels.tilemap(first[..., 2:4].contiguous(), config.video_width, config.video_height, config.tile_num_width, config.tile_num_height)
    g, p = kernels.tilemap_or_groups(g.reshape(-1), freq), kernels.tilemap_or_groups(p.reshape(-1), freq)
    return g, p, kernels.tilemap_iou(g, p)


def predict(args, config, model, videos, users, loader, results_dir, model_path):
    if args.model != 'regression':          # linear regression has no weights to load (predict.py:16)
        model.load_state_dict(torch.load(model_path, map_location=args.device))
        print('Successfully loaded model from', model_path)
    results = {(video, user): [] for video in videos for user in users}
    with torch.no_grad():
        model.eval()
        for history, current, future, video, user, timesteps in loader:
            pred = model.sample(history, current)
            g, p, iou = chunk_maps(config, torch.cat([future, pred], dim=-1), args.dataset_frequency)
            g, p, iou = g.cpu().numpy().view(np.uint64), p.cpu().numpy().view(np.uint64), iou.cpu().numpy()
            for i in range(history.shape[0]):
                results[int(video[i]), int(user[i])].append((g[i], p[i], iou[i]))
    shift = np.arange(config.tile_total_num, dtype=np.uint64)
    for (video, user), value in results.items():
        rows = []
        for i, (g, p, acc) in enumerate(value):
            rows.append((i + args.trim_head // args.dataset_frequency, ((g >> shift) & np.uint64(1)).astype(np.uint8),
                         ((p >> shift) & np.uint64(1)).astype(np.uint8), np.float64(acc)))
        base_dir = os.path.join(results_dir, f'video{video}')
        os.makedirs(base_dir, exist_ok=True)
        pickle.dump(rows, open(os.path.join(base_dir, f'user{user}.pkl'), 'wb'))
        with open(os.path.join(base_dir, f'user{user}.csv'), 'w', encoding='utf-8') as file:
            file.write('chunk,gt,pred,accuracy\n')
            for r in rows:
                file.write(f"{r[0]},{','.join(map(str, list(r[1])))},{','.join(map(str, list(r[2])))},{r[3]}\n")


def run(args, config):
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    random.seed(args.seed)
    results_dir = args.output_dir or os.path.join(config.viewport_datasets_dir[args.dataset], 'prediction')
    os.makedirs(results_dir, exist_ok=True)
    model = create_model(args.model, args.fut_window, args.hidden_dim, args.block_num, args.device, args.seed).to(args.device)
    videos, users = [], []
    for split in ['train', 'valid', 'test']:
        videos += config.video_split[args.dataset][split]
        users += config.user_split[args.dataset][split]
    videos, users = list(set(videos)), list(set(users))
    dataset = create_dataset(args.dataset, config, his_window=args.his_window, fut_window=args.fut_window, sample_step=args.sample_step,
                             frequency=args.dataset_frequency, trim_head=args.trim_head, trim_tail=args.trim_tail,
                             dataset_video_split={'merge': videos}, dataset_user_split={'merge': users}, include=['merge'])[0]
    predict(args, config, model, videos, users, DeviceLoader(dataset, args.bs, device=args.device), results_dir, args.model_path)


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--device', default='cuda:0')
    p.add_argument('--model', default='mtio')
    p.add_argument('--hidden-dim', type=int, default=512)
    p.add_argument('--block-num', type=int, default=2)
    p.add_argument('--model-path', dest='model_path')
    p.add_argument('--compile', action='store_true')
    p.add_argument('--dataset')
    p.add_argument('--his-window', dest='his_window', type=int)
    p.add_argument('--fut-window', dest='fut_window', type=int)
    p.add_argument('--trim-head', dest='trim_head', type=int)
    p.add_argument('--trim-tail', dest='trim_tail', type=int)
    p.add_argument('--dataset-frequency', dest='dataset_frequency', type=int)
    p.add_argument('--sample-step', dest='sample_step', type=int)
    p.add_argument('--bs', type=int, default=512)
    p.add_argument('--seed', type=int, default=5)
    p.add_argument('--config', type=str, default=None)
    p.add_argument('--output-dir', type=str, default=None, help="default: <viewport_datasets_dir>/prediction like the reference")
    args = p.parse_args(argv)
    config = get_config_from_yml(args.config)
    args.trim_head = config.trim_head if args.trim_head is None else args.trim_head
    args.trim_tail = config.trim_tail if args.trim_tail is None else args.trim_tail
    args.dataset_frequency = config.frequency if args.dataset_frequency is None else args.dataset_frequency
    args.sample_step = config.sample_step if args.sample_step is None else args.sample_step
    print(args)
    run(args, config)


if __name__ == '__main__':
    main()
