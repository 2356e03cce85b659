#!/usr/bin/env python3
"""CLI counterpart of viewport_prediction/run_models.py (same flags :158-192, same checkpoint/result file names
:18-19,126-127, same train / validation / test flow :17-97) on the HIP engine.

  python -m mansy_immersivevideostreaming_amd.viewport_prediction.run_models --model mtio --train --test \\
      --train-dataset Jin2022 --test-dataset Jin2022 --his-window 10 --fut-window 10 --bs 512 --device cuda:0 [--config ../config.yml]

Differences that do not change results: batches come from the device-resident `DeviceLoader` (same order as
DataLoader under the same torch seed); each training iteration is ONE fused engine call (`model.train_step`);
`--config` is an extra optional flag (default: the reference's '../config.yml')."""
import argparse
import os
import random
import sys

import numpy as np
import torch

from .models import FusedAdamW, ViewportTransformerMTIO
from .utils.common import get_config_from_yml, mean_square_error
from .utils.load_dataset import DeviceLoader, create_dataset
from .utils.results import Results


class ConsoleLogger:
    def __init__(self, *files):
        self.files = files

    def write(self, obj):
        for f in self.files:
            f.write(obj)
            f.flush()

    def flush(self):
        for f in self.files:
            f.flush()


def train(args, model, loader_train, loader_valid, models_dir, file_prefix):
    checkpoint_path = os.path.join(models_dir, file_prefix + '_checkpoint.pth')
    best_model_path = os.path.join(models_dir, file_prefix + '_best_model.pth')
    if args.resume:
        assert args.resume_path is not None
        model.load_state_dict(torch.load(args.resume_path, map_location=args.device))
        print('Resume model for training from:', args.resume_path)
    train_size, valid_size = len(loader_train), len(loader_valid)
    optimizer = FusedAdamW(model, lr=args.lr)           # torch.optim.AdamW defaults (run_models.py:29)
    best_valid_mse, best_epoch = float('inf'), 0
    print(f'Training {args.model} on {args.train_dataset} - bs: {args.bs} - lr: {args.lr} - seed: {args.seed}')
    for epoch in range(args.epochs):
        print(f"Epoch {epoch + 1}/{args.epochs}\n-------------------------------")
        model.train()
        total = []
        for batch, (history, current, future, video, user, timestep) in enumerate(loader_train):
            train_loss = model.train_step(history, current, future, optimizer)
            total.append(train_loss)
            if args.verbose_steps:
                print(f"\rTrain: [{batch + 1}/{train_size}] - train_loss: {train_loss.item():>9f}", end='')
        print(f'\rTrain: mean train loss: {(torch.stack(total).mean().item()):>9f}')
        if epoch % args.epochs_per_valid == 0:
            model.eval()
            with torch.no_grad():
                mse = []
                for history, current, future, video, user, timestep in loader_valid:
                    pred = model.sample(history, current)
                    mse.append(torch.mean(mean_square_error(pred, future)).item())
                mse = np.sum(mse) / valid_size
                print(f'Valid: mean square error: {mse:>9f}')
                torch.save(model.state_dict(), checkpoint_path)
                print('Checkpoint saved at', checkpoint_path)
                if best_valid_mse > mse:
                    best_valid_mse, best_epoch = mse, epoch + 1
                    torch.save(model.state_dict(), best_model_path)
                print(f'Best model (epoch {best_epoch}, loss {best_valid_mse}) saved at', best_model_path)


def test(args, config, model, loader_seen, loader_unseen, models_dir, results_dir, file_prefix):
    best_model_path = os.path.join(models_dir, file_prefix + '_best_model.pth')
    notebook = Results(args.model, dimension=2, fut_window=args.fut_window, dataset_frequency=args.dataset_frequency, output_dir=results_dir,
                       mse=True, accuracy=True, config=config)
    model.load_state_dict(torch.load(best_model_path, map_location=args.device))
    print('Load model from', best_model_path)
    print(f'Testing {args.model} on {args.test_dataset} - seed: {args.seed}')
    with torch.no_grad():
        model.eval()
        for tag, loader in (('seen', loader_seen), ('unseen', loader_unseen)):
            print(f'On {tag} viewing patterns.')
            for history, current, future, video, user, timesteps in loader:
                pred = model.sample(history, current)
                notebook.record(history.shape[0], pred, future, video, user, timesteps)
            notebook.write(log=True, label=file_prefix + f'_{tag}_')
            notebook.reset()


def create_model(model_name, fut_window, hidden_dim, block_num, device, seed):
    if model_name != 'mtio':
        raise ValueError("only --model mtio is on the MI355X path (the sklearn 'regression' baseline is out of scope)")
    return ViewportTransformerMTIO(in_channel=2, fut_window=fut_window, d_model=hidden_dim, dim_feedforward=hidden_dim,
                                   num_encoder_layers=block_num, num_decoder_layers=block_num, device=device, seed=seed)


def run(args, config):
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed_all(args.seed)
    random.seed(args.seed)
    models_dir = os.path.join(config.vp_models_dir, args.model, args.train_dataset, f'{args.dataset_frequency}Hz')
    results_dir = os.path.join(config.vp_results_dir, args.model, args.test_dataset, f'{args.dataset_frequency}Hz')
    os.makedirs(models_dir, exist_ok=True)
    os.makedirs(results_dir, exist_ok=True)
    file_prefix = f'his_{args.his_window}_fut_{args.fut_window}_hid_{args.hidden_dim}_ss_{args.sample_step}_' \
                  f'epochs_{args.epochs}_bs_{args.bs}_lr_{args.lr}_seed_{args.seed}'
    model = create_model(args.model, args.fut_window, args.hidden_dim, args.block_num, args.device, args.seed).to(args.device)
    common = dict(his_window=args.his_window, fut_window=args.fut_window, frequency=args.dataset_frequency, sample_step=args.sample_step,
                  trim_head=args.trim_head, trim_tail=args.trim_tail)
    if args.train:
        console_log = open(os.path.join(results_dir, file_prefix + 'console.log'), 'w')
        sys.stdout = ConsoleLogger(sys.__stdout__, console_log)
        ds_train, ds_valid = create_dataset(args.train_dataset, config, include=['train', 'valid'], **common)
        train(args, model, DeviceLoader(ds_train, args.bs, shuffle=True, device=args.device),
              DeviceLoader(ds_valid, args.bs, shuffle=False, device=args.device), models_dir, file_prefix)
    if args.test:
        ds_seen, ds_unseen = create_dataset(args.test_dataset, config, include=['test_seen', 'test_unseen'], **common)
        test(args, config, model, DeviceLoader(ds_seen, args.bs, device=args.device), DeviceLoader(ds_unseen, args.bs, device=args.device),
             models_dir, results_dir, file_prefix)


def build_parser():
    p = argparse.ArgumentParser(description='Process the input parameters to train the network.')
    p.add_argument('--train', action='store_true')
    p.add_argument('--test', action='store_true')
    p.add_argument('--device', action='store', dest='device', default='cuda:0')
    p.add_argument('--model', action='store', dest='model', default='mtio')
    p.add_argument('--hidden-dim', type=int, default=512)
    p.add_argument('--block-num', type=int, default=2)
    p.add_argument('--compile', action='store_true', dest='compile', help='accepted and ignored (no tracing compiler on this path)')
    p.add_argument('--resume', action='store_true', dest='resume')
    p.add_argument('--resume-path', type=str, dest='resume_path')
    p.add_argument('--train-dataset', action='store', dest='train_dataset')
    p.add_argument('--test-dataset', action='store', dest='test_dataset')
    p.add_argument('--his-window', action='store', dest='his_window', type=int, default=5)
    p.add_argument('--fut-window', action='store', dest='fut_window', type=int, default=15)
    p.add_argument('--trim-head', action='store', dest='trim_head', type=int)
    p.add_argument('--trim-tail', action='store', dest='trim_tail', type=int)
    p.add_argument('--dataset-frequency', action='store', dest='dataset_frequency', type=int)
    p.add_argument('--sample-step', action='store', dest='sample_step', type=int)
    p.add_argument('--epochs', action='store', dest='epochs', type=int, default=200)
    p.add_argument('--epochs-per-valid', action='store', dest='epochs_per_valid', type=int, default=3)
    p.add_argument('--lr', action='store', dest='lr', type=float, default=1e-4)
    p.add_argument('--weight-decay', action='store', dest='weight_decay', type=float, help='parsed and unused, as in the reference (:190)')
    p.add_argument('--bs', action='store', dest='bs', type=int)
    p.add_argument('--seed', action='store', dest='seed', type=int, default=5)
    p.add_argument('--config', type=str, default=None, help="path of config.yml (default '../config.yml' like the reference)")
    p.add_argument('--verbose-steps', action='store_true', help='print the loss of every step (forces a device sync per step)')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    config = get_config_from_yml(args.config)
    args.trim_head = config.trim_head if args.trim_head is None else args.trim_head
    args.trim_tail = config.trim_tail if args.trim_tail is None else args.trim_tail
    args.dataset_frequency = config.frequency if args.dataset_frequency is None else args.dataset_frequency
    args.sample_step = config.sample_step if args.sample_step is None else args.sample_step
    print(args)
    run(args, config)


if __name__ == '__main__':
    main()
