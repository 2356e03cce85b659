#!/usr/bin/env python3
"""Viewport-prediction driver on the HIP engine -- the counterpart of the reference's `run_models.py`
(flags: viewport_prediction/run_models.py:158-192; file naming :18-19,126-127; train/validate/test flow :17-97).

    python -m mansy_immersivevideostreaming_amd.viewport_prediction.run_models --model mtio --train --test \\
        --train-dataset Jin2022 --test-dataset Jin2022 --his-window 10 --fut-window 10 --bs 512 --device cuda:0

What differs from the reference's script, none of it visible in the results: mini-batches are gathered on the device from an
HBM-resident trace table (same order as torch's DataLoader under the same seed), one training iteration is a single fused
engine call, `--config` may point at a config.yml elsewhere (default `../config.yml`, as in the reference), `--compile` is
accepted and ignored, per-step loss printing is opt-in (`--verbose-steps`) because it forces a device sync per step.
"""
import argparse
import os
import random
import sys

import numpy as np
import torch

from .. import _lib
from .models import FusedAdamW, LinearRegression, ViewportTransformerMTIO
from .utils.common import get_config_from_yml, mean_square_error
from .utils.load_dataset import DeviceLoader, create_dataset
from .utils.results import Results

# (flag, kwargs) -- the reference's command line, as data
_FLAGS = [
    ('--train', dict(action='store_true')), ('--test', dict(action='store_true')),
    ('--device', dict(default='cuda:0')), ('--model', dict(default='mtio')),
    ('--hidden-dim', dict(type=int, default=512)), ('--block-num', dict(type=int, default=2)),
    ('--compile', dict(action='store_true')), ('--resume', dict(action='store_true')), ('--resume-path', dict(type=str)),
    ('--train-dataset', dict()), ('--test-dataset', dict()),
    ('--his-window', dict(type=int, default=5)), ('--fut-window', dict(type=int, default=15)),
    ('--trim-head', dict(type=int)), ('--trim-tail', dict(type=int)), ('--dataset-frequency', dict(type=int)),
    ('--sample-step', dict(type=int)), ('--epochs', dict(type=int, default=200)), ('--epochs-per-valid', dict(type=int, default=3)),
    ('--lr', dict(type=float, default=1e-4)), ('--weight-decay', dict(type=float)),       # parsed, unused (reference :190)
    ('--bs', dict(type=int)), ('--seed', dict(type=int, default=5)),
    ('--config', dict(type=str, default=None)), ('--verbose-steps', dict(action='store_true')),
    # addition of this build: precision of the dense products (f32 = exact fp32 MFMA, the parity mode; bf16x3 / bf16x6 = split-bf16 MFMA)
    ('--precision', dict(choices=('f32', 'bf16', 'bf16x3', 'bf16x6'), default='f32')),
]
_CONFIG_DEFAULTS = {'trim_head': 'trim_head', 'trim_tail': 'trim_tail', 'dataset_frequency': 'frequency', 'sample_step': 'sample_step'}


class _Tee:
    def __init__(self, *sinks):
        self.sinks = sinks

    def write(self, text):
        for s in self.sinks:
            s.write(text)
            s.flush()

    def flush(self):
        for s in self.sinks:
            s.flush()


def create_model(model_name, fut_window, hidden_dim, block_num, device, seed):
    """run_models.py:99-106: the MTIO Transformer, or the linear-regression comparison baseline (one device launch per batch instead
    of scikit-learn per trajectory)."""
    if model_name == 'regression':
        return LinearRegression(fut_window=fut_window, device=device)
    if model_name != 'mtio':
        raise ValueError(f"unknown model '{model_name}': the reference knows 'mtio' and 'regression' (run_models.py:110)")
    return ViewportTransformerMTIO(in_channel=2, fut_window=fut_window, d_model=hidden_dim, dim_feedforward=hidden_dim,
                                   num_encoder_layers=block_num, num_decoder_layers=block_num, device=device, seed=seed)


class Session:
    """One invocation: directories, file stem, model, and the train / test phases."""

    def __init__(self, args, config):
        self.args, self.config = args, config
        a = args
        self.models_dir = os.path.join(config.vp_models_dir, a.model, str(a.train_dataset), f'{a.dataset_frequency}Hz')
        self.results_dir = os.path.join(config.vp_results_dir, a.model, str(a.test_dataset), f'{a.dataset_frequency}Hz')
        for d in (self.models_dir, self.results_dir):
            os.makedirs(d, exist_ok=True)
        self.stem = (f'his_{a.his_window}_fut_{a.fut_window}_hid_{a.hidden_dim}_ss_{a.sample_step}_epochs_{a.epochs}_bs_{a.bs}'
                     f'_lr_{a.lr}_seed_{a.seed}')
        self.model = create_model(a.model, a.fut_window, a.hidden_dim, a.block_num, a.device, a.seed).to(a.device)
        if hasattr(self.model, 'precision'):                      # the model carries its precision into every engine call (no process-wide mode)
            self.model.precision = getattr(a, 'precision', 'f32')
        self.window = dict(his_window=a.his_window, fut_window=a.fut_window, frequency=a.dataset_frequency, sample_step=a.sample_step,
                           trim_head=a.trim_head, trim_tail=a.trim_tail)

    def path(self, kind):
        return os.path.join(self.models_dir, f'{self.stem}_{kind}.pth')

    def loaders(self, dataset, splits, shuffle_first=False):
        sets = create_dataset(dataset, self.config, include=list(splits), **self.window)
        return [DeviceLoader(ds, self.args.bs, shuffle=(shuffle_first and i == 0), device=self.args.device) for i, ds in enumerate(sets)]

    def validate(self, loader):
        """Mean over batches of the mean periodic MSE of sample() (run_models.py:50-58)."""
        self.model.eval()
        per_batch = []
        with torch.no_grad():
            for history, current, future, *_ in loader:
                per_batch.append(torch.mean(mean_square_error(self.model.sample(history, current), future)).item())
        return float(np.sum(per_batch) / len(loader))

    def train(self):
        a, model = self.args, self.model
        if a.resume:
            if a.resume_path is None:
                raise SystemExit('--resume needs --resume-path')
            model.load_state_dict(torch.load(a.resume_path, map_location=a.device))
            print('Resume model for training from:', a.resume_path)
        train_loader, valid_loader = self.loaders(a.train_dataset, ('train', 'valid'), shuffle_first=True)
        optimizer = FusedAdamW(model, lr=a.lr)                # AdamW with torch defaults (run_models.py:29)
        best = (float('inf'), 0)
        print(f'Training {a.model} on {a.train_dataset} - bs: {a.bs} - lr: {a.lr} - seed: {a.seed}')
        for epoch in range(a.epochs):
            print(f'Epoch {epoch + 1}/{a.epochs}\n' + '-' * 31)
            model.train()
            losses = []
            for step, (history, current, future, *_) in enumerate(train_loader):
                losses.append(model.train_step(history, current, future, optimizer))
                if a.verbose_steps:
                    print(f'\rTrain: [{step + 1}/{len(train_loader)}] - train_loss: {losses[-1].item():>9f}', end='')
            print(f'\rTrain: mean train loss: {torch.stack(losses).mean().item():>9f}')
            if epoch % a.epochs_per_valid:
                continue
            mse = self.validate(valid_loader)
            print(f'Valid: mean square error: {mse:>9f}')
            torch.save(model.state_dict(), self.path('checkpoint'))
            print('Checkpoint saved at', self.path('checkpoint'))
            if mse < best[0]:
                best = (mse, epoch + 1)
                torch.save(model.state_dict(), self.path('best_model'))
            print(f'Best model (epoch {best[1]}, loss {best[0]}) saved at', self.path('best_model'))

    def test(self):
        a, model = self.args, self.model
        if a.model != 'regression':          # linear regression has no weights to load (run_models.py:75)
            model.load_state_dict(torch.load(self.path('best_model'), map_location=a.device))
            print('Load model from', self.path('best_model'))
        print(f'Testing {a.model} on {a.test_dataset} - seed: {a.seed}')
        notebook = Results(a.model, dimension=2, fut_window=a.fut_window, dataset_frequency=a.dataset_frequency, output_dir=self.results_dir,
                           mse=True, accuracy=True, config=self.config)
        model.eval()
        with torch.no_grad():
            for tag, loader in zip(('seen', 'unseen'), self.loaders(a.test_dataset, ('test_seen', 'test_unseen'))):
                print(f'On {tag} viewing patterns.')
                for history, current, future, video, user, timesteps in loader:
                    notebook.record(history.shape[0], model.sample(history, current), future, video, user, timesteps)
                notebook.write(log=True, label=f'{self.stem}_{tag}_')
                notebook.reset()


def run(args, config):
    for seeder in (np.random.seed, torch.manual_seed, torch.cuda.manual_seed_all, random.seed):
        seeder(args.seed)
    session = Session(args, config)
    if args.train:
        log = open(os.path.join(session.results_dir, session.stem + 'console.log'), 'w')
        sys.stdout = _Tee(sys.__stdout__, log)
        session.train()
    if args.test:
        session.test()


def build_parser():
    parser = argparse.ArgumentParser(description='MANSY viewport prediction on MI355X')
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    config = get_config_from_yml(args.config)
    for attr, key in _CONFIG_DEFAULTS.items():                 # config.yml supplies what the command line leaves out (:199-203)
        if getattr(args, attr) is None:
            setattr(args, attr, config[key])
    if args.model == 'regression':          # run_models.py:205-209 (the reference also moves to the CPU; here the fit is a device kernel)
        args.train = False
        args.compile = False
        print('Detect model: regression. Automatically disenable train and compile mode.')
    print(args)
    run(args, config)


if __name__ == '__main__':
    main()
