#!/usr/bin/env python3
"""Golden vectors for the bitrate-selection networks, produced by the IMPORTED reference
(bitrate_selection/models/mansy.py, utils/mansy_utils.py) in this container.  Data only; weights are the seeded
synthetic ones of oracle.ppo_oracle.make_policy_state_dict (reference checkpoint layout), observations are real
environment observations from tests/golden/env_reference.npz.

Recorded: actor logits / critic values / identifier outputs on a batch, gradients of seeded linear functionals of them
w.r.t. every parameter, calculate_indentifier_reward on single (un-batched) transitions, and a full
train_identifier(...) call (losses + resulting weights) driven through a duck-typed buffer.
Also the key layout of the shipped example checkpoints (names + shapes only).
"""
import contextlib
import io
import os
import sys

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
from models.mansy import Actor, Critic, FeatureNet, QoEIdentifier, QoEIdentifierFeatureNet  # noqa: E402  (the reference)
from utils.mansy_utils import calculate_indentifier_reward, train_identifier  # noqa: E402
from oracle import ppo_oracle as po  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
SL = {'throughput': (0, 8, (1, 8)), 'next_chunk_size': (8, 328, (5, 64)), 'next_chunk_quality': (328, 648, (5, 64)),
      'pred_viewport': (648, 712, (1, 64)), 'viewport_acc': (712, 720, (1, 8)), 'past_viewport_qualities': (720, 728, (1, 8)),
      'past_quality_variances': (728, 736, (1, 8)), 'past_rebuffering': (736, 744, (1, 8)), 'buffer': (744, 745, (1,)),
      'qoe_weight': (745, 748, (3,)), 'action_one_hot': (748, 763, (15,))}


class B(dict):
    """Minimal stand-in for tianshou's Batch: dict of arrays / nested B, indexable by key or by index array."""

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict.__getitem__(self, k)
        return B({n: v[k] for n, v in self.items()})

    def __getattr__(self, k):
        try:
            return dict.__getitem__(self, k)
        except KeyError:
            raise AttributeError(k)

    def __len__(self):
        return len(next(iter(self.values())))


def obs_batch(rows):
    rows = np.asarray(rows, np.float32)
    return B({k: np.ascontiguousarray(rows[:, a:b].reshape((len(rows),) + shape)) for k, (a, b, shape) in SL.items()})


def obs_single(row):
    return B({k: np.ascontiguousarray(np.asarray(row[a:b], np.float32).reshape(shape)) for k, (a, b, shape) in SL.items()})


class Traj:
    def __init__(self, obs):
        self.b = B({'obs': obs})

    def sample(self, n):
        return self.b, np.arange(len(self.b))


def main():
    z = np.load(os.path.join(OUT, 'env_reference.npz'))
    rows = np.concatenate([z[f'train_id/ep{e}/obs'][1:] for e in range(5)])        # post-step observations (one-hot set)
    rows780 = np.zeros((len(rows), 780), np.float32)
    rows780[:, :779] = rows
    rs = np.random.RandomState(0)
    rs.shuffle(rows780)
    sd = po.make_policy_state_dict(21)
    fn = FeatureNet(8, 64, 5, 128, device='cpu')
    actor = Actor(fn, 1280, 128, 15, 'cpu')
    critic = Critic(fn, 1280, 128, 'cpu')
    ifn = QoEIdentifierFeatureNet(8, 64, 5, 15, 128, device='cpu')
    ident = QoEIdentifier(ifn, 1280, 128, 'cpu')
    actor.load_state_dict({k[len('actor.'):]: v for k, v in sd.items() if k.startswith('actor.')})
    critic.load_state_dict({k[len('critic.'):]: v for k, v in sd.items() if k.startswith('critic.')})
    ident.load_state_dict({k[len('identifier.'):]: v for k, v in sd.items() if k.startswith('identifier.')})
    rec = {'obs': rows780, 'wseed': np.int32(21)}
    Bn = 64
    ob = obs_batch(rows780[:Bn])
    logits, _ = actor(ob)
    value = critic(ob)
    pred = ident(ob, ob['action_one_hot'])
    rec['logits'], rec['value'], rec['ident'] = logits.detach().numpy(), value.detach().numpy(), pred.detach().numpy()
    g = torch.Generator().manual_seed(3)
    c1, c2, c3 = torch.randn(Bn, 15, generator=g), torch.randn(Bn, 1, generator=g), torch.randn(Bn, 3, generator=g)
    rec['ct_logits'], rec['ct_value'], rec['ct_ident'] = c1.numpy(), c2.numpy(), c3.numpy()
    ((logits * c1).sum() + (value * c2).sum()).backward()
    (pred * c3).sum().backward()
    for k, p in actor.named_parameters():
        rec['grad::actor.' + k] = p.grad.numpy().copy()
    for k, p in critic.named_parameters():
        if not k.startswith('feature_net.'):
            rec['grad::critic.' + k] = p.grad.numpy().copy()
    for k, p in ident.named_parameters():
        rec['grad::identifier.' + k] = p.grad.numpy().copy()
    # un-batched identifier reward (the relabel loop's call, mansy_ppo.py:43-47)
    rr = []
    for i in range(8):
        o = obs_single(rows780[100 + i])
        rr.append(float(calculate_indentifier_reward(ident, o, o['action_one_hot'])))
    rec['ident_reward_rows'] = np.arange(100, 108, dtype=np.int32)
    rec['ident_reward'] = np.array(rr, np.float32)
    # train_identifier through a duck-typed buffer
    ident.zero_grad()
    opt = torch.optim.Adam(ident.parameters(), lr=1e-4, weight_decay=1e-2)
    n_tr = 200
    np.random.seed(17)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        train_identifier(ident, opt, Traj(obs_batch(rows780[:n_tr])), update_round=2)
    lines = [l for l in buf.getvalue().splitlines() if 'loss is' in l]
    rec['ti_n'] = np.int32(n_tr)
    rec['ti_npseed'] = np.int32(17)
    rec['ti_losses'] = np.array([float(l.split(':')[-1]) for l in lines], np.float64)      # 2 train + 1 valid
    for k in ('feature_net.conv1d2.0.weight', 'feature_net.fc2.0.weight', 'fc.0.weight', 'out.weight', 'out.bias', 'feature_net.fc1.0.bias'):
        rec['ti_after::identifier.' + k] = ident.state_dict()[k].numpy().copy()
    # shipped checkpoint layout (names/shapes only)
    base = '/root/reference/models/bitrate_selection/mansy/Jin2022_4G/qoe0_1_2_3/' \
           'epochs_1_bs_512_lr_0.0005_gamma_0.95_seed_5_ent_0.02_useid_True_lambda_0.5_ilr_0.0001_iur_2_bc_False/'
    for f in ('best_policy.pth', 'best_identifier.pth'):
        ck = torch.load(base + f, map_location='cpu')
        rec['layout::' + f] = np.array([f'{k}|{"x".join(map(str, v.shape))}' for k, v in ck.items()])
    path = os.path.join(OUT, 'ppo_reference.npz')
    np.savez_compressed(path, **rec)
    print('written', path, os.path.getsize(path) // 1024, 'KiB', rec['ti_losses'])


if __name__ == '__main__':
    main()
