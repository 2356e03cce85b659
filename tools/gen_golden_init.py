#!/usr/bin/env python3
"""Golden vectors for the network initialisation of run_mansy.py:205-226 and run_simple_rl.py:178-190, produced with the IMPORTED
reference nets: torch.manual_seed(seed), construct FeatureNet / Actor / Critic (/ QoEIdentifier) exactly in the reference's order,
then its loop `for m in model.modules(): if isinstance(m, nn.Linear): orthogonal_(m.weight, gain=sqrt(2)); zeros_(m.bias)`.
`model` is tianshou's ActorCritic(actor, critic) there -- an nn.Module holding the two nets, so modules() visits the shared
feature net ONCE; the stand-in below is that container.  Recorded per tensor: sum, sum of absolute values and the first 4
elements (float64) -- enough to pin which tensors were re-initialised, in which order the RNG was consumed, and the values."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refstubs  # noqa: E402
refstubs.install()
REF = '/root/reference/bitrate_selection'
sys.path.insert(0, REF)
os.chdir(REF)
from models import mansy as rm  # noqa: E402  (the reference)
from models import simple_rl as rs  # noqa: E402


class ActorCritic(torch.nn.Module):
    def __init__(self, actor, critic):
        super().__init__()
        self.actor, self.critic = actor, critic


def ortho(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight, gain=np.sqrt(2))
            torch.nn.init.zeros_(m.bias)


def digest(rec, prefix, sd):
    for k, v in sd.items():
        v = v.double().reshape(-1)
        rec[f'{prefix}::{k}'] = np.concatenate([[v.sum().item(), v.abs().sum().item()], v[:4].numpy(), np.zeros(max(0, 4 - v.numel()))])


def main():
    rec = {'seed': np.int32(5)}
    torch.manual_seed(5)
    fn = rm.FeatureNet(8, 64, 5, 128, device='cpu')
    actor = rm.Actor(fn, feature_dim=1280, hidden_dim=128, action_space=15, device='cpu')
    critic = rm.Critic(fn, feature_dim=1280, hidden_dim=128, device='cpu')
    ortho(ActorCritic(actor, critic))
    ifn = rm.QoEIdentifierFeatureNet(8, 64, 5, 15, 128, device='cpu')
    ident = rm.QoEIdentifier(ifn, feature_dim=1280, hidden_dim=128, device='cpu')
    ortho(ident)
    digest(rec, 'mansy/actor', actor.state_dict())
    digest(rec, 'mansy/critic', critic.state_dict())
    digest(rec, 'mansy/identifier', ident.state_dict())
    torch.manual_seed(5)
    fn = rs.FeatureNet(8, 64, 5, device='cpu')
    actor = rs.Actor(fn, feature_dim=5 * 128, action_space=15, device='cpu')
    critic = rs.Critic(fn, feature_dim=5 * 128, device='cpu')
    ortho(ActorCritic(actor, critic))
    digest(rec, 'simple/actor', actor.state_dict())
    digest(rec, 'simple/critic', critic.state_dict())
    path = os.path.join(ROOT, 'tests', 'golden', 'init_reference.npz')
    np.savez_compressed(path, **rec)
    print('written', path, os.path.getsize(path) // 1024, 'KiB', len(rec), 'entries')


if __name__ == '__main__':
    main()
