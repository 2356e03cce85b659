#!/usr/bin/env python3
"""Golden vectors on the TRAINED weights the reference ships (models/bitrate_selection/mansy/Jin2022_4G/qoe0_1_2_3/epochs_1_.../
best_policy.pth, best_identifier.pth): the imported reference nets (bitrate_selection/models/mansy.py) load the two checkpoints
and evaluate real environment observations (tests/golden/env_reference.npz).  Recorded: the unique weight tensors (data; the
.pth files themselves do not travel), actor logits, critic values, identifier outputs, argmax decisions, identifier rewards.
Every other PPO golden uses seeded synthetic weights; this one pins the path on weights with the statistics of a trained model."""
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
from utils.mansy_utils import calculate_indentifier_reward  # noqa: E402
from gen_golden_ppo import obs_batch, obs_single  # noqa: E402

BASE = '/root/reference/models/bitrate_selection/mansy/Jin2022_4G/qoe0_1_2_3/' \
       'epochs_1_bs_512_lr_0.0005_gamma_0.95_seed_5_ent_0.02_useid_True_lambda_0.5_ilr_0.0001_iur_2_bc_False/'


def main():
    pol = torch.load(BASE + 'best_policy.pth', map_location='cpu')
    idn = torch.load(BASE + 'best_identifier.pth', map_location='cpu')
    fn = FeatureNet(8, 64, 5, 128, device='cpu')
    actor, critic = Actor(fn, 1280, 128, 15, 'cpu'), Critic(fn, 1280, 128, 'cpu')
    ident = QoEIdentifier(QoEIdentifierFeatureNet(8, 64, 5, 15, 128, device='cpu'), 1280, 128, 'cpu')
    actor.load_state_dict({k[len('actor.'):]: v for k, v in pol.items() if k.startswith('actor.')})
    critic.load_state_dict({k[len('critic.'):]: v for k, v in pol.items() if k.startswith('critic.')})
    ident.load_state_dict(idn)
    # the policy checkpoint also carries an identifier copy (policy.identifier is a registered sub-module): record whether it
    # equals best_identifier.pth (it is saved at a different moment of the epoch)
    same = all(torch.equal(pol['identifier.' + k], v) for k, v in idn.items()) if any(k.startswith('identifier.') for k in pol) else None
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'env_reference.npz'))
    rows = np.concatenate([z[f'train_id/ep{e}/obs'][1:] for e in range(5)])
    rows780 = np.zeros((len(rows), 780), np.float32)
    rows780[:, :779] = rows
    rs = np.random.RandomState(1)
    rs.shuffle(rows780)
    n = 96
    ob = obs_batch(rows780[:n])
    with torch.no_grad():
        logits, _ = actor(ob)
        value = critic(ob)
        pred = ident(ob, ob['action_one_hot'])
    rr = [float(calculate_indentifier_reward(ident, obs_single(rows780[i]), obs_single(rows780[i])['action_one_hot'])) for i in range(8)]
    rec = {'obs': rows780[:n], 'logits': logits.numpy(), 'value': value.numpy(), 'ident': pred.numpy(), 'ident_reward': np.array(rr, np.float32),
           'policy_identifier_equals_best_identifier': np.array(-1 if same is None else int(same))}
    for k, v in pol.items():                       # unique tensors only: actor.* (incl. the shared feature net), the critic head
        if k.startswith('actor.') or (k.startswith('critic.') and '.feature_net.' not in k):
            rec['w::' + k] = v.numpy()
    for k, v in idn.items():
        rec['w::identifier.' + k] = v.numpy()
    rec['shared_feature_net_identical'] = np.array(int(all(torch.equal(pol['actor.' + k[7:]], v) for k, v in pol.items()
                                                           if k.startswith('critic.feature_net.'))))
    path = os.path.join(ROOT, 'tests', 'golden', 'shipped_checkpoint_reference.npz')
    np.savez_compressed(path, **rec)
    print('written', path, os.path.getsize(path) // 1024, 'KiB; argmax histogram', np.bincount(logits.argmax(-1).numpy(), minlength=15).tolist(),
          'policy identifier == best_identifier:', same, 'shared feature net identical:', int(rec['shared_feature_net_identical']))
    print('logit range', float(logits.min()), float(logits.max()), 'value range', float(value.min()), float(value.max()), 'ident', pred[:3].numpy())


if __name__ == '__main__':
    main()
