#!/usr/bin/env python3
"""Golden vectors for behaviour-cloning pre-training, produced by the IMPORTED reference function
`behavior_cloning_pretraining` (bitrate_selection/utils/mansy_utils.py:52-93) in this container.

The function needs a tianshou policy and tianshou ReplayBuffers only through duck typing: `policy(samples)` must return an
object with `.logits / .act / .dist`, a demonstration must answer `.sample(0) -> (samples, indices)` with `samples.obs`
(a dict-like of numpy arrays) and `samples['act']`.  The stand-ins below wrap the imported reference `Actor` exactly the way
tianshou's PPOPolicy.forward does (logits, _ = actor(batch.obs); dist = Categorical(logits=logits); act = dist.sample()), so
every number recorded here comes out of the reference's own loop: per-step training losses, validation losses, the best-step
choice, the identifier losses of the interleaved train_identifier calls, and the resulting weights.  Data only.
"""
import contextlib
import io
import os
import random
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
from models.mansy import Actor, Critic, FeatureNet, QoEIdentifier, QoEIdentifierFeatureNet  # noqa: E402  (the reference)
from utils.mansy_utils import behavior_cloning_pretraining  # noqa: E402
from oracle import ppo_oracle as po  # noqa: E402
from gen_golden_ppo import B, obs_batch  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


class Demo:
    """ReplayBuffer stand-in: one expert episode."""

    def __init__(self, rows780, act):
        self.b = B({'obs': obs_batch(rows780), 'act': np.asarray(act, np.int64)})

    def sample(self, n):
        assert n == 0
        return self.b, np.arange(len(self.b['act']))


class Out:
    pass


class DuckPolicy:
    """tianshou PPOPolicy.forward + the nn.Module methods the function calls."""

    def __init__(self, actor, critic):
        self.actor, self.critic = actor, critic

    def __call__(self, batch):
        logits, _ = self.actor(batch.obs)
        o = Out()
        o.logits = logits
        o.dist = torch.distributions.Categorical(logits=logits)
        o.act = o.dist.sample()
        return o

    def eval(self):
        self.actor.eval()

    def train(self):
        self.actor.train()

    def state_dict(self):
        return {'actor.' + k: v for k, v in self.actor.state_dict().items()}


class Args:
    device = 'cpu'


def cut(v):
    """Large tensors are kept as a strided slice of their 2-D view (the GPU test applies the same cut)."""
    v = np.asarray(v)
    v2 = v.reshape(v.shape[0], -1)
    return v2[::5, ::7].copy() if v2.size > 20000 else v2.copy()


def main():
    z = np.load(os.path.join(OUT, 'env_reference.npz'))
    rows = np.concatenate([z[f'train_id/ep{e}/obs'][1:] for e in range(5)])
    rows780 = np.zeros((len(rows), 780), np.float32)
    rows780[:, :779] = rows
    rs = np.random.RandomState(4)
    rs.shuffle(rows780)
    lens_train, lens_valid = [51, 33, 40], [37, 29]
    demos, k = [], 0
    for n in lens_train + lens_valid:
        demos.append((rows780[k:k + n].copy(), rs.randint(0, 15, size=n)))
        k += n
    train_demos = [Demo(*d) for d in demos[:3]]
    valid_demos = [Demo(*d) for d in demos[3:]]

    wseed = 33
    sd = po.make_policy_state_dict(wseed)
    fn = FeatureNet(8, 64, 5, 128, device='cpu')
    actor = Actor(fn, 1280, 128, 15, 'cpu')
    critic = Critic(fn, 1280, 128, 'cpu')
    ident = QoEIdentifier(QoEIdentifierFeatureNet(8, 64, 5, 15, 128, device='cpu'), 1280, 128, 'cpu')
    actor.load_state_dict({k[len('actor.'):]: v for k, v in sd.items() if k.startswith('actor.')})
    critic.load_state_dict({k[len('critic.'):]: v for k, v in sd.items() if k.startswith('critic.')})
    ident.load_state_dict({k[len('identifier.'):]: v for k, v in sd.items() if k.startswith('identifier.')})
    lr, ilr, wd = 5e-4, 1e-4, 1e-2
    # run_mansy.py:216: one Adam over the actor (incl. the shared feature net) and the critic head
    optim = torch.optim.Adam(list(actor.parameters()) + [p for n, p in critic.named_parameters() if not n.startswith('feature_net.')], lr=lr,
                             weight_decay=wd)
    ioptim = torch.optim.Adam(ident.parameters(), lr=ilr, weight_decay=wd)
    max_steps, valid_per_step, id_max_steps, id_rounds = 7, 3, 4, 2
    seed = 9
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    buf = io.StringIO()
    with tempfile.TemporaryDirectory() as tmp:
        ppath, ipath = os.path.join(tmp, 'p.pth'), os.path.join(tmp, 'i.pth')
        with contextlib.redirect_stdout(buf):
            behavior_cloning_pretraining(Args(), DuckPolicy(actor, critic), ident, optim, ioptim, train_demos, valid_demos, max_steps,
                                         valid_per_step, id_max_steps, id_rounds, ppath, ipath)
        best_sd = torch.load(ppath)
    lines = buf.getvalue().splitlines()
    tr = [float(l.split('loss=')[1].split(' ')[0]) for l in lines if l.startswith('BC (Training)')]
    va = [(float(l.split('valid loss=')[1].split(' ')[0]), float(l.split('best loss=')[1].split(' ')[0]), int(l.rsplit(' ', 1)[1]))
          for l in lines if l.startswith('BC (Validation)')]
    idl = [float(l.split(':')[-1]) for l in lines if 'identifier loss is' in l]
    idv = [float(l.split(':')[-1]) for l in lines if 'identifier validation loss is' in l]
    # which demonstration each step drew (random.choice on the same seed)
    random.seed(seed)
    picks = [random.choice(range(3)) for _ in range(max_steps)]
    rec = {'wseed': np.int32(wseed), 'seed': np.int32(seed), 'lr': np.float64(lr), 'ilr': np.float64(ilr), 'wd': np.float64(wd),
           'max_steps': np.int32(max_steps), 'valid_per_step': np.int32(valid_per_step), 'id_max_steps': np.int32(id_max_steps),
           'id_rounds': np.int32(id_rounds), 'n_train': np.int32(3), 'n_valid': np.int32(2),
           'train_losses': np.array(tr), 'valid_losses': np.array([v[0] for v in va]), 'best_losses': np.array([v[1] for v in va]),
           'best_steps': np.array([v[2] for v in va], np.int32), 'ident_train_losses': np.array(idl), 'ident_valid_losses': np.array(idv),
           'picks': np.array(picks, np.int32)}
    for i, (o, a) in enumerate(demos):
        rec[f'demo{i}/obs'], rec[f'demo{i}/act'] = o, a.astype(np.int32)
    keep = ('feature_net.conv1d2.0.weight', 'feature_net.conv1d1.0.bias', 'feature_net.fc2.0.weight', 'feature_net.fc1.0.bias', 'fc.0.weight', 'fc.0.bias',
            'out.weight', 'out.bias')
    for k_ in keep:
        rec['after::actor.' + k_] = cut(actor.state_dict()[k_].numpy())
        rec['best::actor.' + k_] = cut(best_sd['actor.' + k_].numpy())
    for k_, v in actor.state_dict().items():
        rec['norm::actor.' + k_] = np.float64(v.double().norm().item())
    for k_ in ('fc.0.weight', 'out.weight'):                          # the critic head is untouched by cloning (no gradient, Adam skips it)
        rec['after::critic.' + k_] = cut(critic.state_dict()[k_].numpy())
    for k_ in ('feature_net.conv1d2.0.weight', 'feature_net.fc2.0.weight', 'fc.0.weight', 'out.weight', 'out.bias'):
        rec['after::identifier.' + k_] = cut(ident.state_dict()[k_].numpy())
    path = os.path.join(OUT, 'bc_reference.npz')
    np.savez_compressed(path, **rec)
    print('written', path, os.path.getsize(path) // 1024, 'KiB')
    print('train', tr, '\nvalid', va, '\nident', idl, idv, '\npicks', picks)


if __name__ == '__main__':
    main()
