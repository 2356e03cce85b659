#!/usr/bin/env python3
"""Golden vectors of the reference's A2C baseline pieces that are importable here (tianshou is not):
  * bitrate_selection/envs/simple_rl_env.py -- SimpleRLEnv episodes (train mode: reward = qoe / sum(w); valid mode: raw qoe)
    with random actions: every observation key flattened into the 416-float row of the build
    [throughput 8 | chunk_sizes 320 | rebuffer 1 | last_bitrates 2 | pred_viewport 64 | zero pad], reward, done, CSV log;
  * bitrate_selection/models/simple_rl.py -- FeatureNet / Actor / Critic on seeded weights: probabilities ("logits" of the
    reference's Actor are softmax outputs), values, and every parameter gradient of a fixed scalar of both.
Data only.  Produced by importing the reference with stubs for gym / munch / prettytable."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refstubs  # noqa: E402
refstubs.install()
import gen_golden_env as gge  # noqa: E402
from envs.simple_rl_env import SimpleRLEnv  # noqa: E402
from models.simple_rl import Actor, Critic, FeatureNet  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
KEYS = [('throughput', 8), ('chunk_sizes', 320), ('rebuffer', 1), ('last_bitrates', 2), ('pred_viewport', 64)]
LD = 416


def flat(state):
    row = np.zeros(LD, np.float32)
    o = 0
    for k, n in KEYS:
        a = np.asarray(state[k], np.float32).reshape(-1)
        assert a.size == n, (k, a.size)
        row[o:o + n] = a
        o += n
    return row


def run_env(config, mode, seed, worker_num, n_ep, qoe_weights, act_seed):
    log = tempfile.mktemp(suffix='.csv')
    env = SimpleRLEnv(config, 'Jin2022', '4G', qoe_weights, log, config.startup_download, mode=mode, seed=seed, worker_num=worker_num)
    env.seed(seed)
    rs = np.random.RandomState(act_seed)
    eps = []
    for _ in range(n_ep):
        st = env.reset()
        rec = dict(sample_id=env.sample_id, video=env.current_video, user=env.current_user, trace=env.current_trace, obs=[flat(st)], act=[],
                   rew=[], done=[])
        over = False
        while not over:
            a = int(rs.randint(0, 15))
            st, r, over, _ = env.step(a)
            rec['act'].append(a)
            rec['rew'].append(np.float32(r))
            rec['done'].append(bool(over))
            rec['obs'].append(flat(st))
        eps.append(rec)
    csv = open(log).read()
    os.remove(log)
    return env, eps, csv


def main():
    config = gge.get_config_from_yml()
    rec = {}
    for tag, mode, seed, wn, n_ep, qoe, act_seed in [('train', 'train', 3, 2, 3, [config.qoe_split['train'][1]], 21),
                                                    ('valid', 'valid', 1, 1, 2, [config.qoe_split['train'][0]], 22)]:
        env, eps, csv = run_env(config, mode, seed, wn, n_ep, qoe, act_seed)
        for k, v in gge.build_tables(config, env, eps).items():
            rec[f'{tag}/{k}'] = v
        rec[f'{tag}/meta'] = np.array([seed, wn, n_ep, int(mode == 'train')], np.int32)
        rec[f'{tag}/qoe_w'] = np.array(qoe, np.float32)
        rec[f'{tag}/csv'] = np.array(csv)
        for i, e in enumerate(eps):
            rec[f'{tag}/ep{i}/sample_id'] = np.int32(e['sample_id'])
            rec[f'{tag}/ep{i}/ids'] = np.array([e['video'], e['user'], e['trace']], np.int32)
            rec[f'{tag}/ep{i}/obs'] = np.stack(e['obs'])
            rec[f'{tag}/ep{i}/act'] = np.array(e['act'], np.int32)
            rec[f'{tag}/ep{i}/rew'] = np.array(e['rew'], np.float32)
            rec[f'{tag}/ep{i}/done'] = np.array(e['done'], np.bool_)
        print(tag, [len(e['act']) for e in eps])
    rec['const/video_rates'] = np.array(config.video_rates, np.int32)
    rec['const/misc'] = np.array([config.startup_download, config.chunk_length, config.max_size, config.max_throughput], np.float64)
    # ---- networks
    torch.manual_seed(17)
    fn = FeatureNet(config.past_k, config.tile_total_num, len(config.video_rates), device='cpu')
    actor = Actor(fn, 5 * 128, config.action_space, 'cpu')
    critic = Critic(fn, 5 * 128, 'cpu')
    for m in list(actor.modules()) + list(critic.modules()):
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight, gain=np.sqrt(2))
            torch.nn.init.normal_(m.bias, std=0.05)
    rows = np.concatenate([rec['train/ep0/obs'], rec['train/ep1/obs'], rec['valid/ep0/obs']])[:96]
    B = rows.shape[0]
    obs = {'throughput': rows[:, 0:8].reshape(B, 1, 8).copy(), 'chunk_sizes': rows[:, 8:328].reshape(B, 5, 64).copy(),
           'rebuffer': rows[:, 328:329].copy(), 'last_bitrates': rows[:, 329:331].copy(), 'pred_viewport': rows[:, 331:395].copy()}
    probs, _ = actor(obs)
    value = critic(obs)
    g = torch.Generator().manual_seed(4)
    c1, c2 = torch.randn(B, 15, generator=g), torch.randn(B, 1, generator=g)
    ((probs * c1).sum() + (value * c2).sum()).backward()
    rec['net/obs'] = rows
    rec['net/probs'] = probs.detach().numpy()
    rec['net/value'] = value.detach().numpy()
    rec['net/c1'], rec['net/c2'] = c1.numpy(), c2.numpy()
    sd = {}
    for prefix, mod in (('actor.', actor), ('critic.', critic)):
        for k, v in mod.state_dict().items():
            sd[prefix + k] = v
    names = {id(p): n for n, p in list(('actor.' + k, v) for k, v in actor.named_parameters()) +
             [('critic.' + k, v) for k, v in critic.named_parameters() if not k.startswith('feature_net.')]}
    for k, v in sd.items():
        rec['net/w::' + k] = v.detach().numpy()
    for prefix, mod in (('actor.', actor), ('critic.', critic)):
        for k, p in mod.named_parameters():
            if prefix == 'critic.' and k.startswith('feature_net.'):
                continue
            rec['net/g::' + prefix + k] = p.grad.numpy()
    rec['net/keys'] = np.array(list(sd.keys()))
    path = os.path.join(OUT, 'a2c_reference.npz')
    np.savez_compressed(path, **rec)
    print('written', path, os.path.getsize(path) // 1024, 'KiB', len(sd), 'state_dict keys')


if __name__ == '__main__':
    main()
