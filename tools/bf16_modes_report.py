#!/usr/bin/env python3
"""Measured error of each precision mode (f32 / bf16x6 / bf16x3) of the HIP path against the goldens captured from the imported
reference: viewport predictor (sample, tile decisions, train-mode prediction, loss, gradients) and bitrate-selection nets
(logits, values, identifier outputs, argmax decisions).  Runs on the GPU box; the output is kept under profiles/.

    python tools/bf16_modes_report.py > profiles/r02_bf16_modes_parity.txt
"""
import glob
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mansy_immersivevideostreaming_amd import kernels as K                                   # noqa: E402
from mansy_immersivevideostreaming_amd.viewport_prediction.models import mtio               # noqa: E402
from oracle import ppo_oracle as po                                                          # noqa: E402
from oracle import vp_oracle as vo                                                           # noqa: E402

MODES = ('f32', 'bf16x6', 'bf16x3')


def vp_rows(path):
    z = np.load(path)
    d = int(z['d'])
    sd = vo.make_state_dict(d, int(z['wseed']), bias=bool(z['bias']))
    h, c, f = (torch.from_numpy(z[k]).cuda() for k in ('history', 'current', 'future'))
    rows = []
    for mode in MODES:
        m = mtio.ViewportTransformerMTIO(in_channel=2, fut_window=int(z['T']), d_model=d, dim_feedforward=d, device='cuda', bias=bool(z['bias']))
        m.load_state_dict(sd)
        m = m.to('cuda')
        m.dropout_p = m.attn_dropout_p = 0.0
        m.precision = mode
        m.eval()
        with torch.no_grad():
            samp = m.sample(h, c)
        e_samp = float(np.abs(samp.cpu().numpy() - z['eval_sample']).max())
        tiles = int((K.tilemap(samp).cpu().numpy() != K.tilemap(torch.from_numpy(z['eval_sample']).cuda()).cpu().numpy()).sum())
        m.train()
        mix_seed = int(z['train_rep_mixseed'])
        random.seed(mix_seed)
        np.random.seed(mix_seed)
        opt = mtio.FusedAdamW(m, lr=1e-4)
        opt.zero_grad()
        pred, gt = m(h, c, f)
        loss = m.loss_function(pred, gt)
        loss.backward()
        e_pred = float(np.abs(pred.detach().cpu().numpy() - z['train_rep_pred']).max())
        e_loss = abs(loss.item() - float(z['train_rep_loss'])) / abs(float(z['train_rep_loss']))
        grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
        worst, worst_name, rels = 0.0, '', []
        for key in z.files:
            full, sl = key.startswith('train_rep_grad::'), key.startswith('train_rep_gradslice::')
            if not (full or sl):
                continue
            k = key.split('::')[1]
            ref = z[key]
            if np.abs(ref).max() < 1e-7:
                continue
            got = grads[k].numpy() if full else grads[k].reshape(grads[k].shape[0], -1)[::37, ::41].numpy()
            rel = float(np.abs(got - ref).max() / np.abs(ref).max())
            rels.append(rel)
            if rel > worst:
                worst, worst_name = rel, k
        rows.append((mode, e_samp, tiles, e_pred, e_loss, float(np.median(rels)), worst, worst_name))
    return rows


def ppo_rows():
    from test_gpu_ppo import build_policy
    from mansy_immersivevideostreaming_amd.bitrate_selection.models import mansy, mansy_ppo
    from mansy_immersivevideostreaming_amd.bitrate_selection.envs import mansy_env

    class NS:
        pass
    M = NS()
    M.mansy, M.ppo, M.env = mansy, mansy_ppo, mansy_env
    Z = np.load(os.path.join(ROOT, 'tests', 'golden', 'ppo_reference.npz'))
    pol = build_policy(M, po.make_policy_state_dict(int(Z['wseed'])))
    obs = torch.from_numpy(Z['obs'][:64]).cuda()
    rows = []
    for mode in MODES:
        with K.precision(mode):
            logits, _ = pol.actor(obs)
            value = pol.critic(obs)
            pred = pol.identifier(obs)
        rows.append((mode, float(np.abs(logits.cpu().numpy() - Z['logits']).max()), float(np.abs(value.cpu().numpy() - Z['value']).max()),
                     float(np.abs(pred.cpu().numpy() - Z['ident']).max()), int((logits.argmax(-1).cpu().numpy() != Z['logits'].argmax(-1)).sum())))
    return rows


if __name__ == '__main__':
    print('# measured error of the HIP path per precision mode against the goldens of the imported reference (north_star bar: 1e-4 on')
    print('# outputs, tile / bitrate decisions bit-exact).  grad rel = max |g - g_ref| / max |g_ref| per tensor: median and worst tensor.')
    print('%-26s %-7s %11s %6s %11s %11s %12s %12s  %s' % ('viewport golden', 'mode', 'sample abs', 'tiles', 'pred abs', 'loss rel', 'grad rel med',
                                                          'grad rel max', 'worst tensor'))
    for p in sorted(q for q in glob.glob(os.path.join(ROOT, 'tests', 'golden', 'vp_*.npz')) if 'vp_loop_' not in q):
        for r in vp_rows(p):
            print('%-26s %-7s %11.3e %6d %11.3e %11.3e %12.3e %12.3e  %s' % ((os.path.basename(p)[:-4],) + r))
    print()
    print('%-26s %-7s %11s %11s %11s %8s' % ('bitrate-selection nets', 'mode', 'logits abs', 'value abs', 'ident abs', 'argmax!='))
    for r in ppo_rows():
        print('%-26s %-7s %11.3e %11.3e %11.3e %8d' % (('ppo_reference (64 obs)',) + r))
