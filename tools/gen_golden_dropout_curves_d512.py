#!/usr/bin/env python3
"""Loss curves of the IMPORTED reference at the REAL width with its dropout ON: d = 512, 2+2 layers, B = 32 real
Jin2022 windows (hist 10, pred 10: BASELINE configs[0]), positional dropout 0.2 + nn.Transformer dropout 0.1 drawn from torch's RNG,
AdamW lr 1e-4 (run_models.py:29), 60 steps over six fixed batches (the 216 train windows of the eight travelling traces hold six full ones), 3 dropout seeds from the same initial weights with identical MTIO
decisions.  The reference re-runs the decoder on the growing target and draws FRESH masks for every recomputed position
(mtio.py:27-29,158-164); the KV-cached engine draws ONE mask per position -- the GPU test
(tests/test_gpu_vp_engine.py::test_dropout_mask_policy_loss_curves_d512) trains the engine the same way and holds its curves inside
the reference's seed-to-seed band.  The fixture holds data only: the six batches (they are windows of the traces that already
travel in dataset_reference.npz) and the loss curves."""
import ast
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refstubs  # noqa: E402
refstubs.install()
REF = '/root/reference/viewport_prediction'
sys.path.insert(0, REF)
os.chdir(REF)
from utils.common import get_config_from_yml  # noqa: E402
from utils.load_dataset import create_dataset  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402
from oracle import vp_oracle as vo  # noqa: E402
import gen_golden_vp as ggv  # noqa: E402

D, S, T, B, SEED, WSEED, STEPS, NB, LR, MIXSEED, NSEED = 512, 10, 10, 32, 5, 41, 60, 6, 1e-4, 7, 3


def main():
    Z = np.load(os.path.join(ROOT, 'tests', 'golden', 'dataset_reference.npz'))
    vsplit, usplit = ast.literal_eval(str(Z['vsplit'])), ast.literal_eval(str(Z['usplit']))
    config = get_config_from_yml()
    train = create_dataset('Jin2022', config, his_window=S, fut_window=T, frequency=5, sample_step=5, trim_head=15, trim_tail=15,
                           dataset_video_split=dict(vsplit), dataset_user_split=dict(usplit), include=['train'])[0]
    torch.manual_seed(SEED)
    batches = []
    for bi, (h, c, f, v, u, t) in enumerate(DataLoader(train, batch_size=B, shuffle=True)):
        if h.shape[0] == B:
            batches.append((h.float(), c.float(), f.float()))
        if len(batches) == NB:
            break
    bias = True
    curves = []
    for dseed in range(NSEED):
        t0 = time.time()
        sd = vo.make_state_dict(D, WSEED, bias=bias)
        model = ggv.build_reference(D, T, bias, sd)          # dropout left ON (p_pe 0.2, transformer 0.1)
        random.seed(MIXSEED); np.random.seed(MIXSEED)         # MTIO repeat / shuffle decisions: identical in every run
        torch.manual_seed(1000 + dseed)                       # dropout masks: differ per run
        opt = torch.optim.AdamW(model.parameters(), lr=LR)
        model.train()
        losses = []
        for i in range(STEPS):
            h, c, f = batches[i % NB]
            pred, gt = model(h, c, f)
            loss = model.loss_function(pred, gt)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        curves.append(losses)
        print('seed', dseed, 'first', losses[0], 'last mean', np.mean(losses[-10:]), '%.0fs' % (time.time() - t0), flush=True)
    path = os.path.join(ROOT, 'tests', 'golden', 'dropout_curves_vp_d512.npz')
    np.savez_compressed(path, curves=np.array(curves, np.float32), d=D, S=S, T=T, B=B, wseed=WSEED, steps=STEPS, nb=NB, lr=LR, mixseed=MIXSEED,
                        bias=int(bias), history=np.stack([b[0].numpy() for b in batches]), current=np.stack([b[1].numpy() for b in batches]),
                        future=np.stack([b[2].numpy() for b in batches]))
    print('written', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
