#!/usr/bin/env python3
"""Loss curves of the IMPORTED reference model trained WITH its dropout on (positional dropout 0.2, nn.Transformer dropout 0.1,
torch's RNG) -- the evidence side of DESIGN section 2's dropout deviation: the reference re-runs the decoder on the growing target
and draws fresh masks for every recomputed position at every decode step (mtio.py:158-164); the KV-cached engine draws one mask
per position.  Five dropout seeds x 200 AdamW steps at d=64 on eight fixed batches (B=64) from the same initial weights, the
MTIO decisions driven by the same host RNG stream in every run.  Recorded: the five loss curves.  The GPU test trains the engine
the same way and compares the curves statistically (tests/test_gpu_vp_engine.py::test_dropout_mask_policy_loss_curves)."""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refstubs  # noqa: E402
refstubs.install()
sys.path.insert(0, '/root/reference/viewport_prediction')
from oracle import vp_oracle as vo  # noqa: E402
import gen_golden_vp as ggv  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def main():
    d, S, T, B, wseed, steps, nb, lr, mixseed = 64, 10, 10, 64, 21, 200, 8, 1e-3, 7
    bias = True
    batches = [vo.synthetic_trajectories(B, S, T, seed=500 + i) for i in range(nb)]
    curves = []
    for dseed in range(5):
        sd = vo.make_state_dict(d, wseed, bias=bias)
        model = ggv.build_reference(d, T, bias, sd)          # dropout left ON (p_pe 0.2, transformer 0.1)
        random.seed(mixseed); np.random.seed(mixseed)         # MTIO repeat / shuffle decisions: identical in every run
        torch.manual_seed(1000 + dseed)                       # dropout masks: differ per run
        opt = torch.optim.AdamW(model.parameters(), lr=lr)
        model.train()
        losses = []
        for i in range(steps):
            h, c, f = batches[i % nb]
            pred, gt = model(h, c, f)
            loss = model.loss_function(pred, gt)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        curves.append(losses)
        print('seed', dseed, 'first', losses[0], 'last mean', np.mean(losses[-20:]))
    path = os.path.join(OUT, 'dropout_curves_vp_d64.npz')
    np.savez_compressed(path, curves=np.array(curves, np.float32), d=d, S=S, T=T, B=B, wseed=wseed, steps=steps, nb=nb, lr=lr, mixseed=mixseed,
                        bias=int(bias), batch_seed0=500)
    print('written', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
