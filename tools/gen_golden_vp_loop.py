#!/usr/bin/env python3
"""Golden capture of the reference's TRAINING LOOP (viewport_prediction/run_models.py:29-58), produced by running
the imported reference model in this container: seeded like run() (np.random / torch / random seeds, :108-112), dropout forced
to 0, `optimizer = AdamW(model.parameters(), lr)`, four consecutive iterations of

    pred, gt = model(history, current, future); loss = model.loss_function(pred, gt)
    optimizer.zero_grad(); loss.backward(); optimizer.step()

on four fixed batches (so the MTIO repeat / shuffle decisions, the AdamW moments and the BatchNorm running statistics evolve
across steps), then the validation metric of :50-58 (mean over batches of mean periodic MSE of sample()).
Data only: inputs, per-step losses, the MTIO decisions taken, final weights / BN statistics, validation MSE."""
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
from models.mtio import ViewportTransformerMTIO  # noqa: E402
from utils.common import mean_square_error  # noqa: E402
from oracle import vp_oracle as vo  # noqa: E402
import gen_golden_vp as ggv  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def main():
    d, S, T, B, seed, wseed, steps = 64, 10, 10, 8, 3, 21, 4
    for bias in (False, True):
        sd = vo.make_state_dict(d, wseed, bias=bias)
        model = ggv.build_reference(d, T, bias, sd)
        ggv.zero_dropout(model)
        batches = [vo.synthetic_trajectories(B, S, T, seed=300 + i) for i in range(steps)]
        valid = [vo.synthetic_trajectories(B, S, T, seed=400 + i) for i in range(2)]
        np.random.seed(seed); torch.manual_seed(seed); random.seed(seed)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
        model.train()
        losses, decisions = [], []
        for h, c, f in batches:
            st = random.getstate()
            decisions.append(random.random() < model.repeat_prob if hasattr(model, 'repeat_prob') else random.random() < 0.5)
            random.setstate(st)                      # peek only: the model draws the same number itself
            pred, gt = model(h, c, f)
            loss = model.loss_function(pred, gt)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(np.float32(loss.item()))
        model.eval()
        with torch.no_grad():
            mse = [torch.mean(mean_square_error(model.sample(h, c), f)).item() for h, c, f in valid]
        rec = dict(d=d, S=S, T=T, B=B, seed=seed, wseed=wseed, bias=int(bias), lr=1e-4, losses=np.array(losses, np.float32),
                   repeat=np.array(decisions, np.bool_), valid_mse=np.float64(np.sum(mse) / len(valid)))
        for i, (h, c, f) in enumerate(batches):
            rec[f'b{i}/history'], rec[f'b{i}/current'], rec[f'b{i}/future'] = h.numpy(), c.numpy(), f.numpy()
        for i, (h, c, f) in enumerate(valid):
            rec[f'v{i}/history'], rec[f'v{i}/current'], rec[f'v{i}/future'] = h.numpy(), c.numpy(), f.numpy()
        for k, v in model.state_dict().items():
            if k != 'positional_embedding.pe':
                rec['final::' + k] = v.numpy().copy()
        path = os.path.join(OUT, f'vp_loop_d64_{"bias" if bias else "nobias"}.npz')
        np.savez_compressed(path, **rec)
        print(path, os.path.getsize(path) // 1024, 'KiB', 'losses', losses, 'repeat', decisions, 'valid', rec['valid_mse'])


if __name__ == '__main__':
    main()
