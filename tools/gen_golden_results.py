#!/usr/bin/env python3
"""Golden files of the reference's evaluation notebook (viewport_prediction/utils/results.py:53-152 `Results.record/write`),
produced by importing it here (stubs for munch / prettytable): fixed ground-truth / prediction batches -> the text of
`results.csv`, `results.log`, `accuracy_result.csv`.  Data only (inputs + the three files' contents)."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refstubs  # noqa: E402
refstubs.install()
REF = '/root/reference/viewport_prediction'
sys.path.insert(0, REF)
os.chdir(REF)
from utils.results import Results  # noqa: E402


def main():
    g = torch.Generator().manual_seed(7)
    batches = []
    for b, n in enumerate((5, 3)):
        gt = torch.rand(n, 10, 2, generator=g)
        pred = (gt + 0.08 * torch.randn(n, 10, 2, generator=g)).remainder(1.0)
        gt[0, 0] = torch.tensor([0.0, 0.0]); pred[0, 0] = torch.tensor([0.999, 0.999])          # wrap-around corners
        gt[0, 1] = torch.tensor([0.125, 0.125]); pred[0, 1] = torch.tensor([0.125, 0.25])       # exact tile boundaries
        batches.append((n, pred, gt, [f'{3 + b}'] * n, torch.arange(n) + 10 * b, torch.arange(n) * 5 + 15))
    out = tempfile.mkdtemp()
    res = Results('mtio', 2, 10, out, 5, mse=True, nll=False, accuracy=True)
    for n, pred, gt, video, user, ts in batches:
        res.record(n, pred, gt, video, user, ts)
    res.write(log=True, label='t_')
    rec = {}
    for i, (n, pred, gt, video, user, ts) in enumerate(batches):
        rec[f'b{i}/pred'], rec[f'b{i}/gt'] = pred.numpy(), gt.numpy()
        rec[f'b{i}/video'] = np.array(video); rec[f'b{i}/user'] = user.numpy(); rec[f'b{i}/timestamp'] = ts.numpy()
    for name in ('t_results.csv', 't_results.log', 't_accuracy_result.csv'):
        rec['file::' + name] = np.array(open(os.path.join(out, name)).read())
    path = os.path.join(ROOT, 'tests', 'golden', 'results_reference.npz')
    np.savez_compressed(path, **rec)
    print('written', path, os.path.getsize(path) // 1024, 'KiB')
    print(open(os.path.join(out, 't_results.csv')).read().splitlines()[1][:200])
    print(open(os.path.join(out, 't_results.log')).read().splitlines()[1][:200])


if __name__ == '__main__':
    main()
