#!/usr/bin/env python3
"""Golden vectors of the reference's linear-regression viewport baseline (viewport_prediction/models/linear_regression.py:18-36,
selected by `run_models.py --model regression`): the IMPORTED class (scikit-learn's LinearRegression per trajectory and coordinate)
on REAL Jin2022 windows -- the test_seen split of the eight traces that already travel in tests/golden/dataset_reference.npz -- at
the reference's default window (his 5 / fut 15) and at the benchmark's (10 / 10), each with the three files the reference's
`Results` notebook writes for that batch (run_models.py:72-85), plus synthetic rows that stress the arithmetic
(constant history, exact lines, values outside [0, 1], wrap-around jumps).
Writes tests/golden/linreg_reference.npz.  Data only."""
import ast
import os
import sys
import tempfile

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
from models.linear_regression import LinearRegression  # noqa: E402
from utils.results import Results  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402


def main():
    Z = np.load(os.path.join(ROOT, 'tests', 'golden', 'dataset_reference.npz'))
    vsplit, usplit = ast.literal_eval(str(Z['vsplit'])), ast.literal_eval(str(Z['usplit']))
    config = get_config_from_yml()
    out = {}
    for tag, S, T in (('s5_t15', 5, 15), ('s10_t10', 10, 10)):
        ds = create_dataset('Jin2022', config, his_window=S, fut_window=T, frequency=5, sample_step=5, trim_head=15, trim_tail=15,
                            dataset_video_split=dict(vsplit), dataset_user_split=dict(usplit), include=['test_seen'])[0]
        h, c, f, v, u, t = next(iter(DataLoader(ds, batch_size=192, shuffle=False)))
        model = LinearRegression(fut_window=T)
        pred = model.sample(h.float(), c.float())
        out[f'{tag}_history'] = h.float().numpy(); out[f'{tag}_current'] = c.float().numpy(); out[f'{tag}_pred'] = pred.numpy()
        out[f'{tag}_video'] = np.asarray(v); out[f'{tag}_user'] = np.asarray(u); out[f'{tag}_timestep'] = np.asarray(t)
        out[f'{tag}_future'] = f.float().numpy()
        # the test driver's notebook over this batch (run_models.py:72-85: Results('regression', ...).record / write), file texts
        tmp = tempfile.mkdtemp()
        res = Results('regression', 2, T, tmp, 5, mse=True, nll=False, accuracy=True)
        res.record(h.shape[0], pred, f.float(), v, u, t)
        res.write(log=True, label='t_')
        for name in ('t_results.csv', 't_results.log', 't_accuracy_result.csv'):
            out[f'{tag}_file::{name}'] = np.array(open(os.path.join(tmp, name)).read())
        print(tag, h.shape, pred.shape, pred.dtype, float(pred.min()), float(pred.max()))
    rs = np.random.RandomState(11)
    S, T, B = 7, 12, 64
    h = rs.rand(B, S, 2).astype(np.float32)
    c = rs.rand(B, 1, 2).astype(np.float32)
    h[0] = 0.25; c[0] = 0.25                                               # constant
    ramp = (np.arange(S + 1, dtype=np.float32) * np.float32(0.125))[:, None] * np.ones(2, np.float32)
    h[1] = ramp[:S]; c[1] = ramp[S:]                                       # exact line leaving [0, 1]
    h[2, :, 0] = np.linspace(0.9, 0.99, S); c[2, 0, 0] = 0.02              # wrap-around jump in x
    h[3] *= 1e-6; c[3] *= 1e-6                                             # tiny values
    h[4] = -h[4]                                                           # negative coordinates
    model = LinearRegression(fut_window=T)
    out['syn_history'] = h; out['syn_current'] = c; out['syn_pred'] = model.sample(torch.from_numpy(h), torch.from_numpy(c)).numpy()
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'linreg_reference.npz'), **out)
    print('wrote', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
