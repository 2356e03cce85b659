#!/usr/bin/env python3
"""Is the run-to-run drift of the bf16-storage step (weight-gradient atomics in another order) as large as its drift against a poisoned workspace?"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import vp_oracle as vo      # (a tools/ probe: input and weight generators only)
from mansy_immersivevideostreaming_amd.viewport_prediction.models import mtio
h, c, f = (t.cuda() for t in vo.synthetic_trajectories(256, 10, 10, seed=4))
def run(poison):
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    m = mtio.ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device='cuda', seed=1)
    m.load_state_dict(vo.make_state_dict(512, 3, bias=False))
    m = m.to('cuda').train(); m.precision = 'bf16'
    opt = mtio.FusedAdamW(m, lr=1e-4)
    losses = []
    for _ in range(4):
        if poison is not None: m._workspace(m._cfg(256, 10)).view(torch.int16).fill_(poison)
        losses.append(m.train_step(h, c, f, opt).item())
    m.eval()
    if poison is not None: m._workspace(m._cfg(256, 10)).view(torch.int16).fill_(poison)
    return losses, m.sample(h, c).cpu(), m._flat_p.clone().cpu()
base = run(None)
for name, p in (('clean again', None), ('clean again', None), ('NaN 0x7FC0', 0x7FC0), ('huge 0x7F00', 0x7F00), ('-huge 0xFF00', -256), ('zeros', 0)):
    o = run(p)
    d = (o[1] - base[1]).abs(); d = torch.minimum(d, 1 - d)
    print(f'{name:14s} losses {["%.6f" % x for x in o[0]]}  sample diff max {d.max().item():.2e} mean {d.mean().item():.2e}  param diff max {(o[2] - base[2]).abs().max().item():.2e}  finite {bool(torch.isfinite(o[1]).all())}', flush=True)
