#!/usr/bin/env python3
"""A/B of tile-order / loop knobs on the VP train step (B = 4096, fp32), variants interleaved in ONE process:
python tools/vp_knob_ab.py col_group 0 12   |   python tools/vp_knob_ab.py f32_wsk 0 1   |   python tools/vp_knob_ab.py mansy_vp_dw_overlap 0 1"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mansy_immersivevideostreaming_amd._lib import lib
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
L = lib()
knob = getattr(L, sys.argv[1] if sys.argv[1].startswith('mansy_') else 'mansy_gemm_' + sys.argv[1])
vals = [int(x) for x in sys.argv[2:]]
torch.manual_seed(5); random.seed(5); np.random.seed(5)
m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device='cuda').to('cuda'); m.train()
opt = FusedAdamW(m, lr=1e-4)
h, c, f = (t.cuda() for t in bench.synthetic_trajectories(4096, 10, 10, seed=5))
for rnd in range(4):
    for v in vals:
        knob(v)
        for _ in range(3): m.train_step(h, c, f, opt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): m.train_step(h, c, f, opt)
        torch.cuda.synchronize()
        print(f'{sys.argv[1]} {v}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms/step', flush=True)
