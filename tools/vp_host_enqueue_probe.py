#!/usr/bin/env python3
"""How long does the HOST take to enqueue one VP train step (B = 4096; ~540 launches on one stream, ~1000 with the two-stream decoder), against the step's
GPU time?  If the enqueue time is a large fraction of the step, the second stream starves while the first one's launches are being submitted."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
torch.manual_seed(5); random.seed(5); np.random.seed(5)
m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device='cuda').to('cuda'); m.train()
opt = FusedAdamW(m, lr=1e-4)
h, c, f = (t.cuda() for t in bench.synthetic_trajectories(4096, 10, 10, seed=5))
for ts in (False, True, False, True):
    m.two_stream = ts
    for _ in range(3): m.train_step(h, c, f, opt)
    torch.cuda.synchronize()
    enq = []
    t0 = time.perf_counter()
    for _ in range(10):
        a = time.perf_counter(); m.train_step(h, c, f, opt); enq.append(time.perf_counter() - a)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    # one step enqueued on an idle GPU: pure host cost (nothing to wait for)
    torch.cuda.synchronize(); a = time.perf_counter(); m.train_step(h, c, f, opt); solo = time.perf_counter() - a; torch.cuda.synchronize()
    print(f'two_stream={int(ts)}: step {t_all / 10 * 1e3:.2f} ms; host enqueue of 10 steps {t_enq * 1e3:.1f} ms (per step: first {enq[0] * 1e3:.2f}, median {sorted(enq)[5] * 1e3:.2f}, last {enq[-1] * 1e3:.2f}); '
          f'one step on an idle GPU: {solo * 1e3:.2f} ms of host time', flush=True)
