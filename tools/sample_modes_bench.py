#!/usr/bin/env python3
"""sample() (KV-cached decode, B = 4096, S = T = 10, d = 512) per precision mode, modes interleaved in one process; max |diff| of the
predictions against the fp32 run."""
import sys, os, time, random
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, bench
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO
torch.manual_seed(5); random.seed(5); np.random.seed(5)
m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device='cuda').to('cuda'); m.eval()
h, c, f = (t.cuda() for t in bench.synthetic_trajectories(4096, 10, 10, seed=5))
ref = None
for mode in ('f32', 'bf16x3', 'bf16x6', 'f32', 'bf16x3', 'bf16x6'):
    m.precision = None if mode == 'f32' else mode
    with torch.no_grad():
        for _ in range(3): out = m.sample(h, c)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): out = m.sample(h, c)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    if ref is None: ref = out.clone()
    print(f'{mode}: sample() B=4096 {ms:.3f} ms = {4096 / ms:.1f} k trajectories/s, max |diff| vs f32 {float((out - ref).abs().max()):.2e}')
