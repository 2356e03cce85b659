#!/usr/bin/env python3
"""Where does the wave-split-K loop stop paying?  The [4096, 512, 512] decoder-step product (512 tiles of 64 x 64: two workgroups per CU on the 64 x 64
loop) and the VP train step with the 'small product' threshold at 256 (default) / 512 / 2048 tiles."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lab_knobs as KN  # noqa: E402  (ABI 8: the A/B knobs live in the -DMANSY_LAB build only)
KN.enter()
import numpy as np, torch
import bench
from mansy_immersivevideostreaming_amd import kernels as K
from mansy_immersivevideostreaming_amd._lib import lib
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
L = lib()
THR = tuple(int(x) for x in sys.argv[1:]) or (256, 512, 2048)
dev = 'cuda'
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (4 * n) * 1e3
for (M, N, Kd, bk) in ((4096, 512, 512, 0), (4096, 512, 512, 1), (2048, 512, 512, 0), (4096, 1536, 512, 0)):
    A = torch.randn(M, Kd, device=dev); B = torch.randn((Kd, N) if bk else (N, Kd), device=dev); out = torch.zeros(M, N, device=dev)
    line = f'gemm M={M} N={N} K={Kd} {"NN" if bk else "NT"}:'
    for thr in THR:
        KN.f32_wsk(thr)
        line += f'  threshold {thr}: {timed(lambda: K.gemm(A, B, False, bool(bk), out=out, force_tile=64)):6.2f} us'
    print(line, flush=True)
torch.manual_seed(5); random.seed(5); np.random.seed(5)
m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device='cuda').to('cuda'); m.train()
opt = FusedAdamW(m, lr=1e-4)
h, c, f = (t.cuda() for t in bench.synthetic_trajectories(4096, 10, 10, seed=5))
for rnd in range(3):
    for thr in THR:
        KN.f32_wsk(thr)
        for _ in range(3): m.train_step(h, c, f, opt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): m.train_step(h, c, f, opt)
        torch.cuda.synchronize()
        print(f'vp threshold {thr}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms/step', flush=True)
KN.f32_wsk(256)
