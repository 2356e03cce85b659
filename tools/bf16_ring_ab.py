#!/usr/bin/env python3
"""A/B of the ring-staged 64 x 64 bf16x3 loop (gemm_bf16h_kernel, mansy_gemm_bf16_variant 1 = 3 stages, the default; 6 = 4 stages) against the
round-2 loop (variant 7): (a) results on the decoder shapes, ragged edges and 1..5 K-tiles, (b) the VP train
step in bf16x3 mode, variants interleaved in ONE process."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lab_knobs as KN  # noqa: E402  (ABI 8: the A/B knobs live in the -DMANSY_LAB build only)
KN.enter()
import numpy as np, torch
from mansy_immersivevideostreaming_amd import kernels as K
from mansy_immersivevideostreaming_amd._lib import lib
import bench
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW

L = lib()
K.set_precision('bf16x3')
torch.manual_seed(0)
worst = 0.0
for (M, N, Kd) in [(4096, 512, 512), (4000, 520, 512), (64, 64, 32), (100, 72, 64), (257, 130, 96), (4096, 1536, 512), (512, 512, 160), (33, 512, 1536)]:
    A = torch.randn(M, Kd, device='cuda'); W = torch.randn(N, Kd, device='cuda')
    pl, pl_t = K.weight_planes(W, 2)
    ref = A.double() @ W.double().t()
    outs = {}
    for v in (7, 1, 6):
        KN.bf16_variant(v)
        outs[v] = K.gemm_planes(A, W, pl, force_tile=64)
    for v in (1, 6):
        same = torch.equal(outs[v], outs[7])
        err = ((outs[v].double() - ref).abs().max() / ref.abs().max()).item()
        worst = max(worst, err)
        print(f'shape {M}x{N}x{Kd} variant {v}: bit-equal to default {same}, max err vs f64 {err:.2e} (default {((outs[7].double() - ref).abs().max() / ref.abs().max()).item():.2e})')
    # transposed form (dX): A [M, N] x W [N, K] with the planes of W^T
    A2 = torch.randn(M, N, device='cuda')
    ref2 = A2.double() @ W.double()
    for v in (7, 1, 6):
        KN.bf16_variant(v)
        outs[v] = K.gemm_planes(A2, W, pl_t, transposed=True, force_tile=64)
    for v in (1, 6):
        err = ((outs[v].double() - ref2).abs().max() / ref2.abs().max()).item()
        worst = max(worst, err)
        print(f'   dX form variant {v}: bit-equal {torch.equal(outs[v], outs[7])}, err {err:.2e}')
print('worst err', worst)
KN.bf16_variant(1)

torch.manual_seed(5); random.seed(5); np.random.seed(5)
m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device='cuda').to('cuda'); m.train()
m.precision = 'bf16x3'
opt = FusedAdamW(m, lr=1e-4)
h, c, f = (t.cuda() for t in bench.synthetic_trajectories(4096, 10, 10, seed=5))
for _ in range(5): m.train_step(h, c, f, opt)
res = {7: [], 1: [], 6: []}
for rep in range(4):
    for v in (7, 1, 6):
        KN.bf16_variant(v)
        for _ in range(2): m.train_step(h, c, f, opt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): loss = m.train_step(h, c, f, opt)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 10 * 1e3)
KN.bf16_variant(1)
for v in (7, 1, 6):
    print(f'variant {v}: bf16x3 train step ms {[round(x, 3) for x in res[v]]} min {min(res[v]):.3f}', 'loss', float(loss))
