#!/usr/bin/env python3
"""Split-K / tile sweep on the dW and forward shapes (tuning aid; not part of the product path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mansy_immersivevideostreaming_amd import kernels as K


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(name, ak, bk, M, N, Kd, acc, tiles, splits):
    A = torch.randn((Kd, M) if ak else (M, Kd), device='cuda')
    B = torch.randn((Kd, N) if bk else (N, Kd), device='cuda')
    out = torch.zeros(M, N, device='cuda')
    for t in tiles:
        for s in splits:
            us = timeit(lambda: K.gemm(A, B, bool(ak), bool(bk), out=out, accumulate=bool(acc), force_tile=t, force_splitk=s))
            print(f'{name:18s} M={M:6d} N={N:5d} K={Kd:6d} tile={t:3d} splitk={s:3d}: {us:8.1f} us {2.0 * M * N * Kd / us / 1e6:6.1f} TF', flush=True)


case('dW 512x512', 1, 1, 512, 512, 40960, 1, [128], [1, 4, 8, 16, 32, 64])
case('dW 512x512', 1, 1, 512, 512, 40960, 1, [64], [4, 8, 16, 32])
case('dW 1536x512', 1, 1, 1536, 512, 40960, 1, [128], [5, 8, 10, 16])
for M in (40960, 20480, 4096):
    for N in (512, 1536):
        case('fwd NT', 0, 0, M, N, 512, 0, [128, 64], [1])
        case('dX NN', 0, 1, M, 512, N, 0, [128, 64], [1])
