#!/usr/bin/env python3
"""LAB (round 5): the weight-gradient products dW = dY^T X of the VP step at the reference's REAL batch sizes (K = B x T or B x S rows: 1280 .. 15360
instead of the 40960 the split heuristics of csrc/gemm_f32.hip were tuned on).  Every (tile, split) form next to what the heuristics pick today.
    python3 tools/dw_mid_sweep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mansy_immersivevideostreaming_amd import kernels as K


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [(512, 512, k) for k in (960, 1280, 2560, 3840, 5120, 7680, 15360)] + [(1536, 512, k) for k in (2560, 7680, 15360)] + \
         [(512, 1536, 2560), (1024, 512, 1536), (1024, 512, 3072)]
for (M, N, Kd) in shapes:
    A = torch.randn(Kd, M, device='cuda'); B = torch.randn(Kd, N, device='cuda'); out = torch.zeros(M, N, device='cuda'); rs = torch.zeros(M, device='cuda')
    base = timeit(lambda: K.gemm(A, B, True, True, out=out, accumulate=True, a_rowsum=rs))
    line = f'dW M={M:4d} N={N:4d} K={Kd:5d}: heuristics {base:6.1f} us |'
    best = (base, 'heur')
    for name, tile, var in (('wsk', 64, 0), ('t64', 64, K.VARIANT_NO_WSK_TN), ('t96', 96, 0)):
        for s in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            if Kd // s < 256:
                continue
            if name == 'wsk' and ((M + 63) // 64) * ((N + 63) // 64) * s > 256:
                continue
            us = timeit(lambda: K.gemm(A, B, True, True, out=out, accumulate=True, a_rowsum=rs, force_tile=tile, force_splitk=s, variant=var))
            line += f' {name}/{s}:{us:5.1f}'
            if us < best[0]:
                best = (us, f'{name}/{s}')
    print(line + f' || best {best[1]} {best[0]:.1f} us ({2.0 * M * N * Kd / best[0] / 1e6:.0f} TF)', flush=True)
