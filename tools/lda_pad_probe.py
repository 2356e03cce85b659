#!/usr/bin/env python3
"""Does a power-of-two row stride of the activations cost L2 bandwidth (channel camping)?  The forward product [M, 512] x [N, 512]^T with
A's row stride 512 (as the engine stores activations) against padded strides, exact fp32 and bf16x3 with pre-split weights; same for
the output's row stride."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mansy_immersivevideostreaming_amd import kernels as K
from mansy_immersivevideostreaming_amd._lib import lib, ptr, stream_ptr, GemmEpilogue, check

L = lib()
def run(M, N, Kd, lda, ldc, prec, planes, W):
    A = torch.randn(M, lda, device='cuda'); C = torch.empty(M, ldc, device='cuda')
    ep = GemmEpilogue(); ep.mask_scale = 1.0
    K.set_precision(prec)
    if planes is not None:
        f = lambda: check(L.mansy_gemm_planes(ptr(A), lda, ptr(W), Kd, 0, ptr(planes), planes.stride(0), planes.stride(1), ptr(C), ldc, M, N, Kd,
                                              ctypes.byref(ep), 0, stream_ptr(A.device)), 'gemm_planes')
    else:
        f = lambda: check(L.mansy_gemm_f32(ptr(A), lda, 0, ptr(W), Kd, 0, ptr(C), ldc, M, N, Kd, ctypes.byref(ep), 0, 0, stream_ptr(A.device)), 'gemm')
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): f()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 30 * 1e6
    ref = A[:, :Kd].double() @ W.double().t()
    err = ((C[:, :N].double() - ref).abs().max() / ref.abs().max()).item()
    return us, err

for (M, N, Kd) in [(40960, 1536, 512), (40960, 512, 512), (4096, 512, 512)]:
    W = torch.randn(N, Kd, device='cuda')
    for prec in ('f32', 'bf16x3'):
        line = f'{prec:7s} {M}x{N}x{Kd}:'
        cfgs = [(Kd, N), (Kd + 16, N), (Kd + 32, N), (Kd + 64, N), (Kd + 32, N + 32), (Kd, N + 32)]
        best = {c: 1e9 for c in cfgs}
        K.set_precision(prec)
        pl = K.weight_planes(W, 2)[0] if prec == 'bf16x3' else None
        for rep in range(3):                     # interleaved rounds, minimum per configuration (the first round also warms the clocks)
            for c in cfgs:
                us, err = run(M, N, Kd, c[0], c[1], prec, pl, W)
                best[c] = min(best[c], us)
                assert err < 1e-4, err
        for c in cfgs:
            line += f'  lda {c[0]} ldc {c[1]}: {best[c]:7.1f} us'
        print(line)
