#!/usr/bin/env python3
"""LAB (round 5): where the time of a [4096, 512, 512] decoder-step launch of the B = 4096 step goes (the LDS-DMA loop, 64 x 64 tiles, two workgroups per CU): wall-clock stamps
of one workgroup (the -DMANSY_LAB build), for an early and a late workgroup of the launch, operands cold as in the chain.
    python3 tools/dma_phase_lab.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lab_knobs as KN  # noqa: E402
lab = KN.enter()
import torch
from mansy_immersivevideostreaming_amd import kernels as K

lab.mansy_lab_set_stamps.argtypes = [ctypes.c_void_p]
lab.mansy_lab_set_stamps.restype = None
stamps = torch.zeros(64, dtype=torch.int64, device='cuda')


def run(M, N, Kd, wg, bk, n=30, prec=None):
    A = torch.randn(M, Kd, device='cuda'); W = torch.randn((Kd, N) if bk else (N, Kd), device='cuda') / 22; b = torch.randn(N, device='cuda'); R = torch.randn(M, N, device='cuda')
    out = torch.empty(M, N, device='cuda'); src = torch.randn(M, Kd, device='cuda')
    if prec is not None:
        K.set_precision(prec)
        assert not bk
        planes = K.weight_planes(W, 3 if prec == 'bf16x6' else 2)          # (plain bf16 reads the leading plane of the two)
    acc = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tt = []
    for it in range(n):
        A.copy_(src); R.mul_(1.0)
        stamps.zero_(); stamps[63] = wg
        lab.mansy_lab_set_stamps(stamps.data_ptr())
        e0.record()
        if prec is None:
            K.gemm(A, W, False, bool(bk), bias=b, resid=R, out=out)
        else:
            K.gemm_planes(A, W, planes[0], transposed=False, bias=b, resid=R)
        e1.record()
        lab.mansy_lab_set_stamps(None)
        torch.cuda.synchronize()
        if it >= 5:
            acc.append(stamps.cpu()[:32].view(4, 8).clone()); tt.append(e0.elapsed_time(e1) * 1e3)
    K.set_precision('f32')
    t = torch.stack(acc).double()
    t[:, :, :5] = t[:, :, :5] - t[:, :1, :1]
    return t.median(dim=0).values, sorted(tt)[len(tt) // 2]


for (M, N, Kd, bk) in ((4096, 512, 512, 0), (4096, 512, 512, 1), (4096, 1536, 512, 0)):
    nwg = (M // 64) * (N // 64)
    for wg in (0, nwg // 2 + 3, nwg - 1):
        med, us = run(M, N, Kd, wg, bk)
        w = med[0]
        print(f'[{M}, {N}, {Kd}] {"NN" if bk else "NT"} launch {us:5.1f} us (event pair); workgroup {wg:4d} of {nwg}: first DMA out {w[1] * 10:5.0f} ns, tile 0 ready {w[2] * 10:5.0f}, K loop done {w[3] * 10:6.0f} '
              f'({int(w[6])} tiles, {w[5] * 10:5.0f} ns of it waiting at the tile barrier), epilogue done {w[4] * 10:6.0f}', flush=True)
for prec in ('bf16', 'bf16x3'):
    for wg in (0, 259):
        med, us = run(4096, 512, 512, wg, 0, prec=prec)
        w = med[0]
        print(f'[4096, 512, 512] NT {prec} (gemm_bf16h_kernel, weight planes): workgroup {wg:4d} of 512: first DMA out {w[1] * 10:5.0f} ns, tile 0 ready {w[2] * 10:5.0f}, K loop done {w[3] * 10:6.0f} '
              f'({int(w[6])} tiles, {w[5] * 10:5.0f} ns of it waiting at the tile barrier), epilogue done {w[4] * 10:6.0f}', flush=True)
