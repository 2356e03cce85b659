#!/usr/bin/env python3
"""LAB (round 5): where the time of ONE small product goes.  The -DMANSY_LAB build of the wave-split-K loop notes the 100 MHz wall clock at its phase boundaries
(workgroup 0, lane 0 of each wave: gemm_wsk.h WSK_STAMP); this prints the phases with COLD operands (rewritten by another launch before
every product, as in the decoder chain) and warm.  (profiles/r05_wsk_phase_lab_two_stage.txt: the same table with the two-stage form that was tried and removed.)
    python3 tools/wsk_phase_lab.py"""
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
NAMES = {0: 'start', 1: 'first DMA issued', 14: 'loop done', 15: 'epilogue done'}
for t in range(4):
    NAMES[2 + 3 * t] = f'tile{t} landed'; NAMES[3 + 3 * t] = f'tile{t} in registers'; NAMES[4 + 3 * t] = f'tile{t} MFMAs issued'


def run(M, N, Kd, cold, epi, n=40):
    A = torch.randn(M, Kd, device='cuda'); W = torch.randn(N, Kd, device='cuda') / 22; b = torch.randn(N, device='cuda'); R = torch.randn(M, N, device='cuda')
    out = torch.empty(M, N, device='cuda'); src = torch.randn(M, Kd, device='cuda')
    acc = []
    for it in range(n):
        if cold:
            A.copy_(src); W.mul_(1.0); R.mul_(1.0)       # another launch rewrites the operands, as in the chain
        lab.mansy_lab_set_stamps(stamps.data_ptr())
        K.gemm(A, W, False, False, bias=b if epi else None, resid=R if epi else None, out=out)
        lab.mansy_lab_set_stamps(None)
        torch.cuda.synchronize()
        if it >= 5:
            acc.append(stamps.cpu().view(4, 16).clone())
    t = torch.stack(acc).double()
    t = torch.where(t > 0, t - t[:, :1, :1], torch.zeros_like(t))
    return t.median(dim=0).values


for (M, N, Kd) in ((32, 512, 512), (512, 512, 512), (256, 1280, 320)):
    for epi in (False, True):
        cols = {cold: run(M, N, Kd, cold, epi) for cold in (True, False)}
        print(f'--- M={M} N={N} K={Kd} {"bias + residual epilogue" if epi else "plain store"}: ns since the start of workgroup 0 (10 ns clock); wave 0 cold | warm | wave 3 cold | warm')
        for i in range(16):
            vals = [cols[c][w, i].item() * 10 for w in (0, 3) for c in (True, False)]
            if i > 0 and all(v == 0 for v in vals):
                continue
            print(f'   {NAMES.get(i, str(i)):24s} ' + ' '.join(f'{v:9.0f}' for v in vals))
