#!/usr/bin/env python3
"""Small-kernel floor probe: LayerNorm fwd/bwd on [4096, 512] vs torch elementwise ops of the same traffic (tuning aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mansy_immersivevideostreaming_amd import kernels as K
from mansy_immersivevideostreaming_amd._lib import lib, ptr, stream_ptr

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for rows in (4096, 40960):
    C = 512
    a = torch.randn(rows, C, device='cuda'); b = torch.randn(rows, C, device='cuda'); c = torch.empty_like(a); d2 = torch.empty_like(a)
    w = torch.randn(C, device='cuda'); bias = torch.randn(C, device='cuda')
    print(f'rows={rows}')
    print('  torch add (2R+1W)      %.1f us' % timeit(lambda: torch.add(a, b, out=c)))
    print('  torch copy (1R+1W)     %.1f us' % timeit(lambda: c.copy_(a)))
    y, z, mean, rstd = K.layernorm_fwd(a, b, w, bias)
    L = lib(); st = stream_ptr(a.device)
    dz = torch.empty_like(a); dzd = torch.empty_like(a); dw = torch.zeros(C, device='cuda'); db = torch.zeros(C, device='cuda')
    print('  LN bwd p=0.1 first     %.1f us' % timeit(lambda: L.mansy_layernorm_bwd(ptr(a), ptr(z), ptr(mean), ptr(rstd), ptr(w), ptr(dz), ptr(dzd), 0.1, 5, 77, ptr(dw), ptr(db), rows, C, st)))
    print('  LN fwd (2R+2W)         %.1f us' % timeit(lambda: L.mansy_layernorm_fwd(ptr(a), ptr(b), ptr(w), ptr(bias), ptr(z), ptr(y), ptr(mean), ptr(rstd), rows, C, 1e-5, st)))
    print('  LN bwd p=0   (2R+2W)   %.1f us' % timeit(lambda: L.mansy_layernorm_bwd(ptr(a), ptr(z), ptr(mean), ptr(rstd), ptr(w), ptr(dz), ptr(dzd), 0.0, 0, 0, ptr(dw), ptr(db), rows, C, st)))
    print('  LN bwd p=0.1 (2R+2W)   %.1f us' % timeit(lambda: L.mansy_layernorm_bwd(ptr(a), ptr(z), ptr(mean), ptr(rstd), ptr(w), ptr(dz), ptr(dzd), 0.1, 5, 77, ptr(dw), ptr(db), rows, C, st)))
    parts = L.mansy_layernorm_bwd_parts(rows); pt = torch.zeros(parts, 2, C, device='cuda')
    print('  LN bwd partial p=0.1 acc  %.1f us (parts %d)' % (timeit(lambda: L.mansy_layernorm_bwd_partial(ptr(a), ptr(z), ptr(mean), ptr(rstd), ptr(w), ptr(dz), ptr(dzd), 0.1, 5, 77, ptr(pt), 1, rows, C, st)), parts))
    print('  LN bwd no dz_drop      %.1f us' % timeit(lambda: L.mansy_layernorm_bwd(ptr(a), ptr(z), ptr(mean), ptr(rstd), ptr(w), ptr(dz), None, 0.0, 0, 0, ptr(dw), ptr(db), rows, C, st)))
