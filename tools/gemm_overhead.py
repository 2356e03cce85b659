#!/usr/bin/env python3
"""Fixed cost of a decoder-step product: [4096, N=512] x K for K = 32 .. 512 (back-to-back launches on one stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mansy_immersivevideostreaming_amd import kernels as K

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

M, N = 4096, 512
out = torch.zeros(M, N, device='cuda')
bias = torch.randn(N, device='cuda'); resid = torch.randn(M, N, device='cuda')
for Kd in (32, 64, 128, 256, 512, 1024):
    A = torch.randn(M, Kd, device='cuda'); B = torch.randn(N, Kd, device='cuda')
    t_plain = timeit(lambda: K.gemm(A, B, out=out))
    t_ep = timeit(lambda: K.gemm(A, B, out=out, bias=bias, resid=resid))
    ideal = 2.0 * M * N * Kd / 157.3e6
    print(f'K={Kd:5d}: plain {t_plain:6.2f} us, bias+resid {t_ep:6.2f} us, MFMA ideal {ideal:6.2f} us')
x = torch.zeros(64, device='cuda')
print('tiny torch kernel back-to-back: %.2f us' % timeit(lambda: x.add_(1.0)))
