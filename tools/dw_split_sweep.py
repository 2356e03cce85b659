import sys, os
sys.path.insert(0, '/root/repo')
import torch
from mansy_immersivevideostreaming_amd import kernels as K
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, Kd) in [(512, 512, 40960), (1536, 512, 40960), (1024, 512, 20480)]:
    A = torch.randn(Kd, M, device='cuda'); B = torch.randn(Kd, N, device='cuda'); out = torch.zeros(M, N, device='cuda')
    for tile in (96, 64):
        for s in (8, 12, 16, 24, 32, 48):
            us = timeit(lambda: K.gemm(A, B, True, True, out=out, accumulate=True, force_tile=tile, force_splitk=s))
            print(f'dW M={M} N={N} K={Kd} tile={tile} splitk={s:3d}: {us:8.1f} us {2.0*M*N*Kd/us/1e6:6.1f} TF', flush=True)
