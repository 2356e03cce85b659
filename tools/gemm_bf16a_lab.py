#!/usr/bin/env python3
"""A/B of bf16-storage product variants (round 6): ring depth of the 128-wide tiles, split count / ring depth of the weight-gradient form."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mansy_immersivevideostreaming_amd import _lib as L
dev = 'cuda'; lib = L.lib(); NSET = 6

def timeit(fn, iters=40):
    for i in range(6): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(iters): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e6

def nn(M, N, K, tile, out):
    A = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(NSET)]
    W = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = [torch.empty(M, N, device=dev) for _ in range(NSET)]
    C16 = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(NSET)]
    ep = L.GemmEpilogue(); st = L.stream_ptr()
    def f(i):
        k = i % NSET
        L.check(lib.mansy_gemm_bf16(L.ptr(A[k]), K, 0, L.ptr(W), K, 0, L.ptr(C[k]) if out != 'bf16' else None, N, L.ptr(C16[k]) if out != 'f32' else None, N, M, N, K, ctypes.byref(ep), tile, 0, st))
    return timeit(f)

def tn(M, N, K, splits, ns):
    dY = [torch.randn(K, M, device=dev).to(torch.bfloat16) for _ in range(NSET)]
    X = [torch.randn(K, N, device=dev).to(torch.bfloat16) for _ in range(NSET)]
    C = torch.zeros(M, N, device=dev); rs = torch.zeros(M, device=dev)
    ep = L.GemmEpilogue(); ep.accumulate = 1; ep.a_rowsum = L.ptr(rs); st = L.stream_ptr()
    def f(i):
        k = i % NSET
        L.check(lib.mansy_gemm_bf16(L.ptr(dY[k]), M, 1, L.ptr(X[k]), N, 1, L.ptr(C), N, None, 0, M, N, K, ctypes.byref(ep), ns, splits, st))
    return timeit(f, 30)

for (M, N, K) in ((4096, 512, 512), (4096, 1536, 512)):
    print(f'NN [{M},{N},{K}] bf16-out: ' + '  '.join(f'tile {t}: {nn(M, N, K, t, "bf16"):6.1f}' for t in (64, 66, 67, 96, 98)) + '   f32-out: ' + '  '.join(f'tile {t}: {nn(M, N, K, t, "f32"):6.1f}' for t in (64, 66, 67, 96, 98)), flush=True)
for (M, N, K) in ((40960, 512, 512), (40960, 1536, 512), (40960, 512, 1536), (20480, 1024, 512)):
    print(f'NN [{M},{N},{K}] bf16-out: ' + '  '.join(f'tile {t}: {nn(M, N, K, t, "bf16"):6.1f}' for t in (96, 98, 128, 130)) + '   f32-out: ' + '  '.join(f'tile {t}: {nn(M, N, K, t, "f32"):6.1f}' for t in (96, 98, 128, 130)), flush=True)
for (M, N, K) in ((512, 512, 40960), (1536, 512, 40960), (512, 1536, 40960), (1024, 512, 20480)):
    print(f'TN [{M},{N},{K}]: ' + '  '.join(f'splits {s or "auto"}: {tn(M, N, K, s, 0):6.1f}' for s in (0, 8, 16, 32)), flush=True)
