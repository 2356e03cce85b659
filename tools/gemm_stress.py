#!/usr/bin/env python3
"""Race screen for the LDS-DMA GEMM loop: many back-to-back launches over shapes / tiles / layouts, every result compared with
an fp64 reference (a rare early LDS read would show as a few wrong tiles on some launches only)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mansy_immersivevideostreaming_amd import kernels as K

torch.manual_seed(0)
shapes = [(4096, 512, 512), (4096, 1536, 512), (4096, 512, 1536), (2048, 1024, 256), (1000, 768, 128), (8192, 512, 512), (512, 512, 8192)]
bad = 0
total = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    for (M, N, Kd), (ak, bk), tile in itertools.product(shapes, [(0, 0), (0, 1), (1, 1)], [0, 64, 96, 128]):
        A = torch.randn((Kd, M) if ak else (M, Kd), device='cuda')
        B = torch.randn((Kd, N) if bk else (N, Kd), device='cuda')
        ref = ((A.double().t() if ak else A.double()) @ (B.double() if bk else B.double().t()))
        outs = [K.gemm(A, B, bool(ak), bool(bk), force_tile=tile) for _ in range(6)]
        torch.cuda.synchronize()
        tol = 2e-6 * ((A.double().abs().t() if ak else A.double().abs()) @ (B.double().abs() if bk else B.double().abs().t())) + 1e-6
        for o in outs:
            total += 1
            nbad = int(((o.double() - ref).abs() > tol).sum().item())
            if nbad:
                bad += 1
                print('MISMATCH', M, N, Kd, ak, bk, tile, nbad, flush=True)
print(f'{total} launches checked, {bad} with mismatches')
sys.exit(1 if bad else 0)
