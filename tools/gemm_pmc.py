#!/usr/bin/env python3
"""One GEMM shape a few times (for `rocprofv3 --pmc ...`): python3 tools/gemm_pmc.py M N K a_kmajor b_kmajor [accumulate]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mansy_immersivevideostreaming_amd import kernels as K
M, N, Kd, ak, bk = (int(x) for x in sys.argv[1:6])
acc = len(sys.argv) > 6 and sys.argv[6] == '1'
A = torch.randn((Kd, M) if ak else (M, Kd), device='cuda'); B = torch.randn((Kd, N) if bk else (N, Kd), device='cuda')
out = torch.zeros(M, N, device='cuda')
for _ in range(6):
    K.gemm(A, B, bool(ak), bool(bk), out=out, accumulate=acc)
torch.cuda.synchronize()
