#!/usr/bin/env python3
"""One bf16-STORAGE product shape a few times (for `rocprofv3 --pmc ...`): python3 tools/gemm_bf16a_pmc.py M N K form   (form: nn = forward / dX with a bf16
output image; tn = weight gradient, K the reduce dimension)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mansy_immersivevideostreaming_amd import _lib as L
M, N, K = (int(x) for x in sys.argv[1:4]); form = sys.argv[4]
lib = L.lib(); st = L.stream_ptr(); ep = L.GemmEpilogue()
if form == 'nn':
    A = torch.randn(M, K, device='cuda').to(torch.bfloat16); W = torch.randn(N, K, device='cuda').to(torch.bfloat16)
    C16 = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    run = lambda: L.check(lib.mansy_gemm_bf16(L.ptr(A), K, 0, L.ptr(W), K, 0, None, N, L.ptr(C16), N, M, N, K, ctypes.byref(ep), None, None, 0, 0, st))
else:
    dY = torch.randn(K, M, device='cuda').to(torch.bfloat16); X = torch.randn(K, N, device='cuda').to(torch.bfloat16)
    C = torch.zeros(M, N, device='cuda'); rs = torch.zeros(M, device='cuda'); ep.accumulate = 1; ep.a_rowsum = L.ptr(rs)
    run = lambda: L.check(lib.mansy_gemm_bf16(L.ptr(dY), M, 1, L.ptr(X), N, 1, L.ptr(C), N, None, 0, M, N, K, ctypes.byref(ep), None, None, 0, 0, st))
for _ in range(6):
    run()
torch.cuda.synchronize()
