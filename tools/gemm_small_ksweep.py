import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from mansy_immersivevideostreaming_amd import kernels as K
dev = 'cuda'
for (M, N) in ((256, 1280), (512, 1280), (256, 128), (512, 256)):
    line = f'M={M} N={N}:'
    for Kd in (32, 64, 128, 320, 640, 1280):
        A = torch.randn(M, Kd, device=dev); B = torch.randn(N, Kd, device=dev); out = torch.zeros(M, N, device=dev)
        for _ in range(5): K.gemm(A, B, False, False, out=out, force_tile=64)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): K.gemm(A, B, False, False, out=out, force_tile=64)
        e1.record(); torch.cuda.synchronize()
        line += f'  K={Kd}: {e0.elapsed_time(e1) / 100 * 1e3:5.1f} us'
    print(line, flush=True)
