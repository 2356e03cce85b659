#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention or attn" > gpurun_out/t_attn.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t_attn.log
timeout 300 python - <<'PY'
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from mansy_immersivevideostreaming_amd import kernels as K
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for L in (10, 5):
    qkv = torch.randn(4096, L, 1536, device='cuda'); dout = torch.randn(4096, L, 512, device='cuda')
    out, P = K.attn_fwd_packed(qkv, 8, drop=(0.1, 7, 100))
    print(f'enc attn L={L}: fwd %.1f us  bwd %.1f us' % (timeit(lambda: K.attn_fwd_packed(qkv, 8, drop=(0.1, 7, 100))), timeit(lambda: K.attn_bwd_packed(qkv, P, dout, 8, drop=(0.1, 7, 100)))))
PY
