#!/usr/bin/env python3
"""Samples rocm-smi clocks/power while a GEMM loop runs (tuning aid)."""
import sys, os, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mansy_immersivevideostreaming_amd import kernels as K

stop = False
def sampler():
    while not stop:
        r = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True)
        keep = [l.strip() for l in r.stdout.splitlines() if 'sclk' in l or 'Power' in l or 'mclk' in l]
        print(time.time(), ' | '.join(keep), flush=True)
        time.sleep(0.3)

A = torch.randn(40960, 512, device='cuda'); B = torch.randn(1536, 512, device='cuda'); out = torch.zeros(40960, 1536, device='cuda')
K.gemm(A, B, out=out); torch.cuda.synchronize()
t = threading.Thread(target=sampler); t.start()
time.sleep(1.0)
print('--- gemm loop start', flush=True)
t0 = time.time(); n = 0
while time.time() - t0 < 4.0:
    for _ in range(50):
        K.gemm(A, B, out=out)
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
print(f'--- gemm loop end: {n} launches, {dt / n * 1e6:.1f} us each, {2.0 * 40960 * 1536 * 512 * n / dt / 1e12:.1f} TF', flush=True)
time.sleep(0.7)
stop = True; t.join()
