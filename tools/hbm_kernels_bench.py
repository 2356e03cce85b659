#!/usr/bin/env python3
"""HBM-side throughput of the byte / integer kernels of the path at sizes large enough to leave the launch floor:
tile hit maps + IoU (V11), trajectory gather (V1), environment step (P8-P12).  Algorithmic bytes / measured time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mansy_immersivevideostreaming_amd import kernels as K
from mansy_immersivevideostreaming_amd._lib import check, lib, ptr, stream_ptr
from mansy_immersivevideostreaming_amd.bitrate_selection.envs.mansy_env import EnvTables, MANSYVecEnv


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


n = 1 << 24
xy = torch.rand(n, 2, device='cuda')
t = timeit(lambda: K.tilemap(xy))
print(f'tilemap       n={n}: {t*1e6:8.1f} us  {n * 16 / t / 1e9:7.1f} GB/s (8 B in + 8 B out per point)')
a, b = K.tilemap(xy), K.tilemap(torch.rand(n, 2, device='cuda'))
t = timeit(lambda: K.tilemap_iou(a, b))
print(f'tilemap_iou   n={n}: {t*1e6:8.1f} us  {n * 24 / t / 1e9:7.1f} GB/s (16 B in + 8 B out per pair)')
# trajectory gather: B windows of 21 samples from a [traces, 300, 2] table
traces, L, B, S, T = 4096, 300, 1 << 18, 10, 10
table = torch.rand(traces, L, 2, device='cuda')
sel = torch.stack([torch.randint(0, traces, (B,), device='cuda'), torch.randint(S, L - T, (B,), device="cuda")], 1).int().contiguous()
hist, cur, fut = (torch.empty(B, k, 2, device='cuda') for k in (S, 1, T))
t = timeit(lambda: check(lib().mansy_traj_gather(ptr(table), L, 2, ptr(sel), B, S, T, ptr(hist), ptr(cur), ptr(fut), stream_ptr(table.device)), 'gather'))
print(f'traj_gather   B={B}: {t*1e6:8.1f} us  {B * (21 * 8 * 2 + 8) / t / 1e9:7.1f} GB/s (168 B read + 168 B written + 8 B index per window)')
# environment step: N environments, one wave each
for N in (256, 16384, 65536):
    tb = EnvTables.synthetic('cuda', seed=5, train_identifier_reward=True, n_sample=max(240, 1024))
    venv = MANSYVecEnv(tb, N, seed=1)
    venv.reset()
    act = torch.randint(0, 15, (N,), device='cuda', dtype=torch.int32)
    t = timeit(lambda: venv.step(act), n=10)
    byt = N * (2 * 3120 + 2 * 1280 + 128 + 64)
    print(f'env_step      N={N:6d}: {t*1e6:8.1f} us  {byt / t / 1e9:7.1f} GB/s ({N / t / 1e6:6.1f} M env-steps/s; 2 x 3 120 B observations written, '
          f'2 x 1 280 B manifest rows + viewport + trace bins read)')
