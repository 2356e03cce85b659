#!/usr/bin/env python3
"""Does the KV-cache LAYOUT matter for the decode attention kernels?  attn_fwd (Lq = 1, Lk = 10, B = 4096, 8 heads x 64) on the engine's step-major
[T][B][q|k|v] slabs (K / V = 4 KB of every 6 KB row) against a K|V-only cache [T][B][k|v] (fully contiguous streams) and a batch-major one."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mansy_immersivevideostreaming_amd import kernels as K
from mansy_immersivevideostreaming_amd._lib import check, lib, ptr, stream_ptr
B, H, d, T = 4096, 8, 512, 10
dev = 'cuda'
def run(name, k_bs, k_rs, koff, voff, buf, qbuf, q_bs):
    out = torch.empty(B, d, device=dev); P = torch.empty(B * H, T, device=dev)
    s = K._attn_shape(B, H, 1, T, d // H, (q_bs, 0), (k_bs, k_rs), (k_bs, k_rs), (d, 0))
    base = buf.data_ptr()
    fn = lambda: check(lib().mansy_attn_fwd(qbuf.data_ptr(), base + 4 * koff, base + 4 * voff, ptr(out), ptr(P), ctypes.byref(s), 0.0, 0, 0, stream_ptr(buf.device)), 'attn')
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f'{name:44s} {us:7.1f} us  {(2 * T + 2) * B * d * 4 / us / 1e6:6.2f} TB/s', flush=True)
    return out
for rnd in range(2):
    slab = torch.randn(T, B, 3 * d, device=dev)                       # engine layout: step-major [q|k|v] rows
    o1 = run('step-major [T][B][q|k|v] (engine)', 3 * d, B * 3 * d, d, 2 * d, slab, slab[T - 1], 3 * d)
    kv = torch.randn(T, B, 2 * d, device=dev); q = torch.randn(B, d, device=dev)
    run('step-major [T][B][k|v], q apart', 2 * d, B * 2 * d, 0, d, kv, q, d)
    kvb = torch.randn(B, T, 2 * d, device=dev)
    run('batch-major [B][T][k|v], q apart', T * 2 * d, 2 * d, 0, d, kvb, q, d)
