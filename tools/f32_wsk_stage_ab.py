#!/usr/bin/env python3
"""A/B of the wave-split-K loop's stages per wave (mansy_gemm_f32_wsk(4) = one 8 KB stage per wave everywhere, (5) = two stages where the launch has
<= 512 workgroups), variants interleaved in ONE process: the small product shapes of the PPO cycle alone, then the whole cycle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lab_knobs as KN  # noqa: E402  (ABI 8: the A/B knobs live in the -DMANSY_LAB build only)
KN.enter()
import torch
import bench
from mansy_immersivevideostreaming_amd import dist as mdist, kernels as K
from mansy_immersivevideostreaming_amd._lib import lib
L = lib()
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(1)
def timed(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(gph, stream=s):
            for _ in range(50): fn()
    gph.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): gph.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 200 * 1e3
for (M, N, Kd, bk) in ((256, 1280, 320, 0), (512, 1280, 320, 0), (512, 1280, 256, 1), (256, 128, 96, 0), (300, 132, 64, 1), (33, 36, 32, 0), (256, 512, 1280, 0)):
    A = torch.randn(M, Kd, generator=g).to(dev); B = torch.randn((Kd, N) if bk else (N, Kd), generator=g).to(dev); out = torch.zeros(M, N, device=dev)
    ref = A.double() @ (B.double() if bk else B.double().t())
    line = f'gemm M={M} N={N} K={Kd} {"NN" if bk else "NT"}:'
    for v in (4, 5):
        KN.f32_wsk(v); out.zero_()
        K.gemm(A, B, False, bool(bk), out=out, force_tile=64)
        err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
        line += f'  stages {v - 3}: {timed(lambda: K.gemm(A, B, False, bool(bk), out=out, force_tile=64)):6.2f} us (err {err:.1e})'
    print(line, flush=True)
for (M, N, Kd) in ((128, 1280, 512), (256, 1280, 512), (36, 132, 96)):
    A = torch.randn(Kd, M, generator=g).to(dev); B = torch.randn(Kd, N, generator=g).to(dev)
    ref = A.double().t() @ B.double() + 1.0
    line = f'gemm TN M={M} N={N} K={Kd}:'
    for v in (4, 5):
        KN.f32_wsk(v)
        out = torch.ones(M, N, device=dev); rs = torch.zeros(M, device=dev)
        K.gemm(A, B, True, True, out=out, accumulate=True, a_rowsum=rs, force_tile=64)
        err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
        rerr = ((rs.double() - A.double().sum(0)).abs().max() / A.double().sum(0).abs().max()).item()
        scratch = torch.zeros(M, N, device=dev)
        line += f'  stages {v - 3}: {timed(lambda: K.gemm(A, B, True, True, out=scratch, accumulate=True, force_tile=64)):6.2f} us (err {err:.1e}, rowsum err {rerr:.1e})'
    print(line, flush=True)
for rnd in range(3):
    for v in (4, 5):
        KN.f32_wsk(v)
        r = bench.bench_ppo(0, 1, dev, mdist, cycles=20, warmup=3, rollout_probe=True)
        print(f'ppo stages {v - 3}: {r["ms_per_cycle"]:.3f} ms/cycle, {r["value"]:.0f} env-steps/s, rollout step {r["rollout_step_latency_us"]} us, loss {r["final_loss"]:.6f}', flush=True)
KN.f32_wsk(5)
