#!/usr/bin/env python3
"""A/B of the operand-ring depth of the 64 x 64 fp32 LDS-DMA loop (mansy_gemm_f32_ring: 2 = one K-tile in flight, 3 / 4 = two / three), variants
interleaved in ONE process: the PPO cycle (bench.bench_ppo, 256 envs x 16 steps), the VP train step (B = 4096, fp32) and the [4096, 512, 512] product."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mansy_immersivevideostreaming_amd import dist as mdist, kernels as K
from mansy_immersivevideostreaming_amd._lib import lib
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
L = lib()
dev = torch.device('cuda', 0)
what = sys.argv[1:] or ['gemm', 'ppo', 'vp']
rings = (2, 3, 4)
if 'gemm' in what:
    for (M, N, Kd, bk) in ((4096, 512, 512, 0), (4096, 512, 512, 1), (256, 1280, 320, 0), (512, 1280, 320, 0), (4096, 1536, 512, 0)):
        A = torch.randn(M, Kd, device=dev); B = torch.randn((Kd, N) if bk else (N, Kd), device=dev); out = torch.zeros(M, N, device=dev)
        line = f'gemm M={M} N={N} K={Kd} {"NN" if bk else "NT"} tile 64:'
        for rg in rings:
            L.mansy_gemm_f32_ring(rg)
            for _ in range(5): K.gemm(A, B, False, bool(bk), out=out, force_tile=64)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): K.gemm(A, B, False, bool(bk), out=out, force_tile=64)
            e1.record(); torch.cuda.synchronize()
            line += f'  ring {rg}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us'
        print(line, flush=True)
if 'ppo' in what:
    for rnd in range(2):
        for rg in rings:
            L.mansy_gemm_f32_ring(rg)
            r = bench.bench_ppo(0, 1, dev, mdist, cycles=20, warmup=3, rollout_probe=True)
            print(f'ppo ring {rg}: {r["ms_per_cycle"]:.3f} ms/cycle, {r["value"]:.0f} env-steps/s, rollout step {r["rollout_step_latency_us"]} us', flush=True)
if 'vp' in what:
    torch.manual_seed(5); random.seed(5); np.random.seed(5)
    m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device='cuda').to('cuda'); m.train()
    opt = FusedAdamW(m, lr=1e-4)
    h, c, f = (t.cuda() for t in bench.synthetic_trajectories(4096, 10, 10, seed=5))
    for rnd in range(2):
        for rg in rings:
            L.mansy_gemm_f32_ring(rg)
            for _ in range(3): m.train_step(h, c, f, opt)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): m.train_step(h, c, f, opt)
            torch.cuda.synchronize()
            print(f'vp ring {rg}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms/step', flush=True)
L.mansy_gemm_f32_ring(2)
