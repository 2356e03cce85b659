#!/usr/bin/env python3
"""The PPO cycle (bench.bench_ppo: 256 envs x 16 steps), a few repeats in one process: ms per cycle, env-steps/s, rollout step latency."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mansy_immersivevideostreaming_amd import dist as mdist
dev = torch.device('cuda', 0)
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    r = bench.bench_ppo(0, 1, dev, mdist, cycles=20, warmup=3, rollout_probe=True)
    print(f'ppo: {r["ms_per_cycle"]:.3f} ms/cycle, {r["value"]:.0f} env-steps/s, rollout step {r["rollout_step_latency_us"]} us, loss {r["final_loss"]:.6f}', flush=True)
