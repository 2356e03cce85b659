#!/usr/bin/env python3
"""The PPO cycle (bench.bench_ppo: 256 envs x 16 steps), repeats in one process: ms per cycle, env-steps/s, rollout step latency.
python tools/ppo_cycle_ab.py 4            -- the current build
python tools/ppo_cycle_ab.py 3 knob 8 9   -- interleaved A/B of mansy_gemm_f32_wsk(v) settings (8 / 9: paired launch off / on)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lab_knobs as KN  # noqa: E402  (ABI 8: the A/B knobs live in the -DMANSY_LAB build only)
KN.enter()
import torch
import bench
from mansy_immersivevideostreaming_amd import dist as mdist
from mansy_immersivevideostreaming_amd._lib import lib
dev = torch.device('cuda', 0)
vals = [int(x) for x in sys.argv[3:]] if len(sys.argv) > 3 and sys.argv[2] == 'knob' else [None]
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    for v in vals:
        if v is not None:
            KN.f32_wsk(v)
        r = bench.bench_ppo(0, 1, dev, mdist, cycles=20, warmup=3, rollout_probe=True)
        print(f'ppo{"" if v is None else f" knob {v}"}: {r["ms_per_cycle"]:.3f} ms/cycle, {r["value"]:.0f} env-steps/s, rollout step {r["rollout_step_latency_us"]} us, '
              f'launches {r.get("launches_per_cycle")}, loss {r["final_loss"]:.6f}', flush=True)
