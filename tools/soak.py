#!/usr/bin/env python3
"""Soak: many VP train steps and PPO cycles in a row -- losses stay finite, VP loss falls on a fixed synthetic set, PPO return
normaliser / parameters stay finite, no memory growth."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mansy_immersivevideostreaming_amd import dist as mdist
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW

torch.manual_seed(5); random.seed(5); np.random.seed(5)
dev = torch.device('cuda', 0)
m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device=dev).to(dev); m.train()
opt = FusedAdamW(m, lr=1e-4)
m.precision = os.environ.get('SOAK_PRECISION') or None          # e.g. bf16: the bf16-storage mode (round 6)
sets = [tuple(t.to(dev) for t in bench.synthetic_trajectories(1024, 10, 10, seed=s)) for s in range(8)]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
losses = []
mem0 = None
for i in range(steps):
    h, c, f = sets[i % len(sets)]
    losses.append(m.train_step(h, c, f, opt))
    if i == 50:
        torch.cuda.synchronize(); mem0 = torch.cuda.memory_allocated()
l = torch.stack(losses).cpu().numpy()
print('VP steps', steps, 'precision', m.precision, 'loss first/last 50 mean', l[:50].mean(), l[-50:].mean(), 'finite', np.isfinite(l).all(),
      'mem growth MB', (torch.cuda.memory_allocated() - mem0) / 1e6)
m.eval()
with torch.no_grad():
    p = m.sample(sets[0][0], sets[0][1])
print('sample finite', torch.isfinite(p).all().item(), 'range', p.min().item(), p.max().item())
r = bench.bench_ppo(0, 1, dev, mdist, cycles=int(sys.argv[2]) if len(sys.argv) > 2 else 300, warmup=2, rollout_probe=False)
print('PPO', r['value'], 'final loss', r['final_loss'], 'finite', np.isfinite(r['final_loss']))
