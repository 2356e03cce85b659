#!/usr/bin/env python3
"""Two ranks (shared GPU, gloo) running the VP bf16-storage step data-parallel as bench.py does; prints per-step loss / finiteness of gradients and parameters.
  MANSY_DIST_BACKEND=gloo MANSY_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/vp_dp2_probe.py"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import numpy as np, torch, torch.distributed as dist
import bench
from mansy_immersivevideostreaming_amd import dist as mdist
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
rank, world, local = mdist.init_process_group()
dev = torch.device('cuda', local); torch.cuda.set_device(dev)
B = int(os.environ.get('B', 256))
torch.manual_seed(5); random.seed(5); np.random.seed(5)
m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device=dev).to(dev); m.train()
if os.environ.get('FAKE') == '1':
    m.set_data_parallel(2, allreduce=lambda t: t.mul_(2.0))
else:
    m.set_data_parallel(world)
opt = FusedAdamW(m, lr=1e-4)
h, c, f = (t.to(dev) for t in bench.synthetic_trajectories(B, 10, 10, seed=5 + rank))
overlapped = os.environ.get('OVERLAP', '1') == '1'
gs = mdist.OverlappedGradSync(world, dev) if overlapped else mdist.make_grad_sync(world)
if os.environ.get('FAKE') == '1':
    gs = lambda g: None
for mode in os.environ.get('MODES', 'f32,bf16x6,bf16').split(','):
    m.precision = mode
    for i in range(4):
        loss = m.train_step(h, c, f, opt, grad_sync=gs)
        torch.cuda.synchronize()
        print(f'rank {rank} {mode} step {i} loss {loss.item():.6f} grads finite {bool(torch.isfinite(m._flat_g).all())} params finite {bool(torch.isfinite(m._flat_p).all())}', flush=True)
        if i == 0 and not bool(torch.isfinite(m._flat_g).all()):
            names = [n for n, _ in m._param_table()]
            bad = []
            for k, n in enumerate(names):
                o = m._offsets[k]; e = m._offsets[k + 1] if k + 1 < len(names) else m._flat_g.numel()
                g = m._flat_g[o:e]
                if not bool(torch.isfinite(g).all()):
                    bad.append((n, int((~torch.isfinite(g)).sum()), g.numel()))
            print(f'rank {rank} non-finite gradients:', bad[:40], flush=True)
dist.barrier(); dist.destroy_process_group()
