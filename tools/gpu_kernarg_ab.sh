#!/bin/bash
# A/B of HIP_FORCE_DEV_KERNARG (kernel arguments in device memory instead of host-coherent memory) on the launch-bound paths:
# the PPO cycle and the VP train step at B = 32 / 4096.  Each setting in fresh processes, twice, interleaved.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
cat > /tmp/ab.py <<'PY'
import sys, os, time, random
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, bench
from mansy_immersivevideostreaming_amd import dist as mdist
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
r = bench.bench_ppo(0, 1, torch.device('cuda', 0), mdist, cycles=10, warmup=3, rollout_probe=False)
out = {'ppo_ms': r['ms_per_cycle']}
for B in (32, 4096):
    torch.manual_seed(5); random.seed(5); np.random.seed(5)
    m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device='cuda').to('cuda'); m.train()
    opt = FusedAdamW(m, lr=1e-4)
    h, c, f = (t.cuda() for t in bench.synthetic_trajectories(B, 10, 10, seed=5))
    for _ in range(5): m.train_step(h, c, f, opt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20 if B == 4096 else 100
    for _ in range(n): m.train_step(h, c, f, opt)
    torch.cuda.synchronize(); out[f'vp_b{B}_ms'] = (time.perf_counter() - t0) / n * 1e3
print(os.environ.get('HIP_FORCE_DEV_KERNARG', 'unset'), out)
PY
for rep in 1 2; do
  for v in unset 0 1; do
    if [ $v = unset ]; then env -u HIP_FORCE_DEV_KERNARG python3 /tmp/ab.py 2>/dev/null | tail -1; else HIP_FORCE_DEV_KERNARG=$v python3 /tmp/ab.py 2>/dev/null | tail -1; fi
  done
done | tee gpurun_out/kernarg_ab.txt
