#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
for B in 256 512 1024; do B=$B timeout 300 python tools/vp_modes_time.py bf16 2>&1 | grep -v amdgpu.ids | sed "s/^/B=$B /"; done
timeout 300 python - <<'PY'
import os, sys, random
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
dev = torch.device('cuda', 0)
for B in (256,):
    for dp in (False, True):
        torch.manual_seed(5); random.seed(5); np.random.seed(5)
        m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device=dev).to(dev); m.train()
        if dp:
            m.set_data_parallel(2, allreduce=lambda t: t.mul_(2.0))       # one rank standing for two identical ones
        opt = FusedAdamW(m, lr=1e-4)
        h, c, f = (t.to(dev) for t in bench.synthetic_trajectories(B, 10, 10, seed=6))
        m.precision = 'bf16'
        gs = (lambda g: None) if dp else None
        ls = [m.train_step(h, c, f, opt, grad_sync=gs).item() for _ in range(4)]
        print('B', B, 'dp', dp, ls, 'params finite', bool(torch.isfinite(m._flat_p).all()), flush=True)
PY
