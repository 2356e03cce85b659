#!/bin/bash
# rocprofv3 kernel-trace + PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs) of the VP bench workload.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/prof gpurun_out/pmc_r gpurun_out/pmc_w
export TMPDIR=/tmp
cat > /tmp/vp_only.py <<'PY'
import sys, os, random
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
from bench import synthetic_trajectories
torch.manual_seed(5); random.seed(5); np.random.seed(5)
m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device='cuda').to('cuda'); m.train()
opt = FusedAdamW(m, lr=1e-4)
h, c, f = (t.cuda() for t in synthetic_trajectories(4096, 10, 10, seed=5))
for _ in range(int(sys.argv[1])): m.train_step(h, c, f, opt)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 /tmp/vp_only.py 5 > gpurun_out/prof.log 2>&1; echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r -- python3 /tmp/vp_only.py 2 > gpurun_out/pmc_r.log 2>&1; echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 /tmp/vp_only.py 2 > gpurun_out/pmc_w.log 2>&1; echo "pmc write rc=$?"
find gpurun_out/prof gpurun_out/pmc_r gpurun_out/pmc_w -name "*.csv" | head -20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
