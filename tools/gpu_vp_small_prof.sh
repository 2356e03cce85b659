#!/bin/bash
# rocprofv3 kernel stats of the VP train step at a small batch: bash tools/gpu_vp_small_prof.sh B S T [dir]   (dir: repo root to run in, default .)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05b; export TMPDIR=/tmp
ROOT=$(pwd)/${4:-.}
cat > /tmp/vp_small.py <<PY
import sys, os, random
sys.path.insert(0, '$ROOT')
import numpy as np, torch
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
from bench import synthetic_trajectories
B, S, T = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
torch.manual_seed(5); random.seed(5); np.random.seed(5)
m = ViewportTransformerMTIO(in_channel=2, fut_window=T, d_model=512, dim_feedforward=512, device='cuda').to('cuda'); m.train()
opt = FusedAdamW(m, lr=1e-4)
h, c, f = (t.cuda() for t in synthetic_trajectories(B, S, T, seed=5))
for _ in range(int(sys.argv[1])): m.train_step(h, c, f, opt)
torch.cuda.synchronize()
PY
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 /tmp/vp_small.py 10 $1 $2 $3 > gpurun_out/r05b/prof_small.log 2>&1; echo "trace rc=$?"
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05b/vp_train_b$1_s$2_t$3_kernel_stats_$(basename $ROOT).csv
head -16 "$f" | cut -c1-150
rm -rf gpurun_out/prof
