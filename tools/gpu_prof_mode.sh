#!/bin/bash
# rocprofv3 kernel stats of the VP train step (B=4096, single stream) in ONE precision mode: bash tools/gpu_prof_mode.sh bf16x6 [tag]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
MODE=${1:-f32}; TAG=${2:-r03x}
cat > /tmp/vp_only.py <<'PY'
import sys, os, random
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
from bench import synthetic_trajectories
torch.manual_seed(5); random.seed(5); np.random.seed(5)
m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device='cuda').to('cuda'); m.train()
m.precision = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != 'f32' else None
m.two_stream = False
opt = FusedAdamW(m, lr=1e-4)
h, c, f = (t.cuda() for t in synthetic_trajectories(4096, 10, 10, seed=5))
for _ in range(int(sys.argv[1])): m.train_step(h, c, f, opt)
torch.cuda.synchronize()
PY
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 /tmp/vp_only.py 5 $MODE > gpurun_out/prof_$MODE.log 2>&1; echo "rc=$?"
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${TAG}_vp_train_b4096_${MODE}_kernel_stats.csv
head -8 "$f" | cut -c1-170
