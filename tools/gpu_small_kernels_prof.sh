#!/bin/bash
# rocprofv3 kernel stats of 4 VP train steps (single stream): durations of the row-reduction kernels (outer_reduce, distill_bwd_stage1, colstats, ...)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/sk; export TMPDIR=/tmp
cat > /tmp/vp_only.py <<'PY'
import sys, os, random
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
from bench import synthetic_trajectories
torch.manual_seed(5); random.seed(5); np.random.seed(5)
m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device='cuda').to('cuda'); m.train()
m.two_stream = False
opt = FusedAdamW(m, lr=1e-4)
h, c, f = (t.cuda() for t in synthetic_trajectories(4096, 10, 10, seed=5))
for _ in range(int(sys.argv[1])): m.train_step(h, c, f, opt)
torch.cuda.synchronize()
PY
rm -rf gpurun_out/sk/*; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sk -- python3 /tmp/vp_only.py 4 > gpurun_out/sk.log 2>&1; echo "rc=$?"
f=$(find gpurun_out/sk -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('outer_reduce', 'distill', 'colstats', 'bn_elu', 'im2col', 'col2im', 'ln_partials', 'embed', 'mtio')):
        print(f"{n[:70]:70s} calls={r['Calls']:>4} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
