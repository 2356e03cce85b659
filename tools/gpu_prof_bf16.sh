cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/pb; export TMPDIR=/tmp
cat > /tmp/vp_only.py <<'PY'
import sys, os, random
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
from bench import synthetic_trajectories
torch.manual_seed(5); random.seed(5); np.random.seed(5)
m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device='cuda').to('cuda'); m.train()
m.precision = 'bf16'; m.two_stream = False
opt = FusedAdamW(m, lr=1e-4)
h, c, f = (t.cuda() for t in synthetic_trajectories(4096, 10, 10, seed=5))
for _ in range(4): m.train_step(h, c, f, opt)
torch.cuda.synchronize()
PY
rm -rf gpurun_out/pb/*; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pb -- python3 /tmp/vp_only.py > gpurun_out/pb.log 2>&1; echo "rc=$?"
f=$(find gpurun_out/pb -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r['TotalDurationNs']) for r in rows) / 4e6
print('kernel ms per step', tot)
for r in rows[:14]:
    print(f"{r['Name'][:80]:80s} calls/step={int(r['Calls'])/4:6.1f} avg_us={float(r['AverageNs'])/1e3:7.1f} ms/step={int(r['TotalDurationNs'])/4e6:6.3f}")
PY
