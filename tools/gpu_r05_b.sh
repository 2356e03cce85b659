#!/bin/bash
# Round 5, second GPU call: the no-copy peer all-reduce (selftest at 1/2/4/8 ranks on the shared GPU, dp_form leg), the small-batch step under rocprofv3.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 600 bash tools/gpu_xg_timing.sh > gpurun_out/r05_xg_timing.txt 2>&1; echo "xg rc=$?"; cat gpurun_out/r05_xg_timing.txt
timeout 600 python tools/r05_legs.py dp small > gpurun_out/r05_legs.json 2> gpurun_out/r05_legs.err; echo "legs rc=$?"; cat gpurun_out/r05_legs.json; tail -5 gpurun_out/r05_legs.err
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_ppo.py -m gpu -q --tb=short -x > gpurun_out/t_dist.log 2>&1; echo "dist+ppo tests rc=$?"; tail -8 gpurun_out/t_dist.log
timeout 600 python -m pytest tests/test_gpu_vp_engine.py -m gpu -q --tb=short -k "dropout or two_stream" > gpurun_out/t_drop.log 2>&1; echo "dropout tests rc=$?"; tail -8 gpurun_out/t_drop.log
cat > /tmp/vp_small.py <<'PY'
import sys, os, random
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
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
for cfg in "32 10 10" "512 5 15"; do
  set -- $cfg
  rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 /tmp/vp_small.py 10 $1 $2 $3 > gpurun_out/prof_small_$1.log 2>&1; echo "trace B=$1 rc=$?"
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_vp_train_b$1_s$2_t$3_kernel_stats.csv
  head -40 gpurun_out/r05_vp_train_b$1_s$2_t$3_kernel_stats.csv
done
rm -rf gpurun_out/prof
