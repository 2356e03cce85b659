#!/bin/bash
# Round-6 profile set (as round 5 + the bf16-storage mode) (as round 3; the VP step is profiled single-stream: two_stream off; new: PPO meta + PMC, per-shape GEMM traffic) (run on the GPU box through gpurun; outputs under gpurun_out/r02, summaries are copied to profiles/ by hand):
#   rocprofv3 kernel-trace + stats of the VP train step (B=4096) per precision mode, step breakdowns, PMC passes (FETCH_SIZE /
#   WRITE_SIZE in SEPARATE runs, no tracing alongside) for the fp32 and the split-bf16 GEMM kernels, the PPO cycle, and the
#   byte / integer kernels (tools/hbm_kernels_bench.py).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/vp_only.py <<'PY'
import sys, os, random
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW
from bench import synthetic_trajectories
torch.manual_seed(5); random.seed(5); np.random.seed(5)
m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device='cuda').to('cuda'); m.train()
m.precision = sys.argv[2] if len(sys.argv) > 2 else None
m.two_stream = False     # per-kernel durations: one stream (concurrent kernels stretch each other)
opt = FusedAdamW(m, lr=1e-4)
h, c, f = (t.cuda() for t in synthetic_trajectories(4096, 10, 10, seed=5))
for _ in range(int(sys.argv[1])): m.train_step(h, c, f, opt)
torch.cuda.synchronize()
PY
for mode in f32 bf16 bf16x3 bf16x6; do
  rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 /tmp/vp_only.py 5 $mode > $OUT/prof_$mode.log 2>&1; echo "trace $mode rc=$?"
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/vp_train_b4096_${mode}_kernel_stats.csv
  python3 tools/step_breakdown.py 45 > $OUT/vp_step_breakdown_$mode.txt 2>&1
done
for mode in f32 bf16 bf16x3 bf16x6; do
  rm -rf gpurun_out/pmc_r gpurun_out/pmc_w; mkdir -p gpurun_out/pmc_r gpurun_out/pmc_w
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r -- python3 /tmp/vp_only.py 2 $mode > $OUT/pmc_r_$mode.log 2>&1; echo "pmc fetch $mode rc=$?"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 /tmp/vp_only.py 2 $mode > $OUT/pmc_w_$mode.log 2>&1; echo "pmc write $mode rc=$?"
  if [ $mode = f32 ]; then python3 tools/pmc_aggregate.py $TAG 50011000 gemm_f32 ""; else python3 tools/pmc_aggregate.py $TAG 50011000 gemm_bf16 _$mode; fi
done
cp profiles/${TAG}_pmc_gemm*.json $OUT/ 2>/dev/null
# PPO cycle
bash tools/gpu_prof_ppo.sh > $OUT/ppo_prof_summary.txt 2>&1
f=$(ls -t gpurun_out/prof_ppo/*/*kernel_stats.csv | head -1); [ -n "$f" ] && cp "$f" $OUT/ppo_kernel_stats.csv
cp gpurun_out/prof_ppo/ppo_kernel_stats.meta.json $OUT/ppo_kernel_stats.meta.json 2>/dev/null
cp gpurun_out/prof_ppo/ppo_pmc.json $OUT/ppo_pmc.json 2>/dev/null
cp gpurun_out/prof_ppo/ppo_traffic_by_kernel.txt $OUT/ppo_traffic_by_kernel.txt 2>/dev/null
# per-shape HBM-side traffic of the fp32 GEMM launches
bash tools/gpu_gemm_traffic.sh > $OUT/gemm_traffic_by_shape.txt 2>&1
# byte / integer kernels
rm -rf gpurun_out/prof_hbm; mkdir -p gpurun_out/prof_hbm
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hbm -- python3 tools/hbm_kernels_bench.py > $OUT/hbm_kernels_bench.txt 2>&1; echo "hbm rc=$?"
f=$(find gpurun_out/prof_hbm -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/hbm_kernels_kernel_stats.csv
python3 tools/hbm_kernels_bench.py > $OUT/hbm_kernels_bench_unprofiled.txt 2>&1
ls -la $OUT
