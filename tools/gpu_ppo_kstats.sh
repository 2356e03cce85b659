#!/bin/bash
# rocprofv3 kernel stats of the PPO cycle only (no PMC passes): per-kernel average durations
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/pk; export TMPDIR=/tmp
cat > /tmp/ppo_only.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench
from mansy_immersivevideostreaming_amd import dist as mdist
r = bench.bench_ppo(0, 1, torch.device('cuda', 0), mdist, cycles=10, warmup=2, rollout_probe=False)
print(r['ms_per_cycle'])
PY
rm -rf gpurun_out/pk/*; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pk -- python3 /tmp/ppo_only.py > gpurun_out/pk.log 2>&1; echo "rc=$?"
f=$(find gpurun_out/pk -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f"{r['Name'][:74]:74s} calls={int(r['Calls'])/14:6.1f}/cyc avg_us={float(r['AverageNs'])/1e3:7.1f}")
PY
