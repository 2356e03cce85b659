#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/prof_ppo; export TMPDIR=/tmp
cat > /tmp/ppo_only.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench
from mansy_immersivevideostreaming_amd import dist as mdist
r = bench.bench_ppo(0, 1, torch.device('cuda', 0), mdist, cycles=int(sys.argv[1]), warmup=2, rollout_probe=False)
print(r['ms_per_cycle'], r['rollout_step_latency_us'])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ppo -- python3 /tmp/ppo_only.py 10 > gpurun_out/prof_ppo.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/prof_ppo.log
f=$(ls -t gpurun_out/prof_ppo/*/*kernel_stats.csv | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r['TotalDurationNs']) for r in rows); n = sum(int(r['Calls']) for r in rows)
NC = 14  # 10 timed + 2 warm-up + 2 roofline-leg cycles
print("total kernel ms per cycle", tot / NC / 1e6, "launches per cycle", n / NC)
for r in rows[:30]:
    print(f"{r['Name'][:70]:70s} calls/cyc={int(r['Calls'])/NC:7.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/cyc={int(r['TotalDurationNs'])/NC/1e6:7.3f}")
PY
