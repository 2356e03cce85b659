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
import json, os, sys as _s
_s.path.insert(0, os.getcwd())
from mansy_immersivevideostreaming_amd import build_ext
import bench
NC = 10 * bench.PPO_BLOCKS + 4  # PPO_BLOCKS timed blocks of 10 cycles + 2 warm-up + 2 roofline-leg cycles
json.dump({'cycles': NC, 'source_digest': build_ext.source_digest(), 'what': 'rocprofv3 --kernel-trace --stats of bench.bench_ppo(cycles=10, warmup=2): %d timed blocks of 10 cycles + 2 warm-up + 2 roofline-leg cycles' % bench.PPO_BLOCKS},
          open('gpurun_out/prof_ppo/ppo_kernel_stats.meta.json', 'w'))
print("total kernel ms per cycle", tot / NC / 1e6, "launches per cycle", n / NC)
for r in rows[:30]:
    print(f"{r['Name'][:70]:70s} calls/cyc={int(r['Calls'])/NC:7.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/cyc={int(r['TotalDurationNs'])/NC/1e6:7.3f}")
PY

# HBM-side traffic of the whole cycle: FETCH_SIZE / WRITE_SIZE in SEPARATE PMC passes (no tracing), every kernel of the process, at TWO cycle
# counts: the per-cycle figure is the difference (the one-off set-up kernels -- synthetic tables, env init, first-call allocations -- cancel)
for c in FETCH_SIZE WRITE_SIZE; do for n in 2 10; do
  rm -rf gpurun_out/prof_ppo_${c}_$n; rocprofv3 --pmc $c --output-format csv -d gpurun_out/prof_ppo_${c}_$n -- python3 /tmp/ppo_only.py $n > gpurun_out/prof_ppo_${c}_$n.log 2>&1; echo "pmc $c $n rc=$?"
done; done
python3 - <<'PY'
import csv, glob, json, os, sys
sys.path.insert(0, os.getcwd())
from mansy_immersivevideostreaming_amd import build_ext
import bench
def tot(c, n):
    f = sorted(glob.glob(f'gpurun_out/prof_ppo_{c}_{n}/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)[-1]
    return sum(float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Counter_Name'] == c) * 1024.0
dn = 8.0 * bench.PPO_BLOCKS                                        # (10 B + 4) - (2 B + 4) cycles, B = bench.PPO_BLOCKS timed blocks per run
fe = 2.0 * (tot('FETCH_SIZE', 10) - tot('FETCH_SIZE', 2)) / dn    # gfx950: FETCH_SIZE reports half of a wide streaming read
wr = (tot('WRITE_SIZE', 10) - tot('WRITE_SIZE', 2)) / dn
out = {'fetch_bytes_per_cycle': fe, 'write_bytes_per_cycle': wr, 'traffic_bytes_per_cycle': fe + wr,
       'traffic_bytes_per_env_step': (fe + wr) / 4096.0, 'source_digest': build_ext.source_digest(),
       'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over every kernel of bench.bench_ppo at 10 and at 2 timed cycles (+ 2 warm-up + 2 roofline-leg '
               'cycles each); per cycle = the difference / (8 x the timed blocks of the bench leg) (set-up kernels cancel); 256 envs x 16 steps = 4096 env-steps per cycle; counters at the L2 <-> fabric boundary '
               '(Infinity-Cache hits included)'}
json.dump(out, open('gpurun_out/prof_ppo/ppo_pmc.json', 'w'), indent=1)
print(json.dumps(out))
PY
python3 tools/ppo_traffic_by_kernel.py > gpurun_out/prof_ppo/ppo_traffic_by_kernel.txt 2>&1; head -12 gpurun_out/prof_ppo/ppo_traffic_by_kernel.txt
