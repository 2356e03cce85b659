#!/bin/bash
# bench.py three times: the dp_form legs of the PPO cycle (host enqueue per cycle under graph replay) run to run
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
for i in 1 2 3; do
  timeout 900 python bench.py > gpurun_out/bench_p$i.log 2>&1
  grep '^{' gpurun_out/bench_p$i.log | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); s=j['secondary']
print('vp', j['ms_per_step'], [(m['dtype'], m['ms_per_step']) for m in j['precision_modes']], 'ppo', s['ms_per_cycle'], s['host_enqueue_ms_per_cycle'])
print({k:(v.get('ms_per_cycle'), v.get('host_enqueue_ms_per_cycle'), v.get('update_graph_replays')) for k,v in s['dp_form'].items() if isinstance(v,dict) and 'ms_per_cycle' in v})"
done
