#!/bin/bash
# round 6, call A: the shipped-run tests on real tables + a bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_shipped_run.py -x -q -rA -s > gpurun_out/t_shipped.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t_shipped.log
grep -E "^pref|episodes identical|PASSED|FAILED" gpurun_out/t_shipped.log | head -40
timeout 900 python bench.py > gpurun_out/bench_r06a.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/bench_r06a.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['secondary']
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print({k:s[k] for k in ('value','ms_per_cycle','data','final_loss')}, s['config']['tables'], s.get('synthetic_tables'), s['cpu_baseline'])
print({k:(v.get('host_enqueue_ms_per_cycle'), v.get('ms_per_cycle')) for k,v in s['dp_form'].items() if isinstance(v,dict)})
"
