#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
MANSY_DIST_BACKEND=gloo MANSY_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --batch 256 > gpurun_out/bench_2r.log 2>gpurun_out/bench_2r.err; echo "rc=$?"
grep '^{' gpurun_out/bench_2r.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('vp spread', d['replica_param_spread'], 'loss', d['final_loss'])
print([(m['dtype'], m['final_loss']) for m in d['precision_modes']])
print('ppo spread', d['secondary']['replica_param_spread'], d['secondary']['final_loss'], d['secondary']['grad_sync'], d['secondary']['update_half'])
print(d['secondary'].get('wire_term'))
"
tail -5 gpurun_out/bench_2r.err
