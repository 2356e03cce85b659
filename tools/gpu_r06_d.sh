#!/bin/bash
# round 6, call D: graph-replayed update half (staging ring), hygiene changes -- PPO / dist / cli / shipped tests, A/B probe, bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 600 python tools/ppo_real_tables_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ppo_graph_ab.txt
timeout 1200 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_shipped_run.py tests/test_gpu_dist.py tests/test_gpu_ppo_cli.py tests/test_gpu_bf16_modes.py tests/test_gpu_kernels.py -q -rA > gpurun_out/t_ppo.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_ppo.log
grep -E "FAILED|^E " gpurun_out/t_ppo.log | head -40
timeout 900 python bench.py > gpurun_out/bench_r06d.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/bench_r06d.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['secondary']
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print({k:s.get(k) for k in ('value','ms_per_cycle','host_enqueue_ms_per_cycle','update_graph_replays','data','final_loss')}, s.get('synthetic_tables'))
print({k:(v.get('host_enqueue_ms_per_cycle'), v.get('ms_per_cycle'), v.get('library_launches_per_cycle')) for k,v in s['dp_form'].items() if isinstance(v,dict)})
"
