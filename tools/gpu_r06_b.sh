#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 600 python tools/ppo_real_tables_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ppo_real_probe.txt
timeout 900 python -m pytest tests/test_gpu_shipped_run.py -x -q -rA -s > gpurun_out/t_shipped.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t_shipped.log
grep -E "episodes identical|PASSED|FAILED|^E " gpurun_out/t_shipped.log | head -40
