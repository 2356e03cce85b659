#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
for k in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_dist.py -q -x -k "two_processes_sharing" 2>&1 | tail -2; done
timeout 900 python -m pytest tests/test_gpu_bf16a.py tests/test_gpu_bf16_modes.py tests/test_gpu_dist.py -q -x 2>&1 | tail -5
