#!/bin/bash
# Full GPU suite + smoke + bench, as the driver does at round end.  Also leaves the list of passed tests (gpurun_out/gpu_tests_passed.txt).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 1800 python -m pytest tests -m gpu -x -q -rA > gpurun_out/t_gpu.log 2>&1; echo "pytest-gpu rc=$?"; tail -1 gpurun_out/t_gpu.log
grep "^PASSED" gpurun_out/t_gpu.log > gpurun_out/gpu_tests_passed.txt; wc -l gpurun_out/gpu_tests_passed.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-600
