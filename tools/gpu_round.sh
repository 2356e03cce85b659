#!/bin/bash
# One GPU-box round: kernel parity, engine parity, smoke, bench (+ optional rocprof).  Logs -> gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x > gpurun_out/t_kernels.log 2>&1; echo "kernels rc=$?"
tail -5 gpurun_out/t_kernels.log
timeout 1200 python -m pytest tests/test_gpu_vp_engine.py -m gpu -q --tb=short > gpurun_out/t_engine.log 2>&1; echo "engine rc=$?"
tail -5 gpurun_out/t_engine.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench.log
