#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 600 python tools/r05_legs.py dp > gpurun_out/r05_legs_dp.json 2> gpurun_out/r05_legs.err; echo "legs rc=$?"; cat gpurun_out/r05_legs_dp.json; tail -5 gpurun_out/r05_legs.err
timeout 600 bash tools/gpu_xg_timing.sh > gpurun_out/r05_xg_timing.txt 2>&1; echo "xg rc=$?"; cat gpurun_out/r05_xg_timing.txt
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_ppo.py -m gpu -q --tb=short -x > gpurun_out/t_dist.log 2>&1; echo "dist+ppo tests rc=$?"; tail -8 gpurun_out/t_dist.log
