#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 600 python tools/r05_legs.py small > gpurun_out/r05_legs_small.json 2> gpurun_out/r05_legs.err; echo "legs rc=$?"; cat gpurun_out/r05_legs_small.json; tail -3 gpurun_out/r05_legs.err
timeout 1500 python -m pytest tests/test_gpu_vp_engine.py tests/test_gpu_vp_fullsize.py tests/test_gpu_abi7_no_global_state.py tests/test_gpu_vp_cli.py tests/test_gpu_kernels.py -m gpu -q --tb=short -x > gpurun_out/t_vp.log 2>&1; echo "vp tests rc=$?"; tail -6 gpurun_out/t_vp.log
