#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "wave_split or wide" > gpurun_out/t_wsk.log 2>&1; echo "wsk tests rc=$?"; tail -8 gpurun_out/t_wsk.log
timeout 600 python tools/r05_legs.py small > gpurun_out/r05_legs_small.json 2> gpurun_out/r05_legs.err; echo "legs rc=$?"; cat gpurun_out/r05_legs_small.json; tail -3 gpurun_out/r05_legs.err
timeout 600 python tools/vp_two_stream_sweep.py 32 128 256 > gpurun_out/r05_small_sweep_wide.txt 2>&1; cat gpurun_out/r05_small_sweep_wide.txt | grep -v amdgpu
timeout 1200 python -m pytest tests/test_gpu_vp_engine.py tests/test_gpu_vp_cli.py tests/test_gpu_dataset.py -m gpu -q --tb=short -x > gpurun_out/t_engine.log 2>&1; echo "engine rc=$?"; tail -8 gpurun_out/t_engine.log
