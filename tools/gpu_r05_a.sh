#!/bin/bash
# Round 5, first GPU call: seam-cost lab, xg timing on the current build, the new bench legs, the tests the ABI 8 change touched.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 120 tools/_bin/team_lab 2000 > gpurun_out/r05_team_lab.txt 2>&1; echo "team_lab rc=$?"; cat gpurun_out/r05_team_lab.txt
timeout 600 bash tools/gpu_xg_timing.sh > gpurun_out/r05_xg_timing.txt 2>&1; echo "xg rc=$?"; cat gpurun_out/r05_xg_timing.txt
timeout 600 python tools/r05_legs.py > gpurun_out/r05_legs.json 2> gpurun_out/r05_legs.err; echo "legs rc=$?"; cat gpurun_out/r05_legs.json; tail -5 gpurun_out/r05_legs.err
timeout 1200 python -m pytest tests/test_gpu_abi7_no_global_state.py tests/test_gpu_kernels.py tests/test_gpu_bf16_modes.py tests/test_gpu_ppo.py tests/test_gpu_a2c.py -m gpu -q --tb=short -x > gpurun_out/t_abi8.log 2>&1; echo "abi8 tests rc=$?"; tail -15 gpurun_out/t_abi8.log
timeout 900 python -m pytest tests/test_gpu_vp_engine.py -m gpu -q --tb=short > gpurun_out/t_engine.log 2>&1; echo "engine rc=$?"; tail -8 gpurun_out/t_engine.log
