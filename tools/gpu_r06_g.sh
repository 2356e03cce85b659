#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 300 python tools/vp_modes_time.py bf16 f32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/vp_modes_time.txt
timeout 900 python -m pytest tests/test_gpu_bf16a.py tests/test_gpu_bf16_modes.py -q -x 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_vp_engine.py tests/test_gpu_vp_fullsize.py tests/test_gpu_abi7_no_global_state.py -q -x 2>&1 | tail -5
