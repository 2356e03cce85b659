#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 300 python -m pytest tests/test_gpu_dist.py -q -x -k "two_processes_sharing" 2>&1 | grep -E "^E |rank|Error" | head -30 | cut -c1-400
timeout 300 python -m pytest tests/test_gpu_bf16_modes.py -q -x -k "reads_no_image" 2>&1 | grep -E "^E " | head -12 | cut -c1-300
