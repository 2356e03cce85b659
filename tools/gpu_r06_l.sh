#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 600 python -m pytest tests/test_gpu_bf16_modes.py -q -x -k "reads_no_image" 2>&1 | grep -E "^E |passed|failed" | head -20
export MANSY_DIST_BACKEND=gloo MANSY_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
echo "--- overlapped"; timeout 300 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/vp_dp2_probe.py 2>&1 | grep "^rank" | sort | head -30
echo "--- plain sync"; OVERLAP=0 MODES=bf16 timeout 300 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 tools/vp_dp2_probe.py 2>&1 | grep "^rank" | sort | head -12
