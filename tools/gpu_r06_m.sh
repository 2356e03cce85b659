#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
export MANSY_DIST_BACKEND=gloo MANSY_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
echo "--- real 2 ranks, plain sync"; OVERLAP=0 MODES=bf16 timeout 300 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 tools/vp_dp2_probe.py 2>&1 | grep "^rank" | sort | head -12 | cut -c1-1500
echo "--- two ranks, fake hooks (no collectives in the step)"; FAKE=1 OVERLAP=0 MODES=bf16 timeout 300 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 tools/vp_dp2_probe.py 2>&1 | grep "^rank" | sort | head -12 | cut -c1-600
