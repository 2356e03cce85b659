#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe | head -40
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 600 python -m pytest tests/test_gpu_bf16a.py -q -x -k weight_gradient 2>&1 | grep -E "^E |passed|failed" | head -20
