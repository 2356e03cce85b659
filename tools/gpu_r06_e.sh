#!/bin/bash
# round 6, call E: hiccup probe, bf16-storage products (tests + per-shape timing), the tests run D did not reach
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo "build rc=$?"
timeout 600 python -m pytest tests/test_gpu_bf16a.py -q -x -rA 2>&1 | tail -15
timeout 600 python tools/gemm_bf16a_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_bf16a_bench.txt
timeout 600 python tools/ppo_hiccup_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ppo_hiccup_probe.txt
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_ppo_cli.py tests/test_gpu_ppo.py -q -x -rA > gpurun_out/t_dist.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_dist.log
grep -E "FAILED|^E " gpurun_out/t_dist.log | head -30
