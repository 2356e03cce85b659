cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16_modes.py -m gpu -q -x --tb=short > gpurun_out/t_bf16.log 2>&1; tail -3 gpurun_out/t_bf16.log
timeout 600 python tools/bf16_ring_ab.py 2>&1 | grep -v "bit-equal True" | tail -12 | tee gpurun_out/bf16_ring_ab.txt
