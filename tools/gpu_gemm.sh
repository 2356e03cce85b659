#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k gemm > gpurun_out/t_gemm.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t_gemm.log
timeout 600 python tools/gemm_bench.py 0 128 96 64 > gpurun_out/gemm_bench.log 2>&1; cat gpurun_out/gemm_bench.log
