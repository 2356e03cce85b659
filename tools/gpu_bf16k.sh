#!/bin/bash
# round 4: the role-split 256 x 128 bf16x3 loop -- parity test, then per-shape timing of variants 8 (round-3 eight-wave loop) / 1 (role-split loop, default) / 11 / 12 (its staging-only / math-only forms)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bf16_modes.py -m gpu -x -q -k "role_split or presplit" > gpurun_out/t_bf16k.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t_bf16k.log
for v in ${VARIANTS:-8 1 11 12}; do
  echo "variant $v"
  timeout 300 python tools/gemm_bench.py 256 --prec bf16x3 --planes --variant $v 2>&1 | grep -v "dW\|Traceback\|dec " 
done > gpurun_out/gemm_bench_bf16k.log 2>&1
cat gpurun_out/gemm_bench_bf16k.log
