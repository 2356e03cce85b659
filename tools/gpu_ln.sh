#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "layernorm or attn" > gpurun_out/t_ln.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t_ln.log
for rpw in 16 8 4; do for rb in 4 2 1; do echo "RPW=$rpw RB=$rb"; MANSY_LN_RPW=$rpw MANSY_LN_RB=$rb timeout 300 python tools/ln_bench.py 2>&1 | grep -A9 "rows=4096" | grep "partial" | head -2; done; done
