cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "tilemap or empty or find_tiles" > gpurun_out/t_tm.log 2>&1
tail -5 gpurun_out/t_tm.log
python tools/hbm_kernels_bench.py > gpurun_out/hbm_bench.txt 2>&1; cat gpurun_out/hbm_bench.txt
