cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_env.py tests/test_gpu_ppo.py tests/test_gpu_expert.py tests/test_gpu_ppo_cli.py -m gpu -q --tb=short -x > gpurun_out/t_env.log 2>&1
tail -5 gpurun_out/t_env.log
python tools/hbm_kernels_bench.py > gpurun_out/hbm_bench.txt 2>&1; cat gpurun_out/hbm_bench.txt
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['secondary'])[:900])" | tee gpurun_out/ppo_sec.txt
