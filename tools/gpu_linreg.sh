cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vp_cli.py -m gpu -q --tb=short > gpurun_out/t_linreg.log 2>&1
tail -30 gpurun_out/t_linreg.log
