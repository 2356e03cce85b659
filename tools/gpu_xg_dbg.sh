cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MANSY_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 XG_SLOT=1
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29502 tools/xg_selftest.py > gpurun_out/xg_dbg2.log 2>&1
grep -v "^\s*$" gpurun_out/xg_dbg2.log | grep -v Warning | tail -40
