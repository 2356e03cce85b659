#!/bin/bash
# One-shot peer-memory all-reduce (tools/xg_selftest.py) with 1 / 2 / 4 / 8 ranks SHARING the one GPU (functional: the ranks' kernels
# time-share the device, so the per-call time is an upper bound of the protocol cost, not an xGMI figure), and the RCCL world-1
# self-test for the library collective's single-rank launch cost.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MANSY_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for form in 0 1; do
export XG_SLOT=$form
for w in 1 2 4 8; do
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29500+w)) tools/xg_selftest.py > /tmp/xg_$w.log 2>&1
  grep '^{"ranks"' /tmp/xg_$w.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('xg', d['ranks'][0].get('form'), 'world', d['ranks'][0]['world'], 'mismatched', sum(r['mismatched_elements'] for r in d['ranks']), 'us_per_call (host loop)', [r['us_per_call'] for r in d['ranks']], 'device', [r['device_us_per_call'] for r in d['ranks']])" || tail -5 /tmp/xg_$w.log
done
done
unset MANSY_SHARE_GPU XG_SLOT
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 python3 tools/rccl_selftest.py 2>/dev/null | grep '^{' | tail -1
