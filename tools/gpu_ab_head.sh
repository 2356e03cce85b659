#!/bin/bash
# A/B on ONE box (boxes differ by up to 4 %): the small-batch legs and the PPO cycle with the library of the worktree _ab_head (a checkout of the commit to
# compare against: `git worktree add _ab_head <commit>` + build) and with the working tree's, alternating.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05b
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rnd in 1 2 3; do
  for d in _ab_head .; do
    ( cd $d && timeout 300 python tools/r05_legs.py small 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())['small_batch']
print('$d', [(x['B'], x['ms_per_step'], x['sample_ms']) for x in d])" )
    ( cd $d && timeout 300 python -c "
import sys; sys.path.insert(0, '.')
import torch, bench
from mansy_immersivevideostreaming_amd import dist as mdist
r = bench.bench_ppo(0, 1, torch.device('cuda', 0), mdist, cycles=20, warmup=3, rollout_probe=True)
print('$d', 'ppo', r['ms_per_cycle'], r['rollout_step_latency_us'])" 2>/dev/null | tail -1 )
  done
done
