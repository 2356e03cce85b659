#!/usr/bin/env python3
"""Round-5 measurement legs of bench.py on their own (one GPU): small-batch VP steps, the configs[4] cycle, the data-parallel form at world 1."""
import json
import os
import sys

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch  # noqa: E402
import bench  # noqa: E402
from mansy_immersivevideostreaming_amd import dist as mdist  # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
which = sys.argv[1:] or ['small', 'c5', 'dp']
out = {}
if 'small' in which:
    out['small_batch'] = bench.bench_vp_small(dev)
    print(json.dumps({'small_batch': out['small_batch']}), flush=True)
if 'c5' in which:
    out['configs4'] = bench.bench_ppo_c5(dev)
    print(json.dumps({'configs4': out['configs4']}), flush=True)
if 'dp' in which:
    out['dp_form'] = bench.bench_ppo_dp_form(dev, mdist)
    print(json.dumps({'dp_form': out['dp_form']}), flush=True)
