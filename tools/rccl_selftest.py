#!/usr/bin/env python3
"""RCCL readiness on ONE GPU: backend "nccl" (= RCCL on ROCm) initialised with WORLD_SIZE=1 in a fresh process, and the exact
collectives of the data-parallel step pushed through it -- same dtypes, sizes, reduce ops and the `device_id=` init path the
8-GPU run uses (dist.init_process_group):
  1. fp32 flat-gradient all-reduce with ReduceOp.AVG over the VP buffer (9.2 M floats, 36.8 MB) and the PPO ones (1.7 / 1.05 MB)
  2. fp64 all-reduce (SUM) of the 2 x 512 SyncBN statistics, issued from INSIDE the engine's hook in the middle of a train step
  3. the all-gather of the 3 float64 return-normaliser moments
  4. a whole data-parallel VP train step (SyncBN hook + gradient sync + AdamW after the collective) in which the one real rank
     stands for two identical ones (the hook doubles the reduced sums): it must reproduce the plain single-process step.
Prints one JSON line.  Launched by tests/test_gpu_dist.py::test_rccl_world1_collectives_of_the_dp_step."""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from mansy_immersivevideostreaming_amd import dist as mdist  # noqa: E402


def main():
    out = {}
    rank, world, local = mdist.init_process_group(backend='nccl', force=True)
    dev = torch.device('cuda', local)
    out['backend'], out['world'] = dist.get_backend(), dist.get_world_size()
    # 1. flat-gradient AVG (the function the trainers call)
    sync = mdist.make_grad_sync(world, force=True)
    for n in (9_212_000, 430_000, 262_000):
        g = torch.randn(n, device=dev)
        ref = g.clone()
        sync(g)
        torch.cuda.synchronize()
        assert torch.equal(g, ref), 'AVG over one rank must be the identity'
    t0 = time.perf_counter()
    g = torch.randn(9_212_000, device=dev)
    for _ in range(10):
        sync(g)
    torch.cuda.synchronize()
    out['allreduce_avg_36MB_us'] = round((time.perf_counter() - t0) / 10 * 1e6, 1)
    g = torch.randn(430_000, device=dev)
    sync(g); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        sync(g)
    torch.cuda.synchronize()
    out['allreduce_avg_1p7MB_us'] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
    # 2. fp64 SUM of the SyncBN statistics
    s = torch.randn(2 * 512, dtype=torch.float64, device=dev)
    ref = s.clone()
    dist.all_reduce(s)
    assert torch.equal(s, ref)
    # 3. 3-double all-gather (return normaliser)
    rms = torch.tensor([0.25, 2.0, 4096.0], dtype=torch.float64, device=dev)
    merged = mdist.global_running_moments(rms, world, force=True)
    torch.cuda.synchronize()
    assert torch.allclose(merged, rms), (merged, rms)
    # 4. a data-parallel VP step through RCCL == the plain step
    from mansy_immersivevideostreaming_amd.viewport_prediction.models import FusedAdamW, ViewportTransformerMTIO
    from oracle import vp_oracle as vo
    res = {}
    hook_calls = []
    for tag in ('plain', 'dp'):
        torch.manual_seed(0); random.seed(0); np.random.seed(0)
        m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=64, dim_feedforward=64, device=dev, seed=3)
        m.load_state_dict(vo.make_state_dict(64, 7, bias=True))
        m = m.to(dev).train()
        m.dropout_p = m.attn_dropout_p = 0.0
        m.repeat_prob = 1.0
        opt = FusedAdamW(m, lr=1e-3)
        h, c, f = (t.to(dev) for t in vo.synthetic_trajectories(32, 10, 10, seed=1))
        if tag == 'dp':
            def bn_allreduce(t):               # one real rank standing for two identical ones: RCCL sum, then x 2
                hook_calls.append(tuple(t.shape))
                assert t.dtype == torch.float64
                dist.all_reduce(t)
                t.mul_(2.0)
            m.set_data_parallel(2, allreduce=bn_allreduce)
            osync = mdist.OverlappedGradSync(world, dev, force=True)        # second communicator + side stream, tail started by the engine hook
            starts = []
            _st = osync.start_tail
            osync.start_tail = lambda t: (starts.append(t.numel()), _st(t))[1]
            loss = m.train_step(h, c, f, opt, grad_sync=osync)
            out['overlap_tail_elems'], out['flat_elems'] = (starts[0] if starts else 0), m._flat_g.numel()
        else:
            loss = m.train_step(h, c, f, opt)
        torch.cuda.synchronize()
        res[tag] = (loss.item(), m._flat_p.clone(), m.transformer.distill_layer.norm.running_mean.clone())
    out['bn_hook_calls'] = len(hook_calls)
    out['bn_hook_shape'] = list(hook_calls[0]) if hook_calls else None
    out['loss_plain'], out['loss_dp'] = res['plain'][0], res['dp'][0]
    dpar = (res['plain'][1] - res['dp'][1]).abs()
    out['param_max_diff'] = float(dpar.max().item())
    out['param_frac_gt_1e-6'] = float((dpar > 1e-6).float().mean().item())
    out['bn_mean_max_diff'] = float((res['plain'][2] - res['dp'][2]).abs().max().item())
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
