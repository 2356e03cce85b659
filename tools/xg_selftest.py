#!/usr/bin/env python3
"""Functional test of the one-shot peer-memory all-reduce (csrc/xgmi.hip, dist.PeerGradSync) with WORLD_SIZE processes that may share
one GPU (MANSY_SHARE_GPU=1; the IPC handles and the reference values travel over gloo): every rank maps every peer's exchange
buffer through hipIpc, then runs `iters` all-reduces of fresh gradients of the PPO actor-critic's flat-buffer size back to back --
no host synchronisation in between, so epochs, slot alternation and the flag handshake are all exercised -- and compares each
result with the average formed on the host in rank order: bit for bit, plus the partial sums of squares.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from mansy_immersivevideostreaming_amd import dist as mdist


def main():
    rank, world, local = mdist.init_process_group(backend='gloo', force=True)
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    n, iters = 427024 // 4 * 4 + 64, int(os.environ.get('XG_ITERS', '40'))
    n = (n + 63) // 64 * 64
    distinct = min(iters, 40)          # long runs (XG_ITERS=3000: a race screen) cycle through 40 distinct gradients
    sync = mdist.PeerGradSync(n, world, rank, dev, timeout_ms=5000)
    gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
    mine, want, got, parts = [], [], [], []
    for it in range(distinct):
        gs = [torch.randn(n, generator=g) * (1.0 + 0.1 * it) for g in gens]        # every rank can form every rank's gradient
        acc = torch.zeros(n)
        for r in range(world):
            acc = acc + gs[r]                                                       # rank order, float32
        want.append(acc * (1.0 / world))
        mine.append(gs[rank].to(dev))
    SLOT = os.environ.get('XG_SLOT') == '1'
    slots = [sync.slot_tensor(0), sync.slot_tensor(1)] if SLOT else None
    torch.cuda.synchronize()
    dist.barrier()
    parts = [torch.empty(64, dtype=torch.float64, device=dev) for _ in range(distinct)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    src = [m.clone() for m in mine]
    bad_rounds = 0
    for it in range(iters):
        k = it % distinct
        if it >= distinct:                      # restore the input of this slot (device copy, stream-ordered) and check the previous use
            if it % distinct == 0:
                torch.cuda.synchronize()
                bad_rounds += sum(int((mine[j].cpu() != want[j]).sum() > 0) for j in range(distinct))
            mine[k].copy_(src[k])
        if SLOT:        # round-5 form: the gradient is produced IN the exchange slot, the collective publishes / waits / sums into mine[k]
            slots[sync.next_slot()].copy_(mine[k])
            sync.reduce_into(mine[k], parts[k])
        else:
            sync(mine[k], parts[k])                                                 # in place
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dev_us = e0.elapsed_time(e1) / iters * 1e3
    sync.check()
    bad, sq_err = 0, 0.0
    for it in range(distinct):
        g = mine[it].cpu()
        bad += int((g != want[it]).sum())
        ref = float((want[it].double() ** 2).sum())
        sq_err = max(sq_err, abs(float(parts[it].sum().item()) - ref) / ref)
    # a plain grad_sync-style call (no sums of squares) works too
    x = torch.full((n,), float(rank + 1), device=dev)
    sync(x)
    torch.cuda.synchronize()
    sync.check()
    mean_ok = bool((x == sum(range(1, world + 1)) / world).all())
    out = dict(rank=rank, world=world, n=n, form='slot (+ a device copy into the slot per call)' if SLOT else 'copy', iters=iters, mismatched_elements=bad, bad_rounds=bad_rounds, sumsq_rel_err=sq_err, plain_call_ok=mean_ok,
               us_per_call=round(dt / iters * 1e6, 1), device_us_per_call=round(dev_us, 1))
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    sync.close()
    if rank == 0:
        print(json.dumps(dict(ranks=gathered)), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
