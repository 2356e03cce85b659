#!/usr/bin/env python3
"""Round 6: the PPO cycle on the real Jin2022 x 4G tables showed a rare ~70 ms cycle (one in a few dozen; the first real-table bench line averaged
13.75 ms over 6 cycles because of one).  400 cycles per configuration, every phase synchronised and timed; outliers (> 4 ms) are printed with their
phase, the garbage collector's counters and whether the collector was enabled."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from mansy_immersivevideostreaming_amd.bitrate_selection.envs.mansy_env import MANSYVecEnv
from mansy_immersivevideostreaming_amd.bitrate_selection.models.mansy_ppo import RolloutBuffer, VecCollector

dev = torch.device('cuda', 0)
for kind, graph, gc_on in (('real', False, True), ('real', 'auto', True), ('real', 'auto', False), ('synthetic', 'auto', True)):
    gc.enable() if gc_on else gc.disable()
    pol = bench._ppo_policy(dev)
    pol.graph_update = graph
    tables = bench._ppo_tables(dev, kind, 256)
    venv = MANSYVecEnv(tables, 256, seed=5, index_offset=0, worker_num=256)
    col = VecCollector(pol, venv, seed=5)
    buf = RolloutBuffer(16, 256, dev)
    tot, outl = [], []
    for i in range(400):
        g0 = gc.get_count()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        col.collect(4096, buf); th1 = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
        pol.train_identifier(buf, 2, verbose=False); th2 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        pol.update(0, buf, is_train=True, batch_size=512, repeat=2); th3 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
        ph = ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3)
        host = ((th1 - t0) * 1e3, (th2 - t1) * 1e3, (th3 - t2) * 1e3)
        tot.append(sum(ph))
        if i >= 3 and max(ph) > 4.0:
            outl.append((i, [round(x, 2) for x in ph], 'host-enqueue part', [round(x, 2) for x in host], 'gc counts before', g0, 'after', gc.get_count()))
    tot = np.array(tot[3:])
    print(f'{kind:9s} graph_update={graph!s:5s} gc={"on " if gc_on else "off"}: median {np.median(tot):.3f} ms, mean {tot.mean():.3f}, max {tot.max():.2f}, cycles > 4 ms: {len(outl)}', flush=True)
    for o in outl[:8]:
        print('    ', *o, flush=True)
gc.enable()
