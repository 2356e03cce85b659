#!/usr/bin/env python3
"""Per-cycle wall times of the PPO cycle on the real Jin2022 x 4G train tables against the synthetic ones (round 6: the first real-table
bench line read 13.75 ms per cycle against 1.75)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from mansy_immersivevideostreaming_amd.bitrate_selection.envs.mansy_env import MANSYVecEnv
from mansy_immersivevideostreaming_amd.bitrate_selection.models.mansy_ppo import RolloutBuffer, VecCollector

dev = torch.device('cuda', 0)
for kind, graph in (('synthetic', False), ('synthetic', 'auto'), ('real', False), ('real', 'auto'), ('synthetic', False), ('synthetic', 'auto')):
    pol = bench._ppo_policy(dev)
    pol.graph_update = graph
    tables = bench._ppo_tables(dev, kind, 256)
    venv = MANSYVecEnv(tables, 256, seed=5, index_offset=0, worker_num=256)
    col = VecCollector(pol, venv, seed=5)
    buf = RolloutBuffer(16, 256, dev)
    times = []
    for i in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        col.collect(4096, buf); torch.cuda.synchronize(); t1 = time.perf_counter()
        pol.train_identifier(buf, 2, verbose=False); torch.cuda.synchronize(); t2 = time.perf_counter()
        res = pol.update(0, buf, is_train=True, batch_size=512, repeat=2); torch.cuda.synchronize(); t3 = time.perf_counter()
        times.append((round((t1 - t0) * 1e3, 3), round((t2 - t1) * 1e3, 3), round((t3 - t2) * 1e3, 3)))
    print(kind, 'graph_update =', graph, 'collect / identifier / update ms per cycle (each synchronised):', times[4:8], 'loss', float(np.mean(res['loss'])), flush=True)
    # un-synchronised cycles: wall time per cycle and the host's enqueue share
    def cyc():
        col.collect(4096, buf); pol.train_identifier(buf, 2, verbose=False); return pol.update(0, buf, is_train=True, batch_size=512, repeat=2)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            cyc()
        th = time.perf_counter() - t0; torch.cuda.synchronize(); tt = time.perf_counter() - t0
        print('   20 cycles: %.3f ms per cycle, host enqueue %.3f ms per cycle, replays %d' % (tt / 20 * 1e3, th / 20 * 1e3, pol.graph_replays), flush=True)
    print('  done fraction', float(buf.done.float().mean()), 'rew mean', float(buf.rew.mean()), 'rew min', float(buf.rew.min()), flush=True)
