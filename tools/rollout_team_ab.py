#!/usr/bin/env python3
"""Collect-only timing: the persistent XCD-team rollout (mansy_policy_rollout) against the hipGraph-replayed per-step launches, same policy and tables."""
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
if os.environ.get('ROLLOUT_PROF') == '1':
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import lab_knobs as KN
    KN.enter()
from mansy_immersivevideostreaming_amd.bitrate_selection.envs.mansy_env import EnvTables, MANSYVecEnv
from mansy_immersivevideostreaming_amd.bitrate_selection.models.mansy_ppo import RolloutBuffer, VecCollector
dev = torch.device('cuda', 0)
N, T = 256, 16
for rnd in range(3):
    for form in ('team', 'graph'):
        pol = bench._ppo_policy(dev)
        tables = EnvTables.synthetic(dev, seed=5, train_identifier_reward=True, n_sample=max(240, N))
        venv = MANSYVecEnv(tables, N, seed=5)
        col = VecCollector(pol, venv, seed=5)
        col.use_team = form == 'team'
        buf = RolloutBuffer(T, N, dev)
        for _ in range(3):
            col.collect(T * N, buf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            col.collect(T * N, buf)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f'{form}: {dt * 1e6:.1f} us per collect, {dt * 1e6 / T:.2f} us per vector step (use_team={col.use_team})', flush=True)
        if form == 'team' and os.environ.get('ROLLOUT_PROF') == '1':
            ctl = pol.engine._rollout_ctl.cpu().numpy().view('uint64')
            prof = ctl[(64 + 512) // 8:(64 + 512) // 8 + 8]
            names = ['A featnet', 'barrier 1', 'B fc slabs', 'barrier 2 (+ C prefetch)', 'C row logic + env', 'barrier 3']
            print('   phase clocks of team 0 / member 0, us per step (last collect):', {n: round(float(p) / 100.0 / T, 2) for n, p in zip(names, prof)}, flush=True)
