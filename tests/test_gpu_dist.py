"""Functional test of the multi-rank code path on ONE GPU: two ranks share the device and talk over gloo
(MANSY_DIST_BACKEND=gloo, MANSY_SHARE_GPU=1 -- RCCL itself refuses two ranks on one GPU).  Everything the 8-GPU bench does
runs for real -- SyncBN statistics hook, flat-gradient all-reduce, sharded environments, return-normaliser merge, the
barrier / max-over-ranks timing protocol and the rank-0 JSON line -- only the transport differs."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu_over_gloo():
    env = dict(os.environ, MANSY_DIST_BACKEND='gloo', MANSY_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    with socket.socket() as s:          # a free rendezvous port on this box
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
           '--batch', '256']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 only
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['parallelism'] == 'dp2' and out['config']['global_batch'] == 512
    assert out['value'] > 0 and out['secondary']['value'] > 0 and out['secondary']['n_gpus'] == 2
    assert out['roofline']['launches_per_step'] > 0
    assert 0 < out['final_loss'] < 10
    # replicas stay bit-identical: averaged gradients, synchronised BatchNorm statistics, merged return normaliser
    assert out['replica_param_spread'] == 0.0 and out['secondary']['replica_param_spread'] == 0.0


@pytest.mark.gpu
def test_two_rank_train_step_equals_whole_batch(tmp_path):
    """a data-parallel VP step over 2 ranks x 32 rows (SyncBN statistics over the global mini-batch + averaged
    flat gradient) computes the gradient, loss and BatchNorm running statistics of ONE process on all 64 rows (dropout off,
    MTIO repeat branch: the same function either way).  fp32 tolerance: summation order over the batch differs."""
    import numpy as np
    env = dict(os.environ, MANSY_DIST_BACKEND='gloo', MANSY_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    probe = os.path.join(ROOT, 'tools', 'dp_equiv.py')
    one, two = str(tmp_path / 'one.npz'), str(tmp_path / 'two.npz')
    single_env = {k: v for k, v in env.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, probe, one], cwd=ROOT, env=single_env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), probe, two]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    a, b = np.load(one), np.load(two)
    assert int(a['world']) == 1 and int(b['world']) == 2
    np.testing.assert_allclose(b['loss'], a['loss'], rtol=2e-6)
    ga, gb = a['grad'].astype(np.float64), b['grad'].astype(np.float64)
    assert np.linalg.norm(ga) > 0
    assert np.linalg.norm(ga - gb) / np.linalg.norm(ga) < 2e-5, np.linalg.norm(ga - gb) / np.linalg.norm(ga)
    assert np.abs(ga - gb).max() < 1e-4 * np.abs(ga).max()
    np.testing.assert_allclose(b['rm'], a['rm'], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(b['rv'], a['rv'], atol=1e-6, rtol=1e-5)
