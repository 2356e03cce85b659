"""CPU, world_size 2 over gloo: the data-parallel plumbing used by bench.py / the trainers -- flat-gradient averaging,
environment sharding by (index_offset, worker_num), and the global merge of the return normaliser."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from mansy_immersivevideostreaming_amd import dist as mdist
    r, w, _ = mdist.init_process_group('gloo')
    assert (r, w) == (rank, world)
    sync = mdist.make_grad_sync(w)
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    sync(g)
    rs = np.random.RandomState(rank)
    x = rs.randn(300 + 50 * rank) * (1 + rank) + rank
    local = torch.tensor([x.mean(), x.var(), float(len(x))], dtype=torch.float64)
    glob = mdist.global_running_moments(local, w)
    off, wn = mdist.shard_envs(256, rank, w)
    cores = mdist.pin_host_cores(rank, w)
    pf = mdist.preflight(w, rank, None)
    out[rank] = (g.numpy().copy(), glob.numpy().copy(), local.numpy().copy(), off, wn, x, pf, cores)
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    g0, glob0, loc0, off0, wn0, x0, pf0, cores0 = out[0]
    g1, glob1, loc1, off1, wn1, x1, pf1, cores1 = out[1]
    # the self-diagnosis object of a multi-rank run (bench.py prints it under dist.preflight): same keys on CPU / gloo, GPU items 'n/a'
    for pf in (pf0, pf1):
        assert {'world', 'backend', 'host_cores', 'device_count', 'device', 'can_access_peer', 'library_allreduce', 'peer_ipc', 'ranks'} <= set(pf)
        assert pf['world'] == 2 and pf['backend'] == 'gloo' and pf['library_allreduce'] == {'sum_of_ones': 2.0, 'ok': True}
        assert pf['peer_ipc'] == 'n/a' and [r['rank'] for r in pf['ranks']] == [0, 1]
    if len(os.sched_getaffinity(0)) >= 2:        # each rank's threads on a core set of its own
        assert cores0 and cores1 and not set(cores0) & set(cores1)
        assert pf0['ranks'][0]['host_cores'] == cores0 and pf0['ranks'][1]['host_cores'] == cores1
    want = np.arange(1000, dtype=np.float32) * 1.5
    np.testing.assert_allclose(g0, want)
    np.testing.assert_allclose(g1, want)
    allx = np.concatenate([x0, x1])
    np.testing.assert_allclose(glob0, [allx.mean(), allx.var(), len(allx)], rtol=1e-12)
    np.testing.assert_array_equal(glob0, glob1)
    assert (off0, wn0, off1, wn1) == (0, 512, 256, 512)
    # env i of rank r starts at sample (seed + off + i) % worker_num: the two shards cover 512 distinct workers
    ids = {(5 + off0 + i) % wn0 for i in range(256)} | {(5 + off1 + i) % wn1 for i in range(256)}
    assert len(ids) == 512


def test_single_process_helpers():
    from mansy_immersivevideostreaming_amd import dist as mdist
    assert mdist.make_grad_sync(1) is None
    assert mdist.shard_envs(256, 0, 1) == (0, 256)
    a = mdist.merge_moments((0.0, 1.0, 0.0), (2.0, 3.0, 10.0))
    assert a == (2.0, 3.0, 10.0)


def test_pooled_moments_equals_chained_merges():
    """The device-side closed form used by the data-parallel return normaliser equals chaining merge_moments (the
    RunningMeanStd.update formula), including empty rows and the all-empty initial state."""
    import numpy as np
    import torch
    from mansy_immersivevideostreaming_amd import dist as mdist
    rs = np.random.RandomState(0)
    rows, acc = [(0.0, 1.0, 0.0)], (0.0, 1.0, 0.0)
    for k in range(5):
        x = rs.randn(100 + 37 * k) * (1 + k) + k
        rows.append((x.mean(), x.var(), float(len(x))))
    rows.append((0.0, 1.0, 0.0))
    for r in rows:
        acc = mdist.merge_moments(acc, r)
    got = mdist.pooled_moments(torch.tensor(rows, dtype=torch.float64)).numpy()
    np.testing.assert_allclose(got, acc, rtol=1e-12)
    np.testing.assert_array_equal(mdist.pooled_moments(torch.tensor([(0.0, 1.0, 0.0)] * 3, dtype=torch.float64)).numpy(), [0.0, 1.0, 0.0])
    rms = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64)
    xs = [torch.from_numpy(rs.randn(64)), torch.from_numpy(rs.randn(200) * 3 + 1)]
    for x in xs:
        mdist.update_running_moments(rms, x)
    allx = torch.cat(xs).numpy()
    np.testing.assert_allclose(rms.numpy(), [allx.mean(), allx.var(), len(allx)], rtol=1e-12)
