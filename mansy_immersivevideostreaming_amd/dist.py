"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in
CPU tests).  The hot path shards by construction -- VP mini-batches over trajectories, PPO over environments (the
reference's worker_id / worker_num scheme, mansy_env.py:55-56,100-101) -- so the only data-path collective is ONE
all-reduce of a flat gradient buffer per optimiser step (VP 36.8 MB, actor-critic 1.7 MB, identifier 1.05 MB), plus a
3-double all-gather to merge the return normaliser."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def local_device_index(local):
    """GPU of this rank: LOCAL_RANK, except in the functional test of the multi-rank path on a box with fewer GPUs than ranks
    (MANSY_SHARE_GPU=1 with MANSY_DIST_BACKEND=gloo: ranks share devices; RCCL itself refuses two ranks on one GPU)."""
    if os.environ.get('MANSY_SHARE_GPU') == '1' and torch.cuda.is_available():
        return local % torch.cuda.device_count()
    return local


def pin_host_cores(local=None, local_world=None):
    """Give this rank's threads a core set of their own: the cores this process may run on, split evenly by LOCAL_RANK (ranks of one node
    then never contend for a core while enqueueing; the HIP / RCCL helper threads created later inherit the set, so it is a SET, not one
    core).  Must run before anything touches the GPU (threads started earlier keep the old mask).  Returns the cores taken (the preflight
    reports them); no-op -- the full mask -- with one rank per node or where sched_setaffinity does not exist."""
    if local is None:
        local = int(os.environ.get('LOCAL_RANK', '0'))
    if local_world is None:
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1')))
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return []
    if local_world <= 1 or len(cores) < local_world:
        return cores
    k = len(cores) // local_world
    mine = cores[local * k:(local + 1) * k]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return cores
    return mine


def preflight(world, rank, device=None, sizes=(1024,)):
    """What the first real multi-GPU run needs to explain itself (one shot, on a box nobody watches): before any timed leg every rank
    gathers -- and rank 0 returns for the bench line --
      device_count / devices    what this process sees (name, index);
      can_access_peer           the hipDeviceCanAccessPeer matrix over the visible devices;
      library_allreduce         an all-reduce of ones over the process group (torch.distributed: RCCL on GPUs): sum == world on every rank;
      peer_ipc                  per rank: a fine-grained allocation exported through hipIpc and opened by EVERY peer, one average of
                                rank-dependent data through it checked against the closed form (csrc/xgmi.hip, both forms);
      host_cores                the core set each rank's threads run on (pin_host_cores).
    Every rank takes part in every exchange whatever failed locally; nothing here raises.  CPU / gloo: the GPU items read 'n/a'."""
    rep = {'world': int(world), 'backend': dist.get_backend() if dist.is_initialized() else None}
    cuda = torch.cuda.is_available() and device is not None and torch.device(device).type == 'cuda'
    try:
        rep['host_cores'] = sorted(os.sched_getaffinity(0))
    except AttributeError:
        rep['host_cores'] = []
    if cuda:
        n_dev = torch.cuda.device_count()
        rep['device_count'] = n_dev
        rep['device'] = [torch.device(device).index, torch.cuda.get_device_name(device)]
        mat = []
        for i in range(n_dev):
            row = []
            for j in range(n_dev):
                try:
                    row.append(1 if i == j else int(torch.cuda.can_device_access_peer(i, j)))
                except Exception:          # noqa: BLE001
                    row.append(-1)
            mat.append(row)
        rep['can_access_peer'] = mat
    else:
        rep.update(device_count=0, device='n/a', can_access_peer='n/a')
    # library collective
    try:
        if world > 1 and dist.is_initialized():
            t = torch.ones(1, dtype=torch.float32, device=device if (cuda and dist.get_backend() == 'nccl') else 'cpu')
            dist.all_reduce(t)
            rep['library_allreduce'] = {'sum_of_ones': float(t.item()), 'ok': float(t.item()) == float(world)}
        else:
            rep['library_allreduce'] = 'n/a (one rank)'
    except Exception as e:          # noqa: BLE001
        rep['library_allreduce'] = {'ok': False, 'error': str(e)[:200]}
    # hipIpc + peer-memory average on small contexts (both forms), rank-dependent data with a closed-form answer
    if cuda and world > 1 and dist.is_initialized():
        ipc = {}
        for n in sizes:
            peer, err = PeerGradSync.try_create(int(n), world, rank, device, timeout_ms=5000)
            ok, why = (1.0, '') if peer is not None else (0.0, str(err)[:200])
            if peer is not None:
                try:
                    want = (world + 1) / 2.0
                    x = torch.full((int(n),), float(rank + 1), device=device)
                    peer(x)                                                  # copy form
                    peer.slot_tensor(peer.next_slot()).fill_(float(rank + 1))
                    y = torch.empty_like(x)
                    peer.reduce_into(y)                                      # exchange-slot form
                    torch.cuda.synchronize(device)
                    peer.check()
                    if abs(float(x[0]) - want) > 1e-6 or abs(float(y[-1]) - want) > 1e-6:
                        ok, why = 0.0, f'rank {rank}: average {float(x[0])} / {float(y[-1])}, expected {want}'
                except Exception as e:          # noqa: BLE001
                    ok, why = 0.0, f'rank {rank}: {e}'[:200]
            gathered = [None] * world
            dist.all_gather_object(gathered, (ok, why))
            if peer is not None:
                peer.close()
            ipc[str(n)] = {'ok': all(g[0] >= 1.0 for g in gathered), 'per_rank': [('ok' if g[0] >= 1.0 else g[1]) for g in gathered]}
        rep['peer_ipc'] = ipc
    else:
        rep['peer_ipc'] = 'n/a'
    if world > 1 and dist.is_initialized():
        allrep = [None] * world
        dist.all_gather_object(allrep, {'rank': rank, 'device': rep['device'], 'host_cores': rep['host_cores']})
        rep['ranks'] = allrep
    return rep


def init_process_group(backend=None, force=False):
    """force=True initialises the process group even with WORLD_SIZE=1 (the RCCL self-test on a one-GPU box:
    tools/rccl_selftest.py)."""
    rank, world, local = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = os.environ.get('MANSY_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    return rank, world, local_device_index(local)


def make_grad_sync(world, force=False):
    """flat gradient buffer -> average over ranks, in place (None when single process, unless force=True: the collective is
    then issued over the one-rank group -- tools/rccl_selftest.py)."""
    if world <= 1 and not force:
        return None
    world = max(world, 1)
    inv = 1.0 / world
    # RCCL averages inside the collective (one launch instead of all-reduce + scale); gloo (CPU tests) has no AVG
    use_avg = dist.is_initialized() and dist.get_backend() == 'nccl'

    state = {'avg': use_avg}

    def grad_sync(flat_g):
        if state['avg']:
            try:
                dist.all_reduce(flat_g, op=dist.ReduceOp.AVG)
                return
            except (RuntimeError, ValueError):      # a backend build without AVG: sum + scale from here on
                state['avg'] = False
        dist.all_reduce(flat_g)
        flat_g.mul_(inv)
    return grad_sync


class OverlappedGradSync:
    """Gradient averaging for the VP step with the big half of the all-reduce hidden under the encoder backward: the engine's
    hook (which = 2) hands over the tail of the flat gradient buffer (decoder, DistillLayer, predictor: two thirds of the 36.8
    MB) as soon as it is final; its all-reduce runs on a SIDE stream through a SECOND communicator (so it can never interleave
    with the SyncBN statistics collectives of the default group), while the compute stream goes on with the encoder backward.
    `finish(head)` reduces the rest on the compute stream and makes it wait for the tail.  Also callable like the plain
    grad_sync (whole buffer, no overlap) -- what the PPO policy and non-engine callers use."""

    def __init__(self, world, device=None, force=False):
        self.world = max(int(world), 1)
        self.inv = 1.0 / self.world
        self.avg = dist.is_initialized() and dist.get_backend() == 'nccl'
        self.group = dist.new_group() if (dist.is_initialized() and (self.world > 1 or force)) else None
        self.stream = torch.cuda.Stream(device=device) if torch.cuda.is_available() else None
        self.done = None

    def _reduce(self, t, group):
        if self.avg:
            try:
                dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
                return
            except (RuntimeError, ValueError):
                self.avg = False
        dist.all_reduce(t, group=group)
        t.mul_(self.inv)

    def __call__(self, flat_g):
        self._reduce(flat_g, None)

    def start_tail(self, tail):
        """Called from the engine hook while the step is being enqueued: everything enqueued so far on the current stream
        produced `tail`; the side stream waits for exactly that point."""
        cur = torch.cuda.current_stream(tail.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        self.stream.wait_event(ready)
        with torch.cuda.stream(self.stream):
            self._reduce(tail, self.group)
            self.done = torch.cuda.Event()
            self.done.record(self.stream)

    def finish(self, head):
        if head.numel():
            self._reduce(head, None)
        if self.done is not None:
            torch.cuda.current_stream(head.device).wait_event(self.done)
            self.done = None


class PeerGradSync:
    """Gradient averaging for the latency-bound collectives of the data-parallel PPO update (16 + 2 per cycle, 1.7 MB / 1.05 MB):
    one hand-written launch per rank (csrc/xgmi.hip, mansy_xg_allreduce_avg) instead of a library all-reduce + a separate
    gradient-norm pass.  Each rank publishes its flat gradient in fine-grained device memory that every peer has mapped through
    hipIpc, waits for the peers' epoch flags and sums all copies in rank order straight over the point-to-point xGMI links; the
    kernel also leaves the partial sums of squares, so the clip + Adam launch that follows needs no norm launch
    (mansy_clip_grad_adam(have_sumsq = 1)).  Every rank gets bit-identical averages.

    One instance per flat buffer size.  The 64-byte IPC handles travel through torch.distributed (any backend: it is host
    data).  Call like the other grad_sync forms -- `sync(flat_g)` -- or `sync(flat_g, scratch)` to collect the sums of squares
    (`fused_sumsq` tells the caller it may).  `check()` surfaces a timed-out wait (the launch itself never hangs)."""
    fused_sumsq = True

    _fail_setup_on_rank = None      # test seam (tests/test_gpu_dist.py sets it on the class): that rank's set-up raises -- the agreed-failure path

    def __init__(self, n_floats, world, rank, device=None, timeout_ms=None):
        # tolerant set-up here too: a rank whose local set-up fails still takes part in both host exchanges, so that EVERY rank gets
        # the error and raises together (a rank raising before all_gather_object would leave the others inside the collective)
        err = self._setup(n_floats, world, rank, device, timeout_ms, tolerant=True)
        if err is not None:                      # this or another rank failed its set-up
            from ._lib import MansyError
            self.close()
            raise MansyError(err)

    @classmethod
    def try_create(cls, n_floats, world, rank, device=None, timeout_ms=None):
        """-> (PeerGradSync, None) or (None, reason): the same set-up, but a failure on ANY rank (fine-grained allocation, hipIpc export /
        import, peer access) is agreed on by all ranks -- every rank takes part in every exchange whatever happened locally -- so that
        the caller can fall back to the library collective everywhere instead of leaving some ranks inside a collective."""
        self = cls.__new__(cls)
        err = self._setup(n_floats, world, rank, device, timeout_ms, tolerant=True)
        if err is not None:
            self.close()
            return None, err
        return self, None

    def _setup(self, n_floats, world, rank, device, timeout_ms, tolerant):
        import ctypes
        from ._lib import XgHandle, check, lib
        self.world, self.rank, self.n = int(world), int(rank), int(n_floats)
        self._lib, self._check = lib(), check
        self.ctx = None
        if device is not None:
            torch.cuda.set_device(device)

        def guarded(fn):
            try:
                fn()
                return None
            except Exception as e:          # noqa: BLE001 -- reported to every rank below
                if not tolerant:
                    raise
                return f'rank {self.rank}: {e}'

        own = XgHandle()

        def create_export():
            if PeerGradSync._fail_setup_on_rank == self.rank:
                raise RuntimeError('injected set-up failure')
            ctx = ctypes.c_void_p()
            check(self._lib.mansy_xg_create(self.n, self.world, self.rank, ctypes.byref(ctx)), 'mansy_xg_create')
            self.ctx = ctx
            if timeout_ms is not None:
                check(self._lib.mansy_xg_set_timeout_ms(self.ctx, float(timeout_ms)), 'mansy_xg_set_timeout_ms')
            if self.world > 1:
                check(self._lib.mansy_xg_export(self.ctx, ctypes.byref(own)), 'mansy_xg_export')
        err = guarded(create_export)
        if self.world <= 1:
            return err
        gathered = [None] * self.world
        dist.all_gather_object(gathered, (err, bytes(own.bytes)))
        errs = [e for e, _ in gathered if e]
        if errs:
            return errs[0]

        def do_import():
            allh = (XgHandle * self.world)()
            for r, (_, h) in enumerate(gathered):
                assert len(h) == 64
                allh[r].bytes[:] = list(h)
            check(self._lib.mansy_xg_import(self.ctx, allh), 'mansy_xg_import')
        err = guarded(do_import)
        flags = [None] * self.world
        dist.all_gather_object(flags, err)       # doubles as the barrier: every rank has mapped every peer before the first flag is read
        errs = [e for e in flags if e]
        return errs[0] if errs else None

    def __call__(self, flat_g, scratch=None):
        from ._lib import ptr, stream_ptr
        assert flat_g.is_cuda and flat_g.dtype == torch.float32 and flat_g.numel() == self.n
        self._check(self._lib.mansy_xg_allreduce_avg(self.ctx, ptr(flat_g), self.n, ptr(scratch), stream_ptr(flat_g.device)), 'mansy_xg_allreduce_avg')

    # ---- round 5: no copy in front of the flag.  The two exchange slots ARE the flat gradient buffers of alternate steps: the step's gradient kernels
    # write into slot `next_slot()`, `reduce_into(out)` publishes it, waits for the peers and leaves the average in `out` (ordinary device memory).
    def slot_ptrs(self):
        """(address of slot 0, address of slot 1): n floats each, fine-grained device memory."""
        import ctypes
        if getattr(self, '_slots', None) is None:
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            self._check(self._lib.mansy_xg_slot_ptrs(self.ctx, ctypes.byref(a), ctypes.byref(b)), 'mansy_xg_slot_ptrs')
            self._slots = (int(a.value), int(b.value))
        return self._slots

    def slot_tensor(self, i):
        """Slot i as a float32 tensor ALIASING the exchange memory (tests / tools: fill a slot, inspect a raw gradient)."""
        class _Mem:
            pass
        m = _Mem()
        m.__cuda_array_interface__ = {'shape': (self.n,), 'typestr': '<f4', 'data': (self.slot_ptrs()[i], False), 'version': 2}
        return torch.as_tensor(m, device=torch.device('cuda', torch.cuda.current_device()))

    def next_slot(self):
        """Index (0 / 1) of the slot the NEXT reduce_into() publishes: where this step's gradients must be produced."""
        return int(self._lib.mansy_xg_next_slot(self.ctx))

    def reduce_into(self, out, scratch=None):
        from ._lib import ptr, stream_ptr
        assert out.is_cuda and out.dtype == torch.float32 and out.numel() == self.n
        self._check(self._lib.mansy_xg_reduce_avg(self.ctx, ptr(out), self.n, ptr(scratch), stream_ptr(out.device)), 'mansy_xg_reduce_avg')

    def check(self):
        self._check(self._lib.mansy_xg_status(self.ctx), 'mansy_xg_status')

    def close(self):
        if getattr(self, 'ctx', None):
            self._lib.mansy_xg_destroy(self.ctx)
            self.ctx = None


class RcclComm:
    """A communicator of the library's own (include/mansy_hip.h: mansy_comm_* / mansy_allreduce_*, thin wrappers over RCCL bound at run time) --
    SURVEY 8b's `mansy_allreduce_*`.  Used as the `sync` context of the one-call data-parallel PPO step in its library-collective form
    (PPOPolicy.set_data_parallel(..., comm=RcclComm(...))): gradients, ncclAllReduce(avg), norm, clip + Adam are then ONE library call per step,
    like the peer-memory form; also callable like a grad_sync (`comm(flat_g)`).  Rank 0 draws the 128-byte id, torch.distributed (any backend:
    it is host data) carries it to the others."""

    def __init__(self, world, rank, device=None):
        import ctypes
        from ._lib import CommId, check, lib
        self.world, self.rank = int(world), int(rank)
        self._lib, self._check = lib(), check
        if device is not None:
            torch.cuda.set_device(device)
        cid = CommId()
        if self.rank == 0:
            check(self._lib.mansy_comm_unique_id(ctypes.byref(cid)), 'mansy_comm_unique_id')
        if self.world > 1:
            box = [bytes(cid.bytes)]
            dist.broadcast_object_list(box, src=0)
            cid.bytes[:] = list(box[0])
        ctx = ctypes.c_void_p()
        check(self._lib.mansy_comm_create(ctypes.byref(cid), self.world, self.rank, ctypes.byref(ctx)), 'mansy_comm_create')
        self.ctx = ctx

    @classmethod
    def try_create(cls, world, rank, device=None):
        """-> (RcclComm, None) or (None, reason), the SAME answer on every rank: before anyone enters the collective communicator set-up, the ranks
        agree (through torch.distributed) that EVERY rank can bind RCCL -- a rank that cannot must not leave the others inside ncclCommInitRank --
        and afterwards that every rank joined and that an average over the new communicator is the average."""
        import ctypes
        from ._lib import CommId, lib
        agree_dev = device if (dist.is_initialized() and dist.get_backend() == 'nccl') else 'cpu'

        def all_min(x):
            if world <= 1 or not dist.is_initialized():
                return float(x)
            t = torch.tensor([float(x)], dtype=torch.float64, device=agree_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return float(t.item())
        probe = CommId()
        can = 1.0 if lib().mansy_comm_unique_id(ctypes.byref(probe)) == 0 else 0.0      # binds librccl.so on this rank (dlopen)
        if all_min(can) < 1.0:
            return None, 'RCCL cannot be bound on every rank (' + (lib().mansy_last_error() or b'').decode() + ')'
        # step by step, agreeing after each: (1) rank 0 draws the id; EVERY rank takes part in the broadcast whatever happened on rank 0;
        # (2) every rank confirms it can select its device; only then (3) the collective ncclCommInitRank, which has no time-out of its own
        self, ok, why = None, 1.0, ''
        cid = CommId()
        id_ok = 1.0
        if rank == 0 and lib().mansy_comm_unique_id(ctypes.byref(cid)) != 0:
            id_ok = 0.0
        box = [(id_ok, bytes(cid.bytes))]
        if world > 1 and dist.is_initialized():
            dist.broadcast_object_list(box, src=0)
        if box[0][0] < 1.0:
            return None, 'rank 0 could not draw a communicator id'
        dev_ok = 1.0
        try:
            if device is not None:
                torch.cuda.set_device(device)
        except Exception as e:          # noqa: BLE001
            dev_ok, why = 0.0, f'rank {rank}: {e}'
        if all_min(dev_ok) < 1.0:
            return None, why or 'another rank could not select its device'
        try:
            self = cls.__new__(cls)
            self.world, self.rank = int(world), int(rank)
            from ._lib import check
            self._lib, self._check = lib(), check
            cid.bytes[:] = list(box[0][1])
            ctx = ctypes.c_void_p()
            check(self._lib.mansy_comm_create(ctypes.byref(cid), self.world, self.rank, ctypes.byref(ctx)), 'mansy_comm_create')
            self.ctx = ctx
            x = torch.full((1024,), float(rank + 1), device=device)
            self(x)
            torch.cuda.synchronize(device)
            if abs(float(x[0]) - (world + 1) / 2.0) > 1e-6:
                ok, why = 0.0, f'rank {rank}: average over the new communicator is {float(x[0])}, expected {(world + 1) / 2.0}'
        except Exception as e:          # noqa: BLE001
            ok, why = 0.0, f'rank {rank}: {e}'
        if all_min(ok) < 1.0:
            if self is not None:
                self.close()
            return None, why or 'another rank failed to join the communicator'
        return self, None

    def __call__(self, flat_g):
        from ._lib import ptr, stream_ptr
        self._check(self._lib.mansy_allreduce_avg_f32(self.ctx, ptr(flat_g), flat_g.numel(), stream_ptr(flat_g.device)), 'mansy_allreduce_avg_f32')

    def sum_f64(self, t):
        from ._lib import ptr, stream_ptr
        assert t.dtype == torch.float64
        self._check(self._lib.mansy_allreduce_sum_f64(self.ctx, ptr(t), t.numel(), stream_ptr(t.device)), 'mansy_allreduce_sum_f64')

    def allgather_f64(self, t):
        from ._lib import ptr, stream_ptr
        out = torch.empty(self.world, t.numel(), dtype=torch.float64, device=t.device)
        self._check(self._lib.mansy_allgather_f64(self.ctx, ptr(t), ptr(out), t.numel(), stream_ptr(t.device)), 'mansy_allgather_f64')
        return out

    def close(self):
        if getattr(self, 'ctx', None):
            self._lib.mansy_comm_destroy(self.ctx)
            self.ctx = None


def probe_peer_grad_sync(sizes, world, rank, device, library_sync, iters=10, probe_timeout_ms=5000, timeout_ms=60000):
    """Measure, do not guess: build the peer-memory all-reduce for flat buffers of `sizes` floats, check it against the library collective
    on the same data, time both (max over ranks), and keep it only if every rank agrees it is correct AND faster.  The wait of the
    peer kernel stays bounded: probe_timeout_ms while it is on trial (a dead link costs the start-up 5 s, then everyone uses the library),
    timeout_ms once it is chosen (generous, so that a rank delayed by a first-call set-up does not poison a step; a library collective
    would simply wait).
    -> ({size: PeerGradSync} or {}, report dict for the bench line).  Every rank calls this; every rank returns the same decision."""
    import time
    report = {'probe': 'peer-memory one-shot (exchange-slot form) vs library all_reduce', 'iters': iters}
    agree_dev = device if dist.get_backend() == 'nccl' else 'cpu'

    def all_min(x):
        t = torch.tensor([float(x)], dtype=torch.float64, device=agree_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item())

    def all_max(x):
        t = torch.tensor([float(x)], dtype=torch.float64, device=agree_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    peers = {}
    for n in sizes:
        p, err = PeerGradSync.try_create(n, world, rank, device, timeout_ms=probe_timeout_ms)
        if p is None:
            for q in peers.values():
                q.close()
            report.update(chosen='library', reason=f'peer set-up failed: {err}')
            return {}, report
        peers[n] = p
    n0 = max(sizes)
    g = torch.Generator(device='cpu').manual_seed(4321 + rank)
    src = torch.randn(n0, generator=g).to(device)
    scratch = torch.zeros(64, dtype=torch.float64, device=device)
    ok, why, agreed = 1.0, '', 1.0
    # correctness on EVERY context (each has its own IPC mappings, flags and epochs: the identifier's 1.05 MB one used to go into
    # production without ever having been launched), speed on the largest
    for n in sorted(sizes):
        ref = src[:n].clone()
        library_sync(ref)
        out = src[:n].clone()
        try:
            peers[n](out, scratch)
            torch.cuda.synchronize(device)
            peers[n].check()
            if not torch.allclose(out, ref, rtol=1e-5, atol=1e-6):
                ok, why = 0.0, f'rank {rank}: peer average ({n} floats) differs from the library average by {float((out - ref).abs().max()):.3e}'
            elif abs(float(scratch.sum()) - float((out.double() ** 2).sum())) > 1e-6 * max(float((out.double() ** 2).sum()), 1e-30):
                ok, why = 0.0, f'rank {rank}: sums of squares differ ({n} floats)'
            else:      # the exchange-slot form (what the engine steps use): the same gradient produced IN the slot, then publish / wait / sum
                peers[n].slot_tensor(peers[n].next_slot()).copy_(src[:n])
                out2 = torch.empty_like(out)
                peers[n].reduce_into(out2, scratch)
                torch.cuda.synchronize(device)
                peers[n].check()
                if not torch.equal(out2, out):
                    ok, why = 0.0, f'rank {rank}: the exchange-slot form differs from the copy form ({n} floats) by {float((out2 - out).abs().max()):.3e}'
        except Exception as e:          # noqa: BLE001
            ok, why = 0.0, f'rank {rank}: {e}'
        # agree AFTER EVERY SIZE: every iteration starts with a collective (library_sync), so a rank-local `break`
        # would leave the failed rank in all_min() while the healthy ones enter the next size's all_reduce + a peer launch that waits
        # for the missing rank -- mismatched collectives.  The agreed value is what every rank leaves the loop on.
        agreed = all_min(ok)
        if agreed < 1.0:
            break
    if agreed < 1.0:
        for q in peers.values():
            q.close()
        report.update(chosen='library', reason='peer self-test failed' + (': ' + why if why else ' on another rank'))
        return {}, report

    def timed(fn):
        buf = src.clone()
        for _ in range(3):
            fn(buf)
        torch.cuda.synchronize(device)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn(buf)
        torch.cuda.synchronize(device)
        return all_max((time.perf_counter() - t0) / iters * 1e6)
    t_lib = timed(library_sync)
    # the form the engine steps use: the gradient is already in the slot, the launch publishes / waits / sums (the library side additionally
    # pays a gradient-norm launch and a second library call per step, which this comparison leaves out: it errs towards the library)
    t_peer = timed(lambda b: peers[n0].reduce_into(b, scratch))
    # once more with NEW data after the timed loops: both slots have been used many times by now, so a reader that kept a stale copy of a
    # peer's slot from an earlier epoch (a cache the acquire did not drop) shows here and nowhere above (the timed loops repeat one buffer).
    # (The library collective of this check runs OUTSIDE the try block: every rank issues it whatever happened to its peer launches.)
    fresh = (src * 0.5 + float(rank + 1)).contiguous()
    ref2 = fresh.clone()
    library_sync(ref2)
    try:
        peers[n0].check()
        timed_ok = 1.0
        for form in ('slot', 'copy'):
            out3 = fresh.clone()
            if form == 'slot':
                peers[n0].slot_tensor(peers[n0].next_slot()).copy_(fresh)
                peers[n0].reduce_into(out3, scratch)
            else:
                peers[n0](out3, scratch)
            torch.cuda.synchronize(device)
            peers[n0].check()
            if not torch.allclose(out3, ref2, rtol=1e-5, atol=1e-6):
                timed_ok = 0.0
    except Exception:               # noqa: BLE001
        timed_ok = 0.0
    report.update(us_library=round(t_lib, 1), us_peer=round(t_peer, 1), floats=n0)
    all_timed_ok = all_min(timed_ok) >= 1.0
    if not all_timed_ok or not t_peer < t_lib:
        for q in peers.values():
            q.close()
        report.update(chosen='library', reason='library collective is not slower' if all_timed_ok else 'a timed peer launch missed its peers or a later average came out wrong')
        return {}, report
    for q in peers.values():                     # in production a late peer is waited for (bounded), not declared dead after 5 s
        q._check(q._lib.mansy_xg_set_timeout_ms(q.ctx, float(timeout_ms)), 'mansy_xg_set_timeout_ms')
    report.update(chosen='peer', reason='correct on every rank and faster')
    return peers, report


def shard_envs(n_env_per_rank, rank, world):
    """-> (index_offset, worker_num): rank r owns global environments [r*n, (r+1)*n) of world*n workers."""
    return rank * n_env_per_rank, world * n_env_per_rank


def merge_moments(a, b):
    """Parallel-variance merge of two (mean, var, count) triples (tianshou RunningMeanStd.update formula)."""
    (am, av, ac), (bm, bv, bc) = a, b
    if bc == 0:
        return am, av, ac
    if ac == 0:
        return bm, bv, bc
    delta = bm - am
    tot = ac + bc
    m2 = av * ac + bv * bc + delta * delta * ac * bc / tot
    return am + delta * bc / tot, m2 / tot, tot


def pooled_moments(triples):
    """[k, 3] tensor of (mean, var, count) rows -> their pooled (mean, var, count) as a [3] tensor, computed where the data
    lives (no host round trip): C = sum c_i, M = sum c_i m_i / C, var = sum c_i (v_i + (m_i - M)^2) / C -- the closed form of
    chaining merge_moments over the rows.  All counts zero: the initial state (0, 1, 0)."""
    m, v, c = triples[:, 0], triples[:, 1], triples[:, 2]
    tot = c.sum()
    safe = torch.where(tot > 0, tot, torch.ones_like(tot))
    mean = (c * m).sum() / safe
    var = (c * (v + (m - mean) ** 2)).sum() / safe
    empty = tot <= 0
    return torch.stack([torch.where(empty, torch.zeros_like(mean), mean), torch.where(empty, torch.ones_like(var), var), tot])


def global_running_moments(rms_local, world, force=False):
    """rms_local: this rank's accumulated [mean, var, count] (float64 tensor, only its own returns).  Returns a NEW tensor
    with the merge over all ranks (identical on every rank); rms_local is not modified.  With one process: a copy.
    Stays on the device: one all-gather of 3 doubles + a handful of tiny tensor ops, no host synchronisation."""
    if world <= 1 and not force:
        return rms_local.clone()
    gathered = [torch.zeros_like(rms_local) for _ in range(max(world, 1))]
    dist.all_gather(gathered, rms_local)
    return pooled_moments(torch.stack(gathered))


def update_running_moments(rms_local, x):
    """rms_local (+)= the statistics of the samples x, in place, on the device (tianshou RunningMeanStd.update)."""
    batch = torch.stack([x.mean(), x.var(unbiased=False), torch.full((), float(x.numel()), dtype=x.dtype, device=x.device)]).to(rms_local.dtype)
    rms_local.copy_(pooled_moments(torch.stack([rms_local, batch])))
    return rms_local
