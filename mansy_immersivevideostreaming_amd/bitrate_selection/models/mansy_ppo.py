"""PPO policy with the QoE-identifier reward relabel (reference: bitrate_selection/models/mansy_ppo.py:14-59 on top of
tianshou==0.4.8's PPOPolicy; net construction at run_mansy.py:205-251), plus the vectorised rollout collector and the
device-resident rollout buffer that replace tianshou's Collector / VectorReplayBuffer for this path.

Everything numerical happens in libmansy_hip.so; this file sequences engine calls:
  collect   : T x [policy forward + Categorical sample (1 call) -> env step (1 call)], transitions written in place into
              step-major slabs [T][N][...] (no per-step host traffic)
  update    : identifier relabel (1 call) -> critic(obs), critic(obs_next), logp_old (3 calls) -> GAE + return normaliser
              (1 call) -> repeat x minibatches of `mansy_ppo_minibatch_step` (1 call each)
tianshou semantics are restated from the 0.4.8 release (T2, parity unpinned -- see oracle/ppo_oracle.py).
"""
import ctypes
from collections.abc import Mapping

import numpy as np
import torch
import torch.nn as nn

from ..._lib import MansyError, check, lib, ptr, stream_ptr
from ..envs.mansy_env import OBS_LD
from .mansy import NetEngine


class RolloutBuffer:
    """[T][N] step-major slabs on the device (what VectorReplayBuffer holds for one collect).
    # T2: tianshou's ReplayBuffer keeps `rew` as float64 (ReplayBuffer.add casts the environment's reward), so the reference's
    # relabel (mansy_ppo.py:47) and the GAE input are float64 there; `rew` is float32 here (the environment's QoE reward is
    # computed in float32 anyway, qoe.py:22-34) and mansy_gae_returns widens it to float64 inside the scan: one float32
    # rounding of the relabelled reward, ~6e-8 relative."""

    def __init__(self, T, N, device):
        self.T, self.N, self.device = T, N, device
        f32 = dict(dtype=torch.float32, device=device)
        # obs and obs_next are the two halves of ONE slab: a full buffer is 2 T N contiguous rows, which process_fn evaluates in one pass
        self.obs2 = torch.zeros(2, T, N, OBS_LD, **f32)
        self.obs, self.obs_next = self.obs2[0], self.obs2[1]
        self.act = torch.zeros(T, N, dtype=torch.int32, device=device)
        self.logp = torch.zeros(T, N, **f32)
        self.rew = torch.zeros(T, N, **f32)
        self.done = torch.zeros(T, N, dtype=torch.uint8, device=device)
        self.filled = 0
        self.generation = 0        # bumped whenever the contents change (reset / collect): keys caches of values computed FROM the contents

    def __len__(self):
        return self.filled * self.N

    def reset(self):
        self.filled = 0
        self.generation += 1


class VecCollector:
    """Collector counterpart: steps N device environments with the policy, filling a RolloutBuffer.  A collect of T vector
    steps is 1 + 3 T kernel launches with no host round trip (pack once; per step: FeatureNet GEMM, head GEMM, and ONE launch
    that forms the logits, samples the action and steps the environment with it, writing observation / reward / done
    straight into the slabs); the whole sequence is captured once into a hipGraph and replayed (sampling uniforms are drawn
    before each replay)."""

    def __init__(self, policy, venv, seed=0, use_graph=True, use_team=False):
        self.policy, self.venv = policy, venv
        self.carry = None                       # observation the next collect starts from
        self.seed = int(seed) & 0x7FFFFFFF
        self.step_count = 0
        self.env_step = 0
        self.use_graph = use_graph
        # round 5: the whole collect as ONE persistent launch on XCD teams (mansy_policy_rollout) where the library takes it (fp32, batch in the
        # wave-split-K range).  Bit-identical to the per-step launches and 49 launches fewer, but NOT faster (24.0 against 22.5 us per vector
        # step: profiles/r05_rollout_team_ab.txt), so it is opt-in (the `use_team` argument / attribute); never on a device shared between
        # ranks (the kernel wants one resident workgroup per CU)
        self.use_team = bool(use_team)
        self._graph = None
        self._graph_key = None
        self.graph_launches = 0
        self._u = None

    @property
    def obs(self):
        return self.carry

    def reset_env(self):
        self.carry = self.venv.reset().clone()

    def _body(self, buffer, T):
        eng = self.policy.engine
        buffer.obs[0].copy_(self.carry)
        for t in range(T):              # policy forward + sampling + environment step: 3 launches per vector step (4 unfused)
            nxt = buffer.obs[t + 1] if t + 1 < T else self.carry
            eng.policy_env_step(self.venv, buffer.obs[t], self._u[t], buffer.act[t], buffer.logp[t], nxt, buffer.obs_next[t], buffer.rew[t],
                                buffer.done[t], reuse_packed=t > 0)

    def collect(self, n_step, buffer):
        """n_step environment steps in total (n_step / N per environment); returns {'n/st': ..}."""
        N = self.venv.n_env
        T = max(1, n_step // N)
        if buffer.T < T or buffer.N != N:
            raise MansyError('rollout buffer too small')
        if self.carry is None:
            self.reset_env()
        dev = self.carry.device
        if self._u is None or self._u.shape != (T, N):
            self._u = torch.empty(T, N, dtype=torch.float32, device=dev)
            self._graph = None
        buffer.reset()
        self._u.uniform_()                      # Categorical sampling uniforms (torch generator => reproducible with manual_seed)
        if self.use_team and buffer.obs.is_contiguous() and buffer.obs.shape[1] == N:
            buffer.obs[0].copy_(self.carry)
            took = self.policy.engine.policy_rollout(self.venv, T, buffer.obs, self._u, buffer.act, buffer.logp, buffer.obs_next, self.carry, buffer.rew,
                                                     buffer.done)
            if took:
                self.step_count += T
                buffer.filled = T
                buffer.generation = getattr(buffer, 'generation', 0) + 1
                self.env_step += T * N
                return {'n/st': T * N}
            self.use_team = False               # the library said the form does not apply here: per-step launches from now on
        key = (id(buffer), T, self.policy.engine.ac.flat_p.data_ptr())
        if self.use_graph and (self._graph is None or self._graph_key != key):
            try:
                self.policy.engine.workspace()
                s = torch.cuda.Stream(device=dev)
                s.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(s):          # warm-up outside capture (allocations, lazy inits); env state is restored below
                    state = self.venv.state.clone()
                    carry = self.carry.clone()
                    self._body(buffer, T)
                    self.venv.state.copy_(state)
                    self.carry.copy_(carry)
                    self.venv.elog_count.zero_()
                torch.cuda.current_stream(dev).wait_stream(s)
                g = torch.cuda.CUDAGraph()
                n0 = lib().mansy_prof_launch_count()
                with torch.cuda.graph(g):
                    self._body(buffer, T)
                self.graph_launches = int(lib().mansy_prof_launch_count() - n0)      # library kernels one replay re-runs (bench.py)
                self._graph, self._graph_key = g, key
                self.venv.state.copy_(state)
                self.carry.copy_(carry)
                self.venv.elog_count.zero_()
            except Exception as e:                  # capture unsupported: fall back to direct launches
                import warnings
                warnings.warn(f'hipGraph capture of the rollout failed ({e}); using direct launches')
                self.use_graph = False
                self._graph = None
        if self.use_graph and self._graph is not None:
            self._graph.replay()
        else:
            self._body(buffer, T)
        self.step_count += T
        buffer.filled = T
        buffer.generation = getattr(buffer, 'generation', 0) + 1      # new contents: values evaluated on the old ones (PPOPolicy._pre_eval) are stale
        self.env_step += T * N
        return {'n/st': T * N}


def split_indices(length, size, shuffle=True, merge_last=True):
    """T2: tianshou Batch.split(size, shuffle=True, merge_last=True) index chunks (np.random.permutation)."""
    if size == -1:
        size = length
    indices = np.random.permutation(length) if shuffle else np.arange(length)
    merge_last = merge_last and length % size > 0
    for idx in range(0, length, size):
        if merge_last and idx + size + size >= length:
            yield indices[idx:]
            break
        yield indices[idx:idx + size]


class LazyLosses(Mapping):
    """The per-minibatch loss statistics of an update ({'loss': [...], ...} like tianshou's learn()), fetched from the device
    on first access: an update that nobody inspects (every collect but the logged ones) costs no device-to-host sync.
    A read-only Mapping (dict(x), x.items(), x['loss'], json all see the values)."""

    def __init__(self, names, stat_tensors):
        self._names, self._pending, self._data = tuple(names), stat_tensors, None
        self.n_steps = sum(int(t.shape[0]) for t in stat_tensors)        # gradient steps taken (known without touching the device)

    def _load(self):
        if self._data is None:
            st = torch.cat(self._pending).cpu().numpy() if self._pending else np.zeros((0, len(self._names)), np.float32)
            self._pending = None
            self._data = {k: st[:, j].tolist() for j, k in enumerate(self._names)}
        return self._data

    def __getitem__(self, k):
        return self._load()[k]

    def __iter__(self):
        return iter(self._names)

    def __contains__(self, k):
        return k in self._names

    def __len__(self):
        return len(self._names)

    def __repr__(self):
        return repr(self._load())


class _ActorCritic(nn.Module):
    """tianshou.utils.net.common.ActorCritic: only here so state_dict() carries the `_actor_critic.*` duplicates."""

    def __init__(self, actor, critic):
        super().__init__()
        self.actor, self.critic = actor, critic


class _Result:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class PPOPolicy(nn.Module):
    """Reference signature (mansy_ppo.py:14-33 + the tianshou keyword arguments used at run_mansy.py:231-251).  `optim` is
    accepted for signature compatibility; its lr / weight_decay are read from it and the fused Adam(L2) kernel does the step."""

    def __init__(self, actor, critic, optim, dist_fn, eps_clip=0.2, dual_clip=None, value_clip=False, advantage_normalization=True,
                 recompute_advantage=False, args=None, identifier=None, discount_factor=0.99, max_grad_norm=None, vf_coef=0.5,
                 ent_coef=0.01, reward_normalization=False, gae_lambda=0.95, max_batchsize=256, action_space=None, action_scaling=False,
                 identifier_optim=None, **kwargs):
        super().__init__()
        if dual_clip is not None and not dual_clip > 1.0:
            raise MansyError('Dual-clip PPO parameter should greater than 1.0.')      # T2: PPOPolicy.__init__'s assertion
        self._dual_clip = dual_clip
        self._recompute_adv = bool(recompute_advantage)
        self.actor, self.critic = actor, critic
        self._actor_critic = _ActorCritic(actor, critic)
        self.identifier = identifier
        self.optim, self.dist_fn, self.args = optim, dist_fn, args
        self.identifier_optim = identifier_optim
        self._eps_clip, self._value_clip, self._norm_adv = eps_clip, bool(value_clip), bool(advantage_normalization)
        self._gamma, self._lambda, self._grad_norm = discount_factor, gae_lambda, max_grad_norm
        self._weight_vf, self._weight_ent, self._rew_norm = vf_coef, ent_coef, bool(reward_normalization)
        self.cnt, self.observe_round = 0, 1000
        self.updating = False
        self.lr_scheduler = None
        self.engine = NetEngine(actor=actor, critic=critic, identifier=identifier, max_batch=8192)
        self._rms = None
        self._seed_ctr = 0
        self.world, self.grad_sync = 1, None
        self.chain_steps = True       # learn(): each minibatch step's last launch prepares the next one (False: self-contained steps)
        self.overlap_identifier_sync = True   # data parallel: the identifier's gradient averages fly under process_fn's evaluation passes
        self._pre_eval = None
        self._pinned = {}
        self.peer_in_slot = True      # peer-memory averages: gradients are produced straight in the exchange slot (no copy in front of the flag)
        # round 6: train_identifier() and update() replayed from captured hipGraphs (the rollout half has been one since round 2): a cycle's host
        # work is then three replays + one staged upload each (the host-drawn shuffles and the Adam bias corrections of the replay's step counts)
        # instead of ~150 engine calls -- the host leaves the critical path (1.42 of a 1.75 ms cycle before).  'auto': wherever the captured
        # sequence is made of this library's own launches only (single process, or peer-memory averages in the exchange-slot form); library
        # collectives (RCCL / torch.distributed) are captured only when forced with True.  False: direct launches.
        self.graph_update = 'auto'
        self._graphs, self._graph_seen = {}, set()
        self.graph_replays = 0
        self.graph_launches = 0       # library launches re-run by the replays so far (mansy_prof_launch_count only sees direct launches)

    def set_data_parallel(self, world, grad_sync, peer=False, force=False, comm=None, in_slot=None):
        """One process per GPU: `grad_sync(flat_grad)` averages a flat gradient buffer over ranks (dist.make_grad_sync).
        peer=True: the hand-written one-shot all-reduce over peer-mapped memory (dist.PeerGradSync, csrc/xgmi.hip) for both flat
        buffers instead -- one launch that also leaves the gradient's sums of squares for the clip.  peer='auto': build it, check it
        against `grad_sync` on the same data, time both, and use it only if every rank finds it correct and faster
        (dist.probe_peer_grad_sync; the decision and the two timings are kept in `self.grad_sync_report`).
        force=True (bench.py's `dp_form` leg): take the data-parallel FORM of the update at world 1 too -- raw gradients, the average really
        issued (grad_sync over a one-rank group, or the peer kernel on a one-rank context), then clip + Adam as a launch of its own --
        i.e. everything a rank pays for data parallelism except the wire time."""
        self.world, self.grad_sync = int(world), grad_sync
        self._graphs, self._graph_seen = {}, set()          # captured sequences hold the previous form's launches
        # in_slot: the exchange-slot form publishes with ONE flag store and relies on the kernel boundary in front of it having
        # written the gradient kernels' stores back to the memory side.  That is validated on THIS machine's links by the start-up probe
        # (peer='auto': fresh data written by many workgroups right before every average, both forms against the library's average); a forced
        # peer=True skips the probe, so it takes the copy form (every workgroup releases its own copy at system scope) unless in_slot=True is
        # asked for explicitly (bench.py's world-1 legs; tests).
        if in_slot is not None:
            self.peer_in_slot = bool(in_slot)
        elif peer is True and int(world) > 1:
            self.peer_in_slot = False
        elif peer == 'auto':
            self.peer_in_slot = True
        # comm: a dist.RcclComm (the library's own RCCL communicator): the library-collective form of the step also becomes ONE call per step
        # (gradients, ncclAllReduce(avg), norm, clip + Adam inside mansy_ppo_minibatch_step); it doubles as grad_sync where the one-call form
        # does not apply (unchained steps)
        self._comm = comm
        if comm is not None and grad_sync is None:
            self.grad_sync = comm
        if force and self.grad_sync is None:
            self.grad_sync = lambda flat_g: None          # the peer kernel does the work; a library sync was not asked for
        for old in (getattr(self, '_peer', None) or {}).values():      # a second call: release the previous hipIpc mappings / slots
            old.close()
        self._peer = {}
        self.grad_sync_report = {'chosen': 'library' if self.world > 1 else 'none'}
        if peer and (self.world > 1 or force):
            import torch.distributed as tdist
            from ...dist import PeerGradSync, probe_peer_grad_sync
            flats = [f for f in (self.engine.ac, self.engine.idn) if f is not None]
            if peer == 'auto':
                sizes = sorted({f.flat_p.numel() for f in flats})
                peers, self.grad_sync_report = probe_peer_grad_sync(sizes, self.world, tdist.get_rank(), flats[0].flat_p.device, grad_sync)
                for f in flats:
                    if f.flat_p.numel() in peers:
                        self._peer[id(f)] = peers[f.flat_p.numel()]
            else:
                rank = tdist.get_rank() if (tdist.is_initialized() and self.world > 1) else 0      # (a forced one-rank context inside a multi-rank job is rank 0 of ITS world)
                for f in flats:
                    self._peer[id(f)] = PeerGradSync(f.flat_p.numel(), self.world, rank, f.flat_p.device)
                self.grad_sync_report = {'chosen': 'peer', 'reason': 'forced'}

    def _check_peers(self):
        """A peer-memory all-reduce whose bounded wait gave up has overwritten the gradient with NaN and raised its context's sticky
        flag (csrc/xgmi.hip); the clip + Adam launch behind it has already run.  Called at the end of every learn() /
        train_identifier() that used a peer context: raises MansyError on the rank that timed out.  mansy_xg_status does NOT synchronise
        (it reads a host-mapped sticky word), so here it only covers launches that have COMPLETED -- a cheap early warning on the hot path.
        The definitive verdict is `sync_check()`, taken wherever parameters leave the process (state_dict(): checkpoints, best-model saves,
        the end of training)."""
        for p in (getattr(self, '_peer', None) or {}).values():
            p.check()

    def sync_check(self):
        """Definitive verdict on the peer-memory averages issued so far: wait for the device, then read the sticky words (a
        time-out in the update that has just been enqueued would otherwise be saved into checkpoint.pth before the next learn() raises, and the
        last update of a run was never checked).  No-op without peer contexts."""
        if getattr(self, '_peer', None):
            torch.cuda.synchronize(self.engine.device)
            self._check_peers()

    def state_dict(self, *args, **kwargs):
        self.sync_check()            # parameters are about to leave the process (torch.save in save_checkpoint_fn / save_best_fn, run_mansy.py:70-84)
        return super().state_dict(*args, **kwargs)

    def _upload_i32(self, key, arr, dev):
        """Host int array -> device int32 tensor through a persistent PINNED staging buffer and a non-blocking copy: the copy engine
        moves it, no staging kernel runs on the compute queue (a pageable upload costs one or two `copyBuffer` launches each; a cycle
        had eleven).  One staging buffer per call site (`key`); its previous upload is waited for before it is overwritten."""
        arr = np.ascontiguousarray(arr, dtype=np.int32)
        slot = self._pinned.get(key)
        if slot is None or slot[0].numel() < arr.size:
            slot = [torch.empty(max(arr.size, 1), dtype=torch.int32).pin_memory(), None]
            self._pinned[key] = slot
        if slot[1] is not None:
            slot[1].synchronize()
        slot[0][:arr.size].copy_(torch.from_numpy(arr))
        out = slot[0][:arr.size].to(dev, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(dev))
        return out

    def _xg_ctx(self, f):
        """The peer-memory context of flat buffer `f` when its data-parallel steps run in the exchange-slot form (round 5): the engine call
        produces the raw gradients straight in the rank's exchange slot, one more launch averages them over the ranks, clip + Adam follow --
        ONE library call per step, like the single-process step (mansy_ppo_minibatch_step / mansy_identifier_train_step, `xg_ctx`).  None:
        single process, a library collective, or the round-4 copy form (peer_in_slot = False)."""
        peer = self._peer.get(id(f)) if getattr(self, '_peer', None) else None
        if peer is not None:
            return peer.ctx if self.peer_in_slot else None
        comm = getattr(self, '_comm', None)          # no peer kernel: the library's own RCCL communicator, if one was handed over
        return comm.ctx if comm is not None else None

    def _sync_clip_adam(self, f, max_norm, lr, wd, tail=None, overlap=None, bias=None):
        """Data-parallel second half of a step: average the raw local gradients over the ranks, then global-norm clip + Adam.
        tail = (data, next_idx or None): the chained form (actor-critic, clipped): the clip + Adam launch also zeroes the
        gradients, re-packs the updated parameters and prepares the next minibatch (mansy_ppo_dp_tail).
        overlap: a callable that enqueues work which does not touch `f` -- it runs on the caller's stream WHILE the average is in
        flight on a side stream (the identifier's two all-reduces hide under the critic / log-prob passes of process_fn)."""
        peer = self._peer.get(id(f)) if getattr(self, '_peer', None) else None
        dev = f.flat_p.device
        scratch = self._pinned.get(('clip_scratch', dev))               # MANSY_CLIP_SCRATCH_DOUBLES, one per device (stream-ordered reuse)
        if scratch is None:
            scratch = self._pinned[('clip_scratch', dev)] = torch.empty(64, dtype=torch.float64, device=dev)

        def average():
            if peer is not None:
                peer(f.flat_g, scratch)
            else:
                self.grad_sync(f.flat_g)
        if overlap is None:
            average()
        else:
            main = torch.cuda.current_stream(dev)
            if getattr(self, '_sync_stream', None) is None or self._sync_stream.device != dev:
                self._sync_stream = torch.cuda.Stream(device=dev)
            ready = torch.cuda.Event()
            ready.record(main)                      # the local gradients are complete
            self._sync_stream.wait_event(ready)
            with torch.cuda.stream(self._sync_stream):
                average()
                done = torch.cuda.Event()
                done.record(self._sync_stream)
            overlap()                               # independent work, enqueued behind the gradients on the caller's stream
            main.wait_event(done)
        if tail is not None:
            data, nxt = tail
            arr, _ = f.pointers()
            check(lib().mansy_ppo_dp_tail(arr, ptr(f.flat_p), ptr(f.flat_g), ptr(f.m), ptr(f.v), f.flat_p.numel(), float(max_norm), lr, wd, f.step,
                                          ptr(scratch), int(peer is not None), ptr(data['obs']), ptr(data['adv']), ptr(nxt),
                                          nxt.numel() if nxt is not None else 0, None, ptr(bias), ptr(self.engine.workspace()), self.engine.max_batch,
                                          self.engine.prec, stream_ptr(f.flat_p.device)), 'mansy_ppo_dp_tail')
            return
        check(lib().mansy_clip_grad_adam(ptr(f.flat_p), ptr(f.flat_g), ptr(f.m), ptr(f.v), f.flat_p.numel(), float(max_norm), lr, wd, f.step,
                                         *f.tail(), ptr(scratch), int(peer is not None), ptr(bias), stream_ptr(f.flat_p.device)), 'mansy_clip_grad_adam')

    # ---- plumbing ---------------------------------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        for f in (self.engine.ac, self.engine.idn):
            if f is not None:
                f.flatten()
        self._rms = None
        return out

    def _hyper(self, optim, default_lr):
        if optim is None:
            return default_lr, 0.0
        g = optim.param_groups[0]
        return g['lr'], g.get('weight_decay', 0.0)

    def ret_rms(self):
        """tianshou RunningMeanStd state [mean, var, count] as device doubles."""
        dev = self.engine.device
        if self._rms is None or self._rms.device != dev:
            self._rms = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64, device=dev)
        return self._rms

    # ---- reference API ----------------------------------------------------------------------------------------
    def forward(self, batch, state=None, **kwargs):
        """tianshou PGPolicy.forward: Batch(logits, act, state, dist); `batch.obs` dict/Batch of numpy arrays or tensor."""
        from .mansy import obs_to_tensor
        obs = batch.obs if hasattr(batch, 'obs') else batch['obs']
        t = obs_to_tensor(obs, self.engine.device)
        self._seed_ctr += 1
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        logits, _, act, _ = self.engine.policy_forward(t, want_value=False, sample=True, seed=seed, site=self._seed_ctr)
        return _Result(logits=logits, act=act.long(), state=state, dist=self.dist_fn(logits) if self.dist_fn else None)

    # ---- hipGraph replay of the update half (round 6) ---------------------------------------------------------------
    def _graph_ok(self, n_steps_per_flat):
        """May this sequence be captured?  Needs an even number of Adam steps per flat buffer (the norm-slot sets and the peer exchange slots
        alternate with the step count: a replay must start on the parity the capture started on), no lagged tail, and -- unless forced --
        no library collective inside (see `graph_update`)."""
        if not self.graph_update or n_steps_per_flat % 2 or self.engine.ac.tail()[0] >= 0:
            return False
        if self.graph_update is True:
            return True
        if self.grad_sync is None:
            return True
        return bool(getattr(self, '_peer', None)) and self.peer_in_slot and self.world >= 1 and getattr(self, '_comm', None) is None and \
            all(self._xg_ctx(f) is not None for f in (self.engine.ac, self.engine.idn) if f is not None)

    @staticmethod
    def _adam_bias(step0, k):
        """[k, 2] float32: (1 - 0.9^t, sqrt(1 - 0.999^t)) for t = step0 + 1 .. step0 + k, in the host arithmetic of csrc/ppo_engine.hip
        (double pow, then float)."""
        t = np.arange(step0 + 1, step0 + k + 1, dtype=np.float64)
        return np.stack([1.0 - np.power(0.9, t), np.sqrt(1.0 - np.power(0.999, t))], 1).astype(np.float32)

    def _stage(self, key, host_i32, dst):
        """Host int32 array -> the persistent device buffer `dst` a captured graph reads, through a pinned staging buffer (stream-ordered,
        no staging kernel).  The previous upload of the same buffer is waited for before the staging memory is overwritten."""
        n = host_i32.size
        ring = self._pinned.get(key)
        if ring is None or ring['buf'].shape[1] < n:
            # EIGHT staging buffers used in turn: the host may run up to seven cycles ahead of the device before it has to wait for an upload to
            # have been consumed (with one buffer every cycle's staging waited for the previous cycle's copy, i.e. for the device: the host's
            # enqueue time then read as the device's cycle time)
            ring = self._pinned[key] = dict(buf=torch.empty(8, max(n, 1), dtype=torch.int32).pin_memory(), ev=[None] * 8, i=0)
        i = ring['i']
        ring['i'] = (i + 1) % 8
        if ring['ev'][i] is not None:
            ring['ev'][i].synchronize()
        ring['buf'][i, :n].copy_(torch.from_numpy(host_i32))
        dst[:n].copy_(ring['buf'][i, :n], non_blocking=True)
        ring['ev'][i] = torch.cuda.Event()
        ring['ev'][i].record(torch.cuda.current_stream(dst.device))

    def _capture(self, dev, body):
        """Capture `body()` (engine calls only, fixed shapes / addresses) on a side stream into a CUDAGraph; returns (graph, body's value,
        library launches one replay re-runs).  Tensors `body` allocates live in the graph's pool for as long as the graph does."""
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        n0 = lib().mansy_prof_launch_count()
        with torch.cuda.graph(g):
            out = body()
        return g, out, int(lib().mansy_prof_launch_count() - n0)

    def train_identifier(self, buffer, update_round=2, verbose=True):
        """utils/mansy_utils.py:9-39 on the collected buffer: np.random.shuffle, 80/20 split, `update_round` full-batch
        MSE steps with Adam(lr, L2), then the validation loss."""
        self.engine.check_rollout(sync=True)       # (no-op unless the persistent-rollout form filled the buffer: its verdict before the data is used)
        n = len(buffer)
        idx = np.arange(n)
        np.random.shuffle(idx)
        f = self.engine.idn
        losses = vloss = None
        fixed = isinstance(buffer, RolloutBuffer)          # (behaviour cloning hands over duck-typed demonstration buffers of varying length: direct launches)
        key = ('ident', id(buffer), buffer.filled, buffer.N, buffer.obs.data_ptr(), int(update_round), f.flat_p.data_ptr(), self.engine.prec) if fixed else None
        if fixed and update_round > 0 and self._graph_ok(update_round):
            G = self._graphs.get(key)
            if G is None and key in self._graph_seen:          # second call with this shape: capture (the first ran direct = the warm-up)
                G = self._capture_identifier(key, buffer, n, update_round)
            if G is not None:
                host = np.concatenate([idx.astype(np.int32), self._adam_bias(f.step, update_round).reshape(-1).view(np.int32)])
                self._stage(('ident_in', key), host, G['inp'])
                f.step += update_round
                G['graph'].replay()
                self.graph_replays += 1
                self.graph_launches += G['launches']
                snap = G['lossbuf'].clone()              # one copy: the graph's own buffer is overwritten by the next replay
                losses, vloss = [snap[r] for r in range(update_round)], (snap[update_round] if G['vloss'] is not None else None)
            else:
                self._graph_seen.add(key)
        if losses is None:
            losses, vloss = self._train_identifier_body(buffer, n, self._upload_i32('ident', idx, buffer.obs.device), update_round)
        if self.grad_sync is not None and update_round > 0:
            self._check_peers()
        if verbose:
            for l in losses:
                print('identifier loss is: ', l.item())
            if vloss is not None:
                print('identifier validation loss is: ', vloss.item())
        return losses, vloss

    def _capture_identifier(self, key, buffer, n, update_round):
        dev = buffer.obs.device
        inp = torch.zeros(n + 2 * update_round, dtype=torch.int32, device=dev)
        bias = inp[n:].view(torch.float32).view(update_round, 2)
        f = self.engine.idn
        step0, pre = f.step, self._pre_eval
        try:
            g, (losses, vloss), nl = self._capture(dev, lambda: self._train_identifier_body(buffer, n, inp[:n], update_round, bias=bias))
        except Exception as e:                      # capture unsupported for this form: direct launches from now on
            import warnings
            warnings.warn(f'hipGraph capture of train_identifier failed ({e}); using direct launches')
            self.graph_update = False
            f.step = step0
            return None
        f.step, self._pre_eval = step0, pre         # the capture executed nothing: the replay that follows takes these steps
        G = self._graphs[key] = dict(graph=g, inp=inp, lossbuf=losses[0]._base if losses[0]._base is not None else losses[0], vloss=vloss, launches=nl)
        return G

    def _train_identifier_body(self, buffer, n, idx_t, update_round, bias=None):
        """The engine calls of train_identifier on shuffled row indices `idx_t` (device int32 [n])."""
        eng = self.engine
        obs = buffer.obs[:buffer.filled].reshape(n, OBS_LD)
        ntr = int(n * 0.8)
        # the shuffled 80 / 20 split as row indices into the buffer: each call gathers its rows in its own prologue launch (no
        # shuffled copy of the 12.8 MB of observations, no separate gather launch)
        tr, va = idx_t[:ntr], idx_t[ntr:n]
        lr, wd = self._hyper(self.identifier_optim, 1e-4)
        f = eng.idn
        losses = []
        lossbuf = torch.empty(update_round + 1, dtype=torch.float32, device=obs.device)       # [round losses ..., validation loss]
        # data parallel: each round's gradient average is on the critical path (1.05 MB, latency-bound).  The critic / log-prob passes
        # of the process_fn that follows (mansy_ppo.py:53: v_s + logp_old on obs, v_s_ on obs_next) depend on the actor-critic only,
        # which this function does not touch: round r's average flies on a side stream while pass r runs here; process_fn then finds
        # the values it needs (self._pre_eval) instead of recomputing them.
        # (peer-memory averages in the exchange-slot form are one ~5 us launch inside the step's own call: nothing worth hiding)
        hide = self.grad_sync is not None and self.overlap_identifier_sync and self._xg_ctx(f) is None and bias is None
        for r in range(update_round):
            f.step += 1
            ov = (lambda part=r: self._pre_evaluate(buffer, part)) if hide and r < 2 else None
            losses.append(self._identifier_step(obs, lr, wd, f.step, rows=tr, overlap=ov, bias=None if bias is None else bias[r], loss=lossbuf[r]))
        vloss = self._identifier_step(obs, lr, wd, 0, rows=va, loss=lossbuf[update_round]) if len(va) else None
        return losses, vloss

    def _identifier_step(self, obs, lr, wd, step, rows=None, overlap=None, bias=None, loss=None):
        """One train_identifier step on `obs` (rows=None) or on its rows `rows` (device int32 indices).  bias: device [2] floats the Adam
        launch reads instead of deriving them from `step` (graph replays)."""
        eng, f = self.engine, self.engine.idn
        B = obs.shape[0] if rows is None else rows.numel()
        if B > eng.max_batch:
            raise MansyError(f'identifier batch {B} exceeds engine max_batch {eng.max_batch}')
        arr, garr = f.pointers(grads=True)
        if loss is None:
            loss = torch.empty((), dtype=torch.float32, device=obs.device)
        dp = self.grad_sync is not None and step > 0
        xg = self._xg_ctx(f) if dp else None
        if xg is not None:          # the data-parallel step as ONE call: gradients into the exchange slot, one launch averages them, Adam
            check(lib().mansy_identifier_train_step(arr, garr, ptr(f.flat_p), ptr(f.flat_g), ptr(f.m), ptr(f.v), f.flat_p.numel(), ptr(obs), ptr(rows), B, lr, wd,
                                                    step, ptr(loss), ptr(eng.workspace()), eng.max_batch, xg, ptr(bias), eng.prec, stream_ptr(obs.device)),
                  'mansy_identifier_train_step')
            return loss
        check(lib().mansy_identifier_train_step(arr, garr, ptr(f.flat_p), ptr(f.flat_g), ptr(f.m), ptr(f.v), f.flat_p.numel(), ptr(obs), ptr(rows), B, lr, wd,
                                                -1 if dp else step, ptr(loss), ptr(eng.workspace()), eng.max_batch, None, ptr(bias), eng.prec, stream_ptr(obs.device)),
              'mansy_identifier_train_step')
        if dp:
            self._sync_clip_adam(f, 0.0, lr, wd, overlap=overlap, bias=bias)
        return loss

    def relabel(self, buffer, lamb):
        """mansy_ppo.py:41-48, batched: rew <- (1 - lamb) rew + lamb (1 - MSE(identifier(obs, obs.action_one_hot), obs.qoe_weight))."""
        eng = self.engine
        n = len(buffer)
        obs = buffer.obs[:buffer.filled].reshape(n, OBS_LD)
        rew = buffer.rew[:buffer.filled].reshape(n)
        arr, _ = eng.idn.pointers()
        for s in range(0, n, eng.max_batch):
            e = min(n, s + eng.max_batch)
            check(lib().mansy_identifier_relabel(arr, ptr(obs[s:e]), ptr(rew[s:e]), None, e - s, float(lamb), ptr(eng.workspace()), eng.max_batch, eng.prec,
                                                 stream_ptr(obs.device)), 'mansy_identifier_relabel')
        self.cnt += n

    def _pre_eval_key(self, buffer):
        return (id(buffer), getattr(buffer, 'generation', None), buffer.filled, buffer.N, self.engine.ac.step, buffer.obs.data_ptr())

    def _pre_evaluate(self, buffer, part):
        """One of process_fn's two evaluation passes ahead of time (part 0: v_s + logp_old on obs, part 1: v_s_ on obs_next), kept for the
        process_fn of the same buffer under the same actor-critic parameters."""
        eng = self.engine
        T, N = buffer.filled, buffer.N
        n = T * N
        dev = buffer.obs.device
        key = self._pre_eval_key(buffer)
        pe = self._pre_eval
        if pe is None or pe['key'] != key:
            pe = self._pre_eval = dict(key=key, done=set(), v_s=torch.empty(n, dtype=torch.float32, device=dev),
                                       v_next=torch.empty(n, dtype=torch.float32, device=dev), logp_old=torch.empty(n, dtype=torch.float32, device=dev))
        arr, _ = eng.ac.pointers()
        src = (buffer.obs if part == 0 else buffer.obs_next)[:T].reshape(n, OBS_LD)
        act = buffer.act[:T].reshape(n)
        for s in range(0, n, eng.max_batch):
            e = min(n, s + eng.max_batch)
            if part == 0:
                check(lib().mansy_policy_evaluate(arr, ptr(src[s:e]), e - s, ptr(act[s:e]), e - s, ptr(pe['logp_old'][s:e]), ptr(pe['v_s'][s:e]),
                                                  ptr(eng.workspace()), eng.max_batch, eng.prec, stream_ptr(dev)), 'mansy_policy_evaluate')
            else:
                check(lib().mansy_policy_evaluate(arr, ptr(src[s:e]), e - s, None, 0, None, ptr(pe['v_next'][s:e]), ptr(eng.workspace()),
                                                  eng.max_batch, eng.prec, stream_ptr(dev)), 'mansy_policy_evaluate')
        pe['done'].add(part)

    def process_fn(self, buffer, out=None):
        """T2: A2CPolicy._compute_returns + PPOPolicy.process_fn: v_s, v_s_, GAE, normalised returns, logp_old.
        out: a previous call's result whose tensors are to be REUSED (same addresses: the captured learn graph reads them)."""
        eng = self.engine
        T, N = buffer.filled, buffer.N
        n = T * N
        dev = buffer.obs.device
        obs = buffer.obs[:T].reshape(n, OBS_LD)
        obs_next = buffer.obs_next[:T].reshape(n, OBS_LD)
        act = buffer.act[:T].reshape(n)
        arr, _ = eng.ac.pointers()
        v_s = torch.empty(n, dtype=torch.float32, device=dev)
        v_next = torch.empty(n, dtype=torch.float32, device=dev)
        logp_old = torch.empty(n, dtype=torch.float32, device=dev)
        joint = getattr(buffer, 'obs2', None)
        pe, self._pre_eval = self._pre_eval, None
        if pe is not None and pe['key'] == self._pre_eval_key(buffer) and pe['done']:
            # (some of) the passes ran under the identifier's gradient averages (train_identifier): take them, compute the rest
            for part in (0, 1):
                if part not in pe['done']:
                    self._pre_eval = pe
                    self._pre_evaluate(buffer, part)
                    self._pre_eval = None
            v_s, v_next, logp_old = pe['v_s'], pe['v_next'], pe['logp_old']
        elif joint is not None and T == buffer.T and 2 * n <= eng.max_batch and obs.data_ptr() == joint.data_ptr():
            # [obs ; obs_next] are 2 n contiguous rows: values of both halves and logp_old of the first in ONE pass (5 launches for 9)
            v_all = torch.empty(2 * n, dtype=torch.float32, device=dev)
            check(lib().mansy_policy_evaluate(arr, ptr(joint), 2 * n, ptr(act), n, ptr(logp_old), ptr(v_all), ptr(eng.workspace()), eng.max_batch, eng.prec,
                                              stream_ptr(dev)), 'mansy_policy_evaluate')
            v_s, v_next = v_all[:n], v_all[n:]
        else:
            for s in range(0, n, eng.max_batch):
                e = min(n, s + eng.max_batch)
                check(lib().mansy_policy_evaluate(arr, ptr(obs[s:e]), e - s, ptr(act[s:e]), e - s, ptr(logp_old[s:e]), ptr(v_s[s:e]),
                                                  ptr(eng.workspace()), eng.max_batch, eng.prec, stream_ptr(dev)), 'mansy_policy_evaluate')
                check(lib().mansy_policy_evaluate(arr, ptr(obs_next[s:e]), e - s, None, 0, None, ptr(v_next[s:e]), ptr(eng.workspace()),
                                                  eng.max_batch, eng.prec, stream_ptr(dev)), 'mansy_policy_evaluate')
        if out is not None:          # persistent outputs: the values land where the captured graph reads them
            for k, t in (('v_s', v_s), ('v_next', v_next), ('logp_old', logp_old)):
                out[k].copy_(t)
            self._returns_from_values(out)
            return out
        returns = torch.empty(n, dtype=torch.float32, device=dev)
        adv = torch.empty(n, dtype=torch.float32, device=dev)
        data = dict(obs=obs, obs_next=obs_next, act=act, v_s=v_s, v_next=v_next, logp_old=logp_old, returns=returns, adv=adv, n=n, buffer=buffer)
        self._returns_from_values(data)
        return data

    def _returns_from_values(self, data):
        """T2: the second half of A2CPolicy._compute_returns -- GAE over data['v_s'] / data['v_next'] with the running return
        statistics, normalised returns and advantages written IN PLACE into data['returns'] / data['adv'], then ret_rms.update."""
        buffer = data['buffer']
        T, N, n = buffer.filled, buffer.N, data['n']
        dev = data['obs'].device
        v_s, v_next, returns, adv = data['v_s'], data['v_next'], data['returns'], data['adv']
        scratch = torch.empty(n + 2, dtype=torch.float64, device=dev)
        rms_local = self.ret_rms()
        rms_use = rms_local
        if self.world > 1:                      # normalise with the statistics of ALL ranks; accumulate only our own returns locally
            from ...dist import global_running_moments
            rms_use = global_running_moments(rms_local, self.world)
        check(lib().mansy_gae_returns(ptr(buffer.rew[:T]), ptr(v_s), ptr(v_next), ptr(buffer.done[:T]), T, N, self._gamma, self._lambda,
                                      int(self._rew_norm), ptr(rms_use), ptr(scratch), ptr(returns), ptr(adv), stream_ptr(dev)),
              'mansy_gae_returns')
        if self.world > 1 and self._rew_norm:      # accumulate only our own (un-normalised) returns locally; no host round trip
            from ...dist import update_running_moments
            update_running_moments(rms_local, scratch[:n])

    def _recompute_returns(self, data):
        """T2: PPOPolicy.learn with recompute_advantage, before every pass but the first: `batch = self._compute_returns(batch, buffer,
        indices)` -- the CURRENT critic's values of obs / obs_next (they also become the value-clip reference v_s), GAE, returns and
        the running return statistics again; logp_old stays the one process_fn took."""
        eng, n, dev = self.engine, data['n'], data['obs'].device
        arr, _ = eng.ac.pointers()
        for src, dst in ((data['obs'], data['v_s']), (data['obs_next'], data['v_next'])):
            for s in range(0, n, eng.max_batch):
                e = min(n, s + eng.max_batch)
                check(lib().mansy_policy_evaluate(arr, ptr(src[s:e]), e - s, None, 0, None, ptr(dst[s:e]), ptr(eng.workspace()), eng.max_batch, eng.prec,
                                                  stream_ptr(dev)), 'mansy_policy_evaluate')
        self._returns_from_values(data)

    def learn(self, data, batch_size, repeat, passes=None, perm=None, bias=None):
        """T2: PPOPolicy.learn: `repeat` passes over shuffled minibatches (np.random.permutation, merge_last).
        passes / perm: the passes' index chunks drawn by the caller and their concatenation as a device int32 tensor (graph capture / replay:
        update() draws and stages them); bias: device [n_steps, 2] Adam bias corrections read by the step's last launch (graph replays)."""
        eng, f = self.engine, self.engine.ac
        n, dev = data['n'], data['obs'].device
        lr, wd = self._hyper(self.optim, 5e-4)
        stats_all = []
        dp = self.grad_sync is not None
        # every pass's permutation is drawn up front, in the order tianshou draws them (one np.random.permutation per pass, nothing
        # else consumes the generator in between), so that each step's last launch can prepare the next step's minibatch
        if passes is None:
            passes = [list(split_indices(n, batch_size)) for _ in range(repeat)]
        chain = self.chain_steps and float(self._grad_norm or 0.0) > 0.0 and f.tail()[0] < 0   # the step forms that end in step_tail / dp_tail
        if perm is None:
            perm = self._upload_i32('perm', np.concatenate([c for chunks in passes for c in chunks]), dev)      # ONE upload for all passes
        flat, off = [], 0                                                    # (pass, k, idx view) of every minibatch step, in order
        for pi, chunks in enumerate(passes):
            for k, chunk in enumerate(chunks):
                flat.append((pi, k, perm[off:off + len(chunk)]))
                off += len(chunk)
        stats_flat = torch.empty(sum(len(chunks) for chunks in passes), 4, dtype=torch.float32, device=dev)      # ONE tensor: a replay's snapshot is one copy
        stats_all = list(stats_flat.split([len(chunks) for chunks in passes]))
        recompute = self._recompute_adv and repeat > 1
        # everything that does not change from step to step is converted for ctypes ONCE: the cycle is a chain of ~8 us launches and this loop's
        # host time per step (pointer tables of 28 tensors, ~35 argument conversions) must stay below the step's GPU time (tools/ppo_host_enqueue_probe.py)
        step_fn = lib().mansy_ppo_minibatch_step
        arr, garr = f.pointers(grads=True)
        fixed_head = (arr, garr, ptr(f.flat_p), ptr(f.flat_g), ptr(f.m), ptr(f.v), f.flat_p.numel(), ptr(data['obs']))
        fixed_mid = (ptr(data['act']), ptr(data['adv']), ptr(data['logp_old']), ptr(data['v_s']), ptr(data['returns']))
        # data parallel, two forms: peer-memory averages in the exchange-slot form (xg) are part of the step's own call, exactly like the
        # single-process step; a library collective (or the round-4 copy form) splits the step: raw gradients (step = 0) -> average -> mansy_ppo_dp_tail
        xg = self._xg_ctx(f) if (dp and chain) else None
        split = dp and xg is None
        hyper = (self._eps_clip, self._weight_vf, self._weight_ent, int(self._norm_adv), int(self._value_clip), float(self._dual_clip or 0.0),
                 0.0 if split else float(self._grad_norm or 0.0), lr, wd)
        ws_ptr, st_ptr = ptr(eng.workspace()), stream_ptr(dev)
        for s, (pi, k, idx) in enumerate(flat):
            first_of_later_pass = recompute and pi > 0 and k == 0
            if first_of_later_pass:
                self._recompute_returns(data)      # new advantages: the previous step must not have prepared this minibatch from the old ones
            f.step += 1
            last_of_pass = s + 1 < len(flat) and flat[s + 1][0] != pi
            nxt = flat[s + 1][2] if (chain and s + 1 < len(flat) and not (recompute and last_of_pass)) else None
            b_s = None if bias is None else bias[s]
            # (_recompute_returns rewrites data['v_s'] / ['returns'] / ['adv'] IN PLACE: the pointers converted above still hold)
            check(step_fn(*fixed_head, ptr(idx), *fixed_mid, idx.numel(), *hyper, 0 if split else f.step, *f.tail(), ptr(stats_all[pi][k]), ws_ptr, eng.max_batch,
                          int(chain and s > 0 and not first_of_later_pass), ptr(None if split else nxt), nxt.numel() if (nxt is not None and not split) else 0, xg,
                          ptr(b_s), eng.prec, st_ptr),
                  'mansy_ppo_minibatch_step')
            if split:                           # raw local gradients -> average over the ranks -> global-norm clip + Adam (+ next prologue)
                self._sync_clip_adam(f, float(self._grad_norm or 0.0), lr, wd, tail=(data, nxt) if chain else None, bias=b_s)
        if dp and bias is None:
            self._check_peers()
        out = LazyLosses(('loss', 'loss/clip', 'loss/vf', 'loss/ent'), stats_all)
        out._flat = stats_flat
        return out

    def bc_step(self, obs, act, ent_coef=0.1, train=True):
        """One behaviour-cloning step on a demonstration (utils/mansy_utils.py:60-69) or, with train=False, its validation
        cross entropy (:71-78).  obs [B,780] float32, act [B] int32 device tensors.  Returns stats [loss, ce, entropy]."""
        eng, f = self.engine, self.engine.ac
        if obs.shape[0] > eng.max_batch:
            raise MansyError(f'demonstration of {obs.shape[0]} transitions exceeds engine max_batch {eng.max_batch}')
        lr, wd = self._hyper(self.optim, 5e-4)
        stats = torch.empty(3, dtype=torch.float32, device=obs.device)
        arr, garr = f.pointers(grads=True)
        n_update = f.offsets[f.critic_head_index()]          # the critic head has no gradient: torch's Adam skips it
        if train:
            f.step += 1
            f.tail_lag += 1
        dp = train and self.grad_sync is not None
        check(lib().mansy_bc_step(arr, garr, ptr(f.flat_p), ptr(f.flat_g), ptr(f.m), ptr(f.v), f.flat_p.numel(), n_update, ptr(obs), ptr(act),
                                  obs.shape[0], float(ent_coef), lr, wd, (f.step if train else 0), ptr(stats), ptr(eng.workspace()),
                                  eng.max_batch, eng.prec, stream_ptr(obs.device)), 'mansy_bc_step')
        if dp:
            raise MansyError('behaviour cloning runs on one rank (the reference does it before training starts)')
        return stats

    def update(self, sample_size, buffer, is_train=False, batch_size=512, repeat=2, **kwargs):
        """mansy_ppo.py:36-59.  From the second call with the same buffer / batch_size / repeat on, the whole sequence (relabel, the
        evaluation passes, GAE, every minibatch step) is one hipGraph replay (`graph_update`); the host draws the permutations (same numpy
        calls in the same order), stages them together with the replay's Adam bias corrections in ONE upload, and replays."""
        if buffer is None or len(buffer) == 0:
            return {}
        self.engine.check_rollout(sync=True)
        relabel = self.args is not None and getattr(self.args, 'use_identifier', False) and is_train
        n = len(buffer)
        f = self.engine.ac
        sizes = [len(c) for c in split_indices(n, batch_size, shuffle=False)]
        n_steps = len(sizes) * repeat
        fixed = isinstance(buffer, RolloutBuffer)
        key = ('update', id(buffer), buffer.filled, buffer.N, buffer.obs.data_ptr(), int(batch_size), int(repeat), bool(relabel),
               float(getattr(self.args, 'lamb', 0.0)) if relabel else 0.0, f.flat_p.data_ptr(), self.engine.prec, self._recompute_adv,
               self.chain_steps) if fixed else None
        if fixed and self._graph_ok(n_steps) and self.chain_steps and float(self._grad_norm or 0.0) > 0.0:
            G = self._graphs.get(key)
            capture = G is None and key in self._graph_seen          # second call with this shape: capture (the first ran direct = the warm-up)
            if G is not None or capture:
                passes = [list(split_indices(n, batch_size)) for _ in range(repeat)]          # the host RNG is consumed exactly as learn() does
                if capture:
                    G = self._capture_update(key, buffer, n, passes, batch_size, repeat, relabel)
                if G is not None:
                    host = np.concatenate([np.concatenate([c for chunks in passes for c in chunks]).astype(np.int32),
                                           self._adam_bias(f.step, n_steps).reshape(-1).view(np.int32)])
                    self._stage(('update_in', key), host, G['inp'])
                    self.updating = True
                    if G['data'] is not None:      # learn-only graph (more than one rank): relabel + process_fn hold a library collective, they run direct
                        if G.pop('fresh', False):
                            pass                   # (the capturing call has just run them)
                        else:
                            if relabel:
                                self.relabel(buffer, self.args.lamb)
                            self.process_fn(buffer, out=G['data'])
                    elif relabel:
                        self.cnt += n
                    f.step += n_steps
                    G['graph'].replay()
                    self.graph_replays += 1
                    self.graph_launches += G['launches']
                    self.updating = False
                    if self.grad_sync is not None:
                        self._check_peers()
                    snap = G['stats_flat'].clone()           # (the graph's own tensor is overwritten by the next replay)
                    return LazyLosses(('loss', 'loss/clip', 'loss/vf', 'loss/ent'), list(snap.split(G['stats_sizes'])))
                # capture failed: the permutations are drawn -- run them directly
                return self._update_body(buffer, batch_size, repeat, relabel, passes=passes)
            self._graph_seen.add(key)
        return self._update_body(buffer, batch_size, repeat, relabel)

    def _update_body(self, buffer, batch_size, repeat, relabel, passes=None, perm=None, bias=None, data=None):
        if data is None:
            if relabel:
                self.relabel(buffer, self.args.lamb)
            self.updating = True
            data = self.process_fn(buffer)
        result = self.learn(data, batch_size, repeat, passes=passes, perm=perm, bias=bias)
        self.updating = False
        return result

    def _capture_update(self, key, buffer, n, passes, batch_size, repeat, relabel):
        dev = buffer.obs.device
        n_perm = sum(len(c) for chunks in passes for c in chunks)
        n_steps = sum(len(chunks) for chunks in passes)
        inp = torch.zeros(n_perm + 2 * n_steps, dtype=torch.int32, device=dev)
        bias = inp[n_perm:].view(torch.float32).view(n_steps, 2)
        f = self.engine.ac
        step0, cnt0, pre = f.step, self.cnt, self._pre_eval
        # more than one rank: process_fn gathers the return statistics with a library collective (dist.global_running_moments).  Unless library
        # collectives in graphs were asked for (graph_update = True), only learn() -- the 16 dependent steps, 90 % of the launches -- is captured;
        # relabel + process_fn run direct into PERSISTENT output tensors the graph reads.
        data = None
        if self.world > 1 and self.graph_update is not True:
            if relabel:
                self.relabel(buffer, self.args.lamb)          # (real work: the replay below must not repeat it)
            data = self.process_fn(buffer)
            data = {k: (v.clone() if torch.is_tensor(v) and k in ('v_s', 'v_next', 'logp_old', 'returns', 'adv') else v) for k, v in data.items()}
            cnt0 = self.cnt
        self._pre_eval = None                      # values evaluated ahead of time belong to one buffer state: a replay recomputes them
        try:
            g, res, nl = self._capture(dev, lambda: self._update_body(buffer, batch_size, repeat, relabel, passes=passes, perm=inp[:n_perm], bias=bias,
                                                                      data=data))
        except Exception as e:                      # capture unsupported for this form: direct launches from now on
            import warnings
            warnings.warn(f'hipGraph capture of the PPO update failed ({e}); using direct launches')
            self.graph_update = False
            f.step, self.cnt, self._pre_eval, self.updating = step0, cnt0, pre, False
            return None
        f.step, self.cnt = step0, cnt0             # the capture executed nothing: the replay that follows takes these steps
        G = self._graphs[key] = dict(graph=g, inp=inp, stats_flat=res._flat, stats_sizes=[int(t.shape[0]) for t in res._pending], launches=nl, data=data,
                                     fresh=data is not None)
        return G
