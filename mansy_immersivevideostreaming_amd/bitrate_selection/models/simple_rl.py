"""A2C baseline ("simple RL") on the HIP engine (csrc/a2c_engine.hip).

* `FeatureNet`, `Actor`, `Critic` -- parameter containers with the reference's constructor signatures and state_dict keys
  (bitrate_selection/models/simple_rl.py:9-63); `Actor.forward` returns softmax outputs as "logits" like the reference.
* `A2CPolicy` -- tianshou==0.4.8 A2CPolicy as run_simple_rl.py:195-209 constructs it (T2, restated; parity unpinned -- see
  oracle/a2c_oracle.py): `forward(batch) -> logits (= probabilities), act, dist`, `update(sample_size, buffer, batch_size,
  repeat)` = returns / GAE (`mansy_gae_returns`) + `repeat` passes of shuffled minibatches through
  `mansy_a2c_minibatch_step` (loss, backward, clip_grad_norm_, RMSprop in one call).
* `A2CBuffer`, `A2CCollector` -- step-major device slabs and the vectorised rollout that replace tianshou's
  VectorReplayBuffer / Collector for this path.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from ... import _lib
from ..._lib import MansyError, check, lib, ptr, stream_ptr
from .mansy import MAXOUT, _Flat, _Seq0, _conv_as_linear_init
from .mansy_ppo import LazyLosses, _ActorCritic, _Result, split_indices

LD = 416
SLICES = {'throughput': (0, 8, (1, 8)), 'chunk_sizes': (8, 328, (5, 64)), 'rebuffer': (328, 329, (1,)), 'last_bitrates': (329, 331, (2,)),
          'pred_viewport': (331, 395, (64,))}


def obs_to_tensor(obs, device):
    """dict of numpy arrays (un-batched like SimpleRLEnv.state, or with a leading batch axis) / [B,416] tensor -> device rows."""
    if torch.is_tensor(obs):
        return obs.to(device=device, dtype=torch.float32).reshape(-1, LD).contiguous()
    get = obs.__getitem__ if hasattr(obs, '__getitem__') else lambda k: getattr(obs, k)
    pv = np.asarray(get('pred_viewport'), np.float32)
    B = 1 if pv.ndim == 1 else pv.shape[0]
    rows = np.zeros((B, LD), np.float32)
    for k, (a, b, _) in SLICES.items():
        rows[:, a:b] = np.asarray(get(k), np.float32).reshape(B, -1)
    return torch.from_numpy(rows).to(device)


def obs_to_dict(row):
    """One row (numpy [416]) -> SimpleRLEnv's state dict (simple_rl_env.py:112-118)."""
    return {k: np.array(row[a:b], dtype=np.float32).reshape(shape) for k, (a, b, shape) in SLICES.items()}


class FeatureNet(nn.Module):
    """simple_rl.py:9-36 (evaluated inside the engine as one block-diagonal MFMA product)."""

    def __init__(self, pask_k, tile_total_num, num_rates, device='cuda'):
        super().__init__()
        if (pask_k, tile_total_num, num_rates) != (8, 64, 5):
            raise MansyError('the HIP FeatureNet is built for past_k=8, 64 tiles, 5 rates (config.yml)')
        self.device = device
        self.conv1d_1 = _Seq0(*_conv_as_linear_init(1, pask_k, 128))
        self.conv1d_2 = _Seq0(*_conv_as_linear_init(1, tile_total_num * num_rates, 128))
        for name, nin in (('fc1', 1), ('fc2', 2), ('fc3', 64)):
            setattr(self, name, _Seq0(nn.Linear(nin, 128)))       # a real nn.Linear: reached by run_simple_rl's orthogonal init loop

    def ordered_parameters(self):
        out = []
        for n in ('conv1d_1', 'conv1d_2', 'fc1', 'fc2', 'fc3'):
            s = getattr(self, n)
            out += [s.weight, s.bias]
        return out


class _Head(nn.Module):
    def __init__(self, feature_net, feature_dim, n_out, device):
        super().__init__()
        if feature_dim != 640:
            raise MansyError('feature_dim must be 5 * 128 (run_simple_rl.py:185-186)')
        self.feature_net = feature_net
        self.fc = _Seq0(nn.Linear(feature_dim, 128))
        self.out = nn.Linear(128, n_out)
        self.device = device
        self._engine = None

    def head_parameters(self):
        return [self.fc.weight, self.fc.bias, self.out.weight, self.out.bias]


class Actor(_Head):
    """simple_rl.py:39-50: forward(batch, state=None, info={}) -> (softmax probabilities [B,15], state)."""

    def __init__(self, feature_net, feature_dim, action_space, device):
        if action_space != 15:
            raise MansyError('the HIP actor is built for action_space=15 (config.yml)')
        super().__init__(feature_net, feature_dim, action_space, device)

    def forward(self, batch, state=None, info={}):
        eng = _engine_of(self)
        probs, _ = eng.forward(obs_to_tensor(batch, eng.device), want_value=False)
        return probs, state


class Critic(_Head):
    """simple_rl.py:53-63: forward(batch) -> [B,1]."""

    def __init__(self, feature_net, feature_dim, device):
        super().__init__(feature_net, feature_dim, 1, device)

    def forward(self, batch, state=None, info={}):
        eng = _engine_of(self)
        _, value = eng.forward(obs_to_tensor(batch, eng.device), want_value=True)
        return value.reshape(-1, 1)


class SimpleEngine:
    """Flat parameter / gradient / RMSprop buffers + workspace of one actor-critic pair; thin wrappers over the C ABI."""

    def __init__(self, actor, critic, max_batch=4096):
        if actor.feature_net is not critic.feature_net:
            raise MansyError('actor and critic must share one FeatureNet instance (run_simple_rl.py:184-186)')
        self.actor, self.critic, self.max_batch = actor, critic, max_batch
        actor._engine = critic._engine = self
        self.f = _Flat(2)
        self.f.attach(actor.feature_net.ordered_parameters() + actor.head_parameters() + critic.head_parameters())
        self._ws = None
        self.precision = None       # 'f32' / 'bf16' / 'bf16x3' / 'bf16x6' (the `precision` argument of every call); None = the thread's host-side default

    @property
    def prec(self):
        return _lib.resolve_precision(self.precision)

    @property
    def device(self):
        return self.actor.fc.weight.device

    def workspace(self):
        dev = self.device
        if dev.type != 'cuda':
            raise MansyError('the A2C networks run on the HIP engine only: move the modules to a cuda (ROCm) device')
        if self._ws is None or self._ws.device != dev:
            self._ws = torch.empty(lib().mansy_a2c_workspace_bytes(self.max_batch), dtype=torch.uint8, device=dev)
        return self._ws

    def forward(self, obs, want_value=True, sample=False, u=None, seed=0, site=0, out=None, reuse_packed=False):
        B, dev = obs.shape[0], obs.device
        if B > self.max_batch:
            parts = [self.forward(obs[s:s + self.max_batch], want_value, sample, None if u is None else u[s:s + self.max_batch], seed, site + s)
                     for s in range(0, B, self.max_batch)]
            return tuple(None if parts[0][i] is None else torch.cat([p[i] for p in parts]) for i in range(len(parts[0])))
        arr, _ = self.f.pointers()
        out = out or {}
        probs = out.get('probs')
        if probs is None:
            probs = torch.empty(B, MAXOUT, dtype=torch.float32, device=dev)
        value = torch.empty(B, dtype=torch.float32, device=dev) if want_value else None
        act = logp = None
        if sample:
            act = out.get('act') if out.get('act') is not None else torch.empty(B, dtype=torch.int32, device=dev)
            logp = out.get('logp') if out.get('logp') is not None else torch.empty(B, dtype=torch.float32, device=dev)
        check(lib().mansy_a2c_forward(arr, ptr(obs), B, ptr(probs), ptr(value), ptr(act), ptr(logp), ptr(u), seed, site, int(reuse_packed),
                                      ptr(self.workspace()), self.max_batch, self.prec, stream_ptr(dev)), 'mansy_a2c_forward')
        return (probs[:, :15], value, act, logp) if sample else (probs[:, :15], value)


def _engine_of(module):
    if module._engine is None:
        raise MansyError('build the actor-critic pair first: A2CPolicy(actor, critic, ...) or SimpleEngine(actor, critic)')
    if not module._engine.f.is_flat():
        module._engine.f.flatten()
    return module._engine


class A2CBuffer:
    """[T][N] step-major slabs on the device."""

    def __init__(self, T, N, device):
        self.T, self.N = T, N
        f32 = dict(dtype=torch.float32, device=device)
        self.obs = torch.zeros(T, N, LD, **f32)
        self.obs_next = torch.zeros(T, N, LD, **f32)
        self.act = torch.zeros(T, N, dtype=torch.int32, device=device)
        self.rew = torch.zeros(T, N, **f32)
        self.done = torch.zeros(T, N, dtype=torch.uint8, device=device)
        self.filled = 0

    def __len__(self):
        return self.filled * self.N

    def reset(self):
        self.filled = 0


class A2CCollector:
    """Steps N device environments (SimpleRLVecEnv) with the policy: per vector step one policy call (FeatureNet product, fc
    product, fused output layers + sampling) and the environment step + two observation-row kernels, all on the device."""

    def __init__(self, policy, venv, seed=0):
        self.policy, self.venv, self.carry = policy, venv, None
        self.env_step = 0

    def reset_env(self):
        self.carry = self.venv.reset().clone()

    def collect(self, n_step, buffer):
        N = self.venv.n_env
        T = max(1, n_step // N)
        if buffer.T < T or buffer.N != N:
            raise MansyError('rollout buffer too small')
        if self.carry is None:
            self.reset_env()
        eng = self.policy.engine
        u = torch.rand(T, N, device=self.carry.device)
        buffer.reset()
        buffer.obs[0].copy_(self.carry)
        for t in range(T):
            eng.forward(buffer.obs[t], want_value=False, sample=True, u=u[t], out={'act': buffer.act[t]}, reuse_packed=t > 0)
            nxt = buffer.obs[t + 1] if t + 1 < T else self.carry
            self.venv.step(buffer.act[t], obs_out=nxt, obs_next_out=buffer.obs_next[t], reward_out=buffer.rew[t], done_out=buffer.done[t])
        buffer.filled = T
        self.env_step += T * N
        return {'n/st': T * N}


class A2CPolicy(nn.Module):
    """tianshou A2CPolicy with the keyword arguments of run_simple_rl.py:195-209.  `optim` is read for lr / alpha / eps
    (torch.optim.RMSprop); the fused kernel does the step."""

    def __init__(self, actor, critic, optim, dist_fn, vf_coef=0.5, ent_coef=0.01, max_grad_norm=None, gae_lambda=0.95, max_batchsize=256,
                 discount_factor=0.99, reward_normalization=False, action_scaling=True, action_bound_method='clip', action_space=None, **kwargs):
        super().__init__()
        self.actor, self.critic = actor, critic
        self._actor_critic = _ActorCritic(actor, critic)
        self.optim, self.dist_fn = optim, dist_fn
        self._weight_vf, self._weight_ent, self._grad_norm = vf_coef, ent_coef, max_grad_norm
        self._gamma, self._lambda, self._rew_norm = discount_factor, gae_lambda, bool(reward_normalization)
        self.engine = SimpleEngine(actor, critic, max_batch=4096)
        self._rms = None
        self._seed_ctr = 0
        self.world, self.grad_sync = 1, None
        self.updating = False

    def set_data_parallel(self, world, grad_sync):
        self.world, self.grad_sync = int(world), grad_sync

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.engine.f.flatten()
        self._sq = None
        self._rms = None
        return out

    def _hyper(self):
        if self.optim is None:
            return 1e-4, 0.99, 1e-8
        g = self.optim.param_groups[0]
        return g['lr'], g.get('alpha', 0.99), g.get('eps', 1e-8)

    def square_avg(self):
        f = self.engine.f
        if getattr(self, '_sq', None) is None or self._sq.numel() != f.flat_p.numel() or self._sq.device != f.flat_p.device:
            self._sq = torch.zeros_like(f.flat_p)
        return self._sq

    def ret_rms(self):
        dev = self.engine.device
        if self._rms is None or self._rms.device != dev:
            self._rms = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64, device=dev)
        return self._rms

    def forward(self, batch, state=None, **kwargs):
        """PGPolicy.forward: logits (= the actor's probabilities), act ~ dist_fn(logits), dist."""
        obs = batch.obs if hasattr(batch, 'obs') else batch['obs']
        t = obs_to_tensor(obs, self.engine.device)
        self._seed_ctr += 1
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        probs, _, act, _ = self.engine.forward(t, want_value=False, sample=True, seed=seed, site=self._seed_ctr)
        return _Result(logits=probs, act=act.long(), state=state, dist=self.dist_fn(probs) if self.dist_fn else None)

    def process_fn(self, buffer):
        """T2: A2CPolicy._compute_returns: v_s, v_s_, GAE(lambda), returns normalised by the running std."""
        eng = self.engine
        T, N = buffer.filled, buffer.N
        n, dev = T * N, buffer.obs.device
        obs, obs_next = buffer.obs[:T].reshape(n, LD), buffer.obs_next[:T].reshape(n, LD)
        v_s = eng.forward(obs, want_value=True)[1]
        v_next = eng.forward(obs_next, want_value=True, reuse_packed=True)[1]
        returns = torch.empty(n, dtype=torch.float32, device=dev)
        adv = torch.empty(n, dtype=torch.float32, device=dev)
        scratch = torch.empty(n + 2, dtype=torch.float64, device=dev)
        rms_local = self.ret_rms()
        rms_use = rms_local
        if self.world > 1:
            from ...dist import global_running_moments
            rms_use = global_running_moments(rms_local, self.world)
        check(lib().mansy_gae_returns(ptr(buffer.rew[:T]), ptr(v_s), ptr(v_next), ptr(buffer.done[:T]), T, N, self._gamma, self._lambda,
                                      int(self._rew_norm), ptr(rms_use), ptr(scratch), ptr(returns), ptr(adv), stream_ptr(dev)),
              'mansy_gae_returns')
        if self.world > 1 and self._rew_norm:      # accumulate only our own (un-normalised) returns locally; no host round trip
            from ...dist import update_running_moments
            update_running_moments(rms_local, scratch[:n])
        return dict(obs=obs, act=buffer.act[:T].reshape(n), returns=returns, adv=adv, n=n)

    def _upload_perm(self, slot_id, arr, dev):
        arr = np.ascontiguousarray(arr, dtype=np.int32)
        pinned = self.__dict__.setdefault('_pinned', {})
        slot = pinned.get(slot_id)
        if slot is None or slot[0].numel() < arr.size:
            slot = [torch.empty(max(arr.size, 1), dtype=torch.int32).pin_memory(), None]
            pinned[slot_id] = slot
        if slot[1] is not None:
            slot[1].synchronize()
        slot[0][:arr.size].copy_(torch.from_numpy(arr))
        out = slot[0][:arr.size].to(dev, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(dev))
        return out

    def learn(self, data, batch_size, repeat):
        """T2: A2CPolicy.learn: `repeat` passes over shuffled minibatches (merge_last)."""
        eng, f = self.engine, self.engine.f
        n, dev = data['n'], data['obs'].device
        lr, alpha, eps = self._hyper()
        sq = self.square_avg()
        stats_all = []
        dp = self.grad_sync is not None
        for ps in range(repeat):
            chunks = list(split_indices(n, batch_size))
            # one upload per pass through a pinned staging buffer, non-blocking (a pageable .to(device) blocks the host until the copy
            # has run, so it cannot enqueue ahead -- the PPO cycle gained 6 % from the same change); minibatches are views
            perm = self._upload_perm(ps, np.concatenate(chunks), dev)
            stats_pass = torch.empty(len(chunks), 4, dtype=torch.float32, device=dev)
            off = 0
            for k, chunk in enumerate(chunks):
                idx = perm[off:off + len(chunk)]
                off += len(chunk)
                stats = stats_pass[k]
                arr, garr = f.pointers(grads=True)
                check(lib().mansy_a2c_minibatch_step(arr, garr, ptr(f.flat_p), ptr(f.flat_g), ptr(sq), f.flat_p.numel(), ptr(data['obs']), ptr(idx),
                                                     ptr(data['act']), ptr(data['adv']), ptr(data['returns']), idx.numel(), self._weight_vf,
                                                     self._weight_ent, 0.0 if dp else float(self._grad_norm or 0.0), lr, alpha, eps, 0 if dp else 1,
                                                     ptr(stats), ptr(eng.workspace()), eng.max_batch, eng.prec, stream_ptr(dev)), 'mansy_a2c_minibatch_step')
                if dp:
                    self.grad_sync(f.flat_g)
                    scratch = torch.empty(64, dtype=torch.float64, device=dev)
                    check(lib().mansy_clip_grad_rmsprop(ptr(f.flat_p), ptr(f.flat_g), ptr(sq), f.flat_p.numel(), float(self._grad_norm or 0.0), lr, alpha,
                                                        eps, ptr(scratch), stream_ptr(dev)), 'mansy_clip_grad_rmsprop')
            stats_all.append(stats_pass)
        return LazyLosses(('loss', 'loss/actor', 'loss/vf', 'loss/ent'), stats_all)

    def update(self, sample_size, buffer, batch_size=256, repeat=2, **kwargs):
        if buffer is None or len(buffer) == 0:
            return {}
        self.updating = True
        result = self.learn(self.process_fn(buffer), batch_size, repeat)
        self.updating = False
        return result
