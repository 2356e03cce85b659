"""Drop-in network classes of the bitrate-selection path (reference: bitrate_selection/models/mansy.py:5-155) --
same constructor signatures, `forward` conventions and state_dict keys -- evaluated by libmansy_hip.so.

Observations may be given as the reference's dict of numpy arrays (tianshou Batch style, batched or un-batched) or,
on the fast path, as a device tensor [B, 780] in the layout of the vectorised environment (OBS_SLICES).
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from ... import _lib
from ..._lib import MansyError, check, lib, ptr, stream_ptr
from ..envs.mansy_env import OBS_LD, OBS_SLICES

HID = 128
MAXOUT = 16


def obs_to_tensor(obs, device):
    """dict of numpy arrays (mansy.py:27-36 keys; batched or not) or tensor -> contiguous cuda tensor [B, 780]."""
    if torch.is_tensor(obs):
        t = obs if obs.dim() == 2 else obs.reshape(1, -1)
        if not t.is_cuda:
            t = t.to(device)
        return t.contiguous().float()
    first = np.asarray(obs['throughput'])
    batched = first.ndim == 3
    B = first.shape[0] if batched else 1
    rows = np.zeros((B, OBS_LD), np.float32)
    for k, (a, b, shape) in OBS_SLICES.items():
        try:
            v = obs[k]
        except (KeyError, IndexError):
            continue
        rows[:, a:b] = np.asarray(v, np.float32).reshape(B, -1)
    return torch.from_numpy(rows).to(device)


class _Flat:
    """Packs the parameters of some modules into one flat fp32 buffer (+ grad / Adam state) and keeps the
    nn.Parameters as views; `pointers()` yields the ctypes arrays the engine wants (engine order)."""

    def __init__(self, kind):
        L = lib()
        self.kind = kind          # 0 actor-critic, 1 identifier, 2 A2C baseline (simple_rl) actor-critic
        n = L.mansy_a2c_num_params() if kind == 2 else L.mansy_net_num_params(kind)
        self.table = []
        for i in range(n):
            buf = ctypes.create_string_buffer(160)
            numel, nd, shape = ctypes.c_longlong(), ctypes.c_int(), (ctypes.c_longlong * 4)()
            if kind == 2:
                check(L.mansy_a2c_param_info(i, buf, 160, ctypes.byref(numel), ctypes.byref(nd), shape), 'mansy_a2c_param_info')
            else:
                check(L.mansy_net_param_info(kind, i, buf, 160, ctypes.byref(numel), ctypes.byref(nd), shape), 'mansy_net_param_info')
            self.table.append((buf.value.decode(), tuple(shape[:nd.value])))
        self.params = None
        self.flat_p = self.flat_g = self.m = self.v = None
        self.step = 0
        self.tail_lag = 0          # behaviour-cloning steps the critic head did not take part in (per-parameter Adam step)

    def critic_head_index(self):
        """Index of the first critic-head tensor (the tail of the actor-critic flat buffer)."""
        return next(i for i, (name, _) in enumerate(self.table) if name.startswith('critic.') and '.feature_net.' not in name)

    def tail(self):
        """(tail_from, tail_step) of mansy_ppo_minibatch_step / mansy_clip_grad_adam."""
        if self.kind != 0 or self.tail_lag == 0:
            return -1, 0
        return self.offsets[self.critic_head_index()], self.step - self.tail_lag

    def attach(self, params):
        """params: list of nn.Parameter in table order."""
        self.params = params
        self.flatten()

    def flatten(self):
        dev = self.params[0].device
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, offs):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1).float())
            p.data = flat[o:o + p.numel()].view(p.shape)
            p.grad = None
        self.flat_p, self.offsets = flat, offs
        self.flat_g = torch.zeros_like(flat)
        if self.m is None or self.m.numel() != total or self.m.device != dev:
            self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)

    def is_flat(self):
        if self.flat_p is None:
            return False
        base = self.flat_p.data_ptr()
        return all(p.data_ptr() == base + 4 * o and p.device == self.flat_p.device for p, o in zip(self.params, self.offsets))

    def pointers(self, grads=False):
        if not self.is_flat():
            self.flatten()
        n = len(self.params)
        arr = (ctypes.c_void_p * n)(*[p.data_ptr() for p in self.params])
        garr = None
        if grads:
            base = self.flat_g.data_ptr()
            garr = (ctypes.c_void_p * n)(*[base + 4 * o for o in self.offsets])
        return arr, garr


def _conv_as_linear_init(cin, k, hidden):
    m = nn.Conv1d(cin, hidden, k)
    return m.weight.detach().clone(), m.bias.detach().clone()


class _Seq0(nn.Module):
    """Holds `0.weight` / `0.bias` like nn.Sequential(layer, activation, ...) does in the reference.  A layer that is an
    nn.Linear in the reference is kept as a real nn.Linear child (pass the module), so that the reference's initialisation loop
    `for m in net.modules(): if isinstance(m, nn.Linear): orthogonal_ / zeros_` (run_mansy.py:205-226) reaches exactly the
    layers it reaches there; the Conv1d layers (not touched by that loop) are plain holders of the flattened weight."""

    def __init__(self, w, b=None):
        super().__init__()
        if isinstance(w, nn.Linear):
            holder = w
        else:
            holder = nn.Module()
            holder.weight = nn.Parameter(w)
            holder.bias = nn.Parameter(b)
        self.add_module('0', holder)

    @property
    def weight(self):
        return self._modules['0'].weight

    @property
    def bias(self):
        return self._modules['0'].bias


def orthogonal_init(*nets):
    """The reference's initialisation loop (run_mansy.py:209-213, 218-222; run_simple_rl.py:182-186):
        for m in model.modules(): if isinstance(m, nn.Linear): orthogonal_(m.weight, gain=sqrt(2)); zeros_(m.bias)
    where `model` is tianshou's ActorCritic(actor, critic) -- ONE container, so modules() visits the feature net the two heads
    share once (two separate traversals would re-draw it and shift the RNG stream).  Pass (actor, critic) or (identifier,)."""
    seen = nn.ModuleList(list(nets))
    for m in seen.modules():
        if isinstance(m, nn.Linear):
            nn.init.orthogonal_(m.weight, gain=np.sqrt(2))
            nn.init.zeros_(m.bias)


class FeatureNet(nn.Module):
    """mansy.py:5-51 (parameter container; evaluated inside the engine as one block-diagonal MFMA product)."""
    IDENTIFIER = False

    def __init__(self, pask_k, tile_total_num, num_rates, hidden_dim=128, device='cuda'):
        super().__init__()
        if (pask_k, tile_total_num, num_rates, hidden_dim) != (8, 64, 5, 128):
            raise MansyError('the HIP FeatureNet is built for past_k=8, 64 tiles, 5 rates, hidden 128 (config.yml)')
        self.past_k, self.tile_total_num, self.num_rates, self.hidden_dim, self.device = pask_k, tile_total_num, num_rates, hidden_dim, device
        for name, cin, k in (('conv1d1', 1, pask_k), ('conv1d2', num_rates, tile_total_num), ('conv1d3', num_rates, tile_total_num),
                             ('conv1d4', 1, tile_total_num), ('conv1d5', 1, pask_k), ('conv1d6', 1, pask_k), ('conv1d7', 1, pask_k),
                             ('conv1d8', 1, pask_k)):
            setattr(self, name, _Seq0(*_conv_as_linear_init(cin, k, hidden_dim)))
        self.fc1 = _Seq0(nn.Linear(1, hidden_dim))
        self.fc2 = _Seq0(nn.Linear(self._last_in(), hidden_dim))

    def _last_in(self):
        return 3

    def ordered_parameters(self):
        out = []
        for n in ('conv1d1', 'conv1d2', 'conv1d3', 'conv1d4', 'conv1d5', 'conv1d6', 'conv1d7', 'conv1d8', 'fc1', 'fc2'):
            s = getattr(self, n)
            out += [s.weight, s.bias]
        return out


class QoEIdentifierFeatureNet(FeatureNet):
    """mansy.py:83-140."""
    IDENTIFIER = True

    def __init__(self, pask_k, tile_total_num, num_rates, action_space, hidden_dim=128, device='cuda'):
        self._action_space = action_space
        if action_space != 15:
            raise MansyError('the HIP identifier is built for action_space=15 (config.yml)')
        super().__init__(pask_k, tile_total_num, num_rates, hidden_dim, device)

    def _last_in(self):
        return self._action_space


class _Head(nn.Module):
    def __init__(self, feature_net, feature_dim, hidden_dim, n_out, device):
        super().__init__()
        self.feature_net = feature_net
        self.feature_dim = feature_dim
        self.fc = _Seq0(nn.Linear(feature_dim, hidden_dim))
        self.out = nn.Linear(hidden_dim, n_out)
        self.device = device
        self._engine = None

    def head_parameters(self):
        return [self.fc.weight, self.fc.bias, self.out.weight, self.out.bias]


class Actor(_Head):
    """mansy.py:54-66: forward(obs, state=None, info={}) -> (logits [B,15], state)."""

    def __init__(self, feature_net, feature_dim, hidden_dim, action_space, device):
        super().__init__(feature_net, feature_dim, hidden_dim, action_space, device)

    def forward(self, batch, state=None, info={}):
        eng = _engine_of(self)
        logits, _ = eng.policy_forward(obs_to_tensor(batch, eng.device), want_value=False)
        return logits, state


class Critic(_Head):
    """mansy.py:69-80: forward(obs) -> [B,1]."""

    def __init__(self, feature_net, feature_dim, hidden_dim, device):
        super().__init__(feature_net, feature_dim, hidden_dim, 1, device)

    def forward(self, batch, state=None, info={}):
        eng = _engine_of(self)
        _, value = eng.policy_forward(obs_to_tensor(batch, eng.device), want_value=True)
        return value.reshape(-1, 1)


class QoEIdentifier(_Head):
    """mansy.py:143-155: forward(observation, action_one_hot) -> sigmoid outputs [B,3]."""

    def __init__(self, feature_net, feature_dim, hidden_dim, device):
        super().__init__(feature_net, feature_dim, hidden_dim, 3, device)

    def forward(self, observation, action_one_hot=None):
        eng = _engine_of(self)
        obs = obs_to_tensor(observation, eng.device)
        if action_one_hot is not None and not torch.is_tensor(observation):
            a, b, _ = OBS_SLICES['action_one_hot']
            obs[:, a:b] = torch.as_tensor(np.asarray(action_one_hot, np.float32).reshape(obs.shape[0], -1), device=obs.device)
        return eng.identifier_forward(obs)


class NetEngine:
    """Owns the flat parameter buffers + workspace of one actor/critic pair and (optionally) one identifier and exposes the
    engine calls on device tensors.  Created lazily by the first forward of any of its modules, or explicitly by PPOPolicy."""

    def __init__(self, actor=None, critic=None, identifier=None, max_batch=4096):
        self.actor, self.critic, self.identifier = actor, critic, identifier
        self.max_batch = max_batch
        self.ac = self.idn = None
        self._ws = None
        # precision of this engine's dense products: 'f32' / 'bf16' / 'bf16x3' / 'bf16x6', passed with every call (the `precision` argument of
        # the PPO entry points); None = the calling thread's host-side default (_lib.current_precision(), 'f32' unless changed)
        self.precision = None
        for m in (actor, critic, identifier):
            if m is not None:
                m._engine = self
        self._bind()

    @property
    def prec(self):
        return _lib.resolve_precision(self.precision)

    @property
    def device(self):
        m = self.actor or self.critic or self.identifier
        return m.fc.weight.device

    def _bind(self):
        if self.actor is not None or self.critic is not None:
            if self.actor is None or self.critic is None:
                # lone actor / critic: pair it with a private twin so the engine's 28-tensor layout is complete
                twin_src = self.actor or self.critic
                if self.actor is None:
                    self.actor = Actor(twin_src.feature_net, 1280, HID, 15, twin_src.device).to(twin_src.fc.weight.device)
                else:
                    self.critic = Critic(twin_src.feature_net, 1280, HID, twin_src.device).to(twin_src.fc.weight.device)
            if self.actor.feature_net is not self.critic.feature_net:
                raise MansyError('actor and critic must share one FeatureNet instance (run_mansy.py:207-209)')
            self.ac = _Flat(0)
            self.ac.attach(self.actor.feature_net.ordered_parameters() + self.actor.head_parameters() + self.critic.head_parameters())
        if self.identifier is not None:
            self.idn = _Flat(1)
            self.idn.attach(self.identifier.feature_net.ordered_parameters() + self.identifier.head_parameters())

    def workspace(self):
        dev = self.device
        if dev.type != 'cuda':
            raise MansyError('the bitrate-selection networks run on the HIP engine only: move the modules to a cuda (ROCm) device')
        if self._ws is None or self._ws.device != dev:
            nbytes = lib().mansy_ppo_workspace_bytes(self.max_batch)
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return self._ws

    # ---- forward paths (no autograd: gradients are produced by the fused update calls) -------------------------
    def policy_forward(self, obs, want_value=True, sample=False, u=None, seed=0, site=0):
        B = obs.shape[0]
        out = []
        for s in range(0, B, self.max_batch):
            out.append(self._policy_forward(obs[s:s + self.max_batch], want_value, sample, None if u is None else u[s:s + self.max_batch], seed, site + s))
        if len(out) == 1:
            return out[0]
        return tuple(None if out[0][i] is None else torch.cat([o[i] for o in out]) for i in range(len(out[0])))

    def _policy_forward(self, obs, want_value, sample, u, seed, site, out=None, reuse_packed=False):
        """out: optional dict of preallocated tensors (logits [B,16], act, logp) -- the rollout writes straight into its slabs."""
        B = obs.shape[0]
        dev = obs.device
        arr, _ = self.ac.pointers()
        out = out or {}
        logits = out.get('logits')
        if logits is None:
            logits = torch.empty(B, MAXOUT, dtype=torch.float32, device=dev)
        value = torch.empty(B, dtype=torch.float32, device=dev) if want_value else None
        act = out.get('act') if sample else None
        logp = out.get('logp') if sample else None
        if sample and act is None:
            act = torch.empty(B, dtype=torch.int32, device=dev)
        if sample and logp is None:
            logp = torch.empty(B, dtype=torch.float32, device=dev)
        check(lib().mansy_policy_forward(arr, ptr(obs), B, ptr(logits), ptr(value), ptr(act), ptr(logp), ptr(u), seed, site, int(reuse_packed),
                                         ptr(self.workspace()), self.max_batch, self.prec, stream_ptr(dev)), 'mansy_policy_forward')
        if sample:
            return logits[:, :15], value, act, logp
        return logits[:, :15], value

    def policy_env_step(self, venv, obs, u, act, logp, obs_out, obs_next_out, reward_out, done_out, reuse_packed=False, seed=0, site=0):
        """One rollout step in one engine call: sample actions for the observation rows `obs` [N, 780] (N = venv.n_env) and step
        every environment with its action inside the same launch (`mansy_policy_env_step`).  Writes act / logp, the post-action
        observations (obs_next_out), the next policy input with auto-reset applied (obs_out), rewards and done flags."""
        import ctypes
        N, dev = venv.n_env, obs.device
        if obs.shape[0] != N or N > self.max_batch:
            raise MansyError(f'policy_env_step: {obs.shape[0]} observation rows for {N} environments (max_batch {self.max_batch})')
        arr, _ = self.ac.pointers()
        check(lib().mansy_policy_env_step(arr, ptr(obs), N, None, ptr(act), ptr(logp), ptr(u), seed, site, int(reuse_packed), ptr(self.workspace()),
                                          self.max_batch, ctypes.byref(venv.tables.c), ptr(venv.state), ptr(obs_next_out), ptr(obs_out), ptr(reward_out),
                                          ptr(done_out), ptr(venv.qoe_parts), ctypes.byref(venv._elog), self.prec, stream_ptr(dev)), 'mansy_policy_env_step')

    def policy_rollout(self, venv, T, obs_slab, u, act, logp, obs_next_slab, carry, reward_out, done_out, reuse_packed=False):
        """A whole collect of T vector steps as ONE persistent launch on XCD teams (`mansy_policy_rollout`, csrc/ppo_engine.hip): block 0 of
        `obs_slab` [T][N][780] holds the starting observations, the call fills the rest like T calls of policy_env_step would (bit-identical).
        -> True, or False where the library says the form does not apply (another precision than fp32, a batch outside the wave-split-K range,
        launch recorder on): the caller then takes the per-step path.  A launch that gave up (a workgroup never became resident within 2 s: the
        device is shared) has raised its host-mapped word: `check_rollout()` -- called by the policy before it trains on the buffer -- raises."""
        import ctypes
        N, dev = venv.n_env, obs_slab.device
        if getattr(self, '_rollout_ctl', None) is None or self._rollout_ctl.device != dev:
            self._rollout_ctl = torch.zeros(1024, dtype=torch.uint8, device=dev)          # MANSY_ROLLOUT_CTL_BYTES
            self._rollout_err = torch.zeros(16, dtype=torch.int32).pin_memory()            # host-mapped: the kernel's give-up word
        if int(self._rollout_err[0]) != 0:
            raise MansyError('a persistent rollout launch gave up waiting for its workgroups (is the device shared?): its outputs were garbage')
        if N > self.max_batch:
            raise MansyError(f'policy_rollout: {N} environments exceed max_batch {self.max_batch}')
        arr, _ = self.ac.pointers()
        rc = lib().mansy_policy_rollout(arr, ptr(obs_slab), N, int(T), ptr(u), ptr(act), ptr(logp), ptr(obs_next_slab), ptr(carry), ptr(reward_out),
                                        ptr(done_out), ptr(venv.qoe_parts), ctypes.byref(venv.tables.c), ptr(venv.state), ctypes.byref(venv._elog),
                                        int(reuse_packed), ptr(self._rollout_ctl), ctypes.c_void_p(self._rollout_err.data_ptr()), ptr(self.workspace()),
                                        self.max_batch, self.prec, stream_ptr(dev))
        if rc != 0:
            msg = lib().mansy_last_error() or b''
            if b'rollout_team' in msg:
                return False
            check(rc, 'mansy_policy_rollout')
        return True

    def check_rollout(self, sync=False):
        """Verdict on the persistent rollout launches issued so far: a launch that gave up left partial / garbage slabs behind.  The
        policy calls it at the top of train_identifier() / process_fn() with sync=True when the team form is in use, i.e. BEFORE the buffer is
        trained on (the word is host-mapped: without the sync it covers completed launches only)."""
        err = getattr(self, '_rollout_err', None)
        if err is None:
            return
        if sync:
            torch.cuda.synchronize(self.device)
        if int(err[0]) != 0:
            raise MansyError('a persistent rollout launch gave up waiting for its workgroups (is the device shared?): its outputs were garbage')

    def identifier_forward(self, obs):
        B, dev = obs.shape[0], obs.device
        outs = []
        arr, _ = self.idn.pointers()
        for s in range(0, B, self.max_batch):
            o = obs[s:s + self.max_batch]
            pred = torch.empty(o.shape[0], MAXOUT, dtype=torch.float32, device=dev)
            check(lib().mansy_identifier_forward(arr, ptr(o), o.shape[0], ptr(pred), ptr(self.workspace()), self.max_batch, self.prec, stream_ptr(dev)),
                  'mansy_identifier_forward')
            outs.append(pred[:, :3])
        return outs[0] if len(outs) == 1 else torch.cat(outs)


def _engine_of(module):
    if module._engine is None:
        if isinstance(module, QoEIdentifier):
            NetEngine(identifier=module)
        elif isinstance(module, Actor):
            NetEngine(actor=module)
        else:
            NetEngine(critic=module)
    eng = module._engine
    for f in (eng.ac, eng.idn):
        if f is not None and not f.is_flat():
            f.flatten()
    return eng
