"""Drop-in `ViewportTransformerMTIO` (reference: viewport_prediction/models/mtio.py:47-166 on top of
models/customized_transformer.py:13-83) whose arithmetic runs in libmansy_hip.so.

Same constructor signature, same methods (`forward`, `loss_function`, `sample`), same
`state_dict()` keys/shapes (both the torch<=2.0 with-bias layout and the torch>=2.1 bias-free
layout produced by the positional-argument slip at customized_transformer.py:46-49), same host
RNG consumption for the MTIO mixing (`random.random()`, two `np.random.shuffle`).

Host side is plumbing only: parameters live in ONE flat fp32 HBM buffer (views exposed as
nn.Parameters), activations in one workspace tensor, and each forward/backward/train step is a single
C-ABI call that enqueues the whole kernel sequence on the current HIP stream.
"""
import ctypes
import random
import weakref

import numpy as np
import torch
import torch.nn as nn

from ... import _lib
from ..._lib import VPConfig, check, lib, ptr, stream_ptr

_PE_MAX_LEN = 5000
_N_HEAD = 8          # nn.Transformer default; mtio.py:56-58 never passes nhead
_ATTN_DROPOUT = 0.1  # nn.Transformer default dropout


# models with SyncBN armed, by the integer that travels as mansy_vp_config::bn_sync_user (weak: a model that is dropped un-registers itself)
_BN_MODELS = weakref.WeakValueDictionary()


def _bn_sync_callback(which, user):
    """mansy_vp_config::bn_sync_fn of this model's calls (per call, no process-wide registration): called by the engine between the
    partial-sum kernel and its consumer: all-reduce the 2*d_model doubles of the DistillLayer BatchNorm statistics over the
    data-parallel ranks."""
    try:
        m = _BN_MODELS[int(user or 0)]
        cfg = m._bn_cfg
        if which == 2:              # decoder-side gradients are final: start their all-reduce under the encoder backward
            if m._grad_ready is not None:
                m._grad_ready()
            return 0
        off = ctypes.c_longlong()
        check(lib().mansy_vp_ws_lookup(ctypes.byref(cfg), b'dis.stats', ctypes.byref(off), None), 'ws_lookup')
        d = m.d_model
        stats = m._workspace(cfg)[off.value:off.value + 6 * d * 8].view(torch.float64)
        m._bn_allreduce(stats[which * 2 * d:(which * 2 + 2) * d])
        return 0
    except Exception:           # never let an exception cross the C boundary
        import traceback
        traceback.print_exc()
        return 1


_BN_SYNC_CFUNC = _lib.BN_SYNC_FN(_bn_sync_callback)


class _Node(nn.Module):
    """Name-space node so parameters get the reference's dotted state_dict keys."""


def _positional_table(max_len, d_model):
    # mtio.py:17-24, same op order
    import math
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


class ViewportTransformerMTIO(nn.Module):
    def __init__(self, in_channel, fut_window, d_model, dim_feedforward, num_head=3, num_encoder_layers=2,
                 num_decoder_layers=2, batch_first=True, dropout=0.2, device='cuda', repeat_prob=0.5, seed=1,
                 bias=True):
        super().__init__()
        if not batch_first:
            raise ValueError('only batch_first=True is supported (the reference never uses anything else)')
        if num_head != 3:
            raise ValueError('the HIP MTIO kernels are built for num_head=3 (reference default)')
        self.in_channel = in_channel
        self.num_head = num_head
        self.fut_window = fut_window
        self.d_model = d_model
        self.dim_feedforward = dim_feedforward
        self.num_encoder_layers = num_encoder_layers
        self.num_decoder_layers = num_decoder_layers
        self.dropout_p = float(dropout)
        self.attn_dropout_p = _ATTN_DROPOUT   # nn.Transformer's own dropout (attention probs + sublayer outputs)
        self.device = device
        self.repeat_prob = repeat_prob
        self.seed = seed
        self.has_bias = bool(bias)
        # precision of the dense products, carried in every call's mansy_vp_config (the library has no process-wide mode): 'f32' (exact
        # fp32 MFMA, the parity mode) / 'bf16' / 'bf16x3' / 'bf16x6' (csrc/gemm_bf16s.hip); None = the calling thread's host-side default
        # (_lib.current_precision(): 'f32' unless a `with kernels.precision(..)` block is open).  A run-time attribute, not part of the checkpoint
        self.precision = None
        self._ws = {}
        self._flat_p = None
        self._flat_g = None
        self.bn_sync_world = 1
        self._bn_allreduce = None
        self._grad_ready = None
        self.two_stream = None
        self._build_parameters()
        self._flatten()

    # ------------------------------------------------------------------ construction
    def _cfg(self, B, S, inference=False):
        # two_stream: None = where it pays (the engine applies it for even B >= 2048: sample() +4 %, train step +2 % at B = 4096; below that the
        # step is a chain of latency-bound launches and a second stream only doubles them); True = wherever the halves are whole (even
        # B >= 256); False = never.  Per-kernel timings (bench.py's roofline leg, rocprof kernel stats) are taken with it off: concurrent
        # kernels stretch each other's durations.
        two = 1 if self.two_stream is None else (2 if self.two_stream else 0)
        if self.precision is not None and self.precision not in _lib.PRECISIONS:
            raise _lib.MansyError(f'unknown precision {self.precision!r}: one of f32, bf16, bf16x3, bf16x6')
        cfg = VPConfig(B=B, S=S, T=self.fut_window, d_model=self.d_model, n_head=_N_HEAD, d_ff=self.dim_feedforward,
                       n_enc=self.num_encoder_layers, n_dec=self.num_decoder_layers, in_ch=self.in_channel * self.num_head,
                       has_bias=int(self.has_bias), p_pe=self.dropout_p, p_drop=self.attn_dropout_p, ln_eps=1e-5, bn_eps=1e-5,
                       bn_momentum=0.1, max_len=_PE_MAX_LEN, bn_sync_world=int(self.bn_sync_world), two_stream=int(two),
                       precision=_lib.resolve_precision(self.precision))
        if self.bn_sync_world > 1:              # the call's own SyncBN / gradient-ready hook + the key that finds this model again
            cfg.bn_sync_fn = _BN_SYNC_CFUNC
            cfg.bn_sync_user = id(self)
        return cfg

    def set_data_parallel(self, world, allreduce=None):
        """SyncBN for the DistillLayer under data parallelism: `world` ranks share batch statistics; `allreduce(t)` sums a
        float64 device tensor over the ranks in place (default: torch.distributed.all_reduce)."""
        self.bn_sync_world = int(world)
        if allreduce is None:
            import torch.distributed as dist
            allreduce = dist.all_reduce
        self._bn_allreduce = allreduce
        if self.bn_sync_world > 1:
            _BN_MODELS[id(self)] = self
        else:
            _BN_MODELS.pop(id(self), None)

    def _arm_bn_sync(self, cfg):
        if self.bn_sync_world > 1:
            self._bn_cfg = cfg

    def _param_table(self):
        L = lib()
        cfg = self._cfg(1, 1)
        n = L.mansy_vp_num_params(ctypes.byref(cfg))
        if n <= 0:
            check(-1, 'mansy_vp_num_params')
        out = []
        for i in range(n):
            buf = ctypes.create_string_buffer(160)
            numel, nd, shape = ctypes.c_longlong(), ctypes.c_int(), (ctypes.c_longlong * 4)()
            check(L.mansy_vp_param_info(ctypes.byref(cfg), i, buf, 160, ctypes.byref(numel), ctypes.byref(nd), shape),
                  'mansy_vp_param_info')
            out.append((buf.value.decode(), tuple(shape[:nd.value])))
        return out

    def _set_nested(self, dotted, value, buffer=False):
        parts = dotted.split('.')
        node = self
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Node())
            node = node._modules[p]
        if buffer:
            node.register_buffer(parts[-1], value)
        else:
            node.register_parameter(parts[-1], value)

    def _build_parameters(self):
        """Initial values come from the same torch constructors, called in the same order, as the reference
        (Linear -> nn.Transformer -> Conv1d/BatchNorm1d -> Linear), so torch.manual_seed(s) gives the
        reference's initial weights bit for bit.  These temporaries are discarded after the copy."""
        d, ff, c6 = self.d_model, self.dim_feedforward, self.in_channel * self.num_head
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            emb = nn.Linear(c6, d)
            tr = nn.Transformer(d_model=d, nhead=_N_HEAD, num_encoder_layers=self.num_encoder_layers,
                                num_decoder_layers=self.num_decoder_layers, dim_feedforward=ff, batch_first=True,
                                bias=self.has_bias)
            conv = nn.Conv1d(d, d, kernel_size=3, padding=1, padding_mode='circular')
            bn = nn.BatchNorm1d(d)
            pred = nn.Linear(d, c6)
        src = {'embedding.linear.' + k: v for k, v in emb.state_dict().items()}
        src.update({'transformer.' + k: v for k, v in tr.state_dict().items()})
        src.update({'transformer.distill_layer.downConv.' + k: v for k, v in conv.state_dict().items()})
        src.update({'transformer.distill_layer.norm.' + k: v for k, v in bn.state_dict().items()})
        src.update({'predictor.0.' + k: v for k, v in pred.state_dict().items()})
        self._param_names = []
        table = self._param_table()
        names = [n for n, _ in table]
        # registration order == reference state_dict order
        for name, shape in table:
            if name.startswith('predictor.'):
                continue
            t = src[name].detach().clone().float()
            assert tuple(t.shape) == shape, (name, t.shape, shape)
            self._set_nested(name, nn.Parameter(t))
            self._param_names.append(name)
            if name == 'transformer.distill_layer.norm.bias':
                self._set_nested('transformer.distill_layer.norm.running_mean', torch.zeros(d), buffer=True)
                self._set_nested('transformer.distill_layer.norm.running_var', torch.ones(d), buffer=True)
                self._set_nested('transformer.distill_layer.norm.num_batches_tracked', torch.tensor(0, dtype=torch.long), buffer=True)
        self._set_nested('positional_embedding.pe', _positional_table(_PE_MAX_LEN, d), buffer=True)
        for name, shape in table:
            if name.startswith('predictor.'):
                self._set_nested(name, nn.Parameter(src[name].detach().clone().float()))
                self._param_names.append(name)
        # engine order (== table order)
        self._engine_names = names

    def _flatten(self):
        """(Re)pack all parameters into one flat, 256-byte-aligned fp32 buffer and re-point the nn.Parameters at
        views of it.  Called after construction and after any .to()/.cuda()."""
        params = [self.get_parameter(n) for n in self._engine_names]
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(params, offs):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1).float())
            p.data = flat[o:o + p.numel()].view(p.shape)
            p.grad = None
        self._flat_p = flat
        self._flat_g = torch.zeros_like(flat)
        self._offsets = offs
        self._params = params
        self._ws = {}
        self._ptr_cache = None

    def _is_flat(self):
        f = self._flat_p
        if f is None:
            return False
        base = f.data_ptr()
        return all(p.data_ptr() == base + 4 * o and p.device == f.device for p, o in zip(self._params, self._offsets))

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flatten()
        return out

    def load_state_dict(self, state_dict, strict=True, assign=False):
        has_bias = 'transformer.encoder.layers.0.self_attn.in_proj_bias' in state_dict
        if has_bias != self.has_bias:
            # checkpoint written under the other torch-version layout: rebuild to match (SURVEY 8c version trap)
            dev = self._flat_p.device
            for n in list(self._modules):
                del self._modules[n]
            self.has_bias = has_bias
            self._build_parameters()
            self._flatten()
            self.to(dev)
        out = super().load_state_dict(state_dict, strict=strict)
        if not self._is_flat():
            self._flatten()
        return out

    # ------------------------------------------------------------------ engine plumbing
    def _require_gpu(self, t):
        if not t.is_cuda:
            raise _lib.MansyError('ViewportTransformerMTIO runs on the HIP engine only: tensors must be on a cuda (ROCm) device')

    def _pointers(self, grads_flat=None):
        if not self._is_flat():
            self._flatten()
        n = len(self._params)
        arr = (ctypes.c_void_p * n)(*[p.data_ptr() for p in self._params])
        garr = None
        if grads_flat is not None:
            base = grads_flat.data_ptr()
            garr = (ctypes.c_void_p * n)(*[base + 4 * o for o in self._offsets])
        return arr, garr

    def _workspace(self, cfg):
        key = (cfg.B, cfg.S, cfg.T)
        ws = self._ws.get(key)
        if ws is None or ws.device != self._flat_p.device:
            nbytes = lib().mansy_vp_workspace_bytes(ctypes.byref(cfg))
            if nbytes == 0:
                check(-1, 'mansy_vp_workspace_bytes')
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self._flat_p.device)
            self._ws = {key: ws}       # keep one live workspace (largest user is B=4096: ~6.5 GB)
        return ws

    def _engine_buffers(self):
        bn = self.transformer.distill_layer.norm
        return self.positional_embedding.pe, bn.running_mean, bn.running_var, bn.num_batches_tracked

    def ws_tensor(self, cfg, name):
        """Named activation slab of the last forward (parity tests)."""
        off, numel = ctypes.c_longlong(), ctypes.c_longlong()
        check(lib().mansy_vp_ws_lookup(ctypes.byref(cfg), name.encode(), ctypes.byref(off), ctypes.byref(numel)), 'ws_lookup')
        ws = self._workspace(cfg)
        return ws[off.value:off.value + 4 * numel.value].view(torch.float32)

    def _next_seed(self):
        return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())

    def _mix_decision(self, B):
        """mtio.py:77-87: host RNG calls in the reference's order."""
        if random.random() < self.repeat_prob:
            return None
        perms = []
        for _ in range(self.num_head - 1):
            indices = np.arange(B)
            np.random.shuffle(indices)
            perms.append(indices)
        return perms

    def _mix(self, x, perms):
        B, L, c = x.shape
        out = torch.empty(B, L, 3 * c, dtype=torch.float32, device=x.device)
        p1 = p2 = None
        if perms is not None:
            p1, p2 = perms
        check(lib().mansy_mtio_mix(ptr(x), ptr(p1), ptr(p2), ptr(out), B, L, c, stream_ptr(x.device)), 'mansy_mtio_mix')
        return out

    def _perms_to_device(self, perms, device):
        """The MTIO row permutations of heads 2 and 3 -> device int32, through a small ring of PINNED staging buffers and ONE
        non-blocking copy (a pageable `.to(device)` blocks the host until the copy has run, i.e. until the GPU has drained
        everything enqueued before it: the host then cannot enqueue the step ahead of time)."""
        if perms is None:
            return None
        B = len(perms[0])
        ring = getattr(self, '_perm_ring', None)
        if ring is None or ring['buf'][0].numel() < 2 * B:
            ring = {'buf': [torch.empty(2 * B, dtype=torch.int32).pin_memory() for _ in range(4)], 'ev': [None] * 4, 'i': 0}
            self._perm_ring = ring
        i = ring['i']
        ring['i'] = (i + 1) % 4
        if ring['ev'][i] is not None:
            ring['ev'][i].synchronize()                      # the copy that last used this slot (four uploads ago) has run
        host = ring['buf'][i][:2 * B]
        host[:B].copy_(torch.from_numpy(np.ascontiguousarray(perms[0], dtype=np.int32)))
        host[B:].copy_(torch.from_numpy(np.ascontiguousarray(perms[1], dtype=np.int32)))
        dev = host.to(device, non_blocking=True)
        ring['ev'][i] = torch.cuda.Event()
        ring['ev'][i].record(torch.cuda.current_stream(device))
        return [dev[:B], dev[B:]]

    # ------------------------------------------------------------------ reference API
    def forward(self, history, current, future):
        """mtio.py:65-92 -> (pred [B,T,6], multi_future [B,T,6])."""
        self._require_gpu(history)
        history, current, future = (t.contiguous().float() for t in (history, current, future))
        perms = self._perms_to_device(self._mix_decision(history.shape[0]), history.device)
        src6 = self._mix(history, perms)
        cur6 = self._mix(current, perms)
        fut6 = self._mix(future, perms)
        pred = self._process_src_current(src6, cur6)
        return pred, fut6

    def _process_src_current(self, src, current):
        """mtio.py:150-166 (src [B,S,6], current [B,1,6])."""
        return _VPFunction.apply(self, src.contiguous(), current.contiguous(), *self._params)

    def loss_function(self, pred, gt):
        """mtio.py:94-104."""
        return _MTIOLoss.apply(pred, gt)

    def sample(self, history, current):
        """mtio.py:106-133 -> [B,T,2] in [0,1]."""
        self._require_gpu(history)
        history, current = history.contiguous().float(), current.contiguous().float()
        B, S, _ = history.shape
        cfg = self._cfg(B, S, inference=True)
        ws = self._workspace(cfg)
        arr, _ = self._pointers()
        pe, rm, rv, _nbt = self._engine_buffers()
        out = torch.empty(B, self.fut_window, self.in_channel, dtype=torch.float32, device=history.device)
        check(lib().mansy_vp_sample(ctypes.byref(cfg), arr, ptr(pe), ptr(rm), ptr(rv), ptr(history), ptr(current), ptr(out),
                                    ptr(ws), stream_ptr(history.device)), 'mansy_vp_sample')
        return out

    # ------------------------------------------------------------------ fused fast path
    def train_step(self, history, current, future, optimizer, grad_sync=None):
        """One iteration of run_models.py:37-44 (mix, zero_grad, forward, loss, backward, AdamW) as a single
        engine call.  `optimizer` must be a FusedAdamW over this model.  Returns the loss (device scalar).
        Data parallel: pass `grad_sync(flat_grad)` (e.g. an RCCL all-reduce of the ONE flat gradient buffer);
        the engine then stops after backward and AdamW runs after the collective."""
        self._require_gpu(history)
        if not self.training:
            raise _lib.MansyError('train_step requires model.train()')
        history, current, future = (t.contiguous().float() for t in (history, current, future))
        B, S, _ = history.shape
        cfg = self._cfg(B, S)
        ws = self._workspace(cfg)
        perms = self._perms_to_device(self._mix_decision(B), history.device)
        p1, p2 = perms if perms is not None else (None, None)
        self._arm_bn_sync(cfg)
        arr, garr = self._pointers(self._flat_g)
        pe, rm, rv, nbt = self._engine_buffers()
        optimizer._ensure_state()
        optimizer.step_count += 1
        g = optimizer.param_groups[0]
        engine_step = optimizer.step_count if grad_sync is None else 0
        loss = torch.empty((), dtype=torch.float32, device=history.device)
        # overlapped gradient sync (dist.OverlappedGradSync): the engine calls back once the decoder-side gradients are final
        tail_started, tail_off = False, 0
        self._grad_ready = None
        if grad_sync is not None and hasattr(grad_sync, 'start_tail') and self.bn_sync_world > 1:
            tail_off = self.grad_tail_offset()

            def _ready():
                nonlocal tail_started
                grad_sync.start_tail(self._flat_g[tail_off:])
                tail_started = True
            self._grad_ready = _ready
        try:
            check(lib().mansy_vp_train_step(
                ctypes.byref(cfg), arr, garr, ptr(self._flat_p), ptr(self._flat_g), ptr(optimizer.exp_avg), ptr(optimizer.exp_avg_sq),
                self._flat_p.numel(), ptr(pe), ptr(rm), ptr(rv), ptr(nbt), ptr(history), ptr(current), ptr(future), ptr(p1), ptr(p2),
                g['lr'], g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], engine_step, ptr(loss), ptr(ws),
                self._next_seed(), stream_ptr(history.device)), 'mansy_vp_train_step')
        finally:
            # the closure belongs to THIS call: a later autograd backward on the same model (hook which = 2 fires there too when
            # bn_sync_world > 1) must not start a tail all-reduce nobody finishes
            self._grad_ready = None
        if grad_sync is not None:
            if tail_started:
                grad_sync.finish(self._flat_g[:tail_off])          # head (embedding + encoder) now; then wait for the tail
            else:
                grad_sync(self._flat_g)
            optimizer.apply_flat()
        return loss

    def grad_tail_offset(self):
        """First element of the flat gradient buffer whose gradient is final before the encoder backward starts: everything from
        transformer.decoder.layers.0.* to the end of the parameter table (engine hook which = 2)."""
        names = [n for n, _ in self._param_table()]
        return self._offsets[next(i for i, n in enumerate(names) if n.startswith('transformer.decoder.layers.0.'))]


class _VPFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, src, cur, *params):
        model._require_gpu(src)
        B, S, _ = src.shape
        cfg = model._cfg(B, S)
        ws = model._workspace(cfg)
        arr, _ = model._pointers()
        pe, rm, rv, nbt = model._engine_buffers()
        seed = model._next_seed() if model.training else 0
        model._arm_bn_sync(cfg)
        pred = torch.empty(B, model.fut_window, cfg.in_ch, dtype=torch.float32, device=src.device)
        check(lib().mansy_vp_forward(ctypes.byref(cfg), arr, ptr(pe), ptr(rm), ptr(rv), ptr(nbt), ptr(src), ptr(cur.reshape(B, -1)),
                                     ptr(pred), ptr(ws), int(model.training), seed, stream_ptr(src.device)), 'mansy_vp_forward')
        ctx.model, ctx.cfg, ctx.seed, ctx.src, ctx.train = model, cfg, seed, src, model.training
        return pred

    @staticmethod
    def backward(ctx, dpred):
        model = ctx.model
        if not ctx.train:
            raise _lib.MansyError('backward through an eval-mode forward is not supported (BatchNorm eval backward is not on the path)')
        gflat = torch.zeros_like(model._flat_p)
        model._grad_ready = None            # this backward writes into its own buffer: no overlapped tail all-reduce belongs to it
        model._arm_bn_sync(ctx.cfg)
        arr, garr = model._pointers(gflat)
        ws = model._workspace(ctx.cfg)
        check(lib().mansy_vp_backward(ctypes.byref(ctx.cfg), arr, garr, ptr(ctx.src), ptr(dpred.contiguous()), ptr(ws), ctx.seed,
                                      stream_ptr(dpred.device)), 'mansy_vp_backward')
        grads = tuple(gflat[o:o + p.numel()].view(p.shape) for p, o in zip(model._params, model._offsets))
        return (None, None, None) + grads


class _MTIOLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        if not pred.is_cuda:
            raise _lib.MansyError('loss_function runs on the HIP path only')
        B, T, C = pred.shape
        pred_c, gt_c = pred.contiguous().float(), gt.contiguous().float()
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred_c)
        scratch = torch.empty(1, dtype=torch.float64, device=pred.device)
        check(lib().mansy_mtio_loss_fwd_bwd(ptr(pred_c), ptr(gt_c), B, T, C, ptr(scratch), ptr(loss), ptr(dpred),
                                            stream_ptr(pred.device)), 'mansy_mtio_loss_fwd_bwd')
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (run_models.py:29 uses its defaults) as ONE kernel over the model's flat
    parameter buffer.  Works both in the reference-style loop (zero_grad / loss.backward / step) and through
    `model.train_step(...)`."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.model = model
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.step_count = 0
        self.exp_avg = None
        self.exp_avg_sq = None

    def _ensure_state(self):
        f = self.model._flat_p
        if self.exp_avg is None or self.exp_avg.device != f.device or self.exp_avg.numel() != f.numel():
            self.exp_avg = torch.zeros_like(f)
            self.exp_avg_sq = torch.zeros_like(f)

    def zero_grad(self, set_to_none=True):
        m = self.model
        if not m._is_flat():
            m._flatten()
        m._flat_g.zero_()
        for p, o in zip(m._params, m._offsets):
            p.grad = m._flat_g[o:o + p.numel()].view(p.shape)

    @torch.no_grad()
    def step(self, closure=None):
        m = self.model
        self._ensure_state()
        for p, o in zip(m._params, m._offsets):      # gradients produced outside the flat buffer (first backward)
            if p.grad is not None and p.grad.data_ptr() != m._flat_g.data_ptr() + 4 * o:
                m._flat_g[o:o + p.numel()].copy_(p.grad.reshape(-1))
        self.step_count += 1
        self.apply_flat()

    def apply_flat(self):
        """AdamW kernel over the flat buffers with the current step_count."""
        m = self.model
        self._ensure_state()
        g = self.param_groups[0]
        f = m._flat_p
        check(lib().mansy_adamw_step(ptr(f), ptr(m._flat_g), ptr(self.exp_avg), ptr(self.exp_avg_sq), f.numel(), g['lr'],
                                     g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], self.step_count, 1,
                                     stream_ptr(f.device)), 'mansy_adamw_step')
