"""ctypes binding of libmansy_hip.so (include/mansy_hip.h).  There is NO CPU fallback: if the
library is missing or a call fails this raises."""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmansy_hip.so')

c_int, c_float, c_void_p, c_ll, c_u32 = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_uint32


class MansyError(RuntimeError):
    pass


BN_SYNC_FN = ctypes.CFUNCTYPE(c_int, c_int, c_void_p)


class VPConfig(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ('B', 'S', 'T', 'd_model', 'n_head', 'd_ff', 'n_enc', 'n_dec', 'in_ch', 'has_bias')] + \
               [(n, c_float) for n in ('p_pe', 'p_drop', 'ln_eps', 'bn_eps', 'bn_momentum')] + \
               [('max_len', c_int), ('bn_sync_world', c_int), ('two_stream', c_int), ('precision', c_int),
                ('bn_sync_fn', BN_SYNC_FN), ('bn_sync_user', c_void_p)]

    def __init__(self, *a, **kw):
        kw.setdefault('precision', 0)           # MANSY_PREC_F32
        super().__init__(*a, **kw)


class GemmEpilogue(ctypes.Structure):
    _fields_ = [('bias', c_void_p), ('relu', c_int), ('mask_src', c_void_p), ('mask_ld', c_int), ('mask_scale', c_float),
                ('drop_p', c_float), ('drop_seed', c_u32), ('drop_site', c_u32), ('resid', c_void_p), ('resid_ld', c_int),
                ('accumulate', c_int), ('a_rowsum', c_void_p), ('prec', c_int), ('variant', c_int)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        if 'prec' not in kw:
            self.prec = PRECISIONS[current_precision()]      # the calling thread's default (host-side; the library has no default)


class CommId(ctypes.Structure):            # mansy_comm_id: an opaque ncclUniqueId
    _fields_ = [('bytes', ctypes.c_ubyte * 128)]


class XgHandle(ctypes.Structure):          # mansy_xg_handle: an opaque hipIpcMemHandle_t
    _fields_ = [('bytes', ctypes.c_ubyte * 64)]


class EnvTables(ctypes.Structure):
    _fields_ = [('size', c_void_p), ('quality', c_void_p), ('video_len', c_void_p), ('n_chunk_max', c_int),
                ('vp_gt', c_void_p), ('vp_pred', c_void_p), ('vp_acc', c_void_p), ('vp_start', c_void_p), ('vp_end', c_void_p),
                ('n_vpchunk_max', c_int), ('trace_bw', c_void_p), ('trace_len', c_void_p), ('trace_len_max', c_int),
                ('samples', c_void_p), ('n_sample', c_int), ('qoe_w', c_void_p), ('video_rates', c_int * 5),
                ('startup_download', c_int), ('chunk_length', c_int), ('max_size', ctypes.c_double),
                ('max_throughput', ctypes.c_double), ('train_identifier_reward', c_int)]


class EpisodeLog(ctypes.Structure):
    _fields_ = [('records', c_void_p), ('count', c_void_p), ('capacity', c_int)]


class AttnShape(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ('nb', 'H', 'Lq', 'Lk', 'dh')] + \
               [(n, c_ll) for n in ('q_bs', 'q_rs', 'k_bs', 'k_rs', 'v_bs', 'v_rs', 'o_bs', 'o_rs')] + [('scale', c_float)]


# name -> argtypes (restype int unless listed in _RESTYPES); must list every symbol of include/mansy_hip.h
P = c_void_p
_PROTOS = {
    'mansy_last_error': [],
    'mansy_abi_version': [],
    'mansy_vp_num_params': [P],
    'mansy_vp_param_info': [P, c_int, ctypes.c_char_p, c_int, P, P, P],
    'mansy_vp_workspace_bytes': [P],
    'mansy_vp_ws_lookup': [P, ctypes.c_char_p, P, P],
    'mansy_vp_forward': [P, P, P, P, P, P, P, P, P, P, c_int, c_u32, P],
    'mansy_vp_backward': [P, P, P, P, P, P, c_u32, P],
    'mansy_vp_sample': [P, P, P, P, P, P, P, P, P, P],
    'mansy_vp_train_step': [P, P, P, P, P, P, P, c_ll, P, P, P, P, P, P, P, P, P, c_float, c_float, c_float, c_float,
                            c_float, c_int, P, P, c_u32, P],
    'mansy_mtio_mix': [P, P, P, P, c_int, c_int, c_int, P],
    'mansy_mtio_loss_fwd_bwd': [P, P, c_int, c_int, c_int, P, P, P, P],
    'mansy_adamw_step': [P, P, P, P, c_ll, c_float, c_float, c_float, c_float, c_float, c_int, c_int, P],
    'mansy_ensemble_wrap': [P, P, c_ll, c_int, c_int, P],
    'mansy_linreg_sample': [P, P, c_int, c_int, c_int, c_int, P, P],
    'mansy_traj_gather': [P, c_int, c_int, P, c_int, c_int, c_int, P, P, P, P],
    'mansy_periodic_mse': [P, P, c_ll, c_int, P, P],
    'mansy_tilemap_metrics': [P, P, c_ll, P, P],
    'mansy_tilemap': [P, c_ll, c_int, c_int, c_int, c_int, c_int, c_int, P, P],
    'mansy_tilemap_iou': [P, P, c_ll, P, P],
    'mansy_tilemap_or_groups': [P, c_ll, c_int, P, P],
    'mansy_gemm_f32': [P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, P, c_int, c_int, P],
    'mansy_weight_planes': [P, c_int, c_int, P, P, c_ll, c_int, P],
    'mansy_gemm_planes': [P, c_int, P, c_int, c_int, P, c_ll, c_int, P, c_int, c_int, c_int, c_int, P, c_int, P],
    'mansy_gemm_bf16': [P, c_int, c_int, P, c_int, c_int, P, c_int, P, c_int, c_int, c_int, c_int, P, P, P, c_int, c_int, P],
    'mansy_attn_fwd': [P, P, P, P, P, P, c_float, c_u32, c_u32, P],
    'mansy_attn_bwd': [P, P, P, P, P, P, P, P, P, c_float, c_u32, c_u32, c_int, P],
    'mansy_layernorm_fwd': [P, P, P, P, P, P, P, P, c_int, c_int, c_float, P],
    'mansy_layernorm_bwd': [P, P, P, P, P, P, P, c_float, c_u32, c_u32, P, P, c_int, c_int, P],
    'mansy_layernorm_bwd_parts': [c_int],
    'mansy_layernorm_bwd_partial': [P, P, P, P, P, P, P, c_float, c_u32, c_u32, P, c_int, c_int, c_int, P],
    'mansy_ln_partials_reduce': [P, c_int, c_int, P, P, P],
    'mansy_attn_bwd_dq': [P, P, P, P, P, P, P, P, P, c_float, c_u32, c_u32, P],
    'mansy_attn_kvgrad': [P, c_ll, P, c_ll, P, P, P, P, P, c_int, c_int, P],
    'mansy_attn_bwd_selfpull': [P, c_ll, P, P, P, P, c_ll, P, P, P, P, P, P, c_int, c_int, c_float, c_u32, c_u32, P],
    'mansy_env_state_bytes': [],
    'mansy_env_init': [P, c_int, c_int, c_int, c_int, P],
    'mansy_env_reset': [P, P, c_int, P, P],
    'mansy_env_step': [P, P, c_int, P, P, P, P, P, P, P, P],
    'mansy_allocate_tile_rates': [P, P, c_int, P, P, P],
    'mansy_expert_profile': [P, P, c_int, P, P, P, P, P, P, P],
    'mansy_expert_choose_action': [P, P, c_int, c_int, P, P, P, P, P, P, P, P],
    'mansy_net_num_params': [c_int],
    'mansy_net_param_info': [c_int, c_int, ctypes.c_char_p, c_int, P, P, P],
    'mansy_ppo_workspace_bytes': [c_int],
    'mansy_policy_forward': [P, P, c_int, P, P, P, P, P, c_u32, c_u32, c_int, P, c_int, c_int, P],
    'mansy_policy_env_step': [P, P, c_int, P, P, P, P, c_u32, c_u32, c_int, P, c_int, P, P, P, P, P, P, P, P, c_int, P],
    'mansy_policy_evaluate': [P, P, c_int, P, c_int, P, P, P, c_int, c_int, P],
    'mansy_policy_rollout': [P, P, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, c_int, P, P, P, c_int, c_int, P],
    'mansy_identifier_forward': [P, P, c_int, P, P, c_int, c_int, P],
    'mansy_identifier_train_step': [P, P, P, P, P, P, c_ll, P, P, c_int, c_float, c_float, c_int, P, P, c_int, P, P, c_int, P],
    'mansy_identifier_relabel': [P, P, P, P, c_int, c_float, P, c_int, c_int, P],
    'mansy_gae_returns': [P, P, P, P, c_int, c_int, ctypes.c_double, ctypes.c_double, c_int, P, P, P, P, P],
    'mansy_ppo_minibatch_step': [P, P, P, P, P, P, c_ll, P, P, P, P, P, P, P, c_int, c_float, c_float, c_float, c_int, c_int, c_float, c_float,
                                 c_float, c_float, c_int, c_ll, c_int, P, P, c_int, c_int, P, c_int, P, P, c_int, P],
    'mansy_bc_step': [P, P, P, P, P, P, c_ll, c_ll, P, P, c_int, c_float, c_float, c_float, c_int, P, P, c_int, c_int, P],
    'mansy_clip_grad_adam': [P, P, P, P, c_ll, c_float, c_float, c_float, c_int, c_ll, c_int, P, c_int, P, P],
    'mansy_ppo_dp_tail': [P, P, P, P, P, c_ll, c_float, c_float, c_float, c_int, P, c_int, P, P, P, c_int, P, P, P, c_int, c_int, P],
    'mansy_comm_unique_id': [P],
    'mansy_comm_create': [P, c_int, c_int, P],
    'mansy_comm_destroy': [P],
    'mansy_allreduce_avg_f32': [P, P, c_ll, P],
    'mansy_allreduce_sum_f64': [P, P, c_ll, P],
    'mansy_allgather_f64': [P, P, P, c_ll, P],
    'mansy_xg_create': [c_ll, c_int, c_int, P],
    'mansy_xg_export': [P, P],
    'mansy_xg_import': [P, P],
    'mansy_xg_set_timeout_ms': [P, ctypes.c_double],
    'mansy_xg_allreduce_avg': [P, P, c_ll, P, P],
    'mansy_xg_slot_ptrs': [P, P, P],
    'mansy_xg_next_slot': [P],
    'mansy_xg_reduce_avg': [P, P, c_ll, P, P],
    'mansy_xg_status': [P],
    'mansy_xg_destroy': [P],
    'mansy_a2c_num_params': [],
    'mansy_a2c_param_info': [c_int, ctypes.c_char_p, c_int, P, P, P],
    'mansy_a2c_workspace_bytes': [c_int],
    'mansy_a2c_obs': [P, P, P, P, c_int, P, P, P],
    'mansy_a2c_forward': [P, P, c_int, P, P, P, P, P, c_u32, c_u32, c_int, P, c_int, c_int, P],
    'mansy_a2c_minibatch_step': [P, P, P, P, P, c_ll, P, P, P, P, P, c_int, c_float, c_float, c_float, c_float, c_float, c_float, c_int, P,
                                 P, c_int, c_int, P],
    'mansy_clip_grad_rmsprop': [P, P, P, c_ll, c_float, c_float, c_float, c_float, P, P],
    'mansy_prof_gemm_enable': [c_int],
    'mansy_prof_gemm_collect': [P, P, P],
    'mansy_prof_launch_count': [],
}
_RESTYPES = {'mansy_last_error': ctypes.c_char_p, 'mansy_prof_launch_count': ctypes.c_ulonglong, 'mansy_vp_workspace_bytes': ctypes.c_size_t,
             'mansy_ppo_workspace_bytes': ctypes.c_size_t, 'mansy_a2c_workspace_bytes': ctypes.c_size_t}

# bumped together with mansy_abi_version() (csrc/capi.hip) whenever a prototype or struct above changes: a stale in-tree
# libmansy_hip.so then fails at load time instead of being called with a wrong argument list
ABI_VERSION = 9

_lib = None


def declared_symbols():
    return sorted(_PROTOS)


def _load(path):
    if not os.path.exists(path):
        raise MansyError(
            f'{path} not found: build it with `python -m mansy_immersivevideostreaming_amd.build_ext` '
            '(there is no CPU fallback for the HIP path)')
    # torch first: it loads the HIP runtime (libamdhip64) that owns the tensors/streams we are handed;
    # loading ours first would bring up a second, disjoint runtime ("no ROCm-capable device").
    import torch  # noqa: F401
    L = ctypes.CDLL(path)
    for name, args in _PROTOS.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, c_int)
    got = L.mansy_abi_version()
    if got != ABI_VERSION:
        raise MansyError(f'{path} has ABI version {got}, these bindings expect {ABI_VERSION}: rebuild it with '
                         '`python -m mansy_immersivevideostreaming_amd.build_ext`')
    return L


def lib():
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().mansy_last_error()
        raise MansyError(f'{what} failed ({rc}): {msg.decode() if msg else ""}')


def ptr(t):
    """Device (or host) pointer of a torch tensor / None."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# ---- precision of the dense products.  The library has no process-wide mode (ABI 8): every call carries its own.  What is kept here is a
# HOST-side, per-thread default for callers that name none (a model with `precision = None`, the kernels.py front-ends): plain Python state.
PRECISIONS = {'f32': 0, 'fp32': 0, 'bf16': 1, 'bf16x3': 3, 'bf16x6': 6, 0: 0, 1: 1, 3: 3, 6: 6}
_NAMES = {0: 'f32', 1: 'bf16', 3: 'bf16x3', 6: 'bf16x6'}
_tls = threading.local()


def current_precision():
    return getattr(_tls, 'precision', 'f32')


def set_precision(mode):
    """Default precision of THIS thread's calls that name none: 'f32' (exact fp32 MFMA), 'bf16', 'bf16x3' or 'bf16x6'.  Returns the previous
    default.  Host-side only -- the value is passed to the library with each call."""
    if mode not in PRECISIONS:
        raise MansyError(f'unknown precision {mode!r}: one of f32, bf16, bf16x3, bf16x6')
    prev = current_precision()
    _tls.precision = _NAMES[PRECISIONS[mode]]
    return prev


def get_precision():
    return current_precision()


def resolve_precision(mode):
    """None -> the thread's default; a name / code -> its MANSY_PREC_* code."""
    if mode is None:
        mode = current_precision()
    if mode not in PRECISIONS:
        raise MansyError(f'unknown precision {mode!r}: one of f32, bf16, bf16x3, bf16x6')
    return PRECISIONS[mode]


class precision:
    """with kernels.precision('bf16x3'): ...   -- every call of this thread that names no precision of its own runs in that mode.
    mode None: leave the default as it is."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = set_precision(self.mode) if self.mode is not None else None
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            set_precision(self.prev)
        return False
