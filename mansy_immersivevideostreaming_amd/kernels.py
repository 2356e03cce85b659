"""Thin torch-tensor front-ends for the single-kernel C-ABI entry points (unit-test surface and building
blocks for the host mirrors).  Tensors must live on the GPU; nothing here computes on the CPU."""
import ctypes
import math

import torch

from ._lib import (AttnShape, GemmEpilogue, MansyError, check, lib, ptr, stream_ptr,   # noqa: F401
                   PRECISIONS, get_precision, precision, set_precision)


# mansy_gemm_epilogue::variant (include/mansy_hip.h MANSY_VARIANT_*): per call, nothing is remembered by the library
VARIANT_NO_WSK, VARIANT_NO_WSK_TN, VARIANT_NO_PLAIN = 0x100, 0x200, 0x400


def VARIANT_BF16(v):
    return (v + 1) & 0xFF


def VARIANT_COL_GROUP(g):
    return ((g + 1) & 0xFF) << 16


def _gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise MansyError('HIP kernels need cuda (ROCm) tensors')


def gemm(A, B, a_kmajor=False, b_kmajor=False, bias=None, relu=False, mask_src=None, mask_scale=1.0, drop=None,
         resid=None, out=None, accumulate=False, force_tile=0, force_splitk=0, a_rowsum=None, prec=None, variant=0):
    """C = A·B with A logical [M,K] (stored [K,M] if a_kmajor) and B logical [K,N]
    (stored [N,K] if not b_kmajor -- a torch Linear weight -- or [K,N] if b_kmajor).
    a_rowsum [M] (a_kmajor only): += the row sums of A over K, the bias-gradient rider of the dW products.
    prec: None = the calling thread's default (kernels.precision), else 0 / 1 / 3 / 6 for this product.
    variant: mansy_gemm_epilogue::variant (VARIANT_* below) -- which loop this one product runs on."""
    _gpu(A, B)
    A, B = A.contiguous(), B.contiguous()
    M, K = (A.shape[1], A.shape[0]) if a_kmajor else A.shape
    N = B.shape[1] if b_kmajor else B.shape[0]
    Kb = B.shape[0] if b_kmajor else B.shape[1]
    assert K == Kb, (A.shape, B.shape)
    if out is None:
        out = torch.zeros(M, N, dtype=torch.float32, device=A.device) if accumulate else \
            torch.empty(M, N, dtype=torch.float32, device=A.device)
    ep = GemmEpilogue()
    ep.bias = ptr(bias).value if bias is not None else None
    ep.relu = int(relu)
    if mask_src is not None:
        ep.mask_src, ep.mask_ld = mask_src.data_ptr(), mask_src.stride(0)
    ep.mask_scale = mask_scale
    if drop is not None:
        ep.drop_p, ep.drop_seed, ep.drop_site = drop
    if resid is not None:
        ep.resid, ep.resid_ld = resid.data_ptr(), resid.stride(0)
    ep.accumulate = int(accumulate)
    if a_rowsum is not None:
        assert a_kmajor and a_rowsum.numel() == M and a_rowsum.dtype == torch.float32
        ep.a_rowsum = a_rowsum.data_ptr()
    if prec is not None:
        ep.prec = int(prec)
    ep.variant = int(variant)
    check(lib().mansy_gemm_f32(ptr(A), A.stride(0), int(a_kmajor), ptr(B), B.stride(0), int(b_kmajor), ptr(out), out.stride(0),
                               M, N, K, ctypes.byref(ep), force_tile, force_splitk, stream_ptr(A.device)), 'mansy_gemm_f32')
    return out


def weight_planes(W, n_planes):
    """bf16 planes of a weight W [N, K] and of its transpose for the split-bf16 products: (planes [n_planes, N, K], planes_t
    [n_planes, K, N]) as int16 tensors (mansy_weight_planes)."""
    _gpu(W)
    W = W.contiguous().float()
    N, K = W.shape
    out = torch.empty(n_planes, N, K, dtype=torch.int16, device=W.device)
    out_t = torch.empty(n_planes, K, N, dtype=torch.int16, device=W.device)
    check(lib().mansy_weight_planes(ptr(W), N, K, ptr(out), ptr(out_t), N * K, n_planes, stream_ptr(W.device)), 'mansy_weight_planes')
    return out, out_t


def gemm_planes(A, W, planes, transposed=False, bias=None, relu=False, resid=None, force_tile=0, variant=0):
    """A [M, K'] times a weight given in fp32 (W [N, K]) AND as bf16 planes (weight_planes): transposed=False -> A W^T with `planes`
    of W; transposed=True -> A W with the planes of W^T.  In fp32 mode the planes are ignored."""
    _gpu(A, W, planes)
    A, W = A.contiguous(), W.contiguous()
    M = A.shape[0]
    Nw, Kw = W.shape
    N, K = (Kw, Nw) if transposed else (Nw, Kw)
    assert A.shape[1] == K and planes.shape[1:] == (N, K), (A.shape, W.shape, planes.shape)
    out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    ep = GemmEpilogue()
    ep.bias = ptr(bias).value if bias is not None else None
    ep.relu = int(relu)
    ep.mask_scale = 1.0
    if resid is not None:
        ep.resid, ep.resid_ld = resid.data_ptr(), resid.stride(0)
    ep.variant = int(variant)
    check(lib().mansy_gemm_planes(ptr(A), A.stride(0), ptr(W), W.stride(0), int(transposed), ptr(planes), planes.stride(0), planes.stride(1),
                                  ptr(out), out.stride(0), M, N, K, ctypes.byref(ep), force_tile, stream_ptr(A.device)), 'mansy_gemm_planes')
    return out


def _attn_shape(nb, H, Lq, Lk, dh, q, k, v, o):
    s = AttnShape(nb=nb, H=H, Lq=Lq, Lk=Lk, dh=dh, scale=1.0 / math.sqrt(dh))
    s.q_bs, s.q_rs = q
    s.k_bs, s.k_rs = k
    s.v_bs, s.v_rs = v
    s.o_bs, s.o_rs = o
    return s


def attn_fwd_packed(qkv, H, drop=None, save_p=True):
    """Encoder-style self attention on a packed projection qkv [B,L,3d] -> (out [B,L,d], P [B*H,L,L])."""
    _gpu(qkv)
    B, L, d3 = qkv.shape
    d = d3 // 3
    s = _attn_shape(B, H, L, L, d // H, (L * d3, d3), (L * d3, d3), (L * d3, d3), (L * d, d))
    out = torch.empty(B, L, d, dtype=torch.float32, device=qkv.device)
    P = torch.empty(B * H, L, L, dtype=torch.float32, device=qkv.device) if save_p else None
    p, seed, site = drop if drop is not None else (0.0, 0, 0)
    base = qkv.data_ptr()
    check(lib().mansy_attn_fwd(base, base + 4 * d, base + 8 * d, ptr(out), ptr(P), ctypes.byref(s), p, seed, site,
                               stream_ptr(qkv.device)), 'mansy_attn_fwd')
    return out, P


def attn_bwd_packed(qkv, P, dout, H, drop=None):
    _gpu(qkv, P, dout)
    B, L, d3 = qkv.shape
    d = d3 // 3
    s = _attn_shape(B, H, L, L, d // H, (L * d3, d3), (L * d3, d3), (L * d3, d3), (L * d, d))
    dqkv = torch.empty_like(qkv)
    p, seed, site = drop if drop is not None else (0.0, 0, 0)
    base, gb = qkv.data_ptr(), dqkv.data_ptr()
    check(lib().mansy_attn_bwd(base, base + 4 * d, base + 8 * d, ptr(P), ptr(dout.contiguous()), gb, gb + 4 * d, gb + 8 * d,
                               ctypes.byref(s), p, seed, site, 0, stream_ptr(qkv.device)), 'mansy_attn_bwd')
    return dqkv


def layernorm_fwd(a, b, w, bias, eps=1e-5):
    _gpu(a, w)
    rows, C = a.shape
    z = torch.empty_like(a)
    y = torch.empty_like(a)
    mean = torch.empty(rows, dtype=torch.float32, device=a.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=a.device)
    check(lib().mansy_layernorm_fwd(ptr(a), ptr(b), ptr(w), ptr(bias), ptr(z), ptr(y), ptr(mean), ptr(rstd), rows, C, eps,
                                    stream_ptr(a.device)), 'mansy_layernorm_fwd')
    return y, z, mean, rstd


def layernorm_bwd(dy, z, mean, rstd, w, drop=None, want_bias=True):
    _gpu(dy, z, w)
    rows, C = dy.shape
    dz = torch.empty_like(dy)
    dz_drop = torch.empty_like(dy)
    dw = torch.zeros(C, dtype=torch.float32, device=dy.device)
    db = torch.zeros(C, dtype=torch.float32, device=dy.device) if want_bias else None
    p, seed, site = drop if drop is not None else (0.0, 0, 0)
    check(lib().mansy_layernorm_bwd(ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(w), ptr(dz), ptr(dz_drop), p, seed, site, ptr(dw),
                                    ptr(db), rows, C, stream_ptr(dy.device)), 'mansy_layernorm_bwd')
    return dz, dz_drop, dw, db


def layernorm_bwd_partial(dy, z, mean, rstd, w, drop=None, partials=None, accumulate=False):
    """Engine form: weight-gradient column sums into per-workgroup slots; returns (dz, dz_drop, partials [parts, 2, C])."""
    _gpu(dy, z, w)
    rows, C = dy.shape
    dz = torch.empty_like(dy)
    dz_drop = torch.empty_like(dy)
    parts = lib().mansy_layernorm_bwd_parts(rows)
    if partials is None:
        partials = torch.zeros(parts, 2, C, dtype=torch.float32, device=dy.device)
    p, seed, site = drop if drop is not None else (0.0, 0, 0)
    check(lib().mansy_layernorm_bwd_partial(ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(w), ptr(dz), ptr(dz_drop), p, seed, site,
                                            ptr(partials), int(accumulate), rows, C, stream_ptr(dy.device)), 'mansy_layernorm_bwd_partial')
    return dz, dz_drop, partials


def ln_partials_reduce(partials, dw, db):
    parts, _, C = partials.shape
    check(lib().mansy_ln_partials_reduce(ptr(partials), parts, C, ptr(dw), ptr(db), stream_ptr(partials.device)), 'mansy_ln_partials_reduce')
    return dw, db


def attn_cross_deferred(q_all, memkv, P_all, dO_all, H, drop_sites=None, p_drop=0.0, seed=0):
    """Decoder cross-attention backward the engine's way: q_all / dO_all [T, B, d], memkv [B, M, 2d] (K | V), P_all [T, B*H, M].
    Returns (dq_all [T, B, d], dmemkv [B, M, 2d])."""
    _gpu(q_all, memkv, P_all, dO_all)
    T, B, d = q_all.shape
    M = memkv.shape[1]
    s = _attn_shape(B, H, 1, M, d // H, (d, 0), (M * 2 * d, 2 * d), (M * 2 * d, 2 * d), (d, 0))
    dq = torch.empty_like(q_all)
    dS = torch.empty(T, B * H, M, dtype=torch.float32, device=q_all.device)
    Pk = torch.empty_like(dS)
    dkv = torch.empty_like(memkv)
    kb = memkv.data_ptr()
    st = stream_ptr(q_all.device)
    for i in range(T):
        site = drop_sites[i] if drop_sites is not None else 0
        check(lib().mansy_attn_bwd_dq(ptr(q_all[i]), kb, kb + 4 * d, ptr(P_all[i]), ptr(dO_all[i]), ptr(dq[i]), ptr(dS[i]), ptr(Pk[i]),
                                      ctypes.byref(s), p_drop, seed, site, st), 'mansy_attn_bwd_dq')
    gb = dkv.data_ptr()
    check(lib().mansy_attn_kvgrad(ptr(q_all), B * d, ptr(dO_all), B * d, ptr(dS), ptr(Pk), gb, gb + 4 * d, ctypes.byref(s), T, 0, st),
          'mansy_attn_kvgrad')
    return dq, dkv


def attn_cross_stepwise(q_all, memkv, P_all, dO_all, H, drop_sites=None, p_drop=0.0, seed=0):
    """Same gradients through T accumulating mansy_attn_bwd calls (the read-modify-write form)."""
    _gpu(q_all, memkv, P_all, dO_all)
    T, B, d = q_all.shape
    M = memkv.shape[1]
    s = _attn_shape(B, H, 1, M, d // H, (d, 0), (M * 2 * d, 2 * d), (M * 2 * d, 2 * d), (d, 0))
    dq = torch.empty_like(q_all)
    dkv = torch.zeros_like(memkv)
    kb, gb = memkv.data_ptr(), dkv.data_ptr()
    st = stream_ptr(q_all.device)
    for i in range(T):
        site = drop_sites[i] if drop_sites is not None else 0
        check(lib().mansy_attn_bwd(ptr(q_all[i]), kb, kb + 4 * d, ptr(P_all[i]), ptr(dO_all[i]), ptr(dq[i]), gb, gb + 4 * d,
                                   ctypes.byref(s), p_drop, seed, site, 1, st), 'mansy_attn_bwd')
    return dq, dkv


def attn_self_decode_bwd(qkv_all, P_all, dO_all, H, drop_sites=None, p_drop=0.0, seed=0, pull=True):
    """KV-cached decoder self-attention backward over T steps.  qkv_all [T, B, 3d] (step-major slab = the KV cache),
    P_all [T, B*H, T] (row i holds i+1 probabilities, packed with stride i+1 like the engine's), dO_all [T, B, d].
    pull=True: the engine's pull form; False: T accumulating read-modify-write calls.  Returns dqkv_all [T, B, 3d]."""
    _gpu(qkv_all, P_all, dO_all)
    T, B, d3 = qkv_all.shape
    d = d3 // 3
    dqkv = torch.empty_like(qkv_all) if pull else torch.zeros_like(qkv_all)
    dS = torch.empty(T, B * H, T, dtype=torch.float32, device=qkv_all.device)
    Pk = torch.empty_like(dS)
    base, gb = qkv_all.data_ptr(), dqkv.data_ptr()
    st = stream_ptr(qkv_all.device)
    for i in range(T - 1, -1, -1):
        s = _attn_shape(B, H, 1, i + 1, d // H, (d3, 0), (d3, B * d3), (d3, B * d3), (d, 0))
        site = drop_sites[i] if drop_sites is not None else 0
        if pull:
            check(lib().mansy_attn_bwd_selfpull(base, B * d3, base + 4 * d, base + 8 * d, ptr(P_all[i]), ptr(dO_all), B * d,
                                                gb + 4 * i * B * d3, gb + 4 * d, gb + 8 * d, ptr(dS), ptr(Pk), ctypes.byref(s), T, i,
                                                p_drop, seed, site, st), 'mansy_attn_bwd_selfpull')
        else:
            check(lib().mansy_attn_bwd(base + 4 * i * B * d3, base + 4 * d, base + 8 * d, ptr(P_all[i]), ptr(dO_all[i]),
                                       gb + 4 * i * B * d3, gb + 4 * d, gb + 8 * d, ctypes.byref(s), p_drop, seed, site, 1, st), 'mansy_attn_bwd')
    return dqkv


def tilemap(xy, W=2560, H=1440, nw=8, nh=8, fov_w=600, fov_h=300):
    """xy [...,2] float32 normalised centres -> uint64 hit maps as int64 tensor [...] (bit row*nw+col)."""
    _gpu(xy)
    xy = xy.contiguous().float()
    n = xy.numel() // 2
    out = torch.empty(xy.shape[:-1], dtype=torch.int64, device=xy.device)
    check(lib().mansy_tilemap(ptr(xy), n, W, H, nw, nh, fov_w, fov_h, ptr(out), stream_ptr(xy.device)), 'mansy_tilemap')
    return out


def tilemap_iou(a, b):
    _gpu(a, b)
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty(a.shape, dtype=torch.float64, device=a.device)
    check(lib().mansy_tilemap_iou(ptr(a), ptr(b), a.numel(), ptr(out), stream_ptr(a.device)), 'mansy_tilemap_iou')
    return out


def tilemap_or_groups(maps, group):
    """maps [n*group] -> [n]: OR of each run of `group` consecutive maps (predict.py:39-45)."""
    _gpu(maps)
    maps = maps.contiguous()
    n = maps.numel() // group
    out = torch.empty(n, dtype=torch.int64, device=maps.device)
    check(lib().mansy_tilemap_or_groups(ptr(maps), n, group, ptr(out), stream_ptr(maps.device)), 'mansy_tilemap_or_groups')
    return out
