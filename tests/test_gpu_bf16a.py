"""GPU: the bf16-STORAGE products of the MANSY_PREC_BF16 perf mode (csrc/gemm_bf16a.hip, round 6) through the C ABI (mansy_gemm_bf16): operands are
bf16 images in HBM, staged by LDS-DMA without conversion, one bf16 MFMA product with fp32 accumulation.  Reference: a plain PyTorch fp32 product
of the SAME bf16-rounded operands -- a product of two bf16 numbers is exact in fp32, so only the summation order differs (tolerance 2e-6 of the
largest element).  Forward / dX form (all three tiles, ragged M, the fused epilogue, the bf16 image of the output) and the weight-gradient form
(K-major operands through ds_read_b64_tr_b16, split-K, the bias-gradient rider)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def L():
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a ROCm device (no CPU fallback exists)')
    from mansy_immersivevideostreaming_amd import _lib
    return _lib


def _bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize('M,N,K,tile', [(4096, 512, 512, 0), (4096, 512, 512, 64), (4096, 512, 512, 96), (4096, 512, 512, 128), (4096, 1536, 512, 0),
                                        (1000, 512, 512, 0), (1000, 192, 64, 64), (160, 64, 192, 0), (40960, 512, 512, 0), (20480, 1024, 512, 0),
                                        (40960, 512, 512, 128), (40000 + 72, 1536, 512, 128), (33000, 200, 1536, 0)])      # (ragged last row panel / column tile)
def test_forward_form_vs_torch(L, M, N, K, tile):
    g = torch.Generator(device='cpu').manual_seed(M + N + K)
    A = _bf(torch.randn(M, K, generator=g)).cuda()
    W = _bf(torch.randn(N, K, generator=g) * 0.05).cuda()
    want = A.float() @ W.float().t()
    C = torch.full((M, N), float('nan'), device='cuda')
    C16 = torch.zeros(M, N, dtype=torch.bfloat16, device='cuda')
    ep = L.GemmEpilogue()
    L.check(L.lib().mansy_gemm_bf16(L.ptr(A), K, 0, L.ptr(W), K, 0, L.ptr(C), N, L.ptr(C16), N, M, N, K, ctypes.byref(ep), None, None, tile, 0, L.stream_ptr()), 'gemm_bf16')
    torch.cuda.synchronize()
    scale = float(want.abs().max())
    assert float((C - want).abs().max()) <= 2e-6 * scale + 1e-6
    assert torch.equal(C16, C.to(torch.bfloat16))                                   # the bf16 image is the rounded final value


@pytest.mark.parametrize('M,tile', [(4096, 0), (40960, 128), (40960 - 56, 96)])
def test_forward_form_fused_epilogue_and_bf16_only_output(L, M, tile):
    N, K = 512, 512
    g = torch.Generator(device='cpu').manual_seed(3)
    A = _bf(torch.randn(M, K, generator=g)).cuda()
    W = _bf(torch.randn(N, K, generator=g) * 0.05).cuda()
    bias = torch.randn(N, generator=g).cuda()
    resid = torch.randn(M, N, generator=g).cuda()
    mask = torch.randn(M, N, generator=g).cuda()
    base = A.float() @ W.float().t() + bias
    for kind in ('relu', 'resid', 'mask'):
        ep = L.GemmEpilogue()
        ep.bias = L.ptr(bias)
        if kind == 'relu':
            ep.relu = 1
            want = torch.relu(base)
        elif kind == 'resid':
            ep.resid, ep.resid_ld = L.ptr(resid), N
            want = base + resid
        else:
            ep.mask_src, ep.mask_ld, ep.mask_scale = L.ptr(mask), N, 1.25
            want = torch.where(mask > 0, base * 1.25, torch.zeros_like(base))
        C16 = torch.zeros(M, N, dtype=torch.bfloat16, device='cuda')
        r16 = m16 = None
        if tile:                  # the residual / the mask source as bf16 images (as the engine's bf16-storage mode keeps them)
            resid16, mask16 = _bf(resid), _bf(mask)
            if kind == 'resid':
                r16, ep.resid, want = L.ptr(resid16), None, base + resid16.float()
            if kind == 'mask':
                m16, ep.mask_src = L.ptr(mask16), None
                want = torch.where(mask16.float() > 0, base * 1.25, torch.zeros_like(base))
        L.check(L.lib().mansy_gemm_bf16(L.ptr(A), K, 0, L.ptr(W), K, 0, None, N, L.ptr(C16), N, M, N, K, ctypes.byref(ep), r16, m16, tile, 0, L.stream_ptr()), 'gemm_bf16')
        torch.cuda.synchronize()
        err = (C16.float() - want).abs().max().item()
        assert err <= 2 ** -8 * float(want.abs().max()) + 1e-6, (kind, err)              # one bf16 rounding of the output
    if tile:
        # two tile shapes on the same operands and images: the same arithmetic in the same order, bit for bit (dropout on: the mask is a function of the
        # element's index, not of the workgroup that draws it); in-place residual (C16 aliases resid16)
        ep = L.GemmEpilogue()
        ep.bias, ep.drop_p, ep.drop_seed, ep.drop_site, ep.resid_ld, ep.mask_ld, ep.mask_scale = L.ptr(bias), 0.1, 77, 5, N, N, 0.5
        outs = []
        for t in (128, 64):
            z16 = _bf(resid).clone()
            Cf = torch.zeros(M, N, device='cuda')
            L.check(L.lib().mansy_gemm_bf16(L.ptr(A), K, 0, L.ptr(W), K, 0, L.ptr(Cf), N, L.ptr(z16), N, M, N, K, ctypes.byref(ep), L.ptr(z16), L.ptr(_bf(mask)), t, 0,
                                            L.stream_ptr()), 'gemm_bf16')
            torch.cuda.synchronize()
            outs.append((Cf, z16))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert 0.4 < float((outs[1][0] == _bf(resid).float()).float().mean()) < 0.7          # masked (~50 %) or dropped (10 %) elements: the residual alone


@pytest.mark.parametrize('tile', [0, 64])      # 0: by shape (eight-wave workgroups with two K groups when the reduce dimension allows); 64: the four-wave kernel
@pytest.mark.parametrize('M,N,K,splits', [(512, 512, 40960, 0), (1536, 512, 8192, 0), (512, 1536, 8192, 4), (64, 64, 640, 0), (192, 64, 1280, 1), (512, 512, 4096, 1),
                                          (256, 128, 64 * 33, 1), (256, 384, 64 * 35, 2), (136, 72, 64 * 16, 0)])      # odd K-tile counts: the second K group sits out the last iteration
def test_weight_gradient_form_vs_torch(L, M, N, K, splits, tile):
    g = torch.Generator(device='cpu').manual_seed(M * 7 + N + K)
    dY = _bf(torch.randn(K, M, generator=g) * 0.1).cuda()
    X = _bf(torch.randn(K, N, generator=g)).cuda()
    want = dY.float().t() @ X.float()
    want_rs = dY.float().sum(0)
    C0 = torch.randn(M, N, generator=g).cuda()
    C = C0.clone()
    rs = torch.zeros(M, device='cuda')
    ep = L.GemmEpilogue()
    ep.accumulate = 1
    ep.a_rowsum = L.ptr(rs)
    L.check(L.lib().mansy_gemm_bf16(L.ptr(dY), M, 1, L.ptr(X), N, 1, L.ptr(C), N, None, 0, M, N, K, ctypes.byref(ep), None, None, tile, splits, L.stream_ptr()), 'gemm_bf16')
    torch.cuda.synchronize()
    scale = float(want.abs().max())
    assert float((C - C0 - want).abs().max()) <= 2e-5 * scale + 1e-5, float((C - C0 - want).abs().max())      # K up to 40 960 fp32 additions in another order (split-K atomics)
    assert float((rs - want_rs).abs().max()) <= 2e-5 * float(want_rs.abs().max()) + 1e-4


def test_bf16_image_arguments_are_validated(L):
    """include/mansy_hip.h: resid16 / mask16 belong to the forward form, take their leading dimensions from ep and must be 8-byte aligned; force_tile takes
    0 / 64 / 96 / 128; the weight-gradient form needs an accumulating epilogue.  Every violation is an error return with a message, never a launch."""
    M = N = K = 256
    A = _bf(torch.randn(M, K)).cuda()
    W = _bf(torch.randn(N, K)).cuda()
    C = torch.zeros(M, N, device='cuda')
    img = _bf(torch.randn(M, N + 8)).cuda()
    lib, st = L.lib(), L.stream_ptr()

    def call(ep, r16=None, m16=None, tile=0, ak=0, bk=0):
        return lib.mansy_gemm_bf16(L.ptr(A), K, ak, L.ptr(W), K, bk, L.ptr(C), N, None, 0, M, N, K, ctypes.byref(ep) if ep is not None else None, r16, m16, tile, 0, st)
    ep = L.GemmEpilogue()
    ep.resid_ld = ep.mask_ld = N + 8
    assert call(ep, L.ptr(img), L.ptr(img)) == 0                                    # (the well-formed call)
    torch.cuda.synchronize()
    assert call(None, L.ptr(img)) != 0                                               # an image without an epilogue to take its leading dimension from
    assert call(ep, ctypes.c_void_p(img.data_ptr() + 2)) != 0 and b'8-byte' in lib.mansy_last_error()     # misaligned image
    ep.resid_ld = N + 6
    assert call(ep, L.ptr(img)) != 0                                                 # leading dimension % 4
    ep.resid_ld = N + 8
    assert call(ep, tile=256) != 0 and call(ep, tile=32) != 0                        # tile codes that do not exist
    ep.accumulate = 1
    assert call(ep, L.ptr(img), ak=1, bk=1) != 0                                     # images on the weight-gradient form
    ep.accumulate = 0
    assert call(ep, ak=1, bk=1) != 0                                                 # weight-gradient form without an accumulating epilogue
    assert call(ep, ak=1, bk=0) != 0                                                 # mixed operand forms
