#!/usr/bin/env python3
"""Per-shape timing of the MFMA GEMM on the shapes the VP train step launches (B=4096).
    python tools/gemm_bench.py [tile ...] [--prec f32|bf16x3|bf16x6 ...] [--check]
--check additionally reports the max error of every (shape, mode) against a float64 product, relative to max|C|."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lab_knobs as KN  # noqa: E402  (ABI 8: the A/B knobs live in the -DMANSY_LAB build only; entered lazily, by the first knob that is set)
import torch
from mansy_immersivevideostreaming_amd import kernels as K

SHAPES = [  # (name, a_kmajor, b_kmajor, M, N, K, accumulate, count per step)
    ('enc qkv fwd NT', 0, 0, 40960, 1536, 512, 0, 2), ('enc 512 fwd NT', 0, 0, 40960, 512, 512, 0, 6), ('conv fwd NT', 0, 0, 40960, 512, 1536, 0, 1),
    ('memkv NT', 0, 0, 20480, 1024, 512, 0, 2), ('dec qkv NT', 0, 0, 4096, 1536, 512, 0, 20), ('dec 512 NT', 0, 0, 4096, 512, 512, 0, 100),
    ('dec dX NN', 0, 1, 4096, 512, 512, 0, 100), ('dec qkv dX NN', 0, 1, 4096, 512, 1536, 0, 20), ('enc dX NN', 0, 1, 40960, 512, 512, 0, 6),
    ('enc qkv dX NN', 0, 1, 40960, 512, 1536, 0, 2), ('conv dX NN', 0, 1, 40960, 1536, 512, 0, 1),
    ('dW 512x512 TN', 1, 1, 512, 512, 40960, 1, 16), ('dW 1536x512 TN', 1, 1, 1536, 512, 40960, 1, 4), ('dW conv 512x1536 TN', 1, 1, 512, 1536, 40960, 1, 1),
    ('dW kv 1024x512 TN', 1, 1, 1024, 512, 20480, 1, 2),
]
args = sys.argv[1:]
CHECK = '--check' in args
VARIANT = None
if '--variant' in args:
    i = args.index('--variant')
    VARIANT = int(args[i + 1])
    del args[i:i + 2]
PLANES = '--planes' in args          # forward / dX shapes with the weight pre-split into bf16 planes (LDS-DMA B)
args = [a for a in args if a != '--planes']
precs = []
while '--prec' in args:
    i = args.index('--prec')
    precs.append(args[i + 1])
    del args[i:i + 2]
precs = precs or ['f32']
tiles = [int(x) for x in args if x != '--check'] or [0]
tiles = [(t, pr) for pr in precs for t in tiles]
if VARIANT is not None:
    from mansy_immersivevideostreaming_amd._lib import lib
    KN.bf16_variant(VARIANT)
tot = {t: 0.0 for t in tiles}
for name, ak, bk, M, N, Kd, acc, cnt in SHAPES:
    A = torch.randn((Kd, M) if ak else (M, Kd), device='cuda')
    B = torch.randn((Kd, N) if bk else (N, Kd), device='cuda')
    out = torch.zeros(M, N, device='cuda')
    line = f'{name:22s} M={M:6d} N={N:5d} K={Kd:6d}'
    ref = None
    if CHECK:
        Ad, Bd = A.double(), B.double()
        ref = (Ad.t() if ak else Ad) @ (Bd if bk else Bd.t())
    for tp in tiles:
        t, pr = tp
        K.set_precision(pr)
        if CHECK:
            out.zero_()
            K.gemm(A, B, bool(ak), bool(bk), out=out, accumulate=bool(acc), force_tile=t)
            err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
        use_pl = PLANES and not ak and pr != 'f32'
        if use_pl:
            Wm = B                                          # the weight behind this operand: [N, K] (forward) or [K, N] (dX: C = A W)
            pl, pl_t = K.weight_planes(Wm, 2 if pr == 'bf16x3' else 3)
            run = (lambda: K.gemm_planes(A, Wm, pl_t if bk else pl, transposed=bool(bk), force_tile=t))
        else:
            run = (lambda: K.gemm(A, B, bool(ak), bool(bk), out=out, accumulate=bool(acc), force_tile=t))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        tf = 2.0 * M * N * Kd / us / 1e6
        tot[tp] += us * cnt
        line += f' | {pr} tile {t:3d}: {us:8.1f} us {tf:6.1f} TF' + (f' err {err:.1e}' if CHECK else '')
    print(line)
K.set_precision('f32')
print('per-step GEMM total (ms):', {f'{pr}/{t}': round(v / 1e3, 2) for (t, pr), v in tot.items()})
