#!/usr/bin/env python3
"""One GEMM shape a few times (for `rocprofv3 --pmc ...`): python3 tools/gemm_pmc.py M N K a_kmajor b_kmajor [accumulate [precision [planes]]]
precision: f32 | bf16x3 | bf16x6; planes = 1: the weight operand pre-split into bf16 planes (the engine's forward / dX form)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lab_knobs as KN  # noqa: E402  (ABI 8: the A/B knobs live in the -DMANSY_LAB build only; entered lazily, by the first knob that is set)
import torch
from mansy_immersivevideostreaming_amd import kernels as K
M, N, Kd, ak, bk = (int(x) for x in sys.argv[1:6])
acc = len(sys.argv) > 6 and sys.argv[6] == '1'
prec = sys.argv[7] if len(sys.argv) > 7 else 'f32'
planes = len(sys.argv) > 8 and sys.argv[8] == '1'
A = torch.randn((Kd, M) if ak else (M, Kd), device='cuda'); B = torch.randn((Kd, N) if bk else (N, Kd), device='cuda')
out = torch.zeros(M, N, device='cuda')
K.set_precision(prec)
if os.environ.get('MANSY_BF16_VARIANT'):
    from mansy_immersivevideostreaming_amd._lib import lib
    KN.bf16_variant(int(os.environ['MANSY_BF16_VARIANT']))
FT = int(os.environ.get('MANSY_FORCE_TILE', '0'))
if os.environ.get('MANSY_COL_GROUP'):
    from mansy_immersivevideostreaming_amd._lib import lib
    KN.col_group(int(os.environ['MANSY_COL_GROUP']))
if planes and not ak and prec != 'f32':
    pl, pl_t = K.weight_planes(B, 2 if prec == 'bf16x3' else 3)
    run = lambda: K.gemm_planes(A, B, pl_t if bk else pl, transposed=bool(bk), force_tile=FT)
else:
    run = lambda: K.gemm(A, B, bool(ak), bool(bk), out=out, accumulate=acc)
for _ in range(6):
    run()
torch.cuda.synchronize()
