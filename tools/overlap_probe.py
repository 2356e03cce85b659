#!/usr/bin/env python3
"""Does an MFMA-bound product overlap with an HBM-bound row kernel when they are issued on two HIP streams?  (Feasibility probe
for pipelining two half-batches of the decoder recurrence: half A's products under half B's attention / LayerNorm kernels.)
Per 'decoder layer-step' of one half batch (2048 rows): 6 products [2048, 512] x [512, 512] and 5 LayerNorm-sized row passes.
  seq   : both halves on one stream, half after half            (what a plain split would cost)
  full  : the unsplit step: 6 products + 5 row passes on 4096 rows
  2str  : half A on stream 1, half B on stream 2, same submission order as a pipelined engine would use"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mansy_immersivevideostreaming_amd import kernels as K


def ln(a, b, lnw, lnb):
    return K.layernorm_fwd(a, b, lnw, lnb)


def work(x, w, lnw, lnb, y, n_gemm=6, n_ln=5):
    for i in range(max(n_gemm, n_ln)):
        if i < n_gemm:
            K.gemm(x, w, out=y)
        if i < n_ln:
            ln(y, x, lnw, lnb)


def timeit(fn, n=30):
    """fn captured once into a hipGraph (no host launch cost in the measurement), replayed n times."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    body, fn = fn, g.replay
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    d = 512
    w = torch.randn(d, d, device='cuda')
    lnw, lnb = torch.ones(d, device='cuda'), torch.zeros(d, device='cuda')
    full = [torch.randn(4096, d, device='cuda'), torch.empty(4096, d, device='cuda')]
    ha = [torch.randn(2048, d, device='cuda'), torch.empty(2048, d, device='cuda')]
    hb = [torch.randn(2048, d, device='cuda'), torch.empty(2048, d, device='cuda')]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    t_full = timeit(lambda: work(full[0], w, lnw, lnb, full[1]))
    t_seq = timeit(lambda: (work(ha[0], w, lnw, lnb, ha[1]), work(hb[0], w, lnw, lnb, hb[1])))

    def two():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        # interleaved submission: a product of half A, then a row pass of half B ...
        for i in range(6):
            with torch.cuda.stream(s1):
                K.gemm(ha[0], w, out=ha[1])
                if i < 5:
                    K.layernorm_fwd(ha[1], ha[0], lnw, lnb)
            with torch.cuda.stream(s2):
                K.gemm(hb[0], w, out=hb[1])
                if i < 5:
                    K.layernorm_fwd(hb[1], hb[0], lnw, lnb)
        cur.wait_stream(s1); cur.wait_stream(s2)
    t_two = timeit(two)
    t_g = timeit(lambda: [K.gemm(full[0], w, out=full[1]) for _ in range(6)])
    t_gh = timeit(lambda: [K.gemm(ha[0], w, out=ha[1]) for _ in range(6)])
    t_l = timeit(lambda: [K.layernorm_fwd(full[1], full[0], lnw, lnb) for _ in range(5)])
    t_lh = timeit(lambda: [K.layernorm_fwd(ha[1], ha[0], lnw, lnb) for _ in range(5)])
    print(f'6 products: full {t_g:.1f} us, half {t_gh:.1f} us | 5 row passes: full {t_l:.1f} us, half {t_lh:.1f} us')
    print(f'layer-step: unsplit {t_full:.1f} us | two halves on one stream {t_seq:.1f} us | two halves on two streams {t_two:.1f} us')


if __name__ == '__main__':
    main()
