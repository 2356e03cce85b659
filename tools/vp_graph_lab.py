#!/usr/bin/env python3
"""LAB (round 5): is the small-batch VP train step HOST-bound?  The step is one library call that enqueues 523 (B = 32, T = 10) .. 1424 (two-stream, T = 15)
launches; the host needs ~4.5 us per launch.  Here the call is captured once into a hipGraph (torch.cuda.CUDAGraph around the SAME library call; seed,
Adam step and MTIO mix frozen -- a timing experiment, not the product form) and replayed, single-stream and with the forced two-stream decoder split.
    python3 tools/vp_graph_lab.py [B,S,T ...]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import synthetic_trajectories
from mansy_immersivevideostreaming_amd._lib import lib
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW

dev = torch.device('cuda', 0)
cases = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]] or [(32, 10, 10), (128, 5, 15), (256, 5, 15), (512, 5, 15), (1024, 5, 15)]
N = 30


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    n0 = lib().mansy_prof_launch_count()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    th = time.perf_counter() - t0
    n1 = lib().mansy_prof_launch_count()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3, th / N * 1e3, (n1 - n0) / N


for B, S, T in cases:
    for two in (False, True):
        if two and (B % 2 or B < 256):
            continue
        torch.manual_seed(5); random.seed(5); np.random.seed(5)
        m = ViewportTransformerMTIO(in_channel=2, fut_window=T, d_model=512, dim_feedforward=512, device=dev).to(dev)
        m.train()
        m.two_stream = two
        m._mix_decision = lambda B_: None            # no host-side permutation upload inside the captured region
        seed = [1234]
        m._next_seed = lambda: seed[0]
        opt = FusedAdamW(m, lr=1e-4)
        h, c, f = (t.to(dev) for t in synthetic_trajectories(B, S, T, seed=5))
        ms_e, host_e, n_e = timed(lambda: m.train_step(h, c, f, opt))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                m.train_step(h, c, f, opt)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=s):
                loss = m.train_step(h, c, f, opt)
            ms_g, host_g, _ = timed(g.replay)
            print(f'B={B:5d} S={S} T={T} two_stream={int(two)}: eager {ms_e:.3f} ms (host {host_e:.3f}, {n_e:.0f} launches)  graph replay {ms_g:.3f} ms '
                  f'(host {host_g:.3f})  ratio {ms_g / ms_e:.3f}  loss {float(loss):.5f}', flush=True)
        except Exception as e:       # noqa: BLE001
            print(f'B={B} S={S} T={T} two_stream={int(two)}: eager {ms_e:.3f} ms; capture failed: {type(e).__name__}: {str(e)[:300]}', flush=True)
