#!/usr/bin/env python3
"""LAB (round 5): do k independent latency-bound VP train-step chains overlap on one MI355X when k host threads enqueue them on k streams?
k models of B / k trajectories each (hist 5, pred 15), one Python thread per model (ctypes releases the GIL for the duration of the library
call), against one model at B.  Prices a k-way row split of the decoder recurrence with one enqueueing thread per part before building it.
    python3 tools/vp_concurrent_lab.py"""
import os, sys, time, random, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import synthetic_trajectories
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW

dev = torch.device('cuda', 0)
N = 30
S, T = 5, 15


def make(B, seed):
    torch.manual_seed(seed); random.seed(seed); np.random.seed(seed)
    m = ViewportTransformerMTIO(in_channel=2, fut_window=T, d_model=512, dim_feedforward=512, device=dev).to(dev)
    m.train()
    m.two_stream = False
    m._mix_decision = lambda B_: None
    m._next_seed = lambda: 1234
    opt = FusedAdamW(m, lr=1e-4)
    data = tuple(t.to(dev) for t in synthetic_trajectories(B, S, T, seed=seed))
    return m, opt, data


def run(models, n):
    bar = threading.Barrier(len(models) + 1)

    def work(m, opt, data, stream):
        torch.cuda.set_device(0)
        with torch.cuda.stream(stream):
            bar.wait()
            for _ in range(n):
                m.train_step(*data, opt)
            stream.synchronize()
        bar.wait()
    streams = [torch.cuda.Stream() for _ in models]
    th = [threading.Thread(target=work, args=(m, o, d, s)) for (m, o, d), s in zip(models, streams)]
    for t in th:
        t.start()
    torch.cuda.synchronize()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    return dt / n * 1e3


for Btot in (512, 1024, 256):
    one = [make(Btot, 5)]
    run(one, 3)
    ms1 = run(one, N)
    line = f'B={Btot}: 1 chain {ms1:.3f} ms'
    for k in (2, 4, 8):
        if Btot // k < 32:
            continue
        ms = [make(Btot // k, 5 + j) for j in range(k)]
        run(ms, 3)
        msk = run(ms, N)
        line += f' | {k} chains of {Btot // k}: {msk:.3f} ms'
        del ms
    print(line, flush=True)
    del one
