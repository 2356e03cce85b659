#!/usr/bin/env python3
"""Does hipGraph replay of sample() (eval: no per-call scalars change) beat direct launches at small batch?"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO
import bench
torch.manual_seed(5); random.seed(5); np.random.seed(5)
m = ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=512, dim_feedforward=512, device='cuda').to('cuda').eval()

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for B in (32, 512, 4096):
    h, c, f = (t.cuda() for t in bench.synthetic_trajectories(B, 10, 10, seed=5))
    with torch.no_grad():
        ref = m.sample(h, c)
        t_direct = timeit(lambda: m.sample(h, c))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m.sample(h, c)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m.sample(h, c)
        t_graph = timeit(lambda: g.replay())
        g.replay(); torch.cuda.synchronize()
        print(f'B={B}: direct {t_direct:.3f} ms, graph {t_graph:.3f} ms, equal {torch.equal(out, ref)}')
