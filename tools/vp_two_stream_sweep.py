#!/usr/bin/env python3
"""Where does the two-stream decoder split pay?  ms per train step and per sample() over batch sizes, one stream vs two (round 5: the split
doubles the launch count; below some batch the step is a latency-bound chain and the second stream only adds host / queue work)."""
import json
import os
import random
import sys
import time

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from mansy_immersivevideostreaming_amd._lib import lib
from mansy_immersivevideostreaming_amd.viewport_prediction.models import ViewportTransformerMTIO, FusedAdamW

dev = torch.device('cuda', 0)
rows = []
for (S, T) in ((10, 10), (5, 15)):
    for B in (int(x) for x in (sys.argv[1:] or [256, 512, 1024, 2048, 4096])):
        for two in (False, True):
            torch.manual_seed(5); random.seed(5); np.random.seed(5)
            m = ViewportTransformerMTIO(in_channel=2, fut_window=T, d_model=512, dim_feedforward=512, device=dev).to(dev)
            m.train()
            m.two_stream = two
            opt = FusedAdamW(m, lr=1e-4)
            h, c, f = (t.to(dev) for t in bench.synthetic_trajectories(B, S, T, seed=5))
            for _ in range(4):
                m.train_step(h, c, f, opt)
            torch.cuda.synchronize()
            n0 = lib().mansy_prof_launch_count(); t0 = time.perf_counter()
            steps = 20
            for _ in range(steps):
                m.train_step(h, c, f, opt)
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            n1 = lib().mansy_prof_launch_count()
            m.eval()
            with torch.no_grad():
                for _ in range(3):
                    m.sample(h, c)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(steps):
                    m.sample(h, c)
                torch.cuda.synchronize()
                ms_s = (time.perf_counter() - t1) / steps * 1e3
            rows.append(dict(S=S, T=T, B=B, two_stream=two, ms_per_step=round(ms, 3), host_enqueue_ms=round(t_host / steps * 1e3, 3),
                             launches=(n1 - n0) // steps, sample_ms=round(ms_s, 3)))
            print(json.dumps(rows[-1]), flush=True)
            del m, opt
            torch.cuda.empty_cache()
