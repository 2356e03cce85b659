#!/usr/bin/env python3
"""MPC expert throughput: look-ahead decisions/s of the device search (all environments per launch) next to the sequential C
oracle on one host core, on the synthetic bench-shaped tables."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mansy_immersivevideostreaming_amd.bitrate_selection.envs import expert_env as X  # noqa: E402
from oracle import env as oenv  # noqa: E402

FIELDS = X.EnvTables.FIELDS


def main():
    T = X.EnvTables.synthetic('cuda', seed=5, train_identifier_reward=False)
    OT = oenv.EnvTables({k: T.host[k] for k in FIELDS}, T.host['qoe_w'], train_identifier_reward=False)
    out = []
    for horizon, n_env in [(2, 256), (3, 256), (4, 256), (5, 256), (4, 1), (5, 1), (6, 1)]:
        venv = X.ExpertVecEnv(T, n_env, horizon, seed=0)
        venv.reset()
        for _ in range(3):
            venv.step(venv.choose_action())
        torch.cuda.synchronize()
        reps = 20 if horizon < 5 else 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            venv.choose_action()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        plans = 15 ** horizon
        rec = dict(horizon=horizon, n_env=n_env, ms_per_call=round(ms, 4), decisions_per_s=round(n_env / ms * 1e3, 1),
                   plans_per_s=round(n_env * plans / ms * 1e3, 0))
        if horizon <= 5 and n_env == 256 or horizon == 4:
            ex = oenv.Expert(OT, venv.cache.vp_video.cpu().numpy(), horizon)
            oe = oenv.Env(OT, seed=0, worker_num=n_env)
            oe.reset()
            k = max(1, int(2e6 // plans))
            t0 = time.perf_counter()
            for _ in range(k):
                ex.choose_action(oe)
            dt = (time.perf_counter() - t0) / k
            rec['cpu_oracle_decisions_per_s_1core'] = round(1 / dt, 2)
        out.append(rec)
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
