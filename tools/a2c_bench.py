#!/usr/bin/env python3
"""A2C baseline throughput: env-steps/s of full collect + update cycles with run_simple_rl.py's defaults scaled to
256 device environments (16 steps per collect = 4096 transitions, batch 256, repeat 2) on the synthetic bench tables."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mansy_immersivevideostreaming_amd.bitrate_selection.envs.mansy_env import EnvTables
from mansy_immersivevideostreaming_amd.bitrate_selection.envs.simple_rl_env import SimpleRLVecEnv
from mansy_immersivevideostreaming_amd.bitrate_selection.models import simple_rl as m

torch.manual_seed(1); np.random.seed(1)
dev = 'cuda'
fn = m.FeatureNet(8, 64, 5, device=dev)
actor, critic = m.Actor(fn, 640, 15, dev), m.Critic(fn, 640, dev)
for mod in list(actor.modules()) + list(critic.modules()):
    if isinstance(mod, torch.nn.Linear):
        torch.nn.init.orthogonal_(mod.weight, gain=np.sqrt(2)); torch.nn.init.zeros_(mod.bias)
optim = torch.optim.RMSprop(torch.nn.ModuleList([actor, critic]).parameters(), lr=1e-4)
pol = m.A2CPolicy(actor, critic, optim, None, discount_factor=0.99, gae_lambda=0.95, max_grad_norm=0.5, vf_coef=0.5, ent_coef=0.1,
                  reward_normalization=True, action_space=15).to(dev)
N, T = 256, 16
venv = SimpleRLVecEnv(EnvTables.synthetic(dev, seed=5, train_identifier_reward=True, n_sample=max(240, N)), N, seed=1)
col, buf = m.A2CCollector(pol, venv), m.A2CBuffer(T, N, dev)

def cycle():
    col.collect(T * N, buf)
    return pol.update(0, buf, batch_size=256, repeat=2)
for _ in range(3): cycle()
torch.cuda.synchronize()
cycles = 20
t0 = time.perf_counter()
for _ in range(cycles): res = cycle()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({'metric': 'A2C env-steps/sec', 'value': round(cycles * N * T / dt, 1), 'ms_per_cycle': round(dt / cycles * 1e3, 3),
                  'final_loss': float(np.mean(res['loss'])), 'workload': f'{N} envs x {T} steps per collect, batch 256, repeat 2 (32 minibatch steps per cycle)'}))
