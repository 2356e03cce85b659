"""Two-stream half-batch decoder (model.two_stream = True) against the single-stream run, dropout ON: every named workspace slab of one
train step compared bit for bit (tests/test_gpu_vp_engine.py::test_two_stream_half_batch_decoder_equals_single_stream is the assertion form).
    python tools/vp_split_probe.py"""
import os, sys, random
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R)
import numpy as np, torch
from mansy_immersivevideostreaming_amd.viewport_prediction.models import mtio as MT
from oracle import vp_oracle as vo
B, d = 512, 256
h, c, f = (t.cuda() for t in vo.synthetic_trajectories(B, 10, 10, seed=9))
names = ['dec.emb', 'dec0.qkv', 'dec0.P1', 'dec0.ao1', 'dec0.z1', 'dec0.y1', 'dec0.qc', 'dec0.P2', 'dec0.ao2', 'dec0.z2', 'dec0.y2', 'dec0.h', 'dec0.z3', 'dec0.y3',
         'dec1.qkv', 'dec1.ao1', 'dec1.z1', 'dec1.h', 'dec1.z3', 'dec.out', 'tok_all', 'dec1.dbr3', 'dec1.da', 'dec1.dbr2', 'dec1.dqc', 'dec1.dbr1', 'dec1.dqkv', 'dec0.dqkv', 'dec.dE']
out = {}
for split in ('0', '1'):
    m = MT.ViewportTransformerMTIO(in_channel=2, fut_window=10, d_model=d, dim_feedforward=d, device='cuda', seed=11)
    m.two_stream = split == '1'
    m.load_state_dict(vo.make_state_dict(d, 4, bias=True)); m = m.to('cuda').train()
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    opt = MT.FusedAdamW(m, lr=1e-4)
    loss = m.train_step(h, c, f, opt).item()
    cfg = m._cfg(B, 10)
    out[split] = (loss, {n: m.ws_tensor(cfg, n).clone() for n in names})
print('loss', out['0'][0], out['1'][0])
for n in names:
    a, b = out['0'][1][n], out['1'][1][n]
    diff = (a - b).abs()
    nz = int((diff > 0).sum())
    print(f'{n:10s} numel {a.numel():9d} differing {nz:8d} max {diff.max().item():.3e}', ('first idx %d' % int((diff > 0).flatten().nonzero()[0])) if nz else '')
