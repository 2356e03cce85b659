"""CPU study for the split-bf16 product modes: every weight product of the VP oracle is replaced by an
emulation of the bf16x3 / bf16x6 MFMA product -- operands split into bf16 terms by round-to-nearest-even
(a = a0 + a1 (+ a2)), the 3 (a0b0 + a0b1 + a1b0) or 6 (+ a1b1 + a0b2 + a2b0) bf16 products summed in a wide accumulator --
forward AND backward (dX and dW products are split products too), and the result is compared with the goldens captured
from the imported reference.  Answers: which variant keeps the 1e-4 bar / the bit-exact tile decisions.

    python tools/bf16_split_study.py            # all goldens, modes f32 / bf16x3 / bf16x6 / bf16x1
"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vp_oracle as vo          # noqa: E402
from oracle import tilemap as otm           # noqa: E402


def split(x, n):
    out = []
    r = x.double()
    for _ in range(n):
        t = r.float().bfloat16()
        out.append(t.double())
        r = r - t.double()
    return out


def split_mm(a, b, mode):
    """a [.., K] @ b [K, N] as the split product; accumulation in f64 (the MFMA accumulates the bf16 products in f32: same order
    of rounding as the exact-f32 path, not what is studied here)."""
    if mode == 'f32':
        return torch.matmul(a, b)
    n = {'bf16x1': 1, 'bf16x3': 2, 'bf16x6': 3}[mode]
    A, B = split(a, n), split(b, n)
    if mode == 'bf16x1':
        pairs = [(0, 0)]
    elif mode == 'bf16x3':
        pairs = [(0, 0), (0, 1), (1, 0)]
    else:
        pairs = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]
    acc = 0
    for i, j in pairs:
        acc = acc + torch.matmul(A[i], B[j])
    return acc.float()


class SplitMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, mode):
        ctx.save_for_backward(a, b)
        ctx.mode = mode
        return split_mm(a, b, mode)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        a2 = a.reshape(-1, a.shape[-1])
        g2 = g.reshape(-1, g.shape[-1])
        da = split_mm(g, b.t(), ctx.mode)
        db = split_mm(a2.t(), g2, ctx.mode)
        return da, db, None


MODE = ['f32']
_orig_matmul = torch.Tensor.__matmul__


def _patched(self, other):
    if MODE[0] != 'f32' and other.dim() == 2 and self.dim() <= 3:      # weight products only (attention operands are 4-D)
        return SplitMM.apply(self, other, MODE[0])
    return _orig_matmul(self, other)


torch.Tensor.__matmul__ = _patched


def study(path):
    z = np.load(path, allow_pickle=False)
    T = int(z['T'])
    sd = vo.make_state_dict(int(z['d']), int(z['wseed']), bias=bool(z['bias']))
    h, c, f = (torch.from_numpy(z[k]) for k in ('history', 'current', 'future'))
    rows = []
    for mode in ('f32', 'bf16x6', 'bf16x3', 'bf16x1'):
        MODE[0] = mode
        with torch.no_grad():
            samp = vo.VPOracle(sd, fut_window=T).sample(h, c)
        e_samp = float(np.abs(samp.numpy() - z['eval_sample']).max())
        tm_ref = otm.tilemap_xy(z['eval_sample'].reshape(-1, 2))
        tm = otm.tilemap_xy(samp.numpy().reshape(-1, 2))
        tile_mismatch = int((tm != tm_ref).sum())
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
                  if v.dtype.is_floating_point and 'running_' not in k and k != 'positional_embedding.pe'}
        full = dict(sd)
        full.update(params)
        orc = vo.VPOracle(full, fut_window=T)
        src, cur, gt = vo.mtio_mix(h, c, f, 3, repeat=True, perms=None)
        pred = orc.process_src_current(src, cur, train=True)
        loss = orc.loss_function(pred, gt)
        loss.backward()
        e_pred = float(np.abs(pred.detach().numpy() - z['train_rep_pred']).max())
        e_loss = abs(loss.item() - float(z['train_rep_loss'])) / abs(float(z['train_rep_loss']))
        worst = 0.0
        for key in z.files:
            if key.startswith('train_rep_grad::') or key.startswith('train_rep_gradslice::'):
                k = key.split('::')[1]
                g = params[k].grad
                ref = z[key]
                got = g.numpy() if key.startswith('train_rep_grad::') else g.reshape(g.shape[0], -1)[::37, ::41].numpy()
                if np.abs(ref).max() < 1e-7:
                    continue
                worst = max(worst, float(np.abs(got - ref).max() / np.abs(ref).max()))
        rows.append((mode, e_samp, tile_mismatch, e_pred, e_loss, worst))
    MODE[0] = 'f32'
    return rows


if __name__ == '__main__':
    paths = sorted(p for p in glob.glob(os.path.join(ROOT, 'tests', 'golden', 'vp_*.npz')) if 'vp_loop_' not in p)
    print('%-28s %-7s %12s %6s %12s %12s %14s' % ('golden', 'mode', 'sample abs', 'tiles', 'pred abs', 'loss rel', 'grad rel(max)'))
    for p in paths:
        for r in study(p):
            print('%-28s %-7s %12.3e %6d %12.3e %12.3e %14.3e' % ((os.path.basename(p)[:-4],) + r))
