"""Pins oracle/vp_oracle.py (the CPU restatement) against golden vectors produced by the
IMPORTED reference (tools/gen_golden_vp.py): eval forward, sample(), train forward with both
MTIO branches, loss, full gradients, BN running stats and one AdamW step."""
import glob
import os
import numpy as np
import pytest
import torch

from oracle import vp_oracle as vo

GOLD = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'vp_*.npz')) if 'vp_loop_' not in os.path.basename(p))


def _load(path):
    z = np.load(path, allow_pickle=False)
    sd = vo.make_state_dict(int(z['d']), int(z['wseed']), bias=bool(z['bias']))
    return z, sd


@pytest.mark.parametrize('path', GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_eval_and_sample(path):
    z, sd = _load(path)
    orc = vo.VPOracle(sd, fut_window=int(z['T']))
    h, c = torch.from_numpy(z['history']), torch.from_numpy(z['current'])
    with torch.no_grad():
        pred = orc.process_src_current(torch.cat([h] * 3, -1), torch.cat([c] * 3, -1), train=False)
        samp = orc.sample(h, c)
    # fp32, tolerance from north_star: 1e-4 (observed ~1e-6)
    np.testing.assert_allclose(pred.numpy(), z['eval_pred'], atol=2e-5, rtol=0)
    np.testing.assert_allclose(samp.numpy(), z['eval_sample'], atol=2e-5, rtol=0)


@pytest.mark.parametrize('branch', ['rep', 'mix'])
@pytest.mark.parametrize('path', GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_train_forward_backward_adamw(path, branch):
    z, sd = _load(path)
    T = int(z['T'])
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and 'running_' not in k and k != 'positional_embedding.pe'}
    full = dict(sd)
    full.update(params)
    orc = vo.VPOracle(full, fut_window=T)
    h, c, f = (torch.from_numpy(z[k]) for k in ('history', 'current', 'future'))
    perms = z[f'train_{branch}_perms']
    src, cur, gt = vo.mtio_mix(h, c, f, 3, repeat=(branch == 'rep'), perms=perms)
    np.testing.assert_array_equal(gt.numpy(), z[f'train_{branch}_gt'])
    pred = orc.process_src_current(src, cur, train=True)
    loss = orc.loss_function(pred, gt)
    loss.backward()
    np.testing.assert_allclose(pred.detach().numpy(), z[f'train_{branch}_pred'], atol=2e-5, rtol=0)
    np.testing.assert_allclose(loss.item(), float(z[f'train_{branch}_loss']), atol=1e-6, rtol=1e-5)
    names = [str(s) for s in z[f'train_{branch}_gradnames']]
    norms = z[f'train_{branch}_gradnorms']
    assert sorted(params) == names
    for k, n in zip(names, norms):
        g = params[k].grad
        assert g is not None, k
        assert abs(g.norm().item() - n) <= 1e-4 * max(n, 1e-3) + 1e-7, (k, g.norm().item(), n)
    for key in z.files:
        if key.startswith(f'train_{branch}_grad::'):
            k = key.split('::')[1]
            ref = z[key]
            tol = 1e-4 * np.abs(ref).max() + 1e-6  # conv bias grad is ~0 under train BN
            np.testing.assert_allclose(params[k].grad.numpy(), ref, atol=tol, rtol=0, err_msg=k)
        if key.startswith(f'train_{branch}_gradslice::'):
            k = key.split('::')[1]
            g = params[k].grad
            ref = z[key]
            tol = 1e-4 * np.abs(ref).max() + 1e-6  # conv bias grad is ~0 under train BN
            np.testing.assert_allclose(g.reshape(g.shape[0], -1)[::37, ::41].numpy(), ref, atol=tol, rtol=0)
    bn_mean, bn_var = orc.last_bn_stats
    np.testing.assert_allclose(bn_mean.numpy(), z[f'train_{branch}_bn_mean'], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(bn_var.numpy(), z[f'train_{branch}_bn_var'], atol=1e-6, rtol=1e-5)
    for key in z.files:
        if key.startswith(f'train_{branch}_adamw::'):
            k = key.split('::')[1]
            p, g = params[k].detach(), params[k].grad
            p1, _, _ = vo.adamw_step(p, g, torch.zeros_like(p), torch.zeros_like(p), step=1)
            np.testing.assert_allclose(p1.numpy(), z[key], atol=2e-7, rtol=1e-6, err_msg=k)
