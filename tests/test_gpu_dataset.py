"""V1 on the GPU against the capture of the imported reference (tests/golden/dataset_reference.npz, tools/gen_golden_dataset.py):
the fixture's eight real Jin2022 traces go into the HBM table, `mansy_traj_gather` (through ViewportDataset.gather and the
DeviceLoader) produces the windows, and every recorded reference item -- history / current / future of
ViewportDataset.__getitem__, load_dataset.py:43-52 -- must come back bit for bit, for both window / step / trim settings and all
five splits.  Also: the loud failure for windows that would leave their trace (the reference fails in collate)."""
import ast
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

Z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'dataset_reference.npz'))
INCLUDE = ['train', 'valid', 'test', 'test_seen', 'test_unseen']


@pytest.fixture(scope='module')
def tree(tmp_path_factory):
    if not torch.cuda.is_available():
        pytest.fail('GPU tests need a ROCm device (no CPU fallback exists)')
    root = str(tmp_path_factory.mktemp('jin'))
    for key in Z.files:
        if key.startswith('trace/'):
            _, v, u = key.split('/')
            d = os.path.join(root, f'video{v}', '5Hz')
            os.makedirs(d, exist_ok=True)
            np.save(os.path.join(d, f'simple_5Hz_user{u}.npy'), Z[key])
    return root


def _sets(tree, tag):
    from mansy_immersivevideostreaming_amd.viewport_prediction.utils.common import Config
    from mansy_immersivevideostreaming_amd.viewport_prediction.utils.load_dataset import create_dataset
    vsplit, usplit = ast.literal_eval(str(Z['vsplit'])), ast.literal_eval(str(Z['usplit']))
    S, T, step, th, tt = (int(x) for x in Z[f'{tag}/params'])
    config = Config(dict(viewport_datasets_dir={'Jin2022': tree}, video_split={'Jin2022': vsplit}, user_split={'Jin2022': usplit},
                         trim_head=99, trim_tail=99, frequency=5, sample_step=99))
    return S, T, create_dataset('Jin2022', config, his_window=S, fut_window=T, frequency=5, sample_step=step, trim_head=th, trim_tail=tt,
                                dataset_video_split=dict(vsplit), dataset_user_split=dict(usplit))


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_device_gather_equals_reference_items(tree, tag):
    from mansy_immersivevideostreaming_amd.viewport_prediction.utils.load_dataset import DeviceLoader
    S, T, sets = _sets(tree, tag)
    for name, ds in zip(INCLUDE, sets):
        ref_idx = Z[f'{tag}/{name}/indices']
        pick = Z[f'{tag}/{name}/pick']
        h, c, f, v, u, t = ds.gather(pick, 'cuda')                       # ONE mansy_traj_gather launch for the picked samples
        assert h.shape == (len(pick), S, 2) and c.shape == (len(pick), 1, 2) and f.shape == (len(pick), T, 2)
        for j, i in enumerate(pick):
            for got, key in ((h, 'history'), (c, 'current'), (f, 'future')):
                np.testing.assert_array_equal(got[j].cpu().numpy(), Z[f'{tag}/{name}/item{int(i)}/{key}'], err_msg=f'{name} item {i} {key}')
            assert (int(v[j]), int(u[j]), int(t[j])) == tuple(int(x) for x in ref_idx[int(i)])
        # the whole split through the DeviceLoader (batches of 64, in order): ids of every sample and the windows of the recorded items
        seen, items = [], {}
        for hb, cb, fb, vb, ub, tb in DeviceLoader(ds, 64, shuffle=False, device='cuda'):
            base = len(seen)
            seen += list(zip(vb.tolist(), ub.tolist(), tb.tolist()))
            for i in pick:
                if base <= int(i) < base + hb.shape[0]:
                    items[int(i)] = (hb[int(i) - base].cpu().numpy(), cb[int(i) - base].cpu().numpy(), fb[int(i) - base].cpu().numpy())
        np.testing.assert_array_equal(np.array(seen, np.int64).reshape(-1, 3), ref_idx, err_msg=name)
        for i, (hh, cc, ff) in items.items():
            np.testing.assert_array_equal(hh, Z[f'{tag}/{name}/item{i}/history'])
            np.testing.assert_array_equal(cc, Z[f'{tag}/{name}/item{i}/current'])
            np.testing.assert_array_equal(ff, Z[f'{tag}/{name}/item{i}/future'])


def test_windows_leaving_their_trace_fail_loudly(tree):
    """trim_tail < fut_window (or trim_head < his_window) makes ragged windows: the reference fails in collate; here the device table
    refuses to build instead of gathering across the trace boundary."""
    from mansy_immersivevideostreaming_amd._lib import MansyError
    from mansy_immersivevideostreaming_amd.viewport_prediction.utils.common import Config
    from mansy_immersivevideostreaming_amd.viewport_prediction.utils.load_dataset import create_dataset
    vsplit, usplit = ast.literal_eval(str(Z['vsplit'])), ast.literal_eval(str(Z['usplit']))
    config = Config(dict(viewport_datasets_dir={'Jin2022': tree}, video_split={'Jin2022': vsplit}, user_split={'Jin2022': usplit},
                         trim_head=15, trim_tail=15, frequency=5, sample_step=5))
    for S, T, th, tt in ((10, 20, 15, 15), (10, 10, 5, 15)):
        ds = create_dataset('Jin2022', config, his_window=S, fut_window=T, trim_head=th, trim_tail=tt, dataset_video_split=dict(vsplit),
                            dataset_user_split=dict(usplit), include=['train'])[0]
        with pytest.raises(MansyError):
            ds.to_device('cuda')


@pytest.mark.parametrize('layout', ['bias', 'nobias'])
def test_c1_real_jin2022_batch_through_train_step_and_sample(tree, layout):
    """BASELINE configs[0] as stated: B = 32 real Jin2022 windows (hist 10, pred 10, step 5, trim 15/15), d = 512, 2+2 layers.
    The batch is rebuilt HERE from the fixture's traces -- HBM trace table -> DeviceLoader(shuffle=True) under the reference's
    seed -> mansy_traj_gather -- and must be the batch the imported DataLoader produced (ids and windows bit for bit); then one
    fused train_step (MTIO decision from the same host RNG stream, dropout off, AdamW) and sample() against the imported model
    (tests/golden/vp_c1_jin2022_b32_*.npz, tools/gen_golden_vp_c1.py): loss, updated weights, BatchNorm running statistics,
    sample() before and after the step within 1e-4, tile maps of it bit-equal."""
    import random
    from oracle import vp_oracle as vo
    from mansy_immersivevideostreaming_amd import kernels
    from mansy_immersivevideostreaming_amd.viewport_prediction.models import mtio
    from mansy_immersivevideostreaming_amd.viewport_prediction.utils.load_dataset import DeviceLoader
    G = np.load(os.path.join(os.path.dirname(__file__), 'golden', f'vp_c1_jin2022_b32_{layout}.npz'))
    S, T, sets = _sets(tree, 'a')
    train = sets[0]
    assert (S, T) == (int(G['S']), int(G['T'])) and len(train) == int(G['n_train'])
    torch.manual_seed(int(G['loader_seed']))
    for bi, (h, c, f, v, u, t) in enumerate(DeviceLoader(train, int(G['B']), shuffle=True, device='cuda')):
        if bi == int(G['batch_index']):
            break
    np.testing.assert_array_equal(np.stack([v.numpy(), u.numpy(), t.numpy()], 1), G['ids'])
    for got, key in ((h, 'history'), (c, 'current'), (f, 'future')):
        np.testing.assert_array_equal(got.cpu().numpy(), G[key])
    d, bias = int(G['d']), bool(G['bias'])
    sd = vo.make_state_dict(d, int(G['wseed']), bias=bias)
    for branch in ('rep', 'mix'):
        m = mtio.ViewportTransformerMTIO(in_channel=2, fut_window=T, d_model=d, dim_feedforward=d, device='cuda', bias=bias)
        m.load_state_dict(sd)
        m = m.to('cuda')
        m.dropout_p = m.attn_dropout_p = 0.0
        m.eval()
        with torch.no_grad():
            s0 = m.sample(h, c)
        np.testing.assert_allclose(s0.cpu().numpy(), G['eval_sample'], atol=1e-4, rtol=0)
        np.testing.assert_array_equal(kernels.tilemap(s0).cpu().numpy(), kernels.tilemap(torch.from_numpy(G['eval_sample']).cuda()).cpu().numpy())
        m.train()
        opt = mtio.FusedAdamW(m, lr=1e-4)
        seed = int(G[f'train_{branch}_mixseed'])
        random.seed(seed)
        np.random.seed(seed)
        loss = m.train_step(h, c, f, opt)
        np.testing.assert_allclose(loss.item(), float(G[f'train_{branch}_loss']), rtol=1e-4, atol=1e-6)
        bn = m.transformer.distill_layer.norm
        np.testing.assert_allclose(bn.running_mean.cpu().numpy(), G[f'train_{branch}_bn_mean'], atol=1e-6, rtol=1e-5)
        np.testing.assert_allclose(bn.running_var.cpu().numpy(), G[f'train_{branch}_bn_var'], atol=1e-6, rtol=1e-5)
        for key in G.files:
            if key.startswith(f'train_{branch}_adamw::'):
                np.testing.assert_allclose(m.state_dict()[key.split('::')[1]].cpu().numpy(), G[key], atol=3e-6, rtol=1e-5, err_msg=key)
        m.eval()
        with torch.no_grad():
            s1 = m.sample(h, c)
        np.testing.assert_allclose(s1.cpu().numpy(), G[f'train_{branch}_after_sample'], atol=1e-4, rtol=0)
        # after one AdamW step of lr 1e-4 every weight moved by ~lr: a point within fp32 rounding of a tile edge could differ
        got_maps = kernels.tilemap(s1).cpu().numpy()
        want_maps = kernels.tilemap(torch.from_numpy(G[f'train_{branch}_after_sample']).cuda()).cpu().numpy()
        assert (got_maps != want_maps).mean() <= 0.01
