#!/usr/bin/env python3
"""BASELINE.json configs[0] as stated (SURVEY 8 "C1"): B = 32 REAL Jin2022 windows (hist 10, pred 10, step 5, trim 15/15) through
the IMPORTED reference model at d = 512, 2+2 layers -- run_models.py:29-44 for one iteration (seeded MTIO decision, dropout forced
to 0, AdamW lr 1e-4) and `sample()` on the same batch (run_models.py:50-58), in both bias layouts (torch <= 2.0 with biases via the
legacy-signature shim; torch >= 2.1 bias-free).

The batch is the FIRST batch of `DataLoader(train_set, batch_size=32, shuffle=True)` under `torch.manual_seed(5)` over the train
split of the eight traces that already travel in tests/golden/dataset_reference.npz (same video / user subset as its tag 'a'), so
the GPU test can rebuild it from those traces through the HBM table + mansy_traj_gather (DeviceLoader) with no new data.
Writes tests/golden/vp_c1_jin2022_b32_{bias,nobias}.npz.  Data only; weights are regenerated from a seed (oracle.vp_oracle)."""
import ast
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refstubs  # noqa: E402
refstubs.install()
REF = '/root/reference/viewport_prediction'
sys.path.insert(0, REF)
os.chdir(REF)
from utils.common import get_config_from_yml  # noqa: E402
from utils.load_dataset import create_dataset  # noqa: E402
from models.mtio import ViewportTransformerMTIO  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402
from oracle import vp_oracle as vo  # noqa: E402

D, S, T, B, SEED = 512, 10, 10, 32, 5


def zero_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0


def maxpool_min_gap(h, c, f, bias, wseed):
    """Smallest difference between the two largest values of any MaxPool window of the DistillLayer (float64 oracle, train mode)."""
    sd = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in vo.make_state_dict(D, wseed, bias=bias).items()}
    orc = vo.VPOracle(sd, fut_window=T)
    src, cur, gt = vo.mtio_mix(h.double(), c.double(), f.double(), 3, True, None)
    with torch.no_grad():
        _, im = orc.process_src_current(src, cur, train=True, want_intermediates=True)
    y = im['dis.act']
    ninf = torch.full((y.shape[0], 1, y.shape[2]), float('-inf'), dtype=y.dtype)
    yp = torch.cat([ninf, y, ninf], 1)
    M = (S - 1) // 2 + 1
    top2 = torch.stack([yp[:, 2 * m:2 * m + 3] for m in range(M)], 1).topk(2, dim=2).values
    return float((top2[:, :, 0] - top2[:, :, 1]).min())


def main():
    Z = np.load(os.path.join(ROOT, 'tests', 'golden', 'dataset_reference.npz'))
    vsplit, usplit = ast.literal_eval(str(Z['vsplit'])), ast.literal_eval(str(Z['usplit']))
    config = get_config_from_yml()
    train = create_dataset('Jin2022', config, his_window=S, fut_window=T, frequency=5, sample_step=5, trim_head=15, trim_tail=15,
                           dataset_video_split=dict(vsplit), dataset_user_split=dict(usplit), include=['train'])[0]
    # MaxPool1d routes each gradient element to the arg-max of its window: a DISCONTINUITY.  On real traces a window's two largest
    # values can be closer than float32 rounding of the conv / BatchNorm chain (first batch of this loader: one window of 81 920 with
    # a gap of 1e-6 at a value of 0.027), where two correct fp32 implementations route differently and every encoder-side gradient
    # moves by up to 2e-3 of its maximum (measured: float32 vs float64 run of the oracle; the imported reference happens to agree
    # with float64 there).  A gradient fixture must not sit on such a point, so the batch taken is the first batch of the seeded
    # loader whose smallest top-2 gap (float64 oracle, either weight set, replicate branch) exceeds 1e-5 (an order of magnitude above the float32 error of the normalised conv output); the skipped batches and
    # their gaps are recorded.
    torch.manual_seed(SEED)
    skipped = []
    for bi, (h, c, f, v, u, t) in enumerate(DataLoader(train, batch_size=B, shuffle=True)):
        h, c, f = h.float(), c.float(), f.float()
        gaps = [maxpool_min_gap(h, c, f, bias, wseed) for bias, wseed in ((True, 31), (False, 32))]
        if min(gaps) > 1e-5:
            break
        skipped.append((bi, min(gaps)))
    print('batch', bi, 'min top-2 gaps', gaps, 'skipped', skipped)
    assert h.shape == (B, S, 2) and c.shape == (B, 1, 2) and f.shape == (B, T, 2)
    for bias, wseed in ((True, 31), (False, 32)):
        sd = vo.make_state_dict(D, wseed, bias=bias)
        if bias:
            with refstubs.legacy_transformer_signature():
                model = ViewportTransformerMTIO(in_channel=2, fut_window=T, d_model=D, dim_feedforward=D, device='cpu')
        else:
            model = ViewportTransformerMTIO(in_channel=2, fut_window=T, d_model=D, dim_feedforward=D, device='cpu')
        model.load_state_dict(sd, strict=True)
        rec = dict(d=D, S=S, T=T, B=B, bias=int(bias), wseed=wseed, loader_seed=SEED, n_train=len(train), batch_index=bi,
                   maxpool_min_gap=np.array(gaps), skipped_batches=np.array(skipped, np.float64).reshape(-1, 2),
                   ids=np.stack([np.asarray(v), np.asarray(u), np.asarray(t)], 1).astype(np.int64),
                   history=h.numpy(), current=c.numpy(), future=f.numpy())
        model.eval()
        with torch.no_grad():
            rec['eval_pred'] = model._process_src_current(torch.cat([h] * 3, -1), torch.cat([c] * 3, -1)).numpy()
            rec['eval_sample'] = model.sample(h, c).numpy()
        for key, mix_seed in (('rep', None), ('mix', None)):
            # find a seed whose first random.random() picks the wanted MTIO branch (mtio.py:77)
            for sidx in range(1, 50):
                random.seed(sidx)
                if (random.random() < 0.5) == (key == 'rep'):
                    mix_seed = sidx
                    break
            # the permutations mtio.py:80-87 will draw for heads 2 and 3 (same stream, replayed below)
            random.seed(mix_seed)
            np.random.seed(mix_seed)
            random.random()
            perms = []
            if key == 'mix':
                for _ in range(2):
                    idx = np.arange(B)
                    np.random.shuffle(idx)
                    perms.append(idx.copy())
            rec[f'train_{key}_perms'] = np.array(perms, dtype=np.int64).reshape(len(perms), B if perms else 0)
            model.load_state_dict(sd)
            model.train()
            zero_dropout(model)
            opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
            random.seed(mix_seed)
            np.random.seed(mix_seed)
            pred, gt = model(h, c, f)
            loss = model.loss_function(pred, gt)
            opt.zero_grad()
            loss.backward()
            grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
            rec[f'train_{key}_mixseed'] = mix_seed
            rec[f'train_{key}_pred'] = pred.detach().numpy()
            rec[f'train_{key}_gt'] = gt.detach().numpy()
            rec[f'train_{key}_loss'] = np.float32(loss.item())
            rec[f'train_{key}_gradnames'] = np.array(sorted(grads))
            rec[f'train_{key}_gradnorms'] = np.array([grads[k].norm().item() for k in sorted(grads)], np.float64)
            for k in ('embedding.linear.weight', 'predictor.0.weight', 'transformer.distill_layer.norm.weight', 'transformer.decoder.norm.weight',
                      'transformer.encoder.layers.0.norm1.weight'):
                rec[f'train_{key}_grad::{k}'] = grads[k].numpy()
            for k in ('transformer.encoder.layers.0.self_attn.in_proj_weight', 'transformer.decoder.layers.0.multihead_attn.in_proj_weight',
                      'transformer.decoder.layers.1.linear2.weight', 'transformer.distill_layer.downConv.weight'):
                rec[f'train_{key}_gradslice::{k}'] = grads[k].reshape(grads[k].shape[0], -1)[::37, ::41].numpy()
            bsd = model.state_dict()
            rec[f'train_{key}_bn_mean'] = bsd['transformer.distill_layer.norm.running_mean'].numpy().copy()
            rec[f'train_{key}_bn_var'] = bsd['transformer.distill_layer.norm.running_var'].numpy().copy()
            opt.step()
            asd = model.state_dict()
            for k in ('embedding.linear.weight', 'predictor.0.weight', 'transformer.decoder.norm.weight'):
                rec[f'train_{key}_adamw::{k}'] = asd[k].numpy().copy()
            # the validation metric of run_models.py:50-58 after that one step, on the same batch
            model.eval()
            with torch.no_grad():
                sp = model.sample(h, c)
            rec[f'train_{key}_after_sample'] = sp.numpy()
        path = os.path.join(ROOT, 'tests', 'golden', f'vp_c1_jin2022_b32_{"bias" if bias else "nobias"}.npz')
        np.savez_compressed(path, **rec)
        print(path, os.path.getsize(path) // 1024, 'KiB', 'loss', float(rec['train_rep_loss']), float(rec['train_mix_loss']))


if __name__ == '__main__':
    main()
