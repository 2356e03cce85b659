#!/usr/bin/env python3
"""Generate tests/golden/vp_*.npz by running the IMPORTED reference VP model
(/root/reference/viewport_prediction) in this container.  The fixtures hold data only
(inputs / expected outputs); weights are regenerated from a seed by
oracle.vp_oracle.make_state_dict (deterministic for this image's torch), so no reference
source or pickled code travels.

Per case we record, from the reference itself:
  * eval:  _process_src_current(src6,cur6)  (mtio.py:150-166), sample(h,c) (mtio.py:106-133)
  * train (all dropout p forced to 0, BatchNorm in train mode, MTIO mixing driven by seeded
    `random`/`np.random` exactly as mtio.py:77-87): forward(history,current,future) -> pred,
    multi_future; loss_function; loss.backward() grads; BN running stats; one AdamW step
    (run_models.py:29,42-44).
Usage: python tools/gen_golden_vp.py
"""
import os
import sys
import random
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refstubs  # noqa: E402
refstubs.install()
sys.path.insert(0, '/root/reference/viewport_prediction')
from models.mtio import ViewportTransformerMTIO  # noqa: E402  (the reference)
from oracle import vp_oracle as vo  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')

CASES = [
    # name, d, S, T, B, bias, wseed, full_grads
    ('vp_d64_s10_t10_nobias', 64, 10, 10, 8, False, 11, True),
    ('vp_d64_s5_t15_bias', 64, 5, 15, 6, True, 12, True),
    ('vp_d512_s10_t10_bias', 512, 10, 10, 4, True, 13, False),
    ('vp_d512_s10_t10_nobias', 512, 10, 10, 4, False, 14, False),
    # the README's own training shape (his 5, fut 15, hid 512: README.md:139, .MISSING_LARGE_BLOBS): Lk = 11..15 self-attention instances and an
    # M = 3 memory meet the imported model at d = 512 (round 5; `python tools/gen_golden_vp.py vp_d512_s5_t15_bias` writes only this case)
    ('vp_d512_s5_t15_bias', 512, 5, 15, 4, True, 15, False),
]


def build_reference(d, T, bias, sd):
    if bias:
        with refstubs.legacy_transformer_signature():
            model = ViewportTransformerMTIO(in_channel=2, fut_window=T, d_model=d, dim_feedforward=d, device='cpu')
    else:
        model = ViewportTransformerMTIO(in_channel=2, fut_window=T, d_model=d, dim_feedforward=d, device='cpu')
    missing = model.load_state_dict(sd, strict=True)
    return model


def zero_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0


def main():
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])
    for name, d, S, T, B, bias, wseed, full in CASES:
        if only and name not in only:
            continue
        sd = vo.make_state_dict(d, wseed, bias=bias)
        model = build_reference(d, T, bias, sd)
        assert set(model.state_dict().keys()) == set(sd.keys()), (
            set(model.state_dict().keys()) ^ set(sd.keys()))
        hist, cur, fut = vo.synthetic_trajectories(B, S, T, seed=wseed + 100)
        rec = dict(d=d, S=S, T=T, B=B, bias=int(bias), wseed=wseed,
                   history=hist.numpy(), current=cur.numpy(), future=fut.numpy())
        # ---- eval -----------------------------------------------------------------
        model.eval()
        with torch.no_grad():
            src6 = torch.cat([hist] * 3, -1)
            cur6 = torch.cat([cur] * 3, -1)
            rec['eval_pred'] = model._process_src_current(src6, cur6).numpy()
            rec['eval_sample'] = model.sample(hist, cur).numpy()
        # ---- train, dropout off, both MTIO branches ---------------------------------
        for tag, mix_seed in (('a', 3), ('b', 1), ('c', 2), ('d', 7)):
            random.seed(mix_seed)
            np.random.seed(mix_seed)
            r = random.random()
            st = random.getstate(), np.random.get_state()
            repeat = r < 0.5
            if tag in ('c', 'd') and f'train_{"rep" if repeat else "mix"}_pred' in rec:
                continue
            key = 'rep' if repeat else 'mix'
            if f'train_{key}_pred' in rec:
                continue
            perms = []
            if not repeat:
                for _ in range(2):
                    idx = np.arange(B)
                    np.random.shuffle(idx)
                    perms.append(idx.copy())
            random.seed(mix_seed)
            np.random.seed(mix_seed)
            model.load_state_dict(sd)
            model.train()
            zero_dropout(model)
            model.zero_grad()
            opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
            pred, gt = model(hist, cur, fut)
            loss = model.loss_function(pred, gt)
            opt.zero_grad()
            loss.backward()
            rec[f'train_{key}_mixseed'] = mix_seed
            rec[f'train_{key}_perms'] = np.array(perms, dtype=np.int64).reshape(len(perms), B if perms else 0)
            rec[f'train_{key}_pred'] = pred.detach().numpy()
            rec[f'train_{key}_gt'] = gt.detach().numpy()
            rec[f'train_{key}_loss'] = np.float32(loss.item())
            grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
            rec[f'train_{key}_gradnorms'] = np.array([grads[k].norm().item() for k in sorted(grads)], dtype=np.float64)
            rec[f'train_{key}_gradnames'] = np.array(sorted(grads))
            if full:
                for k, g in grads.items():
                    rec[f'train_{key}_grad::{k}'] = g.numpy()
            else:
                for k in ('embedding.linear.weight', 'embedding.linear.bias', 'predictor.0.weight',
                          'predictor.0.bias', 'transformer.distill_layer.norm.weight',
                          'transformer.distill_layer.downConv.bias',
                          'transformer.encoder.layers.0.norm1.weight',
                          'transformer.decoder.layers.1.norm3.weight', 'transformer.decoder.norm.weight'):
                    rec[f'train_{key}_grad::{k}'] = grads[k].numpy()
                for k in ('transformer.encoder.layers.0.self_attn.in_proj_weight',
                          'transformer.decoder.layers.0.multihead_attn.in_proj_weight',
                          'transformer.decoder.layers.1.linear2.weight',
                          'transformer.distill_layer.downConv.weight'):
                    rec[f'train_{key}_gradslice::{k}'] = grads[k].reshape(grads[k].shape[0], -1)[::37, ::41].numpy()
            bsd = model.state_dict()
            rec[f'train_{key}_bn_mean'] = bsd['transformer.distill_layer.norm.running_mean'].numpy().copy()
            rec[f'train_{key}_bn_var'] = bsd['transformer.distill_layer.norm.running_var'].numpy().copy()
            opt.step()
            asd = model.state_dict()
            for k in ('embedding.linear.weight', 'predictor.0.weight', 'transformer.decoder.norm.weight'):
                rec[f'train_{key}_adamw::{k}'] = asd[k].numpy().copy()
        assert 'train_rep_pred' in rec and 'train_mix_pred' in rec, name
        path = os.path.join(OUT, name + '.npz')
        np.savez_compressed(path, **rec)
        print(name, 'written', os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
