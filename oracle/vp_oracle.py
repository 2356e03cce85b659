"""TEST INFRASTRUCTURE ONLY (oracle). Not imported by the product path.

CPU restatement (plain torch fp32, explicit matmuls -- no nn.Transformer) of the
reference viewport predictor:

  * ViewportEmbedding / PositionalEncoding      viewport_prediction/models/mtio.py:10-44
  * nn.TransformerEncoder(post-norm, ReLU) x E  models/customized_transformer.py:40-49,74-77
    (third-party torch.nn.Transformer arithmetic, torch 2.10 in this image)
  * DistillLayer                                 models/customized_transformer.py:13-36
  * train path `_process_src_current`            models/mtio.py:150-166
  * `sample`                                     models/mtio.py:106-133
  * MTIO loss / periodic MSE                     models/mtio.py:94-104, utils/common.py:73-80
  * `to_position_normalized_cartesian`           utils/common.py:61-70
  * LinearRegression.sample (the comparison baseline) models/linear_regression.py:18-36
    (third-party scikit-learn LinearRegression(fit_intercept=True): published algorithm restated;
    pinned by tests/golden/linreg_reference.npz = the imported class on real Jin2022 windows)

The decoder is restated as a KV-cached incremental decoder.  In eval mode / with
dropout disabled this computes the same function (and therefore the same
gradients) as the reference's T-step recompute loop (mtio.py:158-164): the
causal mask makes position j's hidden state independent of later positions.
This equivalence is pinned by tests/golden/vp_*.npz, which were produced by
running the *imported reference* (tools/gen_golden_vp.py) -- see
tests/test_oracle_vp.py.

State-dict key names are the reference's (both bias layouts: torch<=2.0 has
in_proj_bias/out_proj.bias/linear*.bias/norm*.bias, torch>=2.1 drops them because
customized_transformer.py:46-49 passes `device` positionally into `bias`).

Dropout (train mode) uses the build's own counter hash (oracle/rng.py) with the
site numbering below, mirrored by csrc/vp_engine.hip.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

from . import rng as _rng

# ---- dropout site numbering (mirrored in csrc/vp_engine.hip) -----------------
SITE_PE_SRC = 1


def site_enc(l, k):          # k: 0 attn prob, 1 dropout1, 2 ffn inner, 3 dropout2
    return 100 + l * 8 + k


def site_pe_tgt(i):
    return 1000 + i


def site_dec(l, i, k):       # k: 0 self prob, 1 dropout1, 2 cross prob, 3 dropout2, 4 ffn inner, 5 dropout3
    return 10000 + (l * 64 + i) * 8 + k


class Dropper:
    """Applies the shared-hash dropout; p_scale=0 disables (parity-with-reference mode)."""

    def __init__(self, seed=0, enabled=False):
        self.seed = seed
        self.enabled = enabled

    def __call__(self, x, site, p):
        if not self.enabled or p <= 0.0:
            return x
        keep = _rng.keep_mask(self.seed, site, x.numel(), p)
        m = torch.from_numpy(keep.astype(np.float32)).reshape(x.shape)
        return x * m * (1.0 / (1.0 - p))


def positional_table(max_len, d_model):
    """mtio.py:17-24 (same float32 op order)."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def _lin(x, sd, wkey, bkey=None):
    y = x @ sd[wkey].t()
    if bkey is not None and bkey in sd:
        y = y + sd[bkey]
    return y


def _ln(x, sd, prefix, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    y = (x - mu) * torch.rsqrt(var + eps) * sd[prefix + '.weight']
    if prefix + '.bias' in sd:
        y = y + sd[prefix + '.bias']
    return y


def _attn(q, k, v, H, drop, site, p):
    """q [B,Lq,d], k,v [B,Lk,d] -> [B,Lq,d]; softmax(q k^T / sqrt(dh)) v, prob dropout."""
    B, Lq, d = q.shape
    Lk = k.shape[1]
    dh = d // H
    qh = q.reshape(B, Lq, H, dh).permute(0, 2, 1, 3)
    kh = k.reshape(B, Lk, H, dh).permute(0, 2, 1, 3)
    vh = v.reshape(B, Lk, H, dh).permute(0, 2, 1, 3)
    s = (qh @ kh.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    P = torch.softmax(s, dim=-1)                       # [B,H,Lq,Lk]
    Pd = drop(P, site, p)
    o = (Pd @ vh).permute(0, 2, 1, 3).reshape(B, Lq, d)
    return o, P


class VPOracle:
    def __init__(self, sd, fut_window, n_head=8, n_enc=2, n_dec=2, in_channel=2, num_head=3,
                 p_pe=0.2, p_drop=0.1):
        self.sd = sd
        self.T = fut_window
        self.H = n_head
        self.E = n_enc
        self.D = n_dec
        self.in_channel = in_channel
        self.num_head = num_head
        self.p_pe = p_pe
        self.p_drop = p_drop
        self.d = sd['embedding.linear.weight'].shape[0]
        self.pe = sd['positional_embedding.pe'][0] if 'positional_embedding.pe' in sd \
            else positional_table(5000, self.d)

    # -- encoder + distill -> memory ------------------------------------------------
    def encode(self, src, train, drop, im):
        sd, d, H = self.sd, self.d, self.H
        B, S, _ = src.shape
        x = _lin(src, sd, 'embedding.linear.weight', 'embedding.linear.bias') + self.pe[:S]
        x = drop(x, SITE_PE_SRC, self.p_pe)
        im['enc.x0'] = x
        for l in range(self.E):
            p = f'transformer.encoder.layers.{l}.'
            qkv = _lin(x, sd, p + 'self_attn.in_proj_weight', p + 'self_attn.in_proj_bias')
            im[f'enc{l}.qkv'] = qkv
            q, k, v = qkv.split(d, dim=-1)
            ao, P = _attn(q, k, v, H, drop, site_enc(l, 0), self.p_drop)
            im[f'enc{l}.P'] = P
            im[f'enc{l}.ao'] = ao
            proj = _lin(ao, sd, p + 'self_attn.out_proj.weight', p + 'self_attn.out_proj.bias')
            z1 = x + drop(proj, site_enc(l, 1), self.p_drop)
            y1 = _ln(z1, sd, p + 'norm1')
            im[f'enc{l}.z1'] = z1
            im[f'enc{l}.y1'] = y1
            h = drop(torch.relu(_lin(y1, sd, p + 'linear1.weight', p + 'linear1.bias')), site_enc(l, 2), self.p_drop)
            im[f'enc{l}.h'] = h
            z2 = y1 + drop(_lin(h, sd, p + 'linear2.weight', p + 'linear2.bias'), site_enc(l, 3), self.p_drop)
            x = _ln(z2, sd, p + 'norm2')
            im[f'enc{l}.z2'] = z2
            im[f'enc{l}.y2'] = x
        x = _ln(x, sd, 'transformer.encoder.norm')
        im['enc.out'] = x
        # DistillLayer (customized_transformer.py:30-36): circular conv k=3 -> BN -> ELU -> maxpool(3,2,1)
        w = sd['transformer.distill_layer.downConv.weight']            # [Co,Ci,3]
        xm1 = torch.roll(x, 1, dims=1)                                  # x[s-1]
        xp1 = torch.roll(x, -1, dims=1)                                 # x[s+1]
        col = torch.stack([xm1, x, xp1], dim=-1).reshape(B, S, d * 3)   # K index = ci*3+t
        im['dis.col'] = col
        conv = col @ w.reshape(d, d * 3).t() + sd['transformer.distill_layer.downConv.bias']
        im['dis.conv'] = conv
        bn = 'transformer.distill_layer.norm.'
        new_stats = None
        if train:
            # torch's CPU BatchNorm (the reference path) accumulates the batch statistics -- and, in backward, sum(dy) and
            # sum(dy * xhat) -- in DOUBLE (at::acc_type<float, is_cuda=false>); on real traces a channel's batch variance can be
            # orders of magnitude below its mean square, where float32 sums lose three digits (found with the B = 32 Jin2022
            # golden: encoder-side gradients 2e-3 off in float32, 1e-8 with the statistics in double).  The normalisation is
            # therefore done in float64 here (autograd then sums in float64 too) and cast back.
            c64 = conv.double()
            flat = c64.reshape(B * S, d)
            mean64 = flat.mean(0)
            var_b = ((flat - mean64) ** 2).mean(0)                      # biased (normalisation)
            n = B * S
            var_u = var_b * (n / max(n - 1, 1))                         # unbiased (running update)
            new_stats = ((0.9 * sd[bn + 'running_mean'] + 0.1 * mean64.to(conv.dtype)).detach(),
                         (0.9 * sd[bn + 'running_var'] + 0.1 * var_u.to(conv.dtype)).detach())
            rstd64 = torch.rsqrt(var_b + 1e-5)
            mean, rstd = mean64.to(conv.dtype), rstd64.to(conv.dtype)
            xhat = ((c64 - mean64) * rstd64).to(conv.dtype)
        else:
            mean, var = sd[bn + 'running_mean'], sd[bn + 'running_var']
            rstd = torch.rsqrt(var + 1e-5)
            xhat = (conv - mean) * rstd
        im['dis.bn_mean'] = mean
        im['dis.bn_rstd'] = rstd
        y = xhat * sd[bn + 'weight'] + sd[bn + 'bias']
        y = F.elu(y)
        im['dis.act'] = y
        M = (S - 1) // 2 + 1
        ninf = torch.full((B, 1, d), float('-inf'))
        yp = torch.cat([ninf, y, ninf], dim=1)
        mem = torch.stack([yp[:, 2 * m:2 * m + 3].max(dim=1).values for m in range(M)], dim=1)
        im['mem'] = mem
        return mem, new_stats

    # -- KV-cached decoder --------------------------------------------------------------
    def decode(self, mem, cur, drop, im, detach_feedback=False):
        sd, d, H, T, D = self.sd, self.d, self.H, self.T, self.D
        B = cur.shape[0]
        memk, memv = [], []
        for l in range(D):
            p = f'transformer.decoder.layers.{l}.multihead_attn.'
            w = sd[p + 'in_proj_weight'][d:]
            kv = mem @ w.t()
            if p + 'in_proj_bias' in sd:
                kv = kv + sd[p + 'in_proj_bias'][d:]
            im[f'dec{l}.memkv'] = kv
            memk.append(kv[..., :d])
            memv.append(kv[..., d:])
        kc = [[] for _ in range(D)]
        vc = [[] for _ in range(D)]
        names = ['x', 'qkv', 'ao1', 'z1', 'y1', 'qc', 'ao2', 'z2', 'y2', 'h', 'z3', 'y3']
        slabs = {f'dec{l}.{n}': [] for l in range(D) for n in names}
        slabs['dec.out'] = []
        tok = cur[:, 0]                                   # [B,6]
        preds = []
        for i in range(T):
            x = _lin(tok, sd, 'embedding.linear.weight', 'embedding.linear.bias') + self.pe[i]
            x = drop(x, site_pe_tgt(i), self.p_pe)
            for l in range(D):
                p = f'transformer.decoder.layers.{l}.'
                slabs[f'dec{l}.x'].append(x)
                qkv = _lin(x, sd, p + 'self_attn.in_proj_weight', p + 'self_attn.in_proj_bias')
                slabs[f'dec{l}.qkv'].append(qkv)
                q, k, v = qkv.split(d, dim=-1)
                kc[l].append(k)
                vc[l].append(v)
                K = torch.stack(kc[l], dim=1)
                V = torch.stack(vc[l], dim=1)
                ao1, _ = _attn(q[:, None], K, V, H, drop, site_dec(l, i, 0), self.p_drop)
                ao1 = ao1[:, 0]
                slabs[f'dec{l}.ao1'].append(ao1)
                z1 = x + drop(_lin(ao1, sd, p + 'self_attn.out_proj.weight', p + 'self_attn.out_proj.bias'),
                              site_dec(l, i, 1), self.p_drop)
                y1 = _ln(z1, sd, p + 'norm1')
                slabs[f'dec{l}.z1'].append(z1)
                slabs[f'dec{l}.y1'].append(y1)
                mp = p + 'multihead_attn.'
                qc = y1 @ sd[mp + 'in_proj_weight'][:d].t()
                if mp + 'in_proj_bias' in sd:
                    qc = qc + sd[mp + 'in_proj_bias'][:d]
                slabs[f'dec{l}.qc'].append(qc)
                ao2, _ = _attn(qc[:, None], memk[l], memv[l], H, drop, site_dec(l, i, 2), self.p_drop)
                ao2 = ao2[:, 0]
                slabs[f'dec{l}.ao2'].append(ao2)
                z2 = y1 + drop(_lin(ao2, sd, mp + 'out_proj.weight', mp + 'out_proj.bias'),
                               site_dec(l, i, 3), self.p_drop)
                y2 = _ln(z2, sd, p + 'norm2')
                slabs[f'dec{l}.z2'].append(z2)
                slabs[f'dec{l}.y2'].append(y2)
                h = drop(torch.relu(_lin(y2, sd, p + 'linear1.weight', p + 'linear1.bias')),
                         site_dec(l, i, 4), self.p_drop)
                slabs[f'dec{l}.h'].append(h)
                z3 = y2 + drop(_lin(h, sd, p + 'linear2.weight', p + 'linear2.bias'),
                               site_dec(l, i, 5), self.p_drop)
                x = _ln(z3, sd, p + 'norm3')
                slabs[f'dec{l}.z3'].append(z3)
                slabs[f'dec{l}.y3'].append(x)
            out = _ln(x, sd, 'transformer.decoder.norm')
            slabs['dec.out'].append(out)
            pred = torch.sigmoid(_lin(out, sd, 'predictor.0.weight', 'predictor.0.bias'))
            preds.append(pred)
            tok = pred.detach() if detach_feedback else pred
        for k_, v_ in slabs.items():
            im[k_] = torch.stack(v_, dim=0)               # [T,B,C]
        pred = torch.stack(preds, dim=1)                  # [B,T,6]
        im['pred'] = pred
        return pred

    # -- public API (same semantics as the reference methods) ---------------------------
    def process_src_current(self, src, cur, train=False, dropout_seed=None, want_intermediates=False):
        """mtio.py:150-166.  src [B,S,6], cur [B,1,6] -> pred [B,T,6]."""
        drop = Dropper(seed=dropout_seed or 0, enabled=(train and dropout_seed is not None))
        im = {}
        mem, new_stats = self.encode(src, train, drop, im)
        pred = self.decode(mem, cur, drop, im)
        self.last_bn_stats = new_stats
        return (pred, im) if want_intermediates else pred

    def sample(self, history, current):
        """mtio.py:106-133 (heads replicated, per-step 3-head mean, final wrap)."""
        src = torch.cat([history] * self.num_head, dim=-1)
        cur = torch.cat([current] * self.num_head, dim=-1)
        pred = self.process_src_current(src, cur, train=False)
        B, T, _ = pred.shape
        ens = pred.reshape(B, T, self.num_head, self.in_channel).sum(dim=2) / self.num_head
        return to_position_normalized_cartesian(ens)

    def loss_function(self, pred, gt):
        """mtio.py:94-104."""
        loss = 0.
        c = self.in_channel
        for i in range(self.num_head):
            loss = loss + torch.mean(mean_square_error(pred[:, :, i * c:(i + 1) * c], gt[:, :, i * c:(i + 1) * c]))
        return loss


def mean_square_error(a, b, dimension=2):
    """utils/common.py:73-80 (periodic)."""
    e = torch.abs(a - b)
    e = torch.minimum(e, torch.abs(a + 1 - b))
    e = torch.minimum(e, torch.abs(a - 1 - b))
    return torch.sum(e * e, dim=-1) / dimension


def to_position_normalized_cartesian(values):
    """utils/common.py:61-70: v<0 -> v - trunc(v) + 1 ; v>1 -> v - trunc(v)."""
    out = values.clone()
    neg = values < 0
    gt1 = values > 1
    out[neg] = values[neg] - values[neg].to(dtype=torch.int) + 1
    out[gt1] = values[gt1] - values[gt1].to(dtype=torch.int)
    return out


def linear_regression_sample(history, current, fut_window):
    """LinearRegression.sample (viewport_prediction/models/linear_regression.py:18-36): per trajectory and coordinate an ordinary
    least-squares line through the S + 1 past samples over t = 0..S, extrapolated to t = S+1..S+T.  scikit-learn's
    LinearRegression(fit_intercept=True).fit works in float64 (the int64 abscissa is converted by _preprocess_data, y follows X's
    dtype): centre X and y on their means, least squares on the centred data (one column: coef = <xc, yc> / <xc, xc>),
    intercept = y_mean - x_mean * coef, predict = t * coef + intercept; the float64 prediction is rounded to float32 when it is
    assigned into the float32 `samples` tensor (:23,35).  numpy arrays in, float32 [B, T, 2] out."""
    merge = np.concatenate([np.asarray(history), np.asarray(current)], axis=1).astype(np.float64)      # [B, L, 2]
    L = merge.shape[1]
    t = np.arange(L, dtype=np.float64)
    tc = t - np.average(t)
    y_mean = np.average(merge, axis=1)                                                                # [B, 2]
    yc = merge - y_mean[:, None, :]
    coef = np.einsum('l,blc->bc', tc, yc) / np.dot(tc, tc)
    intercept = y_mean - np.average(t) * coef
    fut = np.arange(L, L + fut_window, dtype=np.float64)
    return (fut[None, :, None] * coef[:, None, :] + intercept[:, None, :]).astype(np.float32)


def mtio_mix(history, current, future, num_head, repeat, perms):
    """mtio.py:72-90 with the host RNG decisions made explicit:
    repeat=True -> replicate; else heads 1.. use row permutations `perms[k]`."""
    hs, cs, fs = [history], [current], [future]
    for k in range(num_head - 1):
        if repeat:
            hs.append(history), cs.append(current), fs.append(future)
        else:
            idx = torch.as_tensor(perms[k], dtype=torch.long)
            hs.append(history[idx]), cs.append(current[idx]), fs.append(future[idx])
    return torch.cat(hs, -1), torch.cat(cs, -1), torch.cat(fs, -1)


def adamw_step(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW single-tensor math (run_models.py:29 uses torch defaults)."""
    p = p * (1 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def make_state_dict(d_model, seed, bias=True, n_enc=2, n_dec=2, in_ch=6, scale=None):
    """Seeded synthetic weights with the reference's key names/shapes (both bias layouts).
    Deterministic given (torch version, seed): used so large-d fixtures need not store weights."""
    g = torch.Generator().manual_seed(seed)
    d = d_model

    def rn(*shape, s):
        return torch.randn(*shape, generator=g) * s

    sd = {}
    sd['embedding.linear.weight'] = rn(d, in_ch, s=0.4)
    sd['embedding.linear.bias'] = rn(d, s=0.1)
    ws = 1.0 / math.sqrt(d)

    def mha(prefix):
        sd[prefix + 'in_proj_weight'] = rn(3 * d, d, s=ws)
        if bias:
            sd[prefix + 'in_proj_bias'] = rn(3 * d, s=0.05)
        sd[prefix + 'out_proj.weight'] = rn(d, d, s=ws)
        if bias:
            sd[prefix + 'out_proj.bias'] = rn(d, s=0.05)

    def ffn_norms(prefix, norms):
        for n in ('linear1', 'linear2'):
            sd[prefix + n + '.weight'] = rn(d, d, s=ws)
            if bias:
                sd[prefix + n + '.bias'] = rn(d, s=0.05)
        for n in norms:
            sd[prefix + n + '.weight'] = 1.0 + rn(d, s=0.1)
            if bias:
                sd[prefix + n + '.bias'] = rn(d, s=0.05)

    for l in range(n_enc):
        p = f'transformer.encoder.layers.{l}.'
        mha(p + 'self_attn.')
        ffn_norms(p, ('norm1', 'norm2'))
    sd['transformer.encoder.norm.weight'] = 1.0 + rn(d, s=0.1)
    if bias:
        sd['transformer.encoder.norm.bias'] = rn(d, s=0.05)
    for l in range(n_dec):
        p = f'transformer.decoder.layers.{l}.'
        mha(p + 'self_attn.')
        mha(p + 'multihead_attn.')
        ffn_norms(p, ('norm1', 'norm2', 'norm3'))
    sd['transformer.decoder.norm.weight'] = 1.0 + rn(d, s=0.1)
    if bias:
        sd['transformer.decoder.norm.bias'] = rn(d, s=0.05)
    p = 'transformer.distill_layer.'
    sd[p + 'downConv.weight'] = rn(d, d, 3, s=1.0 / math.sqrt(3 * d))
    sd[p + 'downConv.bias'] = rn(d, s=0.05)
    sd[p + 'norm.weight'] = 1.0 + rn(d, s=0.1)
    sd[p + 'norm.bias'] = rn(d, s=0.05)
    sd[p + 'norm.running_mean'] = rn(d, s=0.1)
    sd[p + 'norm.running_var'] = 1.0 + 0.2 * torch.rand(d, generator=g)
    sd[p + 'norm.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)
    sd['positional_embedding.pe'] = positional_table(5000, d).unsqueeze(0)
    sd['predictor.0.weight'] = rn(in_ch, d, s=ws)
    sd['predictor.0.bias'] = rn(in_ch, s=0.1)
    return sd


def synthetic_trajectories(B, S, T, seed=5):
    """SURVEY 8(d) C2 inputs: torus random walks, 21 samples (S + 1 + T), fp32."""
    g = torch.Generator().manual_seed(seed)
    L = S + 1 + T
    p0 = torch.rand(B, 1, 2, generator=g)
    steps = torch.randn(B, L - 1, 2, generator=g) * 0.02
    traj = torch.cat([p0, p0 + torch.cumsum(steps, dim=1)], dim=1)
    traj = traj - torch.floor(traj)
    return traj[:, :S].contiguous(), traj[:, S:S + 1].contiguous(), traj[:, S + 1:].contiguous()
