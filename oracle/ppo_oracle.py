"""TEST INFRASTRUCTURE ONLY (oracle). Not imported by the product path.

CPU restatement (plain torch fp32 / numpy fp64) of the bitrate-selection neural path:
  FeatureNet / Actor / Critic / QoEIdentifierFeatureNet / QoEIdentifier   bitrate_selection/models/mansy.py:5-155
  calculate_indentifier_reward, train_identifier                          bitrate_selection/utils/mansy_utils.py:9-49
  reward relabel loop                                                     bitrate_selection/models/mansy_ppo.py:41-51
-- these are pinned by tests/golden/ppo_reference.npz, produced by the imported reference (tools/gen_golden_ppo.py) --
and of the tianshou==0.4.8 arithmetic behind `PPOPolicy.process_fn/learn` (mansy_ppo.py:53-55; run_mansy.py:231-251):
  A2CPolicy._compute_returns, BasePolicy.compute_episodic_return/_gae_return, RunningMeanStd, PPOPolicy.learn.
tianshou is NOT importable here (not installed, no network) and the reference has no tests at that boundary:
  **PARITY UNPINNED** for gae_returns / RunningMeanStd / ppo_loss below.  They restate the published 0.4.8 algorithm;
  every assumption is marked `# T2:`.  The HIP path is checked against THIS restatement plus hand-derived cases.

Observations are rows of 780 floats (layout of oracle/env.c == the reference's dict keys in a fixed order).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

BRANCHES = [('conv1d1', 0, 8), ('conv1d2', 8, 328), ('conv1d3', 328, 648), ('conv1d4', 648, 712), ('conv1d5', 712, 720),
            ('conv1d6', 720, 728), ('conv1d7', 728, 736), ('conv1d8', 736, 744), ('fc1', 744, 745)]
QOE_W = (745, 748)
ACT_1HOT = (748, 763)
HID = 128
N_ACTION = 15


def _leaky(x):
    return F.leaky_relu(x, 0.01)


def feature_net(sd, prefix, obs, last):
    """mansy.py:26-51 / 103-140: ten dense branches (Conv1d with kernel == length is a Linear over the flattened
    (channel, position) input) -> concat [B,1280]; also returns the 10th branch's output (residual)."""
    feats = []
    for name, a, b in BRANCHES:
        w = sd[prefix + name + '.0.weight'].reshape(HID, -1)
        feats.append(_leaky(obs[:, a:b] @ w.t() + sd[prefix + name + '.0.bias']))
    w = sd[prefix + 'fc2.0.weight']
    lastf = _leaky(obs[:, last[0]:last[1]] @ w.t() + sd[prefix + 'fc2.0.bias'])
    feats.append(lastf)
    return torch.cat(feats, dim=-1), lastf


def head(sd, prefix, feats, resid):
    h = _leaky(feats @ sd[prefix + 'fc.0.weight'].t() + sd[prefix + 'fc.0.bias']) + resid      # residual (mansy.py:65,79,153)
    return h @ sd[prefix + 'out.weight'].t() + sd[prefix + 'out.bias']


def actor_logits(sd, obs, prefix='actor.'):
    f, q = feature_net(sd, prefix + 'feature_net.', obs, QOE_W)
    return head(sd, prefix, f, q)


def critic_value(sd, obs, prefix='critic.'):
    f, q = feature_net(sd, prefix + 'feature_net.', obs, QOE_W)
    return head(sd, prefix, f, q)


def identifier_pred(sd, obs, prefix='identifier.'):
    """QoEIdentifier.forward(obs, obs.action_one_hot) (mansy.py:151-155)."""
    f, a = feature_net(sd, prefix + 'feature_net.', obs, ACT_1HOT)
    return torch.sigmoid(head(sd, prefix, f, a))


def identifier_reward(sd, obs, prefix='identifier.'):
    """calculate_indentifier_reward (mansy_utils.py:42-49): 1 - mean((w_hat - w)^2) per transition."""
    pred = identifier_pred(sd, obs, prefix)
    return 1.0 - ((pred - obs[:, QOE_W[0]:QOE_W[1]]) ** 2).mean(dim=-1)


def relabel_rewards(sd, obs, rew, lamb, prefix='identifier.'):
    """mansy_ppo.py:41-48: rew <- (1 - lamb) * rew + lamb * identifier_reward  (float32)."""
    with torch.no_grad():
        return (1 - lamb) * rew + lamb * identifier_reward(sd, obs, prefix)


def adam_l2_step(p, g, m, v, step, lr, wd, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam(weight_decay=wd): L2 folded into the gradient (run_mansy.py:216,226)."""
    g = g + wd * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    p = p - (lr / bc1) * (m / (v.sqrt() / math.sqrt(bc2) + eps))
    return p, m, v


# ------------------------------------------------------------------------- tianshou 0.4.8 arithmetic (UNPINNED)
class RunningMeanStd:
    """T2: tianshou.utils.RunningMeanStd (mean=0, var=1, count=0; parallel-variance merge)."""

    def __init__(self):
        self.mean, self.var, self.count = 0.0, 1.0, 0

    def update(self, x):
        x = np.asarray(x, dtype=np.float64)
        bm, bv, bc = x.mean(), x.var(), len(x)
        delta = bm - self.mean
        tot = self.count + bc
        new_mean = self.mean + delta * bc / tot
        m2 = self.var * self.count + bv * bc + delta ** 2 * self.count * bc / tot
        self.mean, self.var, self.count = new_mean, m2 / tot, tot


def gae_returns(rew, v_s, v_s_next, done, end_flag, gamma, lam):
    """T2: BasePolicy.compute_episodic_return + _gae_return over ONE time-ordered sequence (float64 numpy).
    `done` masks the bootstrap value, `end_flag` (done or last collected index of an unfinished episode) cuts the trace."""
    rew, v_s, v_s_next = (np.asarray(a, dtype=np.float64) for a in (rew, v_s, v_s_next))
    v_s_next = v_s_next * (1.0 - np.asarray(done, dtype=np.float64))          # T2: value_mask = ~done
    delta = rew + gamma * v_s_next - v_s
    discount = (1.0 - np.asarray(end_flag, dtype=np.float64)) * (gamma * lam)
    adv = np.zeros_like(rew)
    gae = 0.0
    for i in range(len(rew) - 1, -1, -1):
        gae = delta[i] + discount[i] * gae
        adv[i] = gae
    return adv + v_s, adv


def compute_returns(rew, v_s, v_s_next, done, end_flag, ret_rms, gamma=0.95, lam=0.95, rew_norm=True, eps=1e-8):
    """T2: A2CPolicy._compute_returns.  v_s / v_s_next are raw critic outputs; returns are normalised by the running
    std of the UN-normalised returns, which is updated afterwards."""
    v_s = np.asarray(v_s, dtype=np.float64)
    v_s_next = np.asarray(v_s_next, dtype=np.float64)
    if rew_norm:
        scale = np.sqrt(ret_rms.var + eps)
        v_s, v_s_next = v_s * scale, v_s_next * scale
    unnorm, adv = gae_returns(rew, v_s, v_s_next, done, end_flag, gamma, lam)
    if rew_norm:
        returns = unnorm / np.sqrt(ret_rms.var + eps)
        ret_rms.update(unnorm)
    else:
        returns = unnorm
    return returns.astype(np.float32), adv.astype(np.float32)


def ppo_loss(logits, value, act, adv, logp_old, v_old, returns, eps_clip=0.2, vf_coef=0.5, ent_coef=0.02, norm_adv=True,
             value_clip=True, eps=0.0, dual_clip=None):
    """T2: PPOPolicy.learn body for one minibatch -> (loss, clip_loss, vf_loss, ent_loss)."""
    if norm_adv:
        # T2 (release 0.4.8, tianshou/policy/modelfree/ppo.py): `mean, std = b.adv.mean(), b.adv.std(); b.adv = (b.adv - mean) / std
        # # per-batch norm` -- torch's unbiased std, NO epsilon (the `+ self._eps` form is release 0.5.0's); eps stays a parameter
        # for callers restating the later release
        adv = (adv - adv.mean()) / (adv.std() + eps)
    logp_all = torch.log_softmax(logits, dim=-1)
    logp = logp_all.gather(1, act.long()[:, None])[:, 0]
    ratio = (logp - logp_old).exp()
    surr1 = ratio * adv
    surr2 = ratio.clamp(1.0 - eps_clip, 1.0 + eps_clip) * adv
    if dual_clip:
        # T2 (ppo.py): clip1 = min(surr1, surr2); clip2 = max(clip1, dual_clip * adv); clip_loss = -where(adv < 0, clip2, clip1).mean()
        clip1 = torch.min(surr1, surr2)
        clip2 = torch.max(clip1, dual_clip * adv)
        clip_loss = -torch.where(adv < 0, clip2, clip1).mean()
    else:
        clip_loss = -torch.min(surr1, surr2).mean()
    value = value.flatten()
    if value_clip:
        v_clip = v_old + (value - v_old).clamp(-eps_clip, eps_clip)
        vf_loss = torch.max((returns - value).pow(2), (returns - v_clip).pow(2)).mean()
    else:
        vf_loss = (returns - value).pow(2).mean()
    ent = -(logp_all.exp() * logp_all).sum(-1).mean()
    loss = clip_loss + vf_coef * vf_loss - ent_coef * ent
    return loss, clip_loss, vf_loss, ent


def split_indices(length, size, shuffle=True, merge_last=True):
    """T2: tianshou Batch.split(size, shuffle=True, merge_last=True): np.random.permutation, chunks of `size`, a last chunk
    shorter than `size` is merged into the one before it."""
    indices = np.random.permutation(length) if shuffle else np.arange(length)
    merge_last = merge_last and length % size > 0
    out = []
    for idx in range(0, length, size):
        if merge_last and idx + size + size >= length:
            out.append(indices[idx:])
            break
        out.append(indices[idx:idx + size])
    return out


def unique_params(sd):
    """The 28 unique actor-critic tensors of a 120-key policy state dict (the feature net is ONE module shared by actor and
    critic, run_mansy.py:207-209) as leaf tensors, plus the full-key view the forward functions index."""
    uniq, params = {}, {}
    for k, v in sd.items():
        if k.startswith('_actor_critic.') or k.startswith('identifier.'):
            continue
        key = k.replace('critic.feature_net.', 'actor.feature_net.')
        if key not in uniq:
            uniq[key] = v.clone().requires_grad_(True)
        params[k] = uniq[key]
    return uniq, params


def update(sd, obs, obs_next, act, rew, done, ret_rms, opt_state, lamb=0.5, use_identifier=True, batch_size=512, repeat=2, gamma=0.95,
           gae_lambda=0.95, rew_norm=True, eps_clip=0.2, vf_coef=0.5, ent_coef=0.02, max_grad_norm=1.0, lr=5e-4, wd=1e-2, eps=1e-8, on_step=None,
           before_step=None, norm_adv=True, value_clip=True, dual_clip=None, recompute_adv=False):
    """One whole PPOPolicy.update(0, buffer, is_train=True, batch_size, repeat) -- the reference's order of operations
    (bitrate_selection/models/mansy_ppo.py:36-59) over tianshou 0.4.8's process_fn / learn (T2):
      1. relabel: rew <- (1 - lamb) rew + lamb (1 - MSE(identifier(obs, obs.action_one_hot), obs.qoe_weight))      (:41-48)
      2. process_fn: v_s = critic(obs), v_s_ = critic(obs_next) (no grad), both x sqrt(ret_rms.var + eps); GAE per environment
         (end flag = done or last collected step); returns = (adv + v_s) / sqrt(var + eps); ret_rms.update(un-normalised
         returns of the whole buffer); logp_old = log_prob of the taken actions under the current actor
      3. learn: `repeat` passes; each pass np.random.permutation -> minibatches (merge_last); per minibatch ppo_loss, backward,
         clip_grad_norm_(actor_critic parameters, max_grad_norm), Adam with L2 weight decay (one step counter per parameter).
         recompute_adv (T2: `if self._recompute_adv and step > 0: batch = self._compute_returns(batch, self._buffer, self._indices)`):
         before every pass but the first, step 2's values / GAE / returns / ret_rms.update are redone with the current critic (the
         new v_s is also the value-clip reference); logp_old is NOT recomputed.
    Inputs are [T][N] step-major numpy / torch arrays (the build's rollout slabs; tianshou's VectorReplayBuffer.sample(0) would
    hand the same transitions environment-major -- the minibatch permutation is uniform either way).  `sd`: 120-key policy state
    dict; `opt_state`: dict with 'uniq' (leaf tensors, from unique_params), 'params', 'm', 'v', 'step' (per tensor), created on
    first use.  Returns (loss rows [n_minibatch_steps, 4], dict of intermediate arrays).
    `before_step(k, idx, opt_state, inter)` / `on_step(k, uniq)` are called around minibatch step k (teacher-forced tests snapshot
    the weights and Adam moments there).

    # T2: tianshou's ReplayBuffer stores `rew` as float64 (ReplayBuffer.add casts), so in the reference the relabelled reward of
    # mansy_ppo.py:47 and the GAE input are float64 values; this restatement -- like the build's RolloutBuffer.rew -- keeps the
    # reward float32 and widens it inside gae_returns.  The difference is one float32 rounding of the reward (~6e-8 relative),
    # far inside the 1e-4 bar."""
    obs, obs_next = torch.as_tensor(obs), torch.as_tensor(obs_next)
    T, N = obs.shape[0], obs.shape[1]
    n = T * N
    fo, fn_ = obs.reshape(n, -1), obs_next.reshape(n, -1)
    fact = torch.as_tensor(act).reshape(n).long()
    frew = torch.as_tensor(rew).reshape(n).float().clone()
    fdone = np.asarray(done).reshape(T, N).astype(bool)
    if 'uniq' not in opt_state:
        opt_state['uniq'], opt_state['params'] = unique_params(sd)
        opt_state['m'] = {k: torch.zeros_like(v) for k, v in opt_state['uniq'].items()}
        opt_state['v'] = {k: torch.zeros_like(v) for k, v in opt_state['uniq'].items()}
        opt_state['step'] = {k: 0 for k in opt_state['uniq']}
    uniq, params = opt_state['uniq'], opt_state['params']
    if use_identifier:
        frew = relabel_rewards(sd, fo, frew, lamb)
    with torch.no_grad():
        v_s = critic_value(params, fo).flatten()
        v_next = critic_value(params, fn_).flatten()
        logp_old = torch.log_softmax(actor_logits(params, fo), -1).gather(1, fact[:, None])[:, 0]
    def compute_returns_now(v_s, v_next):
        scale = np.sqrt(ret_rms.var + eps) if rew_norm else 1.0
        vs64 = v_s.numpy().astype(np.float64).reshape(T, N) * scale
        vn64 = v_next.numpy().astype(np.float64).reshape(T, N) * scale
        r2 = frew.numpy().reshape(T, N)
        unn = np.zeros((T, N))
        adv = np.zeros((T, N))
        for e in range(N):
            end = fdone[:, e].copy()
            end[-1] = True                                     # T2: the last collected index of an unfinished episode ends the trace
            unn[:, e], adv[:, e] = gae_returns(r2[:, e], vs64[:, e], vn64[:, e], fdone[:, e], end, gamma, gae_lambda)
        returns = unn / scale
        if rew_norm:
            ret_rms.update(unn.reshape(-1))
        return torch.from_numpy(returns.reshape(-1).astype(np.float32)), torch.from_numpy(adv.reshape(-1).astype(np.float32))

    returns_t, adv_t = compute_returns_now(v_s, v_next)
    rows = []
    inter = dict(rew=frew.numpy(), v_s=v_s.numpy(), v_next=v_next.numpy(), logp_old=logp_old.numpy(), returns=returns_t.numpy(),
                 adv=adv_t.numpy())
    for pass_no in range(repeat):
        if recompute_adv and pass_no > 0:
            with torch.no_grad():
                v_s = critic_value(params, fo).flatten()
                v_next = critic_value(params, fn_).flatten()
            returns_t, adv_t = compute_returns_now(v_s, v_next)
            inter.update(v_s=v_s.numpy(), v_next=v_next.numpy(), returns=returns_t.numpy(), adv=adv_t.numpy())
        for idx in split_indices(n, batch_size):
            if before_step is not None:
                before_step(len(rows), np.asarray(idx), opt_state, inter)
            idx = torch.from_numpy(np.asarray(idx)).long()
            for p in uniq.values():
                p.grad = None
            loss, clip, vf, ent = ppo_loss(actor_logits(params, fo[idx]), critic_value(params, fo[idx]), fact[idx], adv_t[idx], logp_old[idx],
                                           v_s[idx], returns_t[idx], eps_clip, vf_coef, ent_coef, norm_adv=norm_adv, value_clip=value_clip,
                                           dual_clip=dual_clip)
            loss.backward()
            if max_grad_norm:
                torch.nn.utils.clip_grad_norm_(list(uniq.values()), max_grad_norm)
            with torch.no_grad():
                for k, p in uniq.items():
                    opt_state['step'][k] += 1
                    p1, opt_state['m'][k], opt_state['v'][k] = adam_l2_step(p, p.grad, opt_state['m'][k], opt_state['v'][k], opt_state['step'][k],
                                                                            lr, wd)
                    p.copy_(p1)
            rows.append([loss.item(), clip.item(), vf.item(), ent.item()])
            if on_step is not None:
                on_step(len(rows) - 1, uniq)
    return np.array(rows), inter


def bc_loss(logits, act, ent_coef=0.1):
    """utils/mansy_utils.py:60-66: CrossEntropyLoss(logits, expert action) - 0.1 * Categorical(logits).entropy().mean()
    -> (loss, cross entropy, mean entropy)."""
    logp_all = torch.log_softmax(logits, dim=-1)
    ce = -logp_all.gather(1, act.long()[:, None])[:, 0].mean()
    ent = -(logp_all.exp() * logp_all).sum(-1).mean()
    return ce - ent_coef * ent, ce, ent


def categorical_sample(logits, u):
    """Inverse-CDF sample from softmax(logits) given uniforms u in [0,1): first index whose cumulative probability
    exceeds u (float32 sequential cumsum) -- the build's externally-driven sampler (SURVEY 8c determinism caveat)."""
    p = torch.softmax(logits.float(), dim=-1)
    c = torch.cumsum(p, dim=-1)
    idx = (c <= u[:, None]).sum(dim=-1)
    return idx.clamp(max=logits.shape[1] - 1)


def make_policy_state_dict(seed, scale=1.0):
    """Seeded synthetic weights with the reference checkpoint layout: 120 keys actor.* / critic.* /
    _actor_critic.{actor,critic}.* / identifier.* (feature_net tensors identical in the actor and critic copies)."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, fan_in):
        return torch.randn(*shape, generator=g) * (scale / math.sqrt(fan_in))

    def fnet(last_in):
        d = {}
        for name, cin, k in (('conv1d1', 1, 8), ('conv1d2', 5, 64), ('conv1d3', 5, 64), ('conv1d4', 1, 64), ('conv1d5', 1, 8), ('conv1d6', 1, 8),
                             ('conv1d7', 1, 8), ('conv1d8', 1, 8)):
            d[f'feature_net.{name}.0.weight'] = rn(HID, cin, k, fan_in=cin * k)
            d[f'feature_net.{name}.0.bias'] = rn(HID, fan_in=16)
        d['feature_net.fc1.0.weight'] = rn(HID, 1, fan_in=1)
        d['feature_net.fc1.0.bias'] = rn(HID, fan_in=16)
        d['feature_net.fc2.0.weight'] = rn(HID, last_in, fan_in=last_in)
        d['feature_net.fc2.0.bias'] = rn(HID, fan_in=16)
        return d
    shared = fnet(3)
    actor = dict(shared)
    actor.update({'fc.0.weight': rn(HID, 1280, fan_in=1280), 'fc.0.bias': rn(HID, fan_in=16), 'out.weight': rn(N_ACTION, HID, fan_in=HID),
                  'out.bias': rn(N_ACTION, fan_in=16)})
    critic = dict(shared)
    critic.update({'fc.0.weight': rn(HID, 1280, fan_in=1280), 'fc.0.bias': rn(HID, fan_in=16), 'out.weight': rn(1, HID, fan_in=HID),
                   'out.bias': rn(1, fan_in=16)})
    ident = fnet(N_ACTION)
    ident.update({'fc.0.weight': rn(HID, 1280, fan_in=1280), 'fc.0.bias': rn(HID, fan_in=16), 'out.weight': rn(3, HID, fan_in=HID),
                  'out.bias': rn(3, fan_in=16)})
    sd = {}
    for k, v in actor.items():
        sd['actor.' + k] = v
    for k, v in critic.items():
        sd['critic.' + k] = v
    for k, v in actor.items():
        sd['_actor_critic.actor.' + k] = v
    for k, v in critic.items():
        sd['_actor_critic.critic.' + k] = v
    for k, v in ident.items():
        sd['identifier.' + k] = v
    return sd
