"""TEST INFRASTRUCTURE ONLY (oracle).  CPU restatement of the reference's A2C baseline:

  * networks: bitrate_selection/models/simple_rl.py:9-63 (FeatureNet: five branches -> concat 640; Actor: softmax of
    out(LeakyReLU(fc(features))) -- the reference calls these probabilities "logits"; Critic) on flat 416-float rows
    [throughput 0:8 | chunk_sizes 8:328 | rebuffer 328 | last_bitrates 329:331 | pred_viewport 331:395 | zero pad].
    Pinned by tests/golden/a2c_reference.npz (tools/gen_golden_a2c.py imports the reference modules).
  * SimpleRLEnv observation (envs/simple_rl_env.py:85-170) as a function of the MANSYEnv oracle's step outputs; pinned by
    the same file (imported SimpleRLEnv episodes).
  * loss / optimiser: tianshou==0.4.8 A2CPolicy.learn behind run_simple_rl.py:190-211 (T2, not installed: PARITY UNPINNED):
    dist = Categorical(probs) (run_simple_rl.py:192-193 passes the softmax output positionally = `probs`),
    loss = -(log_prob * adv).mean() + vf_coef * mse(returns, value) - ent_coef * entropy.mean(), clip_grad_norm_,
    torch.optim.RMSprop(lr).  torch's own Categorical / RMSprop are used here, so only the loss composition is restated."""
import numpy as np
import torch
import torch.nn.functional as F

LD = 416
SL = dict(throughput=(0, 8), chunk_sizes=(8, 328), rebuffer=(328, 329), last_bitrates=(329, 331), pred_viewport=(331, 395))
BRANCHES = [('conv1d_1', 'throughput'), ('conv1d_2', 'chunk_sizes'), ('fc1', 'rebuffer'), ('fc2', 'last_bitrates'), ('fc3', 'pred_viewport')]


def feature_net(sd, prefix, obs):
    outs = []
    for name, key in BRANCHES:
        a, b = SL[key]
        w = sd[f'{prefix}feature_net.{name}.0.weight']
        outs.append(F.leaky_relu(obs[:, a:b] @ w.reshape(w.shape[0], -1).t() + sd[f'{prefix}feature_net.{name}.0.bias']))
    return torch.cat(outs, dim=-1)


def _head(sd, prefix, feats):
    h = F.leaky_relu(feats @ sd[prefix + 'fc.0.weight'].t() + sd[prefix + 'fc.0.bias'])
    return h @ sd[prefix + 'out.weight'].t() + sd[prefix + 'out.bias']


def actor_probs(sd, obs, prefix='actor.'):
    return torch.softmax(_head(sd, prefix, feature_net(sd, prefix, obs)), dim=1)


def critic_value(sd, obs, prefix='critic.'):
    return _head(sd, prefix, feature_net(sd, prefix, obs))


def a2c_loss(probs, value, act, adv, returns, vf_coef=0.5, ent_coef=0.1):
    """T2: A2CPolicy.learn body for one minibatch -> (loss, actor_loss, vf_loss, ent_loss)."""
    dist = torch.distributions.Categorical(probs)
    log_prob = dist.log_prob(act.long())
    actor_loss = -(log_prob * adv).mean()
    vf_loss = F.mse_loss(returns, value.flatten())
    ent_loss = dist.entropy().mean()
    return actor_loss + vf_coef * vf_loss - ent_coef * ent_loss, actor_loss, vf_loss, ent_loss


def categorical_sample(probs, u):
    """Inverse-CDF sample from Categorical(probs) (probs renormalised like torch does) given uniforms u in [0,1)."""
    p = probs.float()
    p = p / p.sum(-1, keepdim=True)
    c = torch.cumsum(p, dim=-1)
    return (c <= u[:, None]).sum(dim=-1).clamp(max=probs.shape[1] - 1)


def simple_obs(mansy_obs, qoe2, action, video_rates=(1, 5, 8, 16, 35), fresh=False):
    """SimpleRLEnv's state (simple_rl_env.py:112-118,148-165) from the MANSYEnv oracle's 779-float observation of the
    same step: the throughput ring, next-chunk sizes and predicted viewport are the same arrays; `rebuffer` is the raw
    rebuffering time of the step (float32) and `last_bitrates` the two chosen bitrates / max bitrate in float32
    (both zero right after reset)."""
    a2r = [(1, 0), (2, 0), (3, 0), (4, 0), (2, 1), (3, 1), (4, 1), (3, 2), (4, 2), (4, 3), (0, 0), (1, 1), (2, 2), (3, 3), (4, 4)]
    row = np.zeros(LD, np.float32)
    row[0:8] = mansy_obs[0:8]
    row[8:328] = mansy_obs[8:328]
    row[331:395] = mansy_obs[648:712]
    if not fresh:
        rin, rout = a2r[action] if 0 <= action < 15 else (0, 0)
        row[328] = np.float32(qoe2)
        row[329] = np.float32(video_rates[rin]) / np.float32(video_rates[-1])
        row[330] = np.float32(video_rates[rout]) / np.float32(video_rates[-1])
    return row


def make_state_dict(seed, scale=1.0):
    """Seeded weights in the reference's state_dict layout (actor.* / critic.* with the shared feature net duplicated)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    shapes = [('conv1d_1', (128, 1, 8)), ('conv1d_2', (128, 1, 320)), ('fc1', (128, 1)), ('fc2', (128, 2)), ('fc3', (128, 64))]
    for name, shp in shapes:
        fan = int(np.prod(shp[1:]))
        sd[f'actor.feature_net.{name}.0.weight'] = torch.randn(*shp, generator=g) * scale / np.sqrt(fan)
        sd[f'actor.feature_net.{name}.0.bias'] = torch.randn(shp[0], generator=g) * 0.05
    for head, nout in (('actor', 15), ('critic', 1)):
        sd[f'{head}.fc.0.weight'] = torch.randn(128, 640, generator=g) * scale / np.sqrt(640)
        sd[f'{head}.fc.0.bias'] = torch.randn(128, generator=g) * 0.05
        sd[f'{head}.out.weight'] = torch.randn(nout, 128, generator=g) * scale / np.sqrt(128)
        sd[f'{head}.out.bias'] = torch.randn(nout, generator=g) * 0.05
    for k in [k for k in sd if k.startswith('actor.feature_net.')]:
        sd['critic.' + k[6:]] = sd[k]
    return sd
