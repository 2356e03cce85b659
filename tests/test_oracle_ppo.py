"""Pins oracle/ppo_oracle.py (network restatement: FeatureNet/Actor/Critic/QoEIdentifier, identifier reward,
train_identifier) against golden vectors produced by the imported reference (tools/gen_golden_ppo.py), and checks the
UNPINNED tianshou-0.4.8 restatements (GAE, running return normaliser, PPO loss) on hand-derived cases."""
import os
import numpy as np
import pytest
import torch
from oracle import ppo_oracle as po

Z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ppo_reference.npz'))


def _sd(grad=False):
    sd = po.make_policy_state_dict(int(Z['wseed']))
    if grad:
        # shared feature net: one leaf per unique tensor
        uniq = {}
        out = {}
        for k, v in sd.items():
            if k.startswith('_actor_critic.'):
                continue
            key = k.replace('critic.feature_net.', 'actor.feature_net.') if k.startswith('critic.feature_net.') else k
            if key not in uniq:
                uniq[key] = v.clone().requires_grad_(True)
            out[k] = uniq[key]
        return out, uniq
    return sd, None


def test_forward_outputs():
    sd, _ = _sd()
    obs = torch.from_numpy(Z['obs'][:64])
    with torch.no_grad():
        np.testing.assert_allclose(po.actor_logits(sd, obs).numpy(), Z['logits'], atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(po.critic_value(sd, obs).numpy(), Z['value'], atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(po.identifier_pred(sd, obs).numpy(), Z['ident'], atol=2e-6, rtol=1e-5)


def test_gradients():
    sd, uniq = _sd(grad=True)
    obs = torch.from_numpy(Z['obs'][:64])
    loss = (po.actor_logits(sd, obs) * torch.from_numpy(Z['ct_logits'])).sum() + (po.critic_value(sd, obs) * torch.from_numpy(Z['ct_value'])).sum() \
        + (po.identifier_pred(sd, obs) * torch.from_numpy(Z['ct_ident'])).sum()
    loss.backward()
    n = 0
    for key in Z.files:
        if key.startswith('grad::'):
            k = key[6:]
            ref = Z[key]
            g = uniq[k].grad.numpy()
            np.testing.assert_allclose(g, ref, atol=2e-5 * max(np.abs(ref).max(), 1e-3), rtol=0, err_msg=k)
            n += 1
    assert n == 24 + 4 + 24


def test_identifier_reward_unbatched_rows():
    sd, _ = _sd()
    rows = Z['ident_reward_rows']
    with torch.no_grad():
        r = po.identifier_reward(sd, torch.from_numpy(Z['obs'][rows]))
    np.testing.assert_allclose(r.numpy(), Z['ident_reward'], atol=1e-6, rtol=0)


def test_train_identifier_two_rounds():
    """mansy_utils.py:9-39: shuffle (np.random), 80/20 split, `update_round` full-batch MSE steps with Adam(1e-4, L2 1e-2)."""
    sd, _ = _sd()
    n = int(Z['ti_n'])
    np.random.seed(int(Z['ti_npseed']))
    idx = np.arange(n)
    np.random.shuffle(idx)
    obs = torch.from_numpy(Z['obs'][:n][idx])
    ntr = int(n * 0.8)
    tr, va = obs[:ntr], obs[ntr:]
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith('identifier.')}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v2 = {k: torch.zeros_like(v) for k, v in params.items()}
    losses = []
    for step in (1, 2):
        for p in params.values():
            p.grad = None
        loss = torch.nn.functional.mse_loss(po.identifier_pred(params, tr), tr[:, 745:748])
        loss.backward()
        losses.append(loss.item())
        with torch.no_grad():
            for k, p in params.items():
                p1, m[k], v2[k] = po.adam_l2_step(p, p.grad, m[k], v2[k], step, lr=1e-4, wd=1e-2)
                p.copy_(p1)
    with torch.no_grad():
        losses.append(torch.nn.functional.mse_loss(po.identifier_pred(params, va), va[:, 745:748]).item())
    np.testing.assert_allclose(losses, Z['ti_losses'], rtol=2e-5, atol=1e-7)
    for key in Z.files:
        if key.startswith('ti_after::'):
            np.testing.assert_allclose(params[key[10:]].detach().numpy(), Z[key], atol=2e-6, rtol=1e-5, err_msg=key)


def test_checkpoint_layout_matches_shipped_files():
    sd = po.make_policy_state_dict(1)
    want = [s.split('|') for s in Z['layout::best_policy.pth']]
    assert [k for k, _ in want] == list(sd.keys())
    for k, shp in want:
        assert 'x'.join(map(str, sd[k].shape)) == shp, k
    ident = [s.split('|')[0] for s in Z['layout::best_identifier.pth']]
    assert ident == [k[len('identifier.'):] for k in sd if k.startswith('identifier.')]


# ---- tianshou restatements: the known answers tianshou's own repository publishes (v0.4.8 test/base/test_returns.py) ----
KA = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'tianshou_known_answers.npz'))


def _ka_case(i):
    g = {k: KA[f'c{i}_{k}'] for k in ('done', 'rew', 'v_next', 'v_s', 'returns')}
    end = g['done'].copy()
    end[-1] = 1                                  # the last collected index cuts the trace of an unfinished episode
    return g, end, float(KA[f'c{i}_gamma']), float(KA[f'c{i}_lambda'])


@pytest.mark.parametrize('i', range(int(KA['n_cases'])))
def test_gae_returns_reproduces_tianshou_published_known_answers(i):
    """P5 pin: BasePolicy.compute_episodic_return's published vectors (discounted returns at gamma 0.1 / lambda 1 over finished,
    unfinished and back-to-back episodes; the 12-step GAE case at gamma 0.99 / lambda 0.95 with bootstrap values)."""
    g, end, gamma, lam = _ka_case(i)
    ret, adv = po.gae_returns(g['rew'], g['v_s'], g['v_next'], g['done'], end, gamma, lam)
    # the published answers carry 4 decimals (tianshou compares them with np.allclose at its default rtol 1e-5 / atol 1e-8)
    np.testing.assert_allclose(ret, g['returns'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(adv, ret - g['v_s'], rtol=1e-12, atol=1e-12)
    # compute_returns without reward normalisation is the same function in float32
    r32, a32 = po.compute_returns(g['rew'], g['v_s'], g['v_next'], g['done'], end, po.RunningMeanStd(), gamma, lam, rew_norm=False)
    np.testing.assert_allclose(r32, g['returns'], rtol=1e-5)


def test_compute_returns_normalisation_is_a_rescaling_of_the_published_case():
    """A2CPolicy._compute_returns with rew_norm: critic outputs are scaled by sqrt(var + eps) on the way in, returns divided by
    it on the way out, the running moments updated with the UN-normalised returns afterwards -- on the published 12-step case
    with a primed normaliser."""
    g, end, gamma, lam = _ka_case(3)
    rms = po.RunningMeanStd()
    rms.update(np.array([1.0, 3.0, 5.0, 11.0]))
    var0 = rms.var
    scale = np.sqrt(var0 + 1e-8)
    ret, adv = po.compute_returns(g['rew'], g['v_s'] / scale, g['v_next'] / scale, g['done'], end, rms, gamma, lam, rew_norm=True)
    np.testing.assert_allclose(ret, g['returns'] / scale, rtol=1e-5)
    allv = np.concatenate([[1.0, 3.0, 5.0, 11.0], g['returns']])
    np.testing.assert_allclose([rms.mean, rms.var, rms.count], [allv.mean(), allv.var(), 16], rtol=1e-5)


# ---- hand-derived known answers for the rest of the tianshou arithmetic ---------------------------------------
def test_gae_hand_case():
    # 3 steps, episode ends at step 1 (done), step 2 is the last collected index of an unfinished episode
    rew = [1.0, 2.0, 3.0]
    v, vn = [0.5, 0.4, 0.3], [0.4, 9.9, 0.2]
    done, end = [0, 1, 0], [0, 1, 1]
    g, l = 0.9, 0.8
    d2 = 3.0 + g * 0.2 - 0.3
    d1 = 2.0 + 0.0 - 0.4                        # bootstrap masked by done
    d0 = 1.0 + g * 0.4 - 0.5
    a2, a1 = d2, d1
    a0 = d0 + g * l * a1
    ret, adv = po.gae_returns(rew, v, vn, done, end, g, l)
    np.testing.assert_allclose(adv, [a0, a1, a2], rtol=1e-12)
    np.testing.assert_allclose(ret, [a0 + 0.5, a1 + 0.4, a2 + 0.3], rtol=1e-12)


def test_running_mean_std_merge_equals_batch_stats():
    rs = np.random.RandomState(0)
    a, b = rs.randn(100) * 3 + 1, rs.randn(57) - 2
    r = po.RunningMeanStd()
    r.update(a)
    r.update(b)
    allv = np.concatenate([a, b])
    np.testing.assert_allclose([r.mean, r.var, r.count], [allv.mean(), allv.var(), 157], rtol=1e-12)


def test_ppo_loss_hand_case():
    logits = torch.tensor([[0.0, 0.0], [np.log(3.0), 0.0]])
    act = torch.tensor([0, 1])
    logp_old = torch.log(torch.tensor([0.25, 0.5]))
    adv = torch.tensor([1.0, -1.0])
    value, v_old, ret = torch.tensor([1.0, 0.0]), torch.tensor([0.5, 0.1]), torch.tensor([2.0, 0.0])
    loss, clip, vf, ent = po.ppo_loss(logits, value, act, adv, logp_old, v_old, ret, norm_adv=False)
    # ratios: 0.5/0.25 = 2 (clipped to 1.2 with adv>0 -> min(2,1.2)=1.2), 0.25/0.5 = 0.5 (adv<0: min(-0.5,-0.8) = -0.8)
    np.testing.assert_allclose(clip.item(), -(1.2 - 0.8) / 2, rtol=1e-6)
    # value clip: v_clip = [0.5+0.2, 0.1-0.1] = [0.7, 0.0]; max((2-1)^2,(2-0.7)^2)=1.69 ; max(0,0)=0
    np.testing.assert_allclose(vf.item(), 1.69 / 2, rtol=1e-6)
    h0 = np.log(2.0)
    h1 = -(0.75 * np.log(0.75) + 0.25 * np.log(0.25))
    np.testing.assert_allclose(ent.item(), (h0 + h1) / 2, rtol=1e-6)
    np.testing.assert_allclose(loss.item(), clip.item() + 0.5 * vf.item() - 0.02 * ent.item(), rtol=1e-6)
