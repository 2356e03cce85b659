"""Pins oracle/a2c_oracle.py (A2C baseline restatement) against vectors produced by the imported reference
(tools/gen_golden_a2c.py): network outputs + every gradient (models/simple_rl.py), and SimpleRLEnv episodes
(envs/simple_rl_env.py) reproduced from the MANSYEnv C oracle + the observation mapping -- bit-exact."""
import os
import numpy as np
import pytest
import torch
from oracle import a2c_oracle as ao
from oracle import env as oenv

Z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'a2c_reference.npz'))
FIELDS = ('size', 'quality', 'video_len', 'vp_gt', 'vp_pred', 'vp_acc', 'vp_start', 'vp_end', 'trace_bw', 'trace_len', 'samples')


def golden_sd(requires_grad=False):
    uniq, sd = {}, {}
    for k in Z['net/keys']:
        k = str(k)
        key = k.replace('critic.feature_net.', 'actor.feature_net.')
        if key not in uniq:
            uniq[key] = torch.from_numpy(Z['net/w::' + k]).clone().requires_grad_(requires_grad)
        sd[k] = uniq[key]
    return sd, uniq


def test_networks_vs_reference():
    sd, uniq = golden_sd(True)
    assert len(sd) == 28
    obs = torch.from_numpy(Z['net/obs'])
    probs, value = ao.actor_probs(sd, obs), ao.critic_value(sd, obs)
    np.testing.assert_allclose(probs.detach().numpy(), Z['net/probs'], atol=2e-6)
    np.testing.assert_allclose(value.detach().numpy(), Z['net/value'], atol=2e-5, rtol=1e-5)
    ((probs * torch.from_numpy(Z['net/c1'])).sum() + (value * torch.from_numpy(Z['net/c2'])).sum()).backward()
    for k, p in uniq.items():
        ref = Z['net/g::' + k]
        np.testing.assert_allclose(p.grad.numpy(), ref, atol=2e-5 * max(np.abs(ref).max(), 1e-3), rtol=0, err_msg=k)


@pytest.mark.parametrize('tag', ['train', 'valid'])
def test_simple_rl_env_bit_exact(tag):
    arrays = {k: Z[f'{tag}/{k}'] for k in FIELDS}
    seed, worker_num, n_ep, norm = (int(x) for x in Z[f'{tag}/meta'])
    T = oenv.EnvTables(arrays, Z[f'{tag}/qoe_w'], train_identifier_reward=bool(norm))       # train: reward = qoe / sum(w)
    env = oenv.Env(T, seed=seed, worker_num=worker_num)
    for e in range(n_ep):
        obs = env.reset()
        assert env.sample_id == int(Z[f'{tag}/ep{e}/sample_id'])
        ref = Z[f'{tag}/ep{e}/obs']
        np.testing.assert_array_equal(ao.simple_obs(obs, 0.0, -1, fresh=True).view(np.uint32), ref[0].view(np.uint32))
        for t, a in enumerate(Z[f'{tag}/ep{e}/act']):
            obs, r, done, parts = env.step(int(a))
            assert done == bool(Z[f'{tag}/ep{e}/done'][t])
            assert np.float32(r).view(np.uint32) == Z[f'{tag}/ep{e}/rew'][t].view(np.uint32), (e, t)
            row = ao.simple_obs(obs, parts[2], int(a))
            bad = np.nonzero(row.view(np.uint32) != ref[t + 1].view(np.uint32))[0]
            assert bad.size == 0, (e, t, bad[:8], row[bad[:8]], ref[t + 1][bad[:8]])


def test_loss_uses_torch_categorical_probs_semantics():
    """Categorical(probs) renormalises and clamps: the restated loss is finite and differentiable for one-hot-ish rows."""
    probs = torch.tensor([[1.0, 0.0, 0.0], [0.2, 0.3, 0.5]], requires_grad=True)
    loss, al, vf, ent = ao.a2c_loss(probs, torch.tensor([[0.1], [0.2]]), torch.tensor([1, 2]), torch.tensor([1.0, -1.0]), torch.tensor([0.0, 1.0]))
    loss.backward()
    assert torch.isfinite(loss) and torch.isfinite(probs.grad).all()
