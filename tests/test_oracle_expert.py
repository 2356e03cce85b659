"""Pins the MPC-expert part of oracle/env.c (profile cache + exhaustive look-ahead search, expert_env.py:126-181,358-422)
against episodes produced by the imported reference (tools/gen_golden_expert.py).  Bit-exact: cache entries, chosen
actions, rewards, observations."""
import os
import numpy as np
import pytest
from oracle import env as oenv

Z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'expert_reference.npz'))
TAGS = ['h1', 'h2', 'h3', 'h4']


def setup(tag):
    arrays = {k: Z[f'{tag}/{k}'] for k in ('size', 'quality', 'video_len', 'vp_gt', 'vp_pred', 'vp_acc', 'vp_start', 'vp_end', 'trace_bw',
                                           'trace_len', 'samples')}
    T = oenv.EnvTables(arrays, Z[f'{tag}/qoe_w'], train_identifier_reward=False)
    horizon, n_ep = (int(x) for x in Z[f'{tag}/meta'])
    return T, oenv.Expert(T, Z[f'{tag}/vp_video'], horizon), n_ep


@pytest.mark.parametrize('tag', TAGS)
def test_expert_cache_bit_exact(tag):
    T, ex, _ = setup(tag)
    filled = Z[f'{tag}/cache/filled']
    assert filled.any()
    for k in ('gt_quality', 'pred_quality', 'gt_var', 'pred_var'):
        np.testing.assert_array_equal(ex.cache[k][filled].view(np.uint32), Z[f'{tag}/cache/{k}'][filled].view(np.uint32), err_msg=k)
    for k in ('gt_size', 'pred_size'):
        np.testing.assert_array_equal(ex.cache[k][filled], Z[f'{tag}/cache/{k}'][filled], err_msg=k)


@pytest.mark.parametrize('tag', TAGS)
def test_expert_episodes_bit_exact(tag):
    T, ex, n_ep = setup(tag)
    env = oenv.Env(T, seed=0, worker_num=1)     # ExpertEnv.reset walks its sample list in order (expert_env.py:185)
    for e in range(n_ep):
        obs = env.reset()
        assert env.sample_id == int(Z[f'{tag}/ep{e}/sample_id'])
        ref_obs = Z[f'{tag}/ep{e}/obs']
        np.testing.assert_array_equal(obs.view(np.uint32), ref_obs[0].view(np.uint32))
        acts, rews, dones = Z[f'{tag}/ep{e}/act'], Z[f'{tag}/ep{e}/rew'], Z[f'{tag}/ep{e}/done']
        for t, a in enumerate(acts):
            assert ex.choose_action(env) == int(a), (tag, e, t)
            obs, r, done, _ = env.step(int(a))
            assert done == bool(dones[t])
            assert np.float32(r).view(np.uint32) == rews[t].view(np.uint32), (e, t, r, rews[t])
            np.testing.assert_array_equal(obs.view(np.uint32), ref_obs[t + 1].view(np.uint32))
