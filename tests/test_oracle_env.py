"""Pins oracle/env.c (sequential C restatement of MANSYEnv + simulator + QoE + tile-rate allocation) against
trajectories produced by the imported reference (tools/gen_golden_env.py).  Bit-exact: observations, rewards,
done flags, bitrate decisions."""
import os
import numpy as np
import pytest
from oracle import env as oenv

Z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'env_reference.npz'))
TAGS = ['train_id', 'valid_w3', 'train_noid']


def tables_for(tag):
    arrays = {k: Z[f'{tag}/{k}'] for k in ('size', 'quality', 'video_len', 'vp_gt', 'vp_pred', 'vp_acc', 'vp_start', 'vp_end', 'trace_bw',
                                           'trace_len', 'samples')}
    meta = Z[f'{tag}/meta']
    return oenv.EnvTables(arrays, Z[f'{tag}/qoe_w'], train_identifier_reward=bool(meta[4])), meta


@pytest.mark.parametrize('tag', TAGS)
def test_env_trajectories_bit_exact(tag):
    T, meta = tables_for(tag)
    seed, worker_num, _, n_ep, _ = (int(x) for x in meta)
    env = oenv.Env(T, seed=seed, worker_num=worker_num)
    for e in range(n_ep):
        obs = env.reset()
        assert env.sample_id == int(Z[f'{tag}/ep{e}/sample_id'])
        ref_obs = Z[f'{tag}/ep{e}/obs']
        np.testing.assert_array_equal(obs.view(np.uint32), ref_obs[0].view(np.uint32))
        acts, rews, dones = Z[f'{tag}/ep{e}/act'], Z[f'{tag}/ep{e}/rew'], Z[f'{tag}/ep{e}/done']
        for t, a in enumerate(acts):
            obs, r, done, _ = env.step(int(a))
            assert done == bool(dones[t]), (e, t)
            assert np.float32(r).view(np.uint32) == rews[t].view(np.uint32), (e, t, r, rews[t])
            bad = np.nonzero(obs.view(np.uint32) != ref_obs[t + 1].view(np.uint32))[0]
            assert bad.size == 0, (e, t, bad[:10], obs[bad[:10]], ref_obs[t + 1][bad[:10]])
        assert done


def test_allocate_tile_rates_all_actions():
    pv, ver = Z['alloc/pred_viewport'], Z['alloc/versions']
    for k in range(pv.shape[0]):
        for a in range(15):
            rin, rout = oenv.ACTION2RATES[a]
            np.testing.assert_array_equal(oenv.allocate_tile_rates(rin, rout, pv[k]), ver[k, a])


def test_sample_enumeration():
    for mode in ('train', 'valid', 'test'):
        nv, nu, nt, nq = (int(x) for x in Z[f'enum/{mode}_lens'])
        fn = oenv.generate_environment_test_samples if mode == 'test' else oenv.generate_environment_samples
        np.testing.assert_array_equal(fn(nv, nu, nt, nq), Z[f'enum/{mode}'])


def test_allocate_tile_rates_all_25_version_pairs_vs_imported_reference():
    """The 15 actions only reach (in, out) pairs with in >= out; the function itself takes any pair (utils/common.py:142-193):
    all 25 on six viewports (empty and full prediction included) through the imported reference, tools/gen_golden_alloc25.py."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'alloc25_reference.npz'))
    for k, pv in enumerate(G['pred_viewport']):
        for i in range(5):
            for o in range(5):
                np.testing.assert_array_equal(oenv.allocate_tile_rates(i, o, pv), G['versions'][k, i, o])
