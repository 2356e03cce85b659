"""CPU: the oracle stack (C environment oracle/env.c + oracle/ppo_oracle.py's actor) driven closed-loop, greedy, over 96 real Jin2022 x 4G
test-split episodes with the weights the reference ships, against the IMPORTED reference doing the same (tests/golden/greedy_real_reference.npz,
tools/gen_golden_greedy_real.py): every decision, every reward bit for bit, the CSV rows -- the oracle pinned at the real tables' scale."""
import os

import numpy as np
import torch

import _jin2022_tree as jt
from oracle import env as oenv
from oracle import ppo_oracle as po

HERE = os.path.dirname(__file__)


def test_oracle_greedy_closed_loop_equals_imported_reference_on_real_tables():
    G = jt.load()
    R = np.load(os.path.join(HERE, 'golden', 'greedy_real_reference.npz'))
    W = np.load(os.path.join(HERE, 'golden', 'shipped_checkpoint_reference.npz'))
    uniq = {k[3:]: torch.from_numpy(W[k]) for k in W.files if k.startswith('w::')}
    sd = {k: uniq[k.replace('_actor_critic.', '').replace('critic.feature_net.', 'actor.feature_net.')] for k in po.make_policy_state_dict(0)}
    fields = ('size', 'quality', 'video_len', 'vp_gt', 'vp_pred', 'vp_acc', 'vp_start', 'vp_end', 'trace_bw', 'trace_len', 'samples')
    OT = oenv.EnvTables({k: G['test/' + k] for k in fields}, G['test/qoe_w'], train_identifier_reward=False)
    top2 = np.sort(R['logits'], -1)
    gap = top2[..., -1] - top2[..., -2]
    row = np.zeros((1, 780), np.float32)
    identical = 0
    for e, entry in enumerate(R['entries'][:48]):
        env = oenv.Env(OT, seed=int(entry), worker_num=1440)
        obs = env.reset()
        assert env.sample_id == int(entry)
        ok = True
        for t in range(51):
            row[0, :779] = obs[:779]
            with torch.no_grad():
                lg = po.actor_logits(sd, torch.from_numpy(row))[0].numpy()
            a = int(lg.argmax())
            np.testing.assert_allclose(lg, R['logits'][e, t], atol=2e-5, rtol=0)
            if a != int(R['act'][e, t]):
                assert gap[e, t] < 1e-4
                ok = False
                break
            obs, r, done, _ = env.step(a)
            assert np.float32(r).view(np.uint32) == R['rew'][e, t].view(np.uint32), (int(entry), t)
        identical += ok
        if ok:
            assert done
    assert identical >= 44
