#!/usr/bin/env python3
"""tests/golden/tianshou_known_answers.npz -- the known answers tianshou's OWN repository publishes for
`BasePolicy.compute_episodic_return` (release v0.4.8, test/base/test_returns.py::test_episodic_returns), the function behind
`A2CPolicy._compute_returns` that the reference's PPO inherits (bitrate_selection/models/mansy_ppo.py:53).

tianshou cannot be installed here (no network), so these vectors are DATA typed in from that published test -- inputs (done,
rew, optional v_s_) and the expected returns its assertions hold -- not output of code run in this container.  In that test
`v_s_` defaults to zeros and `v_s = np.roll(v_s_, 1)`; the last index of the buffer ends the trace of an unfinished episode.
Cases 1-3: gamma = 0.1, gae_lambda = 1 (plain discounted returns); case 4: gamma = 0.99, gae_lambda = 0.95 with a value array.
"""
import os
import numpy as np

CASES = [
    dict(done=[1, 0, 0, 1, 0, 1, 0, 1], rew=[0, 1, 2, 3, 4, 5, 6, 7], v=None, gamma=0.1, lam=1.0,
         ans=[0, 1.23, 2.3, 3, 4.5, 5, 6.7, 7]),
    dict(done=[0, 1, 0, 1, 0, 1, 0], rew=[7, 6, 1, 2, 3, 4, 5], v=None, gamma=0.1, lam=1.0,
         ans=[7.6, 6, 1.2, 2, 3.4, 4, 5]),
    dict(done=[0, 1, 0, 1, 0, 0, 1], rew=[7, 6, 1, 2, 3, 4, 5], v=None, gamma=0.1, lam=1.0,
         ans=[7.6, 6, 1.2, 2, 3.45, 4.5, 5]),
    dict(done=[0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1], rew=[101, 102, 103, 200, 104, 105, 106, 201, 107, 108, 109, 202],
         v=[2, 3, 4, -1, 5, 6, 7, -2, 8, 9, 10, -3], gamma=0.99, lam=0.95,
         ans=[454.8344, 376.1143, 291.298, 200, 464.5610, 383.1085, 295.387, 201, 474.2876, 390.1027, 299.476, 202]),
]


def main():
    out = {'n_cases': np.int64(len(CASES)),
           'source': np.array('tianshou v0.4.8 test/base/test_returns.py::test_episodic_returns (published known answers)')}
    for i, c in enumerate(CASES):
        n = len(c['rew'])
        v_next = np.zeros(n) if c['v'] is None else np.asarray(c['v'], np.float64)
        out[f'c{i}_done'] = np.asarray(c['done'], np.uint8)
        out[f'c{i}_rew'] = np.asarray(c['rew'], np.float64)
        out[f'c{i}_v_next'] = v_next
        out[f'c{i}_v_s'] = np.roll(v_next, 1)              # the test's default: v_s = np.roll(v_s_, 1)
        out[f'c{i}_gamma'] = np.float64(c['gamma'])
        out[f'c{i}_lambda'] = np.float64(c['lam'])
        out[f'c{i}_returns'] = np.asarray(c['ans'], np.float64)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'tianshou_known_answers.npz')
    np.savez_compressed(path, **out)
    print(path)


if __name__ == '__main__':
    main()
