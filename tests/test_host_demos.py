"""Demonstration files (run_expert.py:35-41 / run_mansy.py:260-266): load_demonstrations() reads this build's plain-dict files and
files holding tianshou ReplayBuffer / Batch objects WITHOUT tianshou installed.  The second branch is exercised on a file pickled
from stand-in classes registered under tianshou's module paths with the 0.4.8 attribute layout (T2; a reference-written file
cannot be produced here)."""
import pickle
import sys
import types

import numpy as np


def _fake_tianshou():
    mods = {}
    for name in ('tianshou', 'tianshou.data', 'tianshou.data.batch', 'tianshou.data.buffer', 'tianshou.data.buffer.base'):
        mods[name] = types.ModuleType(name)

    class Batch:
        def __init__(self, **kw):
            self.__dict__.update(kw)

        def __getstate__(self):
            return dict(self.__dict__)

        def __setstate__(self, state):
            self.__dict__.update(state)
    Batch.__module__, Batch.__qualname__ = 'tianshou.data.batch', 'Batch'

    class ReplayBuffer:
        def __setstate__(self, state):
            self.__dict__.update(state)
    ReplayBuffer.__module__, ReplayBuffer.__qualname__ = 'tianshou.data.buffer.base', 'ReplayBuffer'
    mods['tianshou.data.batch'].Batch = Batch
    mods['tianshou.data.buffer.base'].ReplayBuffer = ReplayBuffer
    return mods, Batch, ReplayBuffer


def test_load_demonstrations_both_formats(tmp_path):
    from mansy_immersivevideostreaming_amd.bitrate_selection.envs.mansy_env import OBS_SLICES
    from mansy_immersivevideostreaming_amd.bitrate_selection.utils.mansy_utils import load_demonstrations
    rs = np.random.RandomState(0)
    n, cap = 7, 7
    rows = rs.rand(n, 780).astype(np.float32)
    rows[:, 779] = 0
    act = rs.randint(0, 15, size=n)
    # (a) this build's format
    p1 = tmp_path / 'a.pkl'
    pickle.dump({(1, 2, 3, (7, 1, 1)): {'obs': rows, 'act': act.astype(np.int32), 'done': np.arange(n) == n - 1}}, open(p1, 'wb'))
    d = load_demonstrations(str(p1))
    assert list(d) == [(1, 2, 3, (7, 1, 1))] and np.array_equal(d[(1, 2, 3, (7, 1, 1))]['obs'], rows)
    # (b) the reference's: ReplayBuffer(length) filled by add(Batch(obs=state dict, act, rew=0, done, obs_next, info))
    mods, Batch, ReplayBuffer = _fake_tianshou()
    sys.modules.update(mods)
    try:
        obs = Batch(**{k: rows[:, a:b].reshape((n,) + shape) for k, (a, b, shape) in OBS_SLICES.items()})
        buf = ReplayBuffer()
        buf.__dict__.update(maxsize=cap, _size=n, _index=0, _meta=Batch(obs=obs, act=act.astype(np.int64), rew=np.zeros(n), done=np.arange(n) == n - 1,
                                                                           obs_next=np.arange(1, n + 1), info=Batch()))
        p2 = tmp_path / 'b.pkl'
        pickle.dump({(1, 2, 3, (7, 1, 1)): buf}, open(p2, 'wb'))
    finally:
        for k in mods:
            sys.modules.pop(k, None)
    assert 'tianshou' not in sys.modules                      # read back WITHOUT the package
    d2 = load_demonstrations(str(p2))[(1, 2, 3, (7, 1, 1))]
    np.testing.assert_array_equal(d2['obs'], rows)
    np.testing.assert_array_equal(d2['act'], act)
    assert d2['act'].dtype == np.int32 and d2['done'].tolist() == [False] * (n - 1) + [True]
