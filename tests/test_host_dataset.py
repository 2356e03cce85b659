"""Host mirror of the sliding-window dataset (viewport_prediction/utils/load_dataset.py) against the imported reference on real
Jin2022 traces (tests/golden/dataset_reference.npz, tools/gen_golden_dataset.py): the fixture's eight raw traces are written out in
the dataset's on-disk layout, `create_dataset` runs on them, and every (video, user, timestep) index of every split -- two window /
step / trim settings, the test_seen / test_unseen user rule -- plus a strided sample of items must equal the reference's."""
import ast
import os

import numpy as np
import pytest

Z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'dataset_reference.npz'))
INCLUDE = ['train', 'valid', 'test', 'test_seen', 'test_unseen']


@pytest.fixture(scope='module')
def tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('jin'))
    for key in Z.files:
        if key.startswith('trace/'):
            _, v, u = key.split('/')
            d = os.path.join(root, f'video{v}', '5Hz')
            os.makedirs(d, exist_ok=True)
            np.save(os.path.join(d, f'simple_5Hz_user{u}.npy'), Z[key])
    return root


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_dataset_windows_equal_reference(tree, tag):
    from mansy_immersivevideostreaming_amd.viewport_prediction.utils.common import Config
    from mansy_immersivevideostreaming_amd.viewport_prediction.utils.load_dataset import create_dataset
    vsplit, usplit = ast.literal_eval(str(Z['vsplit'])), ast.literal_eval(str(Z['usplit']))
    S, T, step, th, tt = (int(x) for x in Z[f'{tag}/params'])
    config = Config(dict(viewport_datasets_dir={'Jin2022': tree}, video_split={'Jin2022': vsplit}, user_split={'Jin2022': usplit},
                         trim_head=99, trim_tail=99, frequency=5, sample_step=99))
    sets = create_dataset('Jin2022', config, his_window=S, fut_window=T, frequency=5, sample_step=step, trim_head=th, trim_tail=tt,
                          dataset_video_split=dict(vsplit), dataset_user_split=dict(usplit))
    assert len(sets) == len(INCLUDE)
    for name, ds in zip(INCLUDE, sets):
        ref = Z[f'{tag}/{name}/indices']
        np.testing.assert_array_equal(np.array(ds.trace_indices, np.int64).reshape(-1, 3), ref, err_msg=name)
        for i in Z[f'{tag}/{name}/pick']:
            h, c, f, v, u, t = ds[int(i)]
            assert (v, u, t) == tuple(int(x) for x in ref[int(i)])
            for got, key in ((h, 'history'), (c, 'current'), (f, 'future')):
                want = Z[f'{tag}/{name}/item{int(i)}/{key}']
                assert got.shape == want.shape == ({'history': S, 'current': 1, 'future': T}[key], 2)
                np.testing.assert_array_equal(np.asarray(got), want)


def test_full_jin2022_split_sizes_recorded():
    """Sizes of the full Jin2022 splits as the reference builds them (S=T=10, step 5, trim 15/15) -- SURVEY 8a V1."""
    assert Z['full_sizes'].tolist() == [43152, 7242, 2430, 2430, 2430]
