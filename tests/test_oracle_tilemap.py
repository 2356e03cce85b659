"""Pins oracle/tilemap.c against (a) the imported reference function on 6300 pixel centres
(tools/gen_golden_tilemap.py) and (b) the reference's shipped prediction/*.pkl gt maps."""
import os
import numpy as np
from oracle import tilemap as tm

G = os.path.join(os.path.dirname(__file__), 'golden')


def test_px_grid_bit_exact():
    z = np.load(os.path.join(G, 'tilemap_px.npz'))
    got = tm.tilemap_px(z['px'])
    np.testing.assert_array_equal(got, z['maps'])


def test_dataset_gt_maps_bit_exact():
    z = np.load(os.path.join(G, 'tilemap_dataset.npz'))
    total = 0
    for v, u in z['pairs']:
        chunks, maps = tm.chunk_maps_from_trace(z[f'trace_{v}_{u}'])
        np.testing.assert_array_equal(chunks, z[f'chunk_{v}_{u}'])
        np.testing.assert_array_equal(tm.bits_to_u8(maps), z[f'gt_{v}_{u}'])
        pred_bits = (z[f'pred_{v}_{u}'].astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(1).astype(np.uint64)
        np.testing.assert_allclose(tm.iou(maps, pred_bits), z[f'iou_{v}_{u}'], rtol=0, atol=1e-12)
        total += len(chunks)
    assert total == 637  # 12 (video,user) files, 58-s videos have fewer chunks


def test_px_outside_the_frame_bit_exact():
    """Centres outside the frame (raw predictions; the linear-regression baseline across a wrap jump): Python floor division and
    numpy's negative-slice-bound semantics, against the imported function (tools/gen_golden_tilemap_outside.py)."""
    z = np.load(os.path.join(G, 'tilemap_px_outside.npz'))
    assert not z['raised'].any() and len(z['px']) > 9000
    np.testing.assert_array_equal(tm.tilemap_px(z['px']), z['maps'])
    # the case that distinguishes numpy's semantics from clipping: a stop of -1 keeps all but the last tile of the axis
    k = np.where((z['px'][:, 0] == -700) & (z['px'][:, 1] == 700))[0]
    if len(k):
        m = int(z['maps'][k[0]])
        assert m & 0x7F and not ((m >> 0) & 0x80)
