#!/usr/bin/env python3
"""Golden vectors for the tile hit map, from (a) the reference's own shipped
data (datasets/Jin2022/viewports/prediction/*.pkl `gt` maps + the 5 Hz traces they were derived
from) and (b) the imported reference function on a dense grid of pixel centres incl. every
wrap-around case.  Data only."""
import os
import sys
import pickle
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refstubs  # noqa: E402
refstubs.install()
sys.path.insert(0, '/root/reference/viewport_prediction')
from utils.common import find_tiles_covered_by_viewport  # noqa: E402  (the reference)

DS = '/root/reference/datasets/Jin2022/viewports'
OUT = os.path.join(ROOT, 'tests', 'golden')
PAIRS = [(1, 1), (1, 22), (2, 27), (5, 3), (9, 59), (12, 10), (14, 60), (16, 24), (18, 44), (21, 13), (23, 7), (24, 50)]


def main():
    rec = {}
    for v, u in PAIRS:
        tr = np.load(f'{DS}/video{v}/5Hz/simple_5Hz_user{u}.npy')[:, 1:]
        pk = pickle.load(open(f'{DS}/prediction/video{v}/user{u}.pkl', 'rb'))
        rec[f'trace_{v}_{u}'] = tr.astype(np.float32)
        rec[f'chunk_{v}_{u}'] = np.array([p[0] for p in pk], dtype=np.int32)
        rec[f'gt_{v}_{u}'] = np.stack([p[1] for p in pk]).astype(np.uint8)
        rec[f'pred_{v}_{u}'] = np.stack([p[2] for p in pk]).astype(np.uint8)
        rec[f'iou_{v}_{u}'] = np.array([p[3] for p in pk], dtype=np.float64)
    rec['pairs'] = np.array(PAIRS, dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, 'tilemap_dataset.npz'), **rec)
    # dense pixel grid through the imported reference function
    xs = sorted(set(list(range(0, 2561, 37)) + [0, 1, 299, 300, 301, 319, 320, 321, 639, 640, 2259, 2260, 2261, 2559, 2560]))
    ys = sorted(set(list(range(0, 1441, 23)) + [0, 1, 149, 150, 151, 179, 180, 181, 1289, 1290, 1291, 1439, 1440]))
    px = np.array([(x, y) for x in xs for y in ys], dtype=np.int32)
    maps = np.zeros(len(px), dtype=np.uint64)
    for i, (x, y) in enumerate(px):
        m = find_tiles_covered_by_viewport(int(x), int(y), 2560, 1440, 320, 180, 8, 8).reshape(-1)
        maps[i] = sum(int(b) << k for k, b in enumerate(m))
    np.savez_compressed(os.path.join(OUT, 'tilemap_px.npz'), px=px, maps=maps)
    print('ok', len(px))


if __name__ == '__main__':
    main()
