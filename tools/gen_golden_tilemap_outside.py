#!/usr/bin/env python3
"""Tile hit maps of viewport centres OUTSIDE the frame, through the imported reference function
(viewport_prediction/utils/common.py:37-58,83-127).  `predict.py:40-45` and `results.py` feed raw predictions into it, and the
linear-regression baseline extrapolates across wrap-around jumps (x = -0.3 or 1.4 occur on real traces), so what the reference
does there is part of the behaviour: Python's floor division on negative pixels and numpy's slice semantics on negative tile
indices (`viewport[ty1:ty2+1]` with ty2 + 1 < 0 counts from the END of the axis).  Writes tests/golden/tilemap_px_outside.npz:
px [n,2] int32, maps [n] uint64, raised [n] bool (the reference raised -- UnboundLocalError when no region case matches)."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refstubs  # noqa: E402
refstubs.install()
sys.path.insert(0, '/root/reference/viewport_prediction')
from utils.common import find_tiles_covered_by_viewport  # noqa: E402  (the reference)


def main():
    W, H = 2560, 1440
    xs = sorted(set(list(range(-4000, 7001, 53)) + [k * 320 + d for k in range(-12, 22) for d in (-301, -300, -299, -1, 0, 1, 299, 300, 301)]))
    ys_in = [0, 1, 150, 151, 700, 1289, 1290, 1440]
    ys = sorted(set(list(range(-2500, 4501, 41)) + [k * 180 + d for k in range(-13, 25) for d in (-151, -150, -149, -1, 0, 1, 149, 150, 151)]))
    xs_in = [0, 1, 300, 301, 1000, 2259, 2260, 2560]
    pts = [(x, y) for x in xs for y in ys_in] + [(x, y) for x in xs_in for y in ys]
    rs = np.random.RandomState(4)
    pts += [(int(x), int(y)) for x, y in zip(rs.randint(-4000, 7000, 3000), rs.randint(-2500, 4500, 3000))]
    pts = [p for p in pts if not (0 <= p[0] <= W and 0 <= p[1] <= H)]
    px = np.array(pts, dtype=np.int32)
    maps = np.zeros(len(px), dtype=np.uint64)
    raised = np.zeros(len(px), dtype=bool)
    for i, (x, y) in enumerate(px):
        try:
            m = find_tiles_covered_by_viewport(int(x), int(y), W, H, 320, 180, 8, 8).reshape(-1)
            maps[i] = sum(int(b) << k for k, b in enumerate(m))
        except UnboundLocalError:
            raised[i] = True
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'tilemap_px_outside.npz'), px=px, maps=maps, raised=raised)
    print('ok', len(px), 'raised', int(raised.sum()), 'distinct maps', len(np.unique(maps)))


if __name__ == '__main__':
    main()
