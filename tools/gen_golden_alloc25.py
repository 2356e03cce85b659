#!/usr/bin/env python3
"""allocate_tile_rates (bitrate_selection/utils/common.py:142-193) through the IMPORTED reference for all 25 (rate_version_in,
rate_version_out) pairs -- the 15 actions only reach in >= out -- on four viewports of tests/golden/env_reference.npz, an empty and a
full prediction.  Writes tests/golden/alloc25_reference.npz: pred_viewport [6,64], versions / rates [6,5,5,64] int32.  Data only."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refstubs  # noqa: E402
refstubs.install()
sys.path.insert(0, '/root/reference/bitrate_selection')
os.chdir('/root/reference/bitrate_selection')
from utils.common import allocate_tile_rates  # noqa: E402


def main():
    Z = np.load(os.path.join(ROOT, 'tests', 'golden', 'env_reference.npz'))
    pvs = np.stack([Z['alloc/pred_viewport'][k] for k in (0, 7, 19, 33)] + [np.zeros(64, np.float32), np.ones(64, np.float32)])
    rates = [1, 5, 8, 16, 35]
    ver = np.zeros((len(pvs), 5, 5, 64), np.int32)
    br = np.zeros_like(ver)
    for k, pv in enumerate(pvs):
        for i in range(5):
            for o in range(5):
                v, b = allocate_tile_rates(i, o, pv.copy(), rates, 8, 8)
                ver[k, i, o], br[k, i, o] = v, b
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'alloc25_reference.npz'), pred_viewport=pvs, versions=ver, rates=br)
    print('ok', ver.shape, int(ver.max()))


if __name__ == '__main__':
    main()
