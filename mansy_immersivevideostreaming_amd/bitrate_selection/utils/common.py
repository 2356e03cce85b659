"""Host mirror of bitrate_selection/utils/common.py, name for name: config loading (:13-37), the normalisers (:40-57), the episode
catalogue (:60-98), action2rates / rates2action (:101-139), allocate_tile_rates (:142-193, on the device) and read_log_file
(:196-218).  Inside the environment kernels the same arithmetic runs fused (csrc/env.hip); these are the stand-alone entry points
a caller of the reference module expects to find here."""
import ctypes

import numpy as np
import torch

from ..._lib import check, lib, ptr, stream_ptr
from ...viewport_prediction.utils.common import Config, get_config_from_yml  # noqa: F401  (identical loader, common.py:13-37)
from ..envs.expert_env import action2rates, rates2action  # noqa: F401
from ..envs.mansy_env import generate_environment_samples, generate_environment_test_samples  # noqa: F401


def normalize_quality(config, quality):
    """utils/common.py:40-42: bitrate is the quality measure, so the largest bitrate normalises it."""
    return quality / config.video_rates[-1]


def normalize_size(config, size):
    """utils/common.py:45-47."""
    return size / config.max_size


def normalize_throughput(config, throughput):
    """utils/common.py:50-52."""
    return throughput / config.max_throughput


def normalize_qoe_weight(qoe_weight):
    """utils/common.py:55-57."""
    return qoe_weight / sum(qoe_weight)


def allocate_tile_rates(rate_version_in, rate_version_out, pred_viewport, video_rates, tile_num_width, tile_num_height, device='cuda'):
    """utils/common.py:142-193 for ONE predicted viewport (64 tiles of 0 / 1) -> (tile_rate_versions, tile_rates), both int32 [64]:
    tiles of the viewport get rate_version_in, a tile at ring distance s (8-neighbour BFS on the torus) the version closest to
    video_rates[rate_version_out] // s (ties to the lower bitrate).  The device kernel (mansy_allocate_tile_rates) is keyed by the 15
    actions, so an arbitrary (in, out) pair runs as two rows -- one action with this `in`, one with this `out` -- merged per tile."""
    if tile_num_width * tile_num_height != 64 or len(video_rates) != 5:
        raise ValueError('the tile-rate kernel is built for 8 x 8 tiles and 5 bitrates (config.yml)')
    pv = np.ascontiguousarray(np.asarray(pred_viewport).reshape(-1), dtype=np.float32)
    a_in = rates2action(rate_version_in, 0) if rate_version_in > 0 else 10           # (in, 0) / (0, 0)
    a_out = rates2action(4, rate_version_out)                                        # (4, out)
    P = torch.from_numpy(np.stack([pv, pv])).to(device)
    A = torch.tensor([a_in, a_out], dtype=torch.int32, device=device)
    out = torch.empty(2, 64, dtype=torch.int32, device=device)
    rates = (ctypes.c_int * 5)(*[int(r) for r in video_rates])
    check(lib().mansy_allocate_tile_rates(ptr(P), ptr(A), 2, rates, ptr(out), stream_ptr(P.device)), 'mansy_allocate_tile_rates')
    o = out.cpu().numpy()
    inside = (pv == 1) if (pv == 1).any() else np.ones(64, bool)      # empty prediction: the BFS never starts, every scale stays 0
    versions = np.where(inside, o[0], o[1]).astype(np.int32)
    return versions, np.asarray(video_rates, dtype=np.int32)[versions]


def read_log_file(log_path, verbose=True):
    rows, means = [], [0.0, 0.0, 0.0, 0.0]
    with open(log_path, 'r') as file:
        file.readline()
        for line in file.readlines():
            line = line.strip().split(',')
            video, user, trace = list(map(int, line[:3]))
            vals = list(map(float, line[3:]))
            for i in range(4):
                means[i] += vals[3 + i]
            rows.append([video, user, trace] + vals)
    n = max(len(rows), 1)
    means = [m / n for m in means]
    if verbose:
        print('video user trace qoe_w1 qoe_w2 qoe_w3 qoe qoe1 qoe2 qoe3')
        for r in rows:
            print(' '.join(str(x) for x in r))
        print(-1, -1, -1, -1, -1, -1, *means)
    return means
