"""Host mirror of viewport_prediction/utils/common.py with the arithmetic on the HIP kernels.

get_config_from_yml (common.py:10-34), find_tiles_covered_by_viewport (:46-58), to_position_normalized_cartesian
(:61-70), mean_square_error (:73-80).  Same names, argument meaning and results; tensors must be on a ROCm device.
"""
import ctypes

import numpy as np
import torch
import yaml

from ..._lib import MansyError, check, lib, ptr, stream_ptr
from ... import kernels

DEFAULT_CONFIG_YML_PATH = '../config.yml'


class Config(dict):
    """Attribute dict (the reference uses munch.Munch)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def get_config_from_yml(config_yml_path=None):
    if config_yml_path is None:
        config_yml_path = DEFAULT_CONFIG_YML_PATH
    with open(config_yml_path, 'r', encoding='utf8') as f:
        config = Config(yaml.load(f, Loader=yaml.SafeLoader))
    for datasets_dir in (config.raw_datasets_dir, config.raw_network_datasets_dir, config.viewport_datasets_dir, config.video_datasets_dir,
                         config.network_datasets_dir):
        for key in datasets_dir.keys():
            datasets_dir[key] = config.datasets_base_dir + datasets_dir[key]
    config.vp_results_dir = config.results_base_dir + config.vp_results_dir
    config.bs_results_dir = config.results_base_dir + config.bs_results_dir
    config.vp_models_dir = config.models_base_dir + config.vp_models_dir
    config.bs_models_dir = config.models_base_dir + config.bs_models_dir
    return config


def _gpu(t):
    if not torch.is_tensor(t) or not t.is_cuda:
        raise MansyError('this mirror runs on the HIP kernels: pass cuda (ROCm) tensors')


def mean_square_error(position_a, position_b, dimension=2):
    """Periodic MSE per position: [..., dimension] -> [...]."""
    _gpu(position_a)
    a, b = position_a.contiguous().float(), position_b.contiguous().float()
    rows = a.numel() // a.shape[-1]
    out = torch.empty(a.shape[:-1], dtype=torch.float32, device=a.device)
    check(lib().mansy_periodic_mse(ptr(a), ptr(b), rows, a.shape[-1], ptr(out), stream_ptr(a.device)), 'mansy_periodic_mse')
    if dimension != a.shape[-1]:
        out = out * (a.shape[-1] / dimension)
    return out


def to_position_normalized_cartesian(values):
    """v < 0 -> v - trunc(v) + 1 ; v > 1 -> v - trunc(v)."""
    _gpu(values)
    v = values.contiguous().float()
    out = torch.empty_like(v)
    check(lib().mansy_ensemble_wrap(ptr(v), ptr(out), v.numel(), 1, 1, stream_ptr(v.device)), 'mansy_ensemble_wrap')
    return out


def find_block_covered_by_point(x, y, block_width, block_height):
    """common.py:37-43 (host arithmetic, Python floor division): the tile a pixel falls into; an exact positive multiple belongs to
    the lower tile."""
    w, h = x // block_width, y // block_height
    if x > 0 and x % block_width == 0:
        w -= 1
    if y > 0 and y % block_height == 0:
        h -= 1
    return w, h


def find_tiles_covered_by_viewport(x, y, video_width, video_height, tile_width, tile_height, tile_num_width, tile_num_height,
                                   fov_width=600, fov_height=300, device='cuda'):
    """Single pixel centre (ints) -> uint8 [tile_num_height, tile_num_width] like the reference (device round trip)."""
    # a normalised coordinate that the kernel's int() (truncation toward zero) turns back into exactly this pixel, negative ones included
    half = lambda v: 0.5 if v >= 0 else -0.5
    xy = torch.tensor([[(x + half(x)) / video_width, (y + half(y)) / video_height]], dtype=torch.float32, device=device)
    m = int(kernels.tilemap(xy, video_width, video_height, tile_num_width, tile_num_height, fov_width, fov_height).item())
    bits = [(m >> k) & 1 for k in range(tile_num_width * tile_num_height)]
    return np.array(bits, dtype=np.uint8).reshape(tile_num_height, tile_num_width)


def compute_accuracy(gt, pred, video_width, video_height, tile_num_width, tile_num_height, tile_width=None, tile_height=None,
                     fov_width=600, fov_height=300):
    """utils/results.py:34-50 batched on the device: gt/pred [B,T,2] -> four float64 arrays [B,T]
    (accuracy = IoU, recall, precision, f1)."""
    _gpu(pred)
    g = kernels.tilemap(gt.float(), video_width, video_height, tile_num_width, tile_num_height, fov_width, fov_height)
    p = kernels.tilemap(pred.float(), video_width, video_height, tile_num_width, tile_num_height, fov_width, fov_height)
    out = torch.empty(g.shape + (4,), dtype=torch.float64, device=g.device)
    check(lib().mansy_tilemap_metrics(ptr(g), ptr(p), g.numel(), ptr(out), stream_ptr(g.device)), 'mansy_tilemap_metrics')
    o = out.cpu().numpy()
    return o[..., 0], o[..., 1], o[..., 2], o[..., 3]
