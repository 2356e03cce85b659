"""TEST INFRASTRUCTURE ONLY (oracle).  ctypes front-end for oracle/tilemap.c plus the
per-chunk OR used by predict.py:36-47."""
import ctypes
import numpy as np
from . import build as _b


def tilemap_px(px, W=2560, H=1440, nw=8, nh=8, fov_w=600, fov_h=300):
    L = _b.lib()
    L.oracle_tilemap_px.restype = ctypes.c_uint64
    px = np.asarray(px, dtype=np.int32).reshape(-1, 2)
    return np.array([L.oracle_tilemap_px(int(x), int(y), W, H, W // nw, H // nh, nw, nh, fov_w, fov_h)
                     for x, y in px], dtype=np.uint64)


def tilemap_xy(xy, W=2560, H=1440, nw=8, nh=8, fov_w=600, fov_h=300):
    L = _b.lib()
    xy = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
    out = np.zeros(len(xy), dtype=np.uint64)
    L.oracle_tilemap_xy(xy.ctypes.data_as(ctypes.c_void_p), len(xy), W, H, nw, nh, fov_w, fov_h,
                        out.ctypes.data_as(ctypes.c_void_p))
    return out


def iou(a, b):
    L = _b.lib()
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros(len(a), dtype=np.float64)
    L.oracle_tilemap_iou(a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), len(a),
                         out.ctypes.data_as(ctypes.c_void_p))
    return out


def bits_to_u8(maps, n=64):
    maps = np.asarray(maps, dtype=np.uint64)
    return ((maps[:, None] >> np.arange(n, dtype=np.uint64)[None]) & np.uint64(1)).astype(np.uint8)


def chunk_maps_from_trace(trace_xy, trim_head=15, trim_tail=15, step=5, fut_window=10, freq=5):
    """predict.py:36-47 for the ground-truth side: sample i at t=trim_head+step*i, OR of the maps of
    future steps 1..freq, chunk id = i + trim_head // freq."""
    n = len(trace_xy)
    ts = list(range(trim_head, n - trim_tail, step))
    chunks, maps = [], []
    for i, t in enumerate(ts):
        fut = trace_xy[t + 1:t + 1 + freq]
        m = tilemap_xy(fut)
        maps.append(np.bitwise_or.reduce(m))
        chunks.append(i + trim_head // freq)
    return np.array(chunks, dtype=np.int32), np.array(maps, dtype=np.uint64)
