"""TEST INFRASTRUCTURE ONLY (oracle). Not imported by the product path.

Counter-based hash RNG shared (bit-for-bit) between the HIP kernels
(`csrc/mansy_common.h: mansy_hash_u32`) and this numpy restatement, so that
train-mode dropout masks can be reproduced on the CPU for parity tests.

This is the build's own RNG: the reference uses torch's Philox/MT streams for
dropout (torch.nn.Dropout inside nn.Transformer, viewport_prediction/models/mtio.py:15,29),
which cannot be matched across devices.  Parity with the reference is therefore
defined with dropout disabled; parity between the HIP path and this oracle is
defined with dropout ENABLED through this shared hash.
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _u32(x):
    return np.asarray(x, dtype=np.uint64) & _M32


def hash_u32(seed, site, idx):
    """uint32 hash of (seed, site, idx); idx may be an array.  All arithmetic mod 2^32."""
    seed = np.uint64(int(seed) & 0xFFFFFFFF)
    site = np.uint64(int(site) & 0xFFFFFFFF)
    idx = _u32(idx)
    h = _u32(seed ^ _u32(site * np.uint64(0x9E3779B9)))
    h = _u32(h ^ _u32(idx + np.uint64(0x7F4A7C15) + _u32(h << np.uint64(6)) + (h >> np.uint64(2))))
    h = h ^ (h >> np.uint64(16))
    h = _u32(h * np.uint64(0x85EBCA6B))
    h = h ^ (h >> np.uint64(13))
    h = _u32(h * np.uint64(0xC2B2AE35))
    h = h ^ (h >> np.uint64(16))
    h = _u32(h + _u32(idx * np.uint64(0x27D4EB2F)))
    h = h ^ (h >> np.uint64(15))
    h = _u32(h * np.uint64(0x2C1B3C6D))
    h = h ^ (h >> np.uint64(12))
    h = _u32(h * np.uint64(0x297A2D39))
    h = h ^ (h >> np.uint64(15))
    return h.astype(np.uint32)


def uniform01(seed, site, idx):
    """float32 uniform in [0,1) with 24 random bits (same as the device code)."""
    h = hash_u32(seed, site, idx)
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def keep_mask(seed, site, n, p):
    """Boolean keep-mask for n elements of dropout site `site` (keep iff u >= p)."""
    if p <= 0.0:
        return np.ones(n, dtype=bool)
    u = uniform01(seed, site, np.arange(n, dtype=np.uint64))
    return u >= np.float32(p)
