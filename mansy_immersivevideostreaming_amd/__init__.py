"""MI355X-native (gfx950) implementation of the MANSY neural hot path.

Layout mirrors the reference repo for the path it replaces:
  viewport_prediction/models/mtio.py   -> ViewportTransformerMTIO (drop-in class, HIP engine underneath)
  viewport_prediction/utils/common.py  -> tile hit map / periodic MSE / wrap (HIP kernels)
  csrc/                                -> hand-written HIP kernels + the C ABI (include/mansy_hip.h)
"""
from . import _lib  # noqa: F401

__all__ = ['_lib']
