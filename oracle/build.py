"""TEST INFRASTRUCTURE ONLY.  Builds the plain-C oracle pieces with gcc into oracle/_build/
and loads them with ctypes.  Used by tests/, __graft_entry__ (build + smoke) and bench.py's
cpu_baseline leg only."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, '_build')
SOURCES = ['tilemap.c', 'env.c']


def build(force=False):
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, 'liboracle.so')
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return out
    subprocess.check_call(['gcc', '-O2', '-std=c99', '-ffp-contract=off', '-shared', '-fPIC', '-o', out] + srcs + ['-lm'])
    return out


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib
