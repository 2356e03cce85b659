"""The round-2..4 A/B knobs for the tools in this directory, on top of the -DMANSY_LAB build (ABI 8: the release library has no process-wide
switch).  `enter()` builds libmansy_hip_lab.so if needed and makes every host mirror of this process talk to it; the functions below keep the old
knob vocabulary and translate it into the lab build's one default-variant word (mansy_lab_set_variant; bit layout = MANSY_VARIANT_* of
include/mansy_hip.h + 0x800 no paired launch, 0x1000 generic head_out instance)."""
import os
import sys

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_state = {'word': 0, 'ctx': None, 'lab': None}
_lab = None


class lab_library:
    """Test / tools harness only (it lives here, not in the package: the product's bindings hold no swappable library handle): inside the block every
    host mirror of THIS process talks to the -DMANSY_LAB build of the same sources (libmansy_hip_lab.so: `python -m
    mansy_immersivevideostreaming_amd.build_ext --lab`), the only build that exports mansy_lab_set_variant(v) -- a default kernel-selection variant
    for calls that pass 0, so that whole engine steps can be run on two loops.  `variant` is set on entry and reset to 0 on exit."""

    def __init__(self, variant=0):
        self.variant = int(variant)

    def __enter__(self):
        import ctypes
        from mansy_immersivevideostreaming_amd import _lib, build_ext
        global _lab
        if _lab is None:
            _lab = _lib._load(build_ext.LAB_LIB)
            _lab.mansy_lab_set_variant.argtypes = [ctypes.c_int]
            _lab.mansy_lab_set_variant.restype = ctypes.c_int
        self.prev_lib = _lib.lib()
        _lab.mansy_lab_set_variant(self.variant)
        _lib._lib = _lab
        return _lab

    def __exit__(self, *exc):
        from mansy_immersivevideostreaming_amd import _lib
        _lab.mansy_lab_set_variant(0)
        _lib._lib = self.prev_lib
        return False


def enter():
    from mansy_immersivevideostreaming_amd import build_ext
    if _state['ctx'] is None:
        build_ext.build(lab=True)
        _state['ctx'] = lab_library(0)
        _state['lab'] = _state['ctx'].__enter__()
    return _state['lab']


def _apply():
    enter().mansy_lab_set_variant(_state['word'])


def _bit(mask, on):
    _state['word'] = (_state['word'] | mask) if on else (_state['word'] & ~mask)
    _apply()


def f32_wsk(v):
    """The old mansy_gemm_f32_wsk(v) codes that still exist: 0 / 1 small products on the 64 x 64 loop / the wave-split-K loop; 2 / 3 the same for the
    weight-gradient products; 8 / 9 paired launch off / on; 14 / 15 plain instances off / on."""
    if v in (0, 1):
        _bit(0x100, v == 0)
    elif v in (2, 3):
        _bit(0x200, v == 2)
    elif v in (8, 9):
        _bit(0x800, v == 8)
    elif v in (14, 15):
        _bit(0x400, v == 14)
    elif v >= 0:
        raise SystemExit(f'mansy_gemm_f32_wsk({v}): that experiment is gone from the sources (HISTORY.md has its result)')
    return 1


def bf16_variant(v):
    if v >= 0:
        _state['word'] = (_state['word'] & ~0xFF) | ((v + 1) & 0xFF)
        _apply()
    return 1


def col_group(v):
    if v >= 0:
        _state['word'] = (_state['word'] & ~(0xFF << 16)) | (((v + 1) & 0xFF) << 16)
        _apply()
    return 12


def head_out_generic(on):
    _bit(0x1000, bool(on))
