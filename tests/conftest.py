import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """A fresh clone has no libmansy_hip.so (*.so is git-ignored): compile it once before any test imports the package."""
    from mansy_immersivevideostreaming_amd import build_ext
    build_ext.ensure_built()
