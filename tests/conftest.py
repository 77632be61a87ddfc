import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import gs_oracle
    gs_oracle.build()
    return gs_oracle


_SCENES = {}


def scene(n, seed=None):
    """Cached synthetic scene (float32 [n,80] in the reference's 320-byte record layout)."""
    from gsplat import synth
    key = (n, seed)
    if key not in _SCENES:
        _SCENES[key] = synth.bicycle_like(n) if seed is None else synth.bicycle_like(n, seed)
    return _SCENES[key]
