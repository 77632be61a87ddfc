import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-wgpu_amd"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """On a GPU box torch's HIP runtime has to come up BEFORE libgsplat_hip.so's first HIP call in this process (the other
    order leaves torch with "no ROCm-capable device"); the full-size tests generate their scenes with torch on the device."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass


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
