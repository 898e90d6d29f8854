import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """Tests marked `gpu` need a HIP device and the built library: skip (not fail) them on a host without either,
    so that a plain `pytest tests/` on the CPU container stays green."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:                                   # noqa: BLE001
        have_gpu = False
    lib = os.path.join(ROOT, 'safe_control_gym_amd', 'libscg_hip.so')
    if have_gpu and os.path.exists(lib):
        return
    why = 'no HIP device' if not have_gpu else 'libscg_hip.so not built'
    skip = pytest.mark.skip(reason=f'needs a real MI355X ({why})')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
