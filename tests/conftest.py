import json
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before the HIP runtime starts: the engine's lanes need their own hardware queues
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """a plain `pytest tests` on a box without an MI355X skips the gpu-marked tests instead of failing in lamd_init"""
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="needs an MI355X (lightning_amd has no CPU fallback)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.lib()
    return _orc
