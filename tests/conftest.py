import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "visual-odometry-rs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment (gpu-marked tests run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure both shared libraries exist (prebuilt ones travel to the GPU box; build here if missing)."""
    import __graft_entry__ as g
    lib = os.path.join(ROOT, "visual-odometry-rs_amd", "vors_amd", "libvors_hip.so")
    ora = os.path.join(ROOT, "oracle", "libvors_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        g.build()
