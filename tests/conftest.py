import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a HIP device skips the `gpu` tests instead of failing in them.  (With
    `-m gpu` they are NOT skipped: on the GPU box a missing device or library must fail loudly.)"""
    if "gpu" in (config.getoption("-m") or ""):
        return
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    have = have and os.path.exists(os.path.join(ROOT, "transmission_renderer_amd", "libtr_shade.so"))
    if have:
        return
    skip = pytest.mark.skip(reason="no HIP device / libtr_shade.so: run with -m gpu on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def ggx_lut():
    from transmission_renderer_amd.png import read_png_rgba8
    return read_png_rgba8(os.path.join(ROOT, "transmission_renderer_amd", "assets", "ggx_lut.png"))
