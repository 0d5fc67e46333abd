import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "agrl.pytorch_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _tuning_switches_follow_the_environment():
    """libagrl_hip.so reads its AGRL_* tuning switches once at load time; tests that flip them through monkeypatch call
    _hip.reload_options() themselves, and this fixture (set up before monkeypatch, so torn down after its undo) puts the
    library back on the restored environment afterwards."""
    yield
    from torchreid import _hip
    if _hip._lib is not None:
        _hip.reload_options()
