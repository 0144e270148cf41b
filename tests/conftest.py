import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def variants():
    """Pin kernel-form choices of the library (msgl_set_variant) for one test: variants("extend_wide", 0).  Every choice
    is released again when the test ends."""
    from mini_sglang_amd import ops

    def pin(name, value):
        ops.set_variant(name, None if value is None else int(value))

    yield pin
    ops.reset_variants()
