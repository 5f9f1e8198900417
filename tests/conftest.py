import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "quantized-training_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# producers that hand over FP8 codes only leave the bf16 tensor unwritten (model_fusions._mark_lazy): under test it is filled with NaN, so
# that a consumer that reads it without fake_quantize.materialize_lazy fails loudly instead of reading stale memory
os.environ.setdefault("QT_LAZY_POISON", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU (the CPU build, where sanitizers belong): skip what is marked gpu."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a GPU (marker gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
