import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
MODEL_PB = os.path.join(ROOT, "models", "age_gender_tf2_new-01-0.14-0.92_quantized.pb")
TEST_IMAGE = os.path.join(GOLDEN, "test_image.jpg")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests never run by accident on a CPU-only machine: they are skipped unless a device is visible.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def model_pb():
    return MODEL_PB


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
