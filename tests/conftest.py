import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / at round end)")
    config.addinivalue_line("markers", "reference: needs /root/reference (authoring container only)")


def _have_gpu() -> bool:
    try:
        import torch

        if not torch.cuda.is_available():
            return False
        from lhotse_amd import _lib

        _lib.load()
        return True
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/lhotse")
    skip_ref = pytest.mark.skip(reason="/root/reference not present")
    have_gpu = None
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)
        if "gpu" in item.keywords:
            if have_gpu is None:
                have_gpu = _have_gpu()
            if not have_gpu:  # the product has no CPU fallback: without a HIP device the GPU tests cannot run (they do not fail)
                item.add_marker(pytest.mark.skip(reason="no HIP device / libhipfeat.so: GPU tests run on the MI355X box (python -m pytest tests -m gpu)"))
