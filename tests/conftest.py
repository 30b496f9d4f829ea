import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / at round end)")
    config.addinivalue_line("markers", "reference: needs /root/reference (authoring container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/lhotse")
    skip_ref = pytest.mark.skip(reason="/root/reference not present")
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)
