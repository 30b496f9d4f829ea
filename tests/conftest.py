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


def pytest_sessionfinish(session, exitstatus):
    """Parity artefact of a GPU session: per comparison the achieved error, the reference's own float32 floor and whether
    the 3 x floor clause was needed (tests/_golden.py::record_parity)."""
    try:
        from _golden import PARITY_LOG
    except Exception:
        return
    if not PARITY_LOG:
        return
    import json

    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    by_suite = {}
    for r in PARITY_LOG:
        d = by_suite.setdefault(r["suite"], {"n": 0, "rel_l2_max": 0.0, "max_abs_max": 0.0, "frac_within_min": 1.0, "clause_needed_rel": 0, "clause_needed_abs": 0})
        d["n"] += 1
        d["rel_l2_max"] = max(d["rel_l2_max"], r["rel_l2"])
        d["max_abs_max"] = max(d["max_abs_max"], r["max_abs"])
        d["frac_within_min"] = min(d["frac_within_min"], r["frac_within_rtol1e-4_atol1e-3"])
        d["clause_needed_rel"] += int(r["clause_needed_rel"])
        d["clause_needed_abs"] += int(r["clause_needed_abs"])
    with open(os.path.join(out_dir, "parity_report.json"), "w") as f:
        json.dump({"exitstatus": int(exitstatus), "summary": by_suite, "entries": PARITY_LOG}, f, indent=1)
