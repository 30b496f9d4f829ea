#!/usr/bin/env python3
"""bench.py's PCIe-inclusive leg on its own (GPU box): python tools/host_fed_quick.py [seconds]"""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import lhotse_amd as LA

ex = LA.HipFbank(LA.HipFbankConfig(device="cuda:0"))
print(json.dumps(bench.host_fed(ex, float(sys.argv[1]) if len(sys.argv) > 1 else 2.0)))
