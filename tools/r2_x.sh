#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python tools/bench_defaults.py 2>/dev/null | grep -v amdgpu | grep spectro | cut -c1-200
HIPFEAT_FFT512_VARIANT=b python tools/bench_defaults.py 2>/dev/null | grep -v amdgpu | grep spectro | cut -c1-200
