#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_configs.py tests/test_gpu_kaldifeat.py tests/test_gpu_layers.py -x -q 2>&1 | tail -6
python tools/bench_mfcc.py 2>/dev/null | tail -1 | cut -c1-250
HIPFEAT_FFT512_VARIANT=b python tools/bench_mfcc.py 2>/dev/null | tail -1 | cut -c1-250
python tools/bench_defaults.py 2>/dev/null | grep -v amdgpu | cut -c1-200
