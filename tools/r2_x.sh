#!/bin/bash
HIPFEAT_LIB=$PWD/lhotse_amd/_lib/var_pt.so python tools/phase_timers_w.py 1000 48000 2>&1 | grep -v amdgpu
HIPFEAT_LIB=$PWD/lhotse_amd/_lib/var_pt.so python tools/phase_timers_w.py 1000 24000 2>&1 | grep -v amdgpu
timeout 300 python tools/bench_rates.py --cuts 2000 --rates 22050,24000,32000,44100,48000 2>&1 | grep -v amdgpu | tail -5 | cut -c1-200
python tools/bench_librosa.py 2>&1 | tail -1 | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_fft2048.py tests/test_gpu_librosa.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
