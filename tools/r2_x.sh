#!/bin/bash
mkdir -p gpurun_out/x
timeout 600 python -m pytest tests/test_gpu_fft2048.py tests/test_gpu_librosa.py -x -q > gpurun_out/x/pytest.txt 2>&1; tail -12 gpurun_out/x/pytest.txt
timeout 300 python tools/bench_rates.py --cuts 2000 --rates 44100,48000 2>&1 | grep -v amdgpu | tail -4
HIPFEAT_NO_WAVE_AUTONOMOUS=1 timeout 300 python tools/bench_rates.py --cuts 2000 --rates 48000 2>&1 | grep -v amdgpu | tail -2
