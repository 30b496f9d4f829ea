#!/bin/bash
mkdir -p gpurun_out/w3
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/w3/pytest_all.txt 2>&1; tail -3 gpurun_out/w3/pytest_all.txt
python bench.py --no-cpu-baseline --no-host-fed --steps 100 2>/dev/null | tail -1 | cut -c1-120
HIPFEAT_FFT512_VARIANT=b python bench.py --no-cpu-baseline --no-host-fed --steps 100 2>/dev/null | tail -1 | cut -c1-120
for q in 0 0.3; do
python tools/bench_whisper.py --cuts 4000 --steps 20 --quiet-tail $q 2>/dev/null | tail -1 | cut -c80-120,200-330
done
HIPFEAT_WHISPER_VARIANT=2 python tools/bench_whisper.py --cuts 4000 --steps 20 2>/dev/null | tail -1 | cut -c80-120,180-330
python tools/bench_mfcc.py 2>/dev/null | tail -1 | cut -c1-250
