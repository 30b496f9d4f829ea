#!/bin/bash
mkdir -p gpurun_out/w3
timeout 600 python -m pytest tests/test_gpu_whisper.py -x -q > gpurun_out/w3/pytest.txt 2>&1; tail -3 gpurun_out/w3/pytest.txt
for q in 0 0.3 1.0; do
python tools/bench_whisper.py --cuts 4000 --steps 20 --quiet-tail $q 2>/dev/null | tail -1 | cut -c80-120,200-330
HIPFEAT_WHISPER_VARIANT=2 python tools/bench_whisper.py --cuts 4000 --steps 20 --quiet-tail $q 2>/dev/null | tail -1 | cut -c80-120,180-330
done
python tools/bench_whisper.py --cuts 60 --steps 50 2>/dev/null | tail -1 | cut -c200-330
