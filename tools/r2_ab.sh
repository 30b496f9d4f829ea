#!/bin/bash
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -2
for i in 1 2; do python bench.py --no-cpu-baseline --no-host-fed --steps 150 2>/dev/null | tail -1 | cut -c60-90; done
python tools/bench_whisper.py --cuts 4000 --steps 20 2>/dev/null | tail -1 | cut -c180-260
python tools/bench_defaults.py 2>/dev/null | grep -v amdgpu | cut -c1-190
