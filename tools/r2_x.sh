#!/bin/bash
for i in 1 2; do
python bench.py --no-cpu-baseline --no-host-fed --steps 100 2>/dev/null | tail -1 | cut -c60-110
HIPFEAT_X_NO_NFULL=1 python bench.py --no-cpu-baseline --no-host-fed --steps 100 2>/dev/null | tail -1 | cut -c60-110
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_configs.py -x -q 2>&1 | tail -2
