#!/bin/bash
# round 5, GPU call 7: 12 waves per workgroup (3 / SIMD, no span prefetch) vs 8 waves with the prefetch, for both fft2048c defaults, on the
# experiment-2 build (no spills, 131 / 178 VGPRs)
set -u
OUT=gpurun_out/${1:-r5_run7}
mkdir -p "$OUT"
HIPFEAT_FFT2048_W12=1 timeout 300 python -m pytest tests/test_gpu_fixed_schedule.py tests/test_gpu_fft2048.py -q -x > "$OUT/pytest_w12.txt" 2>&1; tail -2 "$OUT/pytest_w12.txt"
for rep in 1 2 3; do
  for v in 0 1; do
    HIPFEAT_FFT2048_W12=$v python tools/bench_rates.py --rates 44100,48000 --cuts 4000 --steps 20 2>/dev/null | python -c "
import sys,json
print('w12=$v rep$rep', ' '.join('%d:%.3fM(%.3f) %s' % (r['sampling_rate'], r['cuts_per_s']/1e6, r['frac_of_8TBps'], r['kernel']) for r in map(json.loads, sys.stdin)))" | tee -a "$OUT/ab_w12.txt"
  done
done
