#!/bin/bash
# round 5, GPU call 6: fft2048c / fft1024c fixed-schedule instances -- experiment 1 (|X|^2 / ln / Kaldi edges compile-time) and experiment 2
# (default frame geometry compile-time: 44-51 SGPR spills -> 0), same-call A/B against the round-4 build; bit-identity with the generic instances
set -u
OUT=gpurun_out/${1:-r5_run6}
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_fixed_schedule.py tests/test_gpu_fft2048.py tests/test_gpu_librosa.py -q -x > "$OUT/pytest_fixed.txt" 2>&1; tail -3 "$OUT/pytest_fixed.txt"
for rep in 1 2; do
  for v in r4base exp1 product; do
    if [ $v = product ]; then L=""; else L="HIPFEAT_LIB=$PWD/lhotse_amd/_lib/var_$v.so"; fi
    env $L python tools/bench_rates.py --rates 22050,24000,32000,44100,48000 --cuts 4000 --steps 20 2>/dev/null | python -c "
import sys,json
print('$v rep$rep', ' '.join('%d:%.3fM(%.3f)' % (r['sampling_rate'], r['cuts_per_s']/1e6, r['frac_of_8TBps']) for r in map(json.loads, sys.stdin)))" | tee -a "$OUT/ab.txt"
  done
done
