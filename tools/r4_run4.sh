#!/bin/bash
# on-the-fly: the two launches of a mini-batch timed apart (HIPFEAT_MB_SKIP), grid width of the prep launch, K mini-batches per pair
set -u
OUT=gpurun_out/${1:-r4_run4}
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_minibatch.py -x -q 2>&1 | tail -2 | tee "$OUT/pytest_minibatch.txt"
F="--config onthefly --no-cpu-baseline --no-extra --no-parity --steps 40"
run() { tag=$1; shift; env "$@" 2>/dev/null | python -c "
import sys,json
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-22s %9.0f cuts/s  %.3f ms/step  %.1f us/minibatch  frac %.4f' % ('$tag', r['value'], r['roofline']['launch_ms'], r['roofline']['launch_ms']*1e3/64, r['roofline']['frac']))" | tee -a "$OUT/ab.txt"; }
for K in 1 4; do
run k${K}_1s python bench.py $F --prefetch $K --streams 1
run k${K}_1s_prep_only HIPFEAT_MB_SKIP=1 python bench.py $F --prefetch $K --streams 1
run k${K}_1s_fbank_only HIPFEAT_MB_SKIP=2 python bench.py $F --prefetch $K --streams 1
run k${K}_2s python bench.py $F --prefetch $K --streams 2
run k${K}_2s_gx32 HIPFEAT_MB_GRIDX=32 python bench.py $F --prefetch $K --streams 2
run k${K}_2s_gx128 HIPFEAT_MB_GRIDX=128 python bench.py $F --prefetch $K --streams 2
run k${K}_1s_prep_only_gx128 HIPFEAT_MB_SKIP=1 HIPFEAT_MB_GRIDX=128 python bench.py $F --prefetch $K --streams 1
done
run k2_2s python bench.py $F --prefetch 2 --streams 2
run k8_2s python bench.py $F --prefetch 8 --streams 2
run k1_3s python bench.py $F --prefetch 1 --streams 3
