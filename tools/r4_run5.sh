#!/bin/bash
# on-the-fly: item-list prep launch (tables in LDS, bank through the constant address space); parts timed apart; grid size; K; streams
set -u
OUT=gpurun_out/${1:-r4_run5}
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_minibatch.py tests/test_gpu_resample.py tests/test_gpu_speed_arena.py -x -q 2>&1 | tail -2 | tee "$OUT/pytest_minibatch.txt"
F="--config onthefly --no-cpu-baseline --no-extra --no-parity --steps 40"
run() { tag=$1; shift; env "$@" 2>/dev/null | python -c "
import sys,json
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-26s %9.0f cuts/s  %.3f ms/step  %.1f us/minibatch  frac %.4f' % ('$tag', r['value'], r['roofline']['launch_ms'], r['roofline']['launch_ms']*1e3/64, r['roofline']['frac']))" | tee -a "$OUT/ab.txt"; }
for K in 1 4; do
run k${K}_1s python bench.py $F --prefetch $K --streams 1
run k${K}_1s_prep_only HIPFEAT_MB_SKIP=1 python bench.py $F --prefetch $K --streams 1
run k${K}_2s python bench.py $F --prefetch $K --streams 2
run k${K}_3s python bench.py $F --prefetch $K --streams 3
run k${K}_2s_slots1024 HIPFEAT_MB_SLOTS=1024 python bench.py $F --prefetch $K --streams 2
run k${K}_2s_slots3584 HIPFEAT_MB_SLOTS=3584 python bench.py $F --prefetch $K --streams 2
run k${K}_1s_prep_only_slots3584 HIPFEAT_MB_SKIP=1 HIPFEAT_MB_SLOTS=3584 python bench.py $F --prefetch $K --streams 1
done
run k1_2s_staged HIPFEAT_MB_NO_INLINE=1 python bench.py $F --prefetch 1 --streams 2
run k1_per_factor python bench.py $F --route per_factor --streams 1
run k2_3s python bench.py $F --prefetch 2 --streams 3
run k1_4s python bench.py $F --prefetch 1 --streams 4
