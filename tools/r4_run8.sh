#!/bin/bash
# FLAT layout (ragged batches by frame quads): full suite, then same-call A/B on the ragged configs; driver-form headline with the settle phase
set -u
OUT=gpurun_out/${1:-r4_run8}
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.txt" 2>&1
tail -3 "$OUT/pytest_gpu.txt"
run() { tag=$1; shift; env "$@" 2>/dev/null | python -c "
import sys,json
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-28s %9.0f cuts/s  %.4f ms/step  frac %.4f  parity %s' % ('$tag', r['value'], r['roofline']['launch_ms'], r['roofline']['frac'], (r.get('parity') or {}).get('pass')))" | tee -a "$OUT/ab.txt"; }
F="--no-cpu-baseline --no-extra"
for i in 1 2; do
run mfcc40_flat python bench.py --config mfcc40_libri $F
run mfcc40_noflat HIPFEAT_NO_FLAT=1 python bench.py --config mfcc40_libri $F
done
run otf_k1_flat python bench.py --config onthefly $F
run otf_k1_noflat HIPFEAT_NO_FLAT=1 python bench.py --config onthefly $F
run otf_k4_flat python bench.py --config onthefly --prefetch 4 --streams 2 $F
run otf_k4_noflat HIPFEAT_NO_FLAT=1 python bench.py --config onthefly --prefetch 4 --streams 2 $F
run fbank_driver_form python bench.py --steps 20 --warmup 5 $F
run fbank_default python bench.py $F
python tools/bench_speed_fbank.py 2>/dev/null | tee "$OUT/speed_fbank.txt"
HIPFEAT_NO_FLAT=1 python tools/bench_speed_fbank.py 2>/dev/null | tee -a "$OUT/speed_fbank.txt"
