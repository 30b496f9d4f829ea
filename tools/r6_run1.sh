#!/bin/bash
# Round 6, call 1: baseline state of the tree on this round's first box + SQ counters for the MFCC / on-the-fly kernels (VERDICT r5 task 4).
set -u
OUT=gpurun_out/r6_run1
mkdir -p "$OUT"
python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
for cfg in mfcc40_libri onthefly fbank16k; do
  BENCH_SETTLE=0 tools/pmc_any.sh "$OUT/pmc_$cfg" python bench.py --config $cfg --steps 2 --warmup 1 --no-parity --no-extra --no-cpu-baseline --no-host-fed --no-other-configs > "$OUT/pmc_$cfg.txt" 2>&1
done
timeout 1200 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.txt" 2>&1
tail -2 "$OUT/pytest_gpu.txt"; tail -c 400 "$OUT/bench.json"; echo
for cfg in mfcc40_libri onthefly fbank16k; do echo "== $cfg"; cat "$OUT/pmc_$cfg/summary.txt"; done
