#!/bin/bash
# HBM-side counters of the OTHER two BASELINE kernels (VERDICT r4 task 2): fft512c<..., FLAT> on the ragged MFCC batch and the launch
# pair of the on-the-fly mini-batch (minibatch_prep_inline_kernel + the collated feature launch).  FETCH_SIZE and WRITE_SIZE in separate
# passes (MI355X_MICROARCH.md, HBM section); no tracing domain other than --kernel-trace.  Run on the GPU box:
#   tools/r5_traffic.sh gpurun_out/r5_traffic
set -u
OUT=${1:-gpurun_out/r5_traffic}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in mfcc40_libri onthefly; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d="$OUT/$cfg/pass_$ctr"
    mkdir -p "$d"
    BENCH_SETTLE=0 timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$d" -o p -- \
      python bench.py --config $cfg --steps 2 --warmup 1 --no-parity --no-extra --no-cpu-baseline > "$d.log" 2>&1
    echo "$cfg $ctr rc=$?"
  done
  python tools/make_traffic_json.py --config $cfg "$OUT/$cfg" > "$OUT/$cfg/summary.txt" 2>&1
  cat "$OUT/$cfg/summary.txt"
done
