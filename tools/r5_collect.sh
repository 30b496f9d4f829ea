#!/bin/bash
# Round-5 evidence run (GPU box), ONE call: the driver's bench line (all BASELINE configs in it), rocprofv3 kernel stats of the same
# command, the offline path, the N = 2 code path on one GPU (gloo), configs[2] as written, the other sampling rates with their kernel
# stats, the GPU test log with the parity artefact, smoke.
# usage: tools/r5_collect.sh <outdir under gpurun_out>
set -u
OUT=gpurun_out/${1:-r5_final}
mkdir -p "$OUT"
python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-fed > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
db=$(find "$OUT/prof" -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" > "$OUT/kernel_stats.txt" 2>&1
rm -rf "$OUT/prof"
python bench.py --config bulk_save --no-cpu-baseline > "$OUT/bench_bulk_save.json" 2> "$OUT/bench_bulk_save.err"
python bench.py --total-cuts 100000 --steps 10 --no-cpu-baseline --no-extra > "$OUT/bench_total100k.json" 2> "$OUT/bench_total100k.err"
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --gpus 2 --dist-backend gloo --steps 5 --cuts 2000 --no-cpu-baseline > "$OUT/bench_2ranks_gloo.json" 2> "$OUT/bench_2ranks_gloo.err"
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o rates -- python tools/bench_rates.py --cuts 4000 --rates 22050,24000,32000,44100,48000 > "$OUT/rates.txt" 2>> "$OUT/rocprof.err"
db=$(find "$OUT/prof" -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" > "$OUT/rates_kernel_stats.txt" 2>&1
rm -rf "$OUT/prof"
python tools/bench_rates.py --cuts 4000 > "$OUT/rates_all.txt" 2>&1
python tools/bench_defaults.py > "$OUT/defaults.txt" 2>&1
python __graft_entry__.py --smoke > "$OUT/smoke.txt" 2>&1
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.txt" 2>&1
cp gpurun_out/parity_report.json "$OUT/parity_report.json" 2>/dev/null
tail -2 "$OUT/pytest_gpu.txt"; tail -c 300 "$OUT/bench.json"; echo; tail -c 600 "$OUT/bench_2ranks_gloo.json"; echo; tail -3 "$OUT/bench_2ranks_gloo.err"; cat "$OUT/rates.txt"
