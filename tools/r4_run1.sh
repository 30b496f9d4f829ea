#!/bin/bash
# Round-4 first GPU call: the new GPU tests, then the full suite, the driver's bench line, the on-the-fly and bulk-save configs.
set -u
OUT=gpurun_out/${1:-r4_run1}
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_minibatch.py tests/test_gpu_bench_cli.py tests/test_gpu_host_pipeline.py -x -q > "$OUT/pytest_new.txt" 2>&1
tail -5 "$OUT/pytest_new.txt"
timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.txt" 2>&1
tail -3 "$OUT/pytest_gpu.txt"
cp gpurun_out/parity_report.json "$OUT/parity_report.json" 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 400 "$OUT/bench.json"
timeout 600 python bench.py --config onthefly --no-cpu-baseline > "$OUT/bench_onthefly.json" 2> "$OUT/bench_onthefly.err"; tail -c 1500 "$OUT/bench_onthefly.json"; tail -3 "$OUT/bench_onthefly.err"
timeout 600 python bench.py --config bulk_save --no-cpu-baseline > "$OUT/bench_bulk_save.json" 2> "$OUT/bench_bulk_save.err"; tail -c 2500 "$OUT/bench_bulk_save.json"; tail -3 "$OUT/bench_bulk_save.err"
