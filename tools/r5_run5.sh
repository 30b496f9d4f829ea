#!/bin/bash
# round 5, GPU call 5: the whole GPU suite with the native save path in place + the driver-form bench line
set -u
OUT=gpurun_out/${1:-r5_run5}
mkdir -p "$OUT"
timeout 1200 python -m pytest tests -m gpu -q -x -o faulthandler_timeout=120 --deselect tests/test_gpu_bench_cli.py::test_the_default_line_carries_the_other_baseline_configs_and_both_regimes > "$OUT/pytest_gpu.txt" 2>&1
tail -5 "$OUT/pytest_gpu.txt"
cp gpurun_out/parity_report.json "$OUT/parity_report.json" 2>/dev/null
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; tail -8 "$OUT/smoke.txt"
