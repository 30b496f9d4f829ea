#!/bin/bash
# round 5, GPU call 1: the full GPU suite on the re-based parity statement (ref32 = the reference's float32 torch calls, K = 3), the
# driver's form of the bench line (extra.configs, both regimes), and the counter passes of the other two BASELINE kernels
set -u
OUT=gpurun_out/${1:-r5_run1}
mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bench_cli.py::test_the_default_line_carries_the_other_baseline_configs_and_both_regimes > "$OUT/pytest_gpu.txt" 2>&1
tail -5 "$OUT/pytest_gpu.txt"
cp gpurun_out/parity_report.json "$OUT/parity_report.json" 2>/dev/null
( time timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2>&1 | tail -3
python - "$OUT/bench_default.json" <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
p=r['parity']
print('headline %.0f cuts/s frac %.4f  parity pass=%s K=%.2f/%s floor=%.2e hip_f64=%.2e hip_ref=%.2e np32floor=%.2e' % (r['value'], r['roofline']['frac'], p['pass'], p['K_measured'], p['K_allowed'], p['oracle_f32_vs_f64_max_abs'], p['hip_vs_f64_max_abs'], p['max_abs_max'], p['numpy32_floor_of_rounds_1_to_4']['numpy32_vs_f64_max_abs']))
for k,c in r['extra'].get('configs',{}).items():
    print(k, c['value'], c['ms_per_step'], c['roofline']['frac'], c['roofline'].get('frac_end_to_end'), c['parity'])
print('ramp', r['extra'].get('first_launches_after_idle_ms'), r['extra'].get('contract_only'))
print('host_fed', r['extra'].get('host_fed_cuts_per_s'))
print('cpu', r['cpu_baseline']['value'], r['cpu_baseline']['cores'])
PY
tools/r5_traffic.sh "$OUT/traffic" 2>&1 | tail -60
cp profiles/traffic.json "$OUT/traffic.json"
