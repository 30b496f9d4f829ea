#!/bin/bash
# round 5, GPU call 9: clause 2 of the parity statement against float64 truth -- the samples of all eight ranks of the driver's run, and the N = 2 code path
set -u
OUT=gpurun_out/${1:-r5_run9}
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "headline" > "$OUT/pytest_headline.txt" 2>&1; tail -3 "$OUT/pytest_headline.txt"
cp gpurun_out/parity_report.json "$OUT/parity_report_headline.json" 2>/dev/null
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --gpus 2 --dist-backend gloo --steps 5 --cuts 2000 --no-cpu-baseline > "$OUT/bench_2ranks_gloo.json" 2> "$OUT/bench_2ranks_gloo.err"
echo "2 ranks rc=$?"; grep -v "Gloo\|Expected\|socket" "$OUT/bench_2ranks_gloo.err" | tail -5
python - "$OUT/bench_2ranks_gloo.json" <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
p=r['parity']
print('N=2 value', r['value'], 'n_gpus', r['n_gpus'], 'backend', r['config']['dist_backend'], 'numa', r['config']['numa'])
print('parity pass', p['pass'], p['pass_rel_l2'], p['pass_linear'], p['pass_elementwise'], p['linear_domain_vs_f64'], 'K', p['K_measured'])
for k,c in r['extra'].get('configs',{}).items(): print(k, c['value'], c['rank_launch_ms'], c['parity']['pass'])
print('host-fed aggregate', r['extra'].get('aggregate_over_ranks'))
print('per-rank', [x.get('host_fed_cuts_per_s',{}).get('batch_60') for x in r['extra'].get('per_rank',[])])
PY
