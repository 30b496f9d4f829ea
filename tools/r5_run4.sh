#!/bin/bash
# round 5, GPU call 4: the asynchronous host pipeline under a watchdog, then its tests, then the offline path on it
set -u
OUT=gpurun_out/${1:-r5_run4}
mkdir -p "$OUT"
timeout 200 python tools/host_pipeline_stress.py > "$OUT/stress.txt" 2>&1; echo "stress rc=$?"; tail -25 "$OUT/stress.txt"
timeout 300 python -m pytest tests/test_gpu_bulk_save.py tests/test_gpu_host_pipeline.py tests/test_gpu_driver_threading.py -q -x > "$OUT/pytest_bulk.txt" 2>&1
tail -5 "$OUT/pytest_bulk.txt"
for st in 1 8; do
  timeout 400 python bench.py --config bulk_save --stripes $st --no-cpu-baseline > "$OUT/bulk_save_stripes$st.json" 2> "$OUT/bulk_save_stripes$st.err"
  tail -2 "$OUT/bulk_save_stripes$st.err"
  python - "$OUT/bulk_save_stripes$st.json" <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('value', r['value'], 'parity', r['parity']['pass'])
for k,v in r['extra']['bulk_save'].items():
    print(' ', k, v if not isinstance(v,dict) else {a:b for a,b in v.items() if a!='binds'}, '|', v.get('binds') if isinstance(v,dict) else '')
PY
done
