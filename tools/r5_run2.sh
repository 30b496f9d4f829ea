#!/bin/bash
# round 5, GPU call 2: the offline path with the native save stages (hipfeat_archive_append + hipfeat_manifest_lines), 1 and 4 stripes,
# and what a page-cache file takes from one / several writers on this host
set -u
OUT=gpurun_out/${1:-r5_run2}
mkdir -p "$OUT"
python tools/tmpfs_write_probe.py /dev/shm 40 | tee "$OUT/tmpfs_write_probe.json"
for st in 1 4; do
  timeout 600 python bench.py --config bulk_save --stripes $st --no-cpu-baseline > "$OUT/bulk_save_stripes$st.json" 2> "$OUT/bulk_save_stripes$st.err"
  python - "$OUT/bulk_save_stripes$st.json" <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('value', r['value'], 'parity', r['parity']['pass'])
for k,v in r['extra']['bulk_save'].items():
    print(' ', k, v if not isinstance(v,dict) else {a:b for a,b in v.items() if a!='binds'})
PY
done
