#!/bin/bash
# round 5, GPU call 10: what limits the offline path's pipeline thread -- packing threads x NUMA placement (float32 -> hip_archive and int16 -> f16, 8 stripes)
set -u
OUT=gpurun_out/${1:-r5_run10}
mkdir -p "$OUT"
for numa in off on; do
  for th in 8 12 24; do
    HIPFEAT_COPY_THREADS=$th timeout 300 python bench.py --config bulk_save --numa $numa --steps 3 --no-cpu-baseline --no-parity > "$OUT/b_${numa}_$th.json" 2>/dev/null
    python - "$OUT/b_${numa}_$th.json" "$numa" "$th" <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
b=r['extra']['bulk_save']
pick=lambda k: [v for kk,v in b.items() if kk.startswith(k) and '(8 files)' in kk][0]
f=pick('float32->hip_archive '); i=pick('int16->hip_archive_f16')
print('numa=%s threads=%s value %.0f | f32 %.0f (caller %.2f arch %.2f wait %.2f man %.2f) | i16f16 %.0f (caller %.2f arch %.2f wait %.2f man %.2f) | %s' % (sys.argv[2], sys.argv[3], r['value'], f['cuts_per_s'], f['main_thread_extract_share'], f['archive_thread_busy_share'], f['archive_thread_waiting_for_the_device_share'], f['manifest_thread_busy_share'], i['cuts_per_s'], i['main_thread_extract_share'], i['archive_thread_busy_share'], i['archive_thread_waiting_for_the_device_share'], i['manifest_thread_busy_share'], r['config']['numa'].get('why','')[:60]))
PY
  done
done | tee "$OUT/copy_threads.txt"
