#!/bin/bash
# round 5, GPU call 11: chunks per batch in the library's host pipeline (its thread binds the float32 variants at 0.86-0.95 busy, half of it enqueue calls)
set -u
OUT=gpurun_out/${1:-r5_run11}
mkdir -p "$OUT"
for rep in 1 2; do
for ch in 4 2 1; do
  HIPFEAT_PIPE_CHUNKS=$ch timeout 300 python bench.py --config bulk_save --steps 3 --no-cpu-baseline --no-parity > "$OUT/b_$ch.json" 2>/dev/null
  python - "$OUT/b_$ch.json" "$ch" <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
b=r['extra']['bulk_save']
pick=lambda k: [v for kk,v in b.items() if kk.startswith(k) and '(8 files)' in kk][0]
s=' | '.join('%s %.0f (arch %.2f wait %.2f man %.2f pipe %.2f pack %.2f)' % (n, v['cuts_per_s'], v['archive_thread_busy_share'], v['archive_thread_waiting_for_the_device_share'], v['manifest_thread_busy_share'], v['pipeline_thread_busy_share'], v['pipeline_thread_packing_share']) for n,v in (('f32',pick('float32->hip_archive ')),('i16',pick('int16->hip_archive ')),('f32f16',pick('float32->hip_archive_f16')),('i16f16',pick('int16->hip_archive_f16'))))
print('chunks=%s value %.0f | %s' % (sys.argv[2], r['value'], s))
PY
done
done | tee "$OUT/chunks.txt"
