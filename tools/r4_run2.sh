#!/bin/bash
# on-the-fly config: where does a mini-batch's time go?  kernel trace (durations per kernel) + same-call A/B of the rounds rule
set -u
OUT=gpurun_out/${1:-r4_run2}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F="--config onthefly --no-cpu-baseline --no-host-fed --no-parity"
for tag in new r3rule r8 r6 r3; do
  case $tag in new) E="";; r3rule) E="HIPFEAT_ROUNDS_R3=1";; r8) E="HIPFEAT_ROUNDS=8";; r6) E="HIPFEAT_ROUNDS=6";; r3) E="HIPFEAT_ROUNDS=3";; esac
  env $E python bench.py $F --steps 40 2>/dev/null | python -c "
import sys,json
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); x=r['extra']['routes']
print('$tag', r['value'], r['roofline']['launch_ms'], {k:(v['cuts_per_s'],v['host_us_per_minibatch']) for k,v in x.items() if k!='what'})" | tee -a "$OUT/ab.txt"
done
python bench.py $F --steps 40 --streams 1 2>/dev/null | python -c "
import sys,json
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('1stream', r['value'], r['roofline']['launch_ms'])" | tee -a "$OUT/ab.txt"
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o otf -- python bench.py $F --steps 20 --streams 1 > "$OUT/otf_rocprof.json" 2> "$OUT/rocprof.err"
db=$(find "$OUT/prof" -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" > "$OUT/otf_kernel_stats.txt" 2>&1
head -30 "$OUT/otf_kernel_stats.txt"
ls "$OUT/prof"/* | head; find "$OUT/prof" -name "*kernel_trace*" | head -2
