#!/bin/bash
# Leg C (DataLoader, packed transport) vs leg D (shared ring) of the plumbing workload on the GPU box, one process and several sharing the GPU.
# Usage (GPU box): bash tools/ring_sweep.sh <out dir>
OUT=${1:-gpurun_out/r6_ring2}; mkdir -p $OUT
python tools/loader_worker_probe.py > $OUT/worker_probe.txt 2>&1
python - <<'PY'
import sys; sys.path.insert(0,'tools'); import plumbing as P
P.write_corpus('/dev/shm/ring_wav', 64)
PY
stat() { grep -E "usage_usec|user_usec|system_usec|nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; }
one() {  # leg workers extra...
  local leg=$1 w=$2; shift 2
  HIPFEAT_NO_FORK_WARNING=1 python tools/plumbing.py --leg $leg --wav-dir /dev/shm/ring_wav --repeat 1000 --workers $w --passes 1 "$@" 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('leg $leg workers $w $*', r['cuts_per_s'], r['seconds_to_first_batch'], {k:v for k,v in r.items() if k.endswith('share') or k.startswith(('container','quota','cpus','ring_slots','batches'))})"
}
{
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max)"
for rep in 1 2; do
  for w in 6 8 12; do
    for extra in "" "--pcm16 --half"; do
      one D $w $extra; one D $w $extra --no-pin
    done
  done
  one C 8; one C 8 --pcm16 --half
done
} 2>&1 | tee $OUT/ab.txt
# several processes sharing the GPU, leg D each (own plan, pipeline, archive, ring), started together
for cfg in "2 6" "4 3"; do
  set -- $cfg; procs=$1; w=$2
  for extra in "" "--pcm16 --half"; do
    at=$(python -c "import time; print(time.time()+12)")
    for k in $(seq 1 $procs); do
      HIPFEAT_NO_FORK_WARNING=1 python tools/plumbing.py --leg D --wav-dir /dev/shm/ring_wav --repeat 1000 --workers $w --passes 1 --start-at $at $extra 2>/dev/null > $OUT/mp_${procs}x${w}_$k.json &
    done
    wait
    python - $OUT $procs $w "$extra" <<'PY'
import sys, json
out, procs, w, extra = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
rs = [json.loads(open(f"{out}/mp_{procs}x{w}_{k}.json").readline()) for k in range(1, procs + 1)]
a = max(r["steady_region_epoch"][0] for r in rs); b = min(r["steady_region_epoch"][1] for r in rs)
tot = sum(r["steady_cuts"] for r in rs) / (max(r["steady_region_epoch"][1] for r in rs) - min(r["steady_region_epoch"][0] for r in rs))
print(f"{procs} processes x {w} workers {extra}: {tot:.0f} cuts/s in total (each {[r['cuts_per_s'] for r in rs]}), overlap of the steady regions {(b - a):.2f} s")
PY
  done
done 2>&1 | tee $OUT/multi.txt
rm -rf /dev/shm/ring_wav
