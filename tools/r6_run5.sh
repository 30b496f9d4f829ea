#!/bin/bash
# Round 6, call 5: MFCC MODE 2 A/B; plumbing with packed transport; full GPU suite on the no-scratch build.
set -u
OUT=gpurun_out/r6_run5
mkdir -p "$OUT"
tools/r6_mfcc_ab.sh "$OUT/mfcc_ab" > "$OUT/mfcc_ab.txt" 2>&1
cat "$OUT/mfcc_ab.txt"
timeout 900 python bench.py --config plumbing > "$OUT/bench_plumbing.json" 2> "$OUT/bench_plumbing.err"
echo "plumbing rc=$?"; tail -c 600 "$OUT/bench_plumbing.err"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6_run5/bench_plumbing.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("parity",{}).get("pass"))
for k,v in d["extra"].get("plumbing", d["extra"]).items():
    if isinstance(v,dict): print("  ",k[:110], v.get("cuts_per_s"), v.get("cuts_per_s_incl_worker_start"), v.get("seconds_to_first_batch"), {a:b for a,b in v.items() if a.endswith("share")})
    else: print("  ", k, str(v)[:200])
PY
timeout 1400 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.txt" 2>&1
tail -5 "$OUT/pytest_gpu.txt"
