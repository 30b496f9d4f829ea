#!/bin/bash
# Round 6, call 4: the plumbing config, then the default driver-form line with everything in it.
set -u
OUT=gpurun_out/r6_run4
mkdir -p "$OUT"
timeout 900 python bench.py --config plumbing > "$OUT/bench_plumbing.json" 2> "$OUT/bench_plumbing.err"
echo "plumbing rc=$?"; tail -c 1500 "$OUT/bench_plumbing.err"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -c 800 "$OUT/bench.err"
python - <<'PY'
import json
for f in ("gpurun_out/r6_run4/bench_plumbing.json","gpurun_out/r6_run4/bench.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("parity",{}).get("pass"))
        if "plumbing" in f:
            for k,v in d["extra"]["plumbing"].items():
                if isinstance(v,dict): print("  ",k[:90], v.get("cuts_per_s"), {a:b for a,b in v.items() if a.endswith("share")})
        else:
            p=d["extra"]["configs"].get("plumbing",{})
            print("  plumbing in default line:", p.get("value"), p.get("error"), p.get("parity"))
            for k,v in (p.get("legs") or {}).items():
                if isinstance(v,dict): print("  ",k[:90], v.get("cuts_per_s"))
    except Exception as e:
        print(f, "ERR", e)
PY
