#!/bin/bash
# full GPU suite after the workgroup->cut map / rounds rule / librosa instance; the three BASELINE configs
set -u
OUT=gpurun_out/${1:-r4_run7}
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest_gpu.txt" 2>&1
tail -3 "$OUT/pytest_gpu.txt"
cp gpurun_out/parity_report.json "$OUT/parity_report.json" 2>/dev/null
for cfg in fbank16k mfcc40_libri; do
  python bench.py --config $cfg --no-cpu-baseline --no-extra > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
  HIPFEAT_ROUNDS_R3=1 python bench.py --config $cfg --no-cpu-baseline --no-extra > "$OUT/bench_${cfg}_r3rule.json" 2>/dev/null
  python - <<PY
import json
for f in ("$OUT/bench_$cfg.json", "$OUT/bench_${cfg}_r3rule.json"):
    r=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f.split("/")[-1], r["value"], r["roofline"]["frac"], r["parity"]["pass"], r["config"]["kernel"])
PY
done
python bench.py --config onthefly --no-cpu-baseline > "$OUT/bench_onthefly.json" 2> "$OUT/bench_onthefly.err"
python - <<PY
import json
r=json.loads([l for l in open("$OUT/bench_onthefly.json") if l.startswith("{")][-1]); print("onthefly", r["value"], r["roofline"]["frac"], r["parity"]["pass"]); print(json.dumps(r["extra"]["routes"], indent=0)[:1500]); print(r["extra"].get("host_fed"))
PY
