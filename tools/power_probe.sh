#!/bin/bash
# Package power and shader clock while bench.py runs, for the three synthetic inputs (GPU box).  The kernel is the same; only the data
# the FFT datapath toggles differs.  usage: tools/power_probe.sh <outfile>
OUT=${1:-gpurun_out/power_probe.txt}
: > "$OUT"
for inp in uniform sine zeros; do
  ( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power \(W\)" | tr '\n' ' '; echo; sleep 0.5; done ) > /tmp/smi_$inp.txt &
  SMI=$!
  line=$(python bench.py --no-cpu-baseline --no-host-fed --no-parity --steps 2000 --input $inp 2>/dev/null | tail -1)
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  python - "$inp" "$line" /tmp/smi_$inp.txt >> "$OUT" <<'PY'
import sys, json, re, statistics
inp, line, path = sys.argv[1], sys.argv[2], sys.argv[3]
r = json.loads(line)
pw, ck = [], []
for l in open(path):
    m = re.search(r"Power \(W\): ([0-9.]+)", l); c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", l)
    if m and c and float(m.group(1)) > 600:  # samples taken while the kernel runs
        pw.append(float(m.group(1))); ck.append(int(c.group(1)))
print(json.dumps({"input": inp, "cuts_per_s": r["value"], "frac_of_hbm": r["roofline"]["frac"], "samples_under_load": len(pw),
                  "package_power_W_median": statistics.median(pw) if pw else None, "sclk_MHz_median": statistics.median(ck) if ck else None,
                  "sclk_MHz_min_max": [min(ck), max(ck)] if ck else None}))
PY
done
cat "$OUT"
