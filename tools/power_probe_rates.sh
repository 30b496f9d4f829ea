#!/bin/bash
# Package power and shader clock while tools/bench_rates.py runs one sampling rate for a few seconds (GPU box): tells a power-limited
# kernel (package at the cap, clock well under 2.4 GHz) from a latency-limited one.  usage: tools/power_probe_rates.sh <outfile> [rates...]
OUT=${1:-gpurun_out/power_probe_rates.txt}; shift
RATES=${@:-8000 16000 24000 48000}
: > "$OUT"
for sr in $RATES; do
  steps=$(( 16000 * 1500 / sr ))
  ( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power \(W\)" | tr '\n' ' '; echo; sleep 0.5; done ) > /tmp/smi_$sr.txt &
  SMI=$!
  line=$(python tools/bench_rates.py --rates $sr --cuts 4000 --steps $steps 2>/dev/null | tail -1)
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  python - "$line" /tmp/smi_$sr.txt >> "$OUT" <<'PY'
import sys, json, re, statistics
line, path = sys.argv[1], sys.argv[2]
r = json.loads(line)
pw, ck = [], []
for l in open(path):
    m = re.search(r"Power \(W\): ([0-9.]+)", l); c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", l)
    if m and c and float(m.group(1)) > 600:
        pw.append(float(m.group(1))); ck.append(int(c.group(1)))
print(json.dumps({"sampling_rate": r["sampling_rate"], "kernel": r["kernel"], "cuts_per_s": r["cuts_per_s"], "frac_of_8TBps": r["frac_of_8TBps"], "samples_under_load": len(pw),
                  "package_power_W_median": statistics.median(pw) if pw else None, "sclk_MHz_median": statistics.median(ck) if ck else None}))
PY
done
cat "$OUT"
