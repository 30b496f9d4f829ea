#!/bin/bash
# Package power and shader clock while tools/bench_defaults.py runs ONE default extractor for a few seconds (GPU box): power-limited
# (package at the ~1.4 kW cap, clock under 2.4 GHz) or latency-limited (full clock, power under the cap)?
# usage: tools/power_probe_defaults.sh <outfile> [extractor names...]
OUT=${1:-gpurun_out/power_probe_defaults.txt}; shift
NAMES=${@:-hip-mfcc hip-spectrogram hip-whisper-fbank hip-librosa-fbank}
: > "$OUT"
for n in $NAMES; do
  ( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power \(W\)" | tr '\n' ' '; echo; sleep 0.5; done ) > /tmp/smi_$n.txt &
  SMI=$!
  line=$(python tools/bench_defaults.py --only $n --cuts 4000 --steps 1500 2>/dev/null | tail -1)
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  python - "$line" /tmp/smi_$n.txt >> "$OUT" <<'PY'
import sys, json, re, statistics
line, path = sys.argv[1], sys.argv[2]
r = json.loads(line)
pw, ck = [], []
for l in open(path):
    m = re.search(r"Power \(W\): ([0-9.]+)", l); c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", l)
    if m and c and float(m.group(1)) > 600:
        pw.append(float(m.group(1))); ck.append(int(c.group(1)))
r.update(samples_under_load=len(pw), package_power_W_median=statistics.median(pw) if pw else None, sclk_MHz_median=statistics.median(ck) if ck else None)
print(json.dumps(r))
PY
done
cat "$OUT"
