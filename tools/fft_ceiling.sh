#!/bin/bash
# FFT-only ceiling of the fft512c design under the package power cap (GPU box).  Runs tools/ubench/fft_ceiling at every level on noise
# (and levels 1 / 3 on zeros) with rocm-smi sampled every 0.5 s, then the product kernel (bench.py, 2000 steps) on the same box for reference.
# usage: tools/fft_ceiling.sh <outfile> [seconds per level]
OUT=${1:-gpurun_out/fft_ceiling.txt}; SEC=${2:-5}
BIN=tools/ubench/fft_ceiling
[ -x $BIN ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -I lhotse_amd/csrc tools/ubench/fft_ceiling.hip -o $BIN || exit 1
: > "$OUT"
probe() {  # probe <label> <command...>: run the command under rocm-smi sampling, append one JSON line (command's last line + power / clock medians)
  local label=$1; shift
  ( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power \(W\)" | tr '\n' ' '; echo; sleep 0.5; done ) > /tmp/smi_probe.txt &
  local SMI=$!
  local line; line=$("$@" 2>/dev/null | tail -1)
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  python - "$label" "$line" /tmp/smi_probe.txt >> "$OUT" <<'PY'
import sys, json, re, statistics
label, line, path = sys.argv[1], sys.argv[2], sys.argv[3]
r = json.loads(line)
pw, ck = [], []
for l in open(path):
    m = re.search(r"Power \(W\): ([0-9.]+)", l); c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", l)
    if m and c and float(m.group(1)) > 600:
        pw.append(float(m.group(1))); ck.append(int(c.group(1)))
keep = {k: r[k] for k in ("level", "input", "frames_per_s", "cut_equiv_per_s", "frac_of_hbm_if_rest_free", "value", "ms_per_step") if k in r}
if "roofline" in r: keep["frac_of_hbm"] = r["roofline"]["frac"]
keep.update(what=label, samples_under_load=len(pw), package_power_W_median=statistics.median(pw) if pw else None,
            sclk_MHz_median=statistics.median(ck) if ck else None)
print(json.dumps(keep))
PY
}
for lvl in 0 1 2 3; do probe "ceiling level $lvl" $BIN $lvl $SEC; done
for lvl in 1 3; do probe "ceiling level $lvl (zeros)" $BIN $lvl $SEC zeros; done
probe "product fft512c via bench.py (uniform noise)" python bench.py --no-cpu-baseline --no-host-fed --no-parity --steps 1500 --input uniform
cat "$OUT"
