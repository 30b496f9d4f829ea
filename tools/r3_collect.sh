#!/bin/bash
# Round-3 evidence run (GPU box), ONE call: bench lines of the three configs, rocprofv3 kernel stats of the headline command, PMC passes,
# the FFT-only ceiling under the power cap, secondary benches, the GPU test log with the parity artefact.
# usage: tools/r3_collect.sh <outdir under gpurun_out>
set -u
OUT=gpurun_out/${1:-r3_final}
mkdir -p "$OUT"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python bench.py --config mfcc40_libri > "$OUT/bench_mfcc40_libri.json" 2> "$OUT/bench_mfcc40_libri.err"
python bench.py --config onthefly > "$OUT/bench_onthefly.json" 2> "$OUT/bench_onthefly.err"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python bench.py --no-cpu-baseline --no-host-fed --steps 50 > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
db=$(find "$OUT/prof" -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" > "$OUT/kernel_stats.txt" 2>&1
rm -rf "$OUT/prof"
tools/pmc_profile.sh "$OUT/pmc" --no-host-fed --no-parity > /dev/null 2>&1
cp "$OUT/pmc/summary.txt" "$OUT/pmc.txt" 2>/dev/null
rm -rf "$OUT/pmc"
tools/fft_ceiling.sh "$OUT/fft_ceiling.txt" 5 > "$OUT/fft_ceiling.log" 2>&1
python tools/bench_defaults.py > "$OUT/defaults.txt" 2>&1
python tools/bench_rates.py --cuts 4000 > "$OUT/rates.txt" 2>&1
python tools/bench_librosa.py > "$OUT/librosa.txt" 2>&1
{ python tools/bench_whisper.py --cuts 4000 --steps 20; python tools/bench_whisper.py --cuts 4000 --steps 20 --quiet-tail 0.3; python tools/bench_whisper.py --cuts 60 --steps 50; } > "$OUT/whisper.txt" 2>&1
python tools/bench_8k.py > "$OUT/8k.txt" 2>&1
python tools/bench_speed_fbank.py > "$OUT/speed_fbank.txt" 2>&1
python tools/parity_probe.py 64 > "$OUT/parity_probe.txt" 2>&1
python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.txt" 2>&1
cp gpurun_out/parity_report.json "$OUT/parity_report.json" 2>/dev/null
tail -2 "$OUT/pytest_gpu.txt"; tail -c 600 "$OUT/bench.json"
