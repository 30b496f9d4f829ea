#!/bin/bash
# Round-4 evidence run (GPU box), ONE call: the driver's bench line, the other configs, rocprofv3 kernel stats of the headline command, PMC
# passes (-> traffic.json), the power / clock probe, secondary benches, the GPU test log with the parity artefact.
# usage: tools/r4_collect.sh <outdir under gpurun_out>
set -u
OUT=gpurun_out/${1:-r4_final}
mkdir -p "$OUT"
python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
python bench.py > "$OUT/bench_default_steps.json" 2> /dev/null
python bench.py --config mfcc40_libri > "$OUT/bench_mfcc40_libri.json" 2> "$OUT/bench_mfcc40_libri.err"
HIPFEAT_NO_FLAT=1 python bench.py --config mfcc40_libri --no-cpu-baseline --no-extra > "$OUT/bench_mfcc40_libri_noflat.json" 2> /dev/null
python bench.py --config onthefly > "$OUT/bench_onthefly.json" 2> "$OUT/bench_onthefly.err"
python bench.py --config onthefly --prefetch 4 --streams 2 --no-cpu-baseline --no-extra > "$OUT/bench_onthefly_prefetch4.json" 2> /dev/null
python bench.py --config bulk_save --no-cpu-baseline > "$OUT/bench_bulk_save.json" 2> "$OUT/bench_bulk_save.err"
python bench.py --total-cuts 100000 --steps 10 --no-cpu-baseline --no-extra > "$OUT/bench_total100k.json" 2> "$OUT/bench_total100k.err"
python tools/host_profile_minibatch.py 1 > "$OUT/host_profile_minibatch.txt" 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python bench.py --no-cpu-baseline --no-extra --steps 50 > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
db=$(find "$OUT/prof" -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" > "$OUT/kernel_stats.txt" 2>&1
rm -rf "$OUT/prof"
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o otf -- python bench.py --config onthefly --no-cpu-baseline --no-extra --no-parity --steps 30 > "$OUT/bench_onthefly_under_rocprof.json" 2>> "$OUT/rocprof.err"
db=$(find "$OUT/prof" -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" > "$OUT/kernel_stats_onthefly.txt" 2>&1
rm -rf "$OUT/prof"
tools/pmc_profile.sh "$OUT/pmc" --no-extra --no-parity > /dev/null 2>&1
cp "$OUT/pmc/summary.txt" "$OUT/pmc.txt" 2>/dev/null
rm -rf "$OUT/pmc"
( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power \(W\)" | tr '\n' ' '; echo; sleep 0.5; done ) > "$OUT/smi_uniform.txt" &
SMI=$!
python bench.py --no-cpu-baseline --no-extra --no-parity --steps 2000 > "$OUT/bench_2000steps.json" 2>/dev/null
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
python tools/bench_rates.py --cuts 4000 > "$OUT/rates.txt" 2>&1
python tools/bench_defaults.py > "$OUT/defaults.txt" 2>&1
python tools/bench_librosa.py --cuts 4000 --steps 20 > "$OUT/librosa.txt" 2>&1
HIPFEAT_NO_FIXED_SCHEDULE=1 python tools/bench_librosa.py --cuts 4000 --steps 20 >> "$OUT/librosa.txt" 2>&1
{ python tools/bench_whisper.py --cuts 4000 --steps 20; python tools/bench_whisper.py --cuts 60 --steps 50; } > "$OUT/whisper.txt" 2>&1
python tools/bench_speed_fbank.py > "$OUT/speed_fbank.txt" 2>&1
python tools/parity_probe.py 64 > "$OUT/parity_probe.txt" 2>&1
python tools/launch_ramp2.py > "$OUT/launch_ramp.txt" 2>&1
python __graft_entry__.py --smoke > "$OUT/smoke.txt" 2>&1
python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.txt" 2>&1
cp gpurun_out/parity_report.json "$OUT/parity_report.json" 2>/dev/null
tail -2 "$OUT/pytest_gpu.txt"; tail -c 400 "$OUT/bench.json"; head -5 "$OUT/host_profile_minibatch.txt"
