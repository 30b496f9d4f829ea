#!/bin/bash
# Round-2 evidence run (GPU box): bench line, rocprofv3 kernel stats of the same command, PMC passes, secondary benches, GPU test log.
# usage: tools/r2_collect.sh <outdir under gpurun_out>
set -u
OUT=gpurun_out/${1:-r2_final}
mkdir -p "$OUT"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python bench.py --no-cpu-baseline --no-host-fed --steps 50 > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
db=$(find "$OUT/prof" -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" > "$OUT/kernel_stats.txt" 2>&1
rm -rf "$OUT/prof"
tools/pmc_profile.sh "$OUT/pmc" --no-host-fed --no-parity > /dev/null 2>&1
cp "$OUT/pmc/summary.txt" "$OUT/pmc.txt" 2>/dev/null
rm -rf "$OUT/pmc"
python tools/bench_defaults.py > "$OUT/defaults.txt" 2>&1
python tools/bench_rates.py --cuts 4000 > "$OUT/rates.txt" 2>&1
python tools/bench_librosa.py > "$OUT/librosa.txt" 2>&1
{ python tools/bench_whisper.py --cuts 4000 --steps 20; python tools/bench_whisper.py --cuts 4000 --steps 20 --quiet-tail 0.3; python tools/bench_whisper.py --cuts 4000 --steps 20 --quiet-tail 2
  HIPFEAT_WHISPER_VARIANT=2 python tools/bench_whisper.py --cuts 4000 --steps 20; python tools/bench_whisper.py --cuts 60 --steps 50; } > "$OUT/whisper.txt" 2>&1
if [ -f lhotse_amd/_lib/var_pt.so ]; then  # experiment build with -DHIPFEAT_PHASE_TIMERS (tools/variants.py pt:"-DHIPFEAT_PHASE_TIMERS")
  { HIPFEAT_LIB=$PWD/lhotse_amd/_lib/var_pt.so python tools/phase_timers_c.py 4000; HIPFEAT_LIB=$PWD/lhotse_amd/_lib/var_pt.so python tools/phase_timers_w3.py 4000
    HIPFEAT_LIB=$PWD/lhotse_amd/_lib/var_pt.so python tools/phase_timers_w.py 1000 24000; HIPFEAT_LIB=$PWD/lhotse_amd/_lib/var_pt.so python tools/phase_timers_w.py 1000 48000; } > "$OUT/phase_timers.txt" 2>&1
fi
{ python tools/bench_mfcc.py; HIPFEAT_FFT512_VARIANT=b python tools/bench_mfcc.py; } > "$OUT/mfcc.txt" 2>&1
{ HIPFEAT_FFT512_VARIANT=b python bench.py --no-cpu-baseline --no-host-fed --steps 100; HIPFEAT_NO_WAVE_AUTONOMOUS=1 python tools/bench_rates.py --cuts 2000 --rates 24000,48000; python tools/fft_accuracy.py; } > "$OUT/ab_old_kernels.txt" 2>&1
python tools/bench_8k.py > "$OUT/8k.txt" 2>&1
python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.txt" 2>&1
cp gpurun_out/parity_report.json "$OUT/parity_report.json" 2>/dev/null
tail -2 "$OUT/pytest_gpu.txt"; tail -c 600 "$OUT/bench.json"
