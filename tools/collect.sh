#!/bin/bash
# ONE evidence-collection script for the GPU box (replaces the per-round tools/r2_* ... r5_run*.sh of rounds 2-5, which live on in git history).
#
#   tools/collect.sh <outdir under gpurun_out> <section> [<section> ...]        e.g.  gpurun -- 'tools/collect.sh r6_final bench stats suite smoke'
#
# sections (each writes its files under gpurun_out/<outdir>/; copy what is to be judged into profiles/ with the round's prefix):
#   bench        the driver's form of the bench line (`--steps 20 --warmup 5`, every BASELINE config in it) + the default-steps form
#   stats        rocprofv3 --kernel-trace --stats of the headline command -> kernel_stats.txt (tools/rocprof_summary.py)
#   pmc          SQ counters of the three BASELINE kernels in separate --pmc passes (tools/pmc_any.sh) -> pmc_<config>/summary.txt
#   traffic      FETCH_SIZE / WRITE_SIZE passes of the three BASELINE kernels -> traffic_<config>/summary.txt (tools/make_traffic_json.py)
#   configs      bench.py --config mfcc40_libri / onthefly / bulk_save / plumbing, and --total-cuts 100000 (configs[2] on one GPU)
#   stripes      the offline path with 1 / 8 / 16 / 32 archive stripes
#   n2           the N = 2 code path on ONE GPU over gloo (bench.py --gpus 2 --dist-backend gloo)
#   rates        the other sampling rates (tools/bench_rates.py) with their kernel stats, defaults, librosa, whisper
#   power        package power / shader clock while the headline kernel runs (tools/power_probe.sh)
#   suite        python -m pytest tests -m gpu  (+ the parity artefact gpurun_out/parity_report.json)
#   smoke        __graft_entry__.smoke()
#   host         CPUs, cgroup quota, memory, /dev/shm of the box (tools/host_limits_probe.sh with the current ring loader)
#   ring         leg C (DataLoader) vs leg D (shared ring, page-locked / not) of the plumbing workload, 1 / 2 / 4 processes per GPU (tools/ring_sweep.sh)
#   mfcc_ab      the MODE 2 twiddle A/B of round 6 (tools/r6_mfcc_ab.sh; needs lhotse_amd/_lib/var_mfcc_tws.so)
set -u
NAME=${1:?outdir}; shift
OUT=gpurun_out/$NAME
mkdir -p "$OUT"
prof_env() { cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; }
summarise() { db=$(find "$1" -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py "$db" > "$2" 2>&1; rm -rf "$1"; }

for section in "$@"; do
  echo "== $section"
  case $section in
    bench)
      timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "rc=$?"
      timeout 600 python bench.py --no-cpu-baseline --no-extra > "$OUT/bench_default_steps.json" 2> /dev/null
      tail -c 300 "$OUT/bench.json"; echo ;;
    stats)
      prof_env
      # (BENCH_NO_PLUMBING: under rocprofv3's preloaded tool a multiprocessing fork server cannot restore its signal handlers, its workers die
      #  and the server itself never exits -- rocprofv3 then waits for it for ever; the kernels of the line are what is profiled here)
      BENCH_NO_PLUMBING=1 timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-fed > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
      summarise "$OUT/prof" "$OUT/kernel_stats.txt"; head -12 "$OUT/kernel_stats.txt" ;;
    pmc)
      for cfg in fbank16k mfcc40_libri onthefly; do
        BENCH_SETTLE=0 tools/pmc_any.sh "$OUT/pmc_$cfg" python bench.py --config $cfg --steps 2 --warmup 1 --no-parity --no-extra --no-cpu-baseline --no-host-fed --no-other-configs > "$OUT/pmc_$cfg.log" 2>&1
        find "$OUT/pmc_$cfg" -name "*.csv" -size +2M -delete 2>/dev/null
      done ;;
    traffic)
      prof_env
      for cfg in mfcc40_libri onthefly; do
        for ctr in FETCH_SIZE WRITE_SIZE; do
          d="$OUT/traffic_$cfg/pass_$ctr"; mkdir -p "$d"
          BENCH_SETTLE=0 timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$d" -o p -- python bench.py --config $cfg --steps 2 --warmup 1 --no-parity --no-extra --no-cpu-baseline > "$d.log" 2>&1
        done
        python tools/make_traffic_json.py --config $cfg "$OUT/traffic_$cfg" "$OUT/pmc_$cfg/summary.txt" > "$OUT/traffic_$cfg/summary.txt" 2>&1; tail -5 "$OUT/traffic_$cfg/summary.txt"
        find "$OUT/traffic_$cfg" -name "*.csv" -size +2M -delete 2>/dev/null
      done
      tools/pmc_profile.sh "$OUT/pmc_headline" --no-extra --no-parity --no-other-configs > /dev/null 2>&1; cp "$OUT/pmc_headline/summary.txt" "$OUT/pmc_headline.txt" 2>/dev/null
      find "$OUT/pmc_headline" -name "*.csv" -size +2M -delete 2>/dev/null
      # (profiles/traffic.json is written HERE, on the box: copy it back through gpurun_out)
      probe=""; [ -f "$OUT/power_probe.txt" ] && probe="$OUT/power_probe.txt"
      python tools/make_traffic_json.py "$OUT/pmc_headline.txt" 4000 "${PMC_LABEL:-profiles/r06_pmc.txt}" $probe > "$OUT/traffic_headline.json" 2>&1
      cp profiles/traffic.json "$OUT/traffic.json" ;;
    configs)
      for cfg in mfcc40_libri onthefly bulk_save plumbing; do
        timeout 900 python bench.py --config $cfg > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"; echo "$cfg rc=$?"
      done
      timeout 900 python bench.py --total-cuts 100000 --steps 10 --no-cpu-baseline --no-extra > "$OUT/bench_total100k.json" 2> "$OUT/bench_total100k.err" ;;
    stripes)
      for n in 1 8 16 32; do
        timeout 600 python bench.py --config bulk_save --stripes $n --no-cpu-baseline > "$OUT/bench_bulk_save_stripes$n.json" 2> /dev/null; echo "stripes $n rc=$?"
      done ;;
    n2)
      HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --gpus 2 --dist-backend gloo --steps 5 --cuts 2000 --no-cpu-baseline > "$OUT/bench_2ranks_gloo.json" 2> "$OUT/bench_2ranks_gloo.err"
      echo "rc=$?"; tail -c 400 "$OUT/bench_2ranks_gloo.json"; echo ;;
    rates)
      prof_env
      rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o rates -- python tools/bench_rates.py --cuts 4000 --rates 22050,24000,32000,44100,48000 > "$OUT/rates.txt" 2>> "$OUT/rocprof.err"
      summarise "$OUT/prof" "$OUT/rates_kernel_stats.txt"
      python tools/bench_rates.py --cuts 4000 > "$OUT/rates_all.txt" 2>&1
      python tools/bench_defaults.py > "$OUT/defaults.txt" 2>&1
      python tools/bench_librosa.py --cuts 4000 --steps 20 > "$OUT/librosa.txt" 2>&1
      python tools/bench_whisper.py --cuts 4000 --steps 20 > "$OUT/whisper.txt" 2>&1
      cat "$OUT/rates.txt" ;;
    power)
      tools/power_probe.sh "$OUT/power_probe.txt" ;;
    suite)
      timeout 1800 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.txt" 2>&1; echo "rc=$?"
      cp gpurun_out/parity_report.json "$OUT/parity_report.json" 2>/dev/null; tail -3 "$OUT/pytest_gpu.txt" ;;
    smoke)
      timeout 600 python __graft_entry__.py --smoke > "$OUT/smoke.txt" 2>&1; echo "rc=$?"; tail -8 "$OUT/smoke.txt" ;;
    host)
      bash tools/host_limits_probe.sh > "$OUT/host_limits.txt" 2>&1; head -12 "$OUT/host_limits.txt" ;;
    ring)
      bash tools/ring_sweep.sh "$OUT/ring" > /dev/null 2>&1; tail -4 "$OUT/ring/multi.txt" ;;
    mfcc_ab)
      tools/r6_mfcc_ab.sh "$OUT/mfcc_ab" > "$OUT/mfcc_ab.txt" 2>&1; cat "$OUT/mfcc_ab.txt" ;;
    *) echo "unknown section $section" ;;
  esac
done
