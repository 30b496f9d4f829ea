#!/bin/bash
# on-the-fly, second look: the 2-D prep grid, tables staged vs in the kernel arguments, K mini-batches per launch pair
set -u
OUT=gpurun_out/${1:-r4_run3}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_minibatch.py -x -q 2>&1 | tail -3 | tee "$OUT/pytest_minibatch.txt"
F="--config onthefly --no-cpu-baseline --no-host-fed --no-parity --steps 40"
run() { tag=$1; shift; env "$@" 2>/dev/null | python -c "
import sys,json
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); x=r.get('extra',{}).get('routes',{})
print('$tag', r['value'], r['roofline']['launch_ms'], r['roofline']['frac'], {k:(v['cuts_per_s'],v['host_us_per_minibatch']) for k,v in x.items() if k!='what'})" | tee -a "$OUT/ab.txt"; }
run k1 python bench.py $F
run k1_staged HIPFEAT_MB_NO_INLINE=1 python bench.py $F
run k1_devkernarg HIP_FORCE_DEV_KERNARG=1 python bench.py $F
run k2 python bench.py $F --prefetch 2
run k4 python bench.py $F --prefetch 4
run k4_1stream python bench.py $F --prefetch 4 --streams 1
run k8 python bench.py $F --prefetch 8
for k in 1 4; do
rocprofv3 --kernel-trace --stats -d "$OUT/prof$k" -o otf -- python bench.py $F --steps 20 --streams 1 --prefetch $k > /dev/null 2> "$OUT/rocprof$k.err"
db=$(find "$OUT/prof$k" -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" > "$OUT/otf_kernel_stats_k$k.txt" 2>&1
rm -rf "$OUT/prof$k"
grep -A3 "dispatches=" "$OUT/otf_kernel_stats_k$k.txt" | grep -v "^--" | head -12; tail -2 "$OUT/otf_kernel_stats_k$k.txt"
done
python bench.py --config onthefly --no-cpu-baseline --prefetch 4 > "$OUT/bench_onthefly_k4.json" 2> "$OUT/bench_onthefly_k4.err"; tail -c 300 "$OUT/bench_onthefly_k4.json" | head -c 300
