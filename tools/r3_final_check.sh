set -u
O=gpurun_out/r3e; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python tools/bench_rates.py --cuts 4000 > $O/rates.txt 2>&1; cut -c1-150 $O/rates.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d $O/prof -o rates -- python tools/bench_rates.py --cuts 4000 --rates 24000,48000 > /dev/null 2> $O/rocprof_rates.err
db=$(find $O/prof -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py "$db" > $O/rates_kernel_stats.txt 2>&1; rm -rf $O/prof; head -6 $O/rates_kernel_stats.txt | cut -c1-200
