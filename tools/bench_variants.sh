#!/bin/bash
# usage: tools/bench_variants.sh name1 name2 ...   (run on the GPU box; variants built by tools/variants.py)
for v in "$@"; do
  HIPFEAT_LIB=$PWD/lhotse_amd/_lib/var_$v.so python bench.py --cuts 4000 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); print('%-14s %10.0f cuts/s  launch %.3f ms  frac %.4f  %s' % ('$v', r['value'], r['roofline']['launch_ms'], r['roofline']['frac'], r['config']['kernel']))
except Exception as e: print('$v', 'FAILED', e)
"
done
