#!/bin/bash
# usage (GPU box): tools/r2_variants.sh name1 name2 ...  -- bench + in-run oracle parity of experiment builds (tools/variants.py)
for v in "$@"; do
  lib=$PWD/lhotse_amd/_lib/var_$v.so
  [ "$v" = base ] && lib=$PWD/lhotse_amd/_lib/libhipfeat.so
  HIPFEAT_LIB=$lib python bench.py --cuts 4000 --steps 30 --warmup 3 --no-cpu-baseline --no-host-fed $R2_FLAGS 2>gpurun_out/err_$v.txt | tail -1 | python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); p=r.get('parity') or {}
    print('%-14s %10.0f cuts/s  launch %.3f ms  frac %.4f  rel_l2 %.2e max_abs %.2e  %s' % ('$v', r['value'], r['roofline']['launch_ms'], r['roofline']['frac'], p.get('rel_l2_max',-1), p.get('max_abs_max',-1), r['config']['kernel']))
except Exception as e: print('$v', 'FAILED', e)
"
done
