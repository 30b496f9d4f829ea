#!/bin/bash
# usage (GPU box): tools/r2_env.sh "ENV=.. ENV2=.." label [bench flags]   -- bench.py (4000 cuts) under an environment setting
envs="$1"; label="$2"; shift 2
env $envs python bench.py --cuts 4000 --steps 30 --warmup 3 --no-cpu-baseline --no-host-fed "$@" 2>gpurun_out/err_$label.txt | tail -1 | python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); p=r.get('parity') or {}
    print('%-14s %10.0f cuts/s  launch %.3f ms  frac %.4f  rel_l2 %.2e max_abs %.2e  %s' % ('$label', r['value'], r['roofline']['launch_ms'], r['roofline']['frac'], p.get('rel_l2_max',-1), p.get('max_abs_max',-1), r['config']['kernel']))
except Exception as e: print('$label', 'FAILED', e)
"
