#!/bin/bash
# VERDICT r3 task 4(ii): which phase of fft512c owns its LDS bank conflicts (23 % of the LDS-active cycles for two rounds)?
# Experiment builds with ONE phase's LDS accesses removed (wrong results, valid counters):
#   var_abl_ex = no exchange, var_abl_pw = no power-row writes, var_abl_ma / var_abl_mb = no A / B operand reads of the mel phase.
# Build (CPU): python -c "from lhotse_amd import build; build.build(extra_flags=['-DHIPFEAT_ABL_NO_EXCHANGE'], output='lhotse_amd/_lib/var_abl_ex.so')" etc.
set -u
OUT=gpurun_out/${1:-r4_lds}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python bench.py --cuts 2000 --steps 3 --warmup 1 --no-cpu-baseline --no-host-fed --no-parity --no-extra"
for v in base abl_ex abl_pw abl_ma abl_mb; do
  lib=lhotse_amd/_lib/var_$v.so; [ $v = base ] && lib=lhotse_amd/_lib/libhipfeat.so
  HIPFEAT_LIB=$lib rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$OUT/$v/pass1" -o p -- $CMD > "$OUT/$v.log" 2>&1
  python tools/pmc_summary.py "$OUT/$v" > "$OUT/$v.txt" 2>&1
  echo "== $v"; grep -A12 "fft512c" "$OUT/$v.txt" | grep "median" 
  rm -rf "$OUT/$v"
done
