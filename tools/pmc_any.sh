#!/bin/bash
# Counters for any command in separate rocprofv3 --pmc passes (run on the GPU box).
# usage: tools/pmc_any.sh <outdir> <command...>
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for group in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" ; do
  i=$((i+1))
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/pass$i" -o p -- "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass $i rc=$?"
done
python tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
