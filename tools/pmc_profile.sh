#!/bin/bash
# Collect hardware counters for the bench workload in separate rocprofv3 --pmc passes
# (run on the GPU box).  usage: tools/pmc_profile.sh <outdir> [bench args...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--cuts 4000 --steps 3 --warmup 1 --no-cpu-baseline $*"
i=0
for group in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
  "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/pass$i" -o p -- python bench.py $ARGS > "$OUT/pass$i.log" 2>&1
  echo "pass $i rc=$?"
done
python tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
