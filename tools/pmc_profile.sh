#!/bin/bash
# GPU box: collect PMC counters for the engine's kernels (separate passes, as
# the MI355X guide prescribes; --pmc never combined with trace domains other
# than --kernel-trace). Numerator and denominator of every ratio
# tools/pmc_rooflines.py forms sit in the same pass.
#   usage (from the repo root): tools/pmc_profile.sh OUTDIR -- SCRIPT.py [args]
set -u
REPO=$(pwd)
OUT=$REPO/$1; shift; shift
SCRIPT=$REPO/$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
cd /tmp
i=0
for PMC in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ATOMIC_RETURN SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE TCC_EA0_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/pass$i" -- python3 "$SCRIPT" "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass $i rc=$? ($PMC)"
done
cd "$OUT"
find . -name "*kernel_trace.csv" -delete
