#!/bin/bash
# SQ / TCC counters of the q4 K1 kernel on the real bench inputs (scripts/dev/k1_q4.py --part real), separate PMC passes.
# usage: k1_q4_pmc.sh TAG [variant]   -> gpurun_out/${TAG}_k1_sq.txt
TAG=${1:?tag}; VAR=${2:-0}; PART=${3:-real}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/${TAG}_k1_sq.txt
: > $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/k1pmc$i
  (cd /tmp && rocprofv3 --pmc $set -d /tmp/k1pmc$i -o p --output-format csv -- python $R/scripts/dev/k1_q4.py c2 --no-old --q4 $VAR --part $PART --reps 2 > /tmp/k1pmc$i.log 2>&1)
  csv=$(find /tmp/k1pmc$i -name '*counter_collection.csv' | head -1)
  [ -n "$csv" ] && python $R/scripts/pmc_ours.py "$csv" >> $OUT
done
cat $OUT
