#!/bin/bash
# SQ counters of the K1 kernels alone (scripts/k1_bench.py), two PMC passes; output: gpurun_out/$1_k1_sq.txt
TAG=${1:?tag}
export TMPDIR=/tmp
R=$PWD
: > $R/gpurun_out/${TAG}_k1_sq.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); rm -rf /tmp/k1pmc$i
  (cd /tmp && rocprofv3 --pmc $set -d /tmp/k1pmc$i -o p --output-format csv -- python $R/scripts/k1_bench.py --reps 3 > /tmp/k1pmc$i.log 2>&1)
  csv=$(find /tmp/k1pmc$i -name '*counter_collection.csv' | head -1)
  [ -n "$csv" ] && python $R/scripts/pmc_ours.py "$csv" >> $R/gpurun_out/${TAG}_k1_sq.txt
done
cat $R/gpurun_out/${TAG}_k1_sq.txt
