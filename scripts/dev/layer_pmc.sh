#!/bin/bash
# SQ / GRBM counters of chosen layers in isolation (scripts/layer_bench.py --only ...), separate PMC passes (never combined
# with tracing).  usage: layer_pmc.sh TAG "s2.main.conv1,s2.main.conv11" -> gpurun_out/${TAG}_layer_sq.txt
TAG=${1:?tag}; ONLY=${2:-s2.main.conv1}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/${TAG}_layer_sq.txt
: > $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/lpmc$i
  (cd /tmp && rocprofv3 --pmc $set -d /tmp/lpmc$i -o p --output-format csv -- python $R/scripts/layer_bench.py --only $ONLY --reps 2 $LB_ARGS > /tmp/lpmc$i.log 2>&1)
  csv=$(find /tmp/lpmc$i -name '*counter_collection.csv' | head -1)
  if [ -n "$csv" ]; then python $R/scripts/pmc_ours.py "$csv" >> $OUT; else echo "set $i: no csv"; tail -3 /tmp/lpmc$i.log; fi
done
grep "ms " /tmp/lpmc1.log | head -20 >> $OUT
cat $OUT
