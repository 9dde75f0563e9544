#!/bin/bash
export TMPDIR=/tmp
for ko in 0 1 2 4 8 16 3 7; do
echo "== K3z ring 3, ZKO=$ko (1 no tile loads, 2 no stores, 4 no MFMAs (2 VALU each instead), 8 no barrier, 16 no finish)"
DMVS_ALLOW_DEV_BUILD=1 DMVS_LIB=dmvsnet_amd/csrc/dev/libdmvs_z3ko$ko.so python scripts/layer_bench.py --only s1.main.conv2,s2.main.conv2,s3.main.conv2 2>/dev/null | grep conv2
done > gpurun_out/r06_f_k3z_knockouts.txt 2>&1
cat gpurun_out/r06_f_k3z_knockouts.txt
