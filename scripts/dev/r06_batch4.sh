#!/bin/bash
export TMPDIR=/tmp
bash scripts/dev/layer_pmc.sh r06_d_k3z_s2conv2 s2.main.conv2 > /dev/null 2>&1
LB_ARGS=--no-zmarch bash scripts/dev/layer_pmc.sh r06_d_k3w_s2conv2 s2.main.conv2 > /dev/null 2>&1
cat gpurun_out/r06_d_k3z_s2conv2_layer_sq.txt; echo; cat gpurun_out/r06_d_k3w_s2conv2_layer_sq.txt
