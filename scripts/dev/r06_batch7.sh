#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -x -q -k "zmarch" 2>&1 | tail -3
for rep in 1 2; do
echo "== K3w (--no-zmarch)"; python scripts/layer_bench.py --only main.conv2,refine.conv2 --no-zmarch 2>/dev/null | grep conv2
echo "== K3z (product build)"; python scripts/layer_bench.py --only main.conv2,refine.conv2 2>/dev/null | grep conv2
done > $O/r06_g_layers_conv2.txt 2>&1
cat $O/r06_g_layers_conv2.txt
bash scripts/dev/layer_pmc.sh r06_g_k3z_s2conv2 s2.main.conv2 > /dev/null 2>&1
cat gpurun_out/r06_g_k3z_s2conv2_layer_sq.txt
