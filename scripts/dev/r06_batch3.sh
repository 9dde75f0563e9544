#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -x -q -k "zmarch" > $O/r06_c_pytest_zmarch.txt 2>&1
tail -15 $O/r06_c_pytest_zmarch.txt
for rep in 1 2; do
echo "== K3w (--no-zmarch)"; python scripts/layer_bench.py --only conv2 --no-zmarch 2>/dev/null | grep conv2
echo "== K3z ring 2 (product build)"; python scripts/layer_bench.py --only conv2 2>/dev/null | grep conv2
echo "== K3z ring 3"; DMVS_ALLOW_DEV_BUILD=1 DMVS_LIB=dmvsnet_amd/csrc/dev/libdmvs_z3.so python scripts/layer_bench.py --only conv2 2>/dev/null | grep conv2
done > $O/r06_c_layers_conv2.txt 2>&1
cat $O/r06_c_layers_conv2.txt
for zs in 2 4 8 16; do echo "== K3z ring 2 zs=$zs"; python scripts/layer_bench.py --only main.conv2 --tune k3z_zs=$zs 2>/dev/null | grep conv2; done > $O/r06_c_layers_conv2_zs.txt 2>&1
cat $O/r06_c_layers_conv2_zs.txt
