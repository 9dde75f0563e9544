#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -x -q -k "zmarch" 2>&1 | tail -8
for rep in 1 2; do
echo "== conv0x2 on K3w (--no-zmarch)"; python scripts/layer_bench.py --only conv0x2 --no-zmarch 2>/dev/null | grep conv0x2
echo "== conv0x2 on K3z0 (forced for every depth)"; python - <<'PY'
import subprocess, sys
PY
python scripts/layer_bench.py --only conv0x2 --zmarch0-min-depth 1 2>/dev/null | grep conv0x2
done > $O/r06_l_conv0_layers.txt 2>&1
cat $O/r06_l_conv0_layers.txt
