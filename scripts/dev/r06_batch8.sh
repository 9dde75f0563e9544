#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -x -q -k "zmarch or coarse or wino" 2>&1 | tail -3
# same-box alternating A/B of the whole forward: K3z for conv2 (D >= 8) vs K3w
L=dmvsnet_amd/csrc/libdmvs_hip.so
bash scripts/dev/ab_bench.sh 5 k3z=$L k3w=$L@--no-zmarch > $O/r06_h_ab_k3z_vs_k3w.txt 2>&1
tail -4 $O/r06_h_ab_k3z_vs_k3w.txt
python scripts/layer_bench.py --only conv2 2>/dev/null | grep "conv2 " > $O/r06_h_conv2_layers.txt
python scripts/layer_bench.py --only conv2 --no-zmarch 2>/dev/null | grep "conv2 " >> $O/r06_h_conv2_layers.txt
cat $O/r06_h_conv2_layers.txt
# FeatureNet: SQ counters of its three worst launches alone
bash scripts/dev/layer_pmc.sh r06_h_feat feat.conv0.fused,feat.out3.fpn.q4,feat.conv1.1 > /dev/null 2>&1
cat $O/r06_h_feat_layer_sq.txt
