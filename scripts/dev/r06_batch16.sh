#!/bin/bash
# start stagger on the K3 / prob / Winograd kernels: layers alone (scripts/layer_bench.py) and a same-box A/B of the whole forward
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r06_q_stagger_layers.txt
: > $O
ONLY="s2.main.conv11,s2.main.conv1,s2.main.conv3,s2.main.conv9,s2.main.prob,s3.main.prob,s3.main.conv11,s3.main.conv1,s2.main.conv0x2,s3.main.conv0x2,s3.main.conv2,feat.out3.fpn,feat.out2.q4"
for u in 0 8 32 128; do
  echo "== k3_stagger=$u prob_stagger=$u wino_stagger=$u" >> $O
  python scripts/layer_bench.py --only $ONLY --tune k3_stagger=$u --tune prob_stagger=$u --tune wino_stagger=$u 2>/dev/null | cut -c1-75 >> $O
done
cat $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "start_stagger" 2>&1 | tail -3
L=dmvsnet_amd/csrc/libdmvs_hip.so
bash scripts/dev/ab_bench.sh 4 base=$L k3s32=$L:k3_stagger=32 prob32=$L:prob_stagger=32 wino64=$L:wino_stagger=64 > gpurun_out/r06_q_ab_stagger.txt 2>&1
tail -6 gpurun_out/r06_q_ab_stagger.txt
