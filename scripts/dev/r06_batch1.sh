#!/bin/bash
# r06 batch 1: baseline of the box + three cheap experiments (VERDICT r05 items 3, 4, and the conv2 LDS-stage A/B)
export TMPDIR=/tmp
O=gpurun_out
python bench.py --no-cpu-baseline --no-aten-gpu-baseline --no-live-traffic --no-full-outputs > $O/r06_a_bench_quick.json 2> $O/r06_a_bench_quick.err
./scripts/dev/mfma_valu_overlap.bin > $O/r06_a_mfma_valu_overlap_bf16.txt 2>&1
python scripts/layer_bench.py --only conv2,conv0x2,feat.conv1.1,feat.conv2.1,feat.conv2.2,feat.conv0.fused 2>/dev/null > $O/r06_a_layers_base.txt
python scripts/layer_bench.py --only main.conv2,refine.conv2 --tune wino_stages=2 2>/dev/null > $O/r06_a_layers_conv2_two_stages.txt
python scripts/layer_bench.py --feat-coarse --only feat.conv2.1,feat.conv2.2 2>/dev/null > $O/r06_a_layers_feat_k3r.txt
tail -n 30 $O/r06_a_*.txt
python -c "import json;d=json.load(open('$O/r06_a_bench_quick.json'));print(d['value'],d['ms_per_step'],d['ms_per_stage'])"
