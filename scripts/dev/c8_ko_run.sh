#!/bin/bash
# run on the GPU box: times FeatureNet conv0.0 / conv0.1 with the product library and every dev/libdmvs_c8ko*.so present
for lib in "" $(ls dmvsnet_amd/csrc/dev/libdmvs_c8ko*.so 2>/dev/null); do
  echo "== ${lib:-product}"
  DMVS_LIB=${lib:+$PWD/$lib} python scripts/layer_bench.py --only feat.conv0 --reps ${REPS:-50} 2>&1 | grep "feat.conv0" | awk '{print $1, $4}'
done
