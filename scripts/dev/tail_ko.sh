#!/bin/bash
# knock-out builds of the fused tail kernel (DMVS_TAIL_KO bits) timed with scripts/dev/tail_bench.py
cd dmvsnet_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize"
for ko in "$@"; do
  /opt/rocm/bin/hipcc $F -DDMVS_TAIL_KO=$ko -c reg_tail.hip -o /tmp/reg_tail_ko.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdmvs_ko.so layout.o warp_corr.o depth_regress.o conv3d_direct.o conv3d_mfma.o /tmp/reg_tail_ko.o fusion.o
  echo "== DMVS_TAIL_KO=$ko"
  (cd ../.. && DMVS_LIB=/tmp/libdmvs_ko.so python scripts/dev/tail_bench.py 2>&1 | grep -E "^s|one branch" | awk '{print $1,$2,$3,$4,$5,$6}')
done
