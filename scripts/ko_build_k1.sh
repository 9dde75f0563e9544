#!/bin/bash
# Development: knock-out variants of the K1 LDS kernel (DMVS_KW bit mask, see warp_corr.hip); DMVS_LIB selects one.
set -e
cd "$(dirname "$0")/../dmvsnet_amd/csrc"
make -s
mkdir -p dev
for ko in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DDMVS_KW=$ko -c warp_corr.hip -o dev/warp_kw$ko.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dev/libdmvs_kw$ko.so layout.o dev/warp_kw$ko.o depth_regress.o conv3d_direct.o fusion.o conv3d_mfma.o
  echo built dev/libdmvs_kw$ko.so
done
rm -f dev/*.o
