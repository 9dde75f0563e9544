#!/bin/bash
# Development: knock-out builds of the Winograd conv kernel (DMVS_WKO bit mask, see conv3d_wino.hip) next to the product
# library; select one at run time with DMVS_ALLOW_DEV_BUILD=1 DMVS_LIB=dmvsnet_amd/csrc/dev/libdmvs_wko<N>.so.
set -e
cd "$(dirname "$0")/../../dmvsnet_amd/csrc"
make -s
mkdir -p dev
for ko in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DDMVS_DEV_BUILD -DDMVS_WKO=$ko $EXTRA -c conv3d_wino.hip -o dev/conv3d_wino_ko$ko.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dev/libdmvs_wko$ko.so $(ls *.o | grep -v conv3d_wino.o) dev/conv3d_wino_ko$ko.o
  echo built dev/libdmvs_wko$ko.so
done
