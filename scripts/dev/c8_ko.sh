#!/bin/bash
# Development: knock-out builds of the K3s row-sweep kernel (DMVS_C8_KO bit mask, see conv2d_c8.hip) next to the product
# library; select one at run time with DMVS_ALLOW_DEV_BUILD=1 DMVS_LIB=dmvsnet_amd/csrc/dev/libdmvs_c8ko<N>.so.   EXTRA="-D..." adds defines.
set -e
cd "$(dirname "$0")/../../dmvsnet_amd/csrc"
make -s
mkdir -p dev
OBJS=$(ls *.o | grep -v conv2d_c8.o)
for ko in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DDMVS_DEV_BUILD -DDMVS_C8_KO=$ko $EXTRA -c conv2d_c8.hip -o dev/conv2d_c8_ko$ko.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dev/libdmvs_c8ko$ko.so $OBJS dev/conv2d_c8_ko$ko.o
  echo built dev/libdmvs_c8ko$ko.so
done
