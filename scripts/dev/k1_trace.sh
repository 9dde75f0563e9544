#!/bin/bash
# builds libdmvs with -DDMVS_Q4_TRACE into /tmp and runs scripts/dev/k1_trace.py against it
set -e
cd dmvsnet_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DDMVS_DEV_BUILD -DDMVS_Q4_TRACE"
/opt/rocm/bin/hipcc $F -c warp_corr.hip -o /tmp/warp_corr_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdmvs_trace.so $(ls *.o | grep -v warp_corr.o) /tmp/warp_corr_trace.o
cd ../..
if [ "$1" = modes ]; then shift; DMVS_ALLOW_DEV_BUILD=1 DMVS_LIB=/tmp/libdmvs_trace.so python scripts/dev/k1_modes.py "$@"; else DMVS_ALLOW_DEV_BUILD=1 DMVS_LIB=/tmp/libdmvs_trace.so python scripts/dev/k1_trace.py "$@"; fi
