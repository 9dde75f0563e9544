#!/bin/bash
# builds libdmvs with -DDMVS_K3R_TRACE (per-wave s_memtime phase sums in the register-stationary coarse kernel) into /tmp and
# runs scripts/dev/k3r_trace.py against it.  EXTRA="-D..." adds experiment switches to the traced build.
set -e
cd dmvsnet_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DDMVS_DEV_BUILD -DDMVS_K3R_TRACE $EXTRA"
/opt/rocm/bin/hipcc $F -c conv3d_coarse.hip -o /tmp/conv3d_coarse_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdmvs_k3rtrace.so $(ls *.o | grep -v conv3d_coarse.o) /tmp/conv3d_coarse_trace.o
cd ../..
DMVS_ALLOW_DEV_BUILD=1 DMVS_LIB=/tmp/libdmvs_k3rtrace.so python scripts/dev/k3r_trace.py "$@"
