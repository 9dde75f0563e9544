#!/bin/bash
# builds libdmvs with -DDMVS_K3R_TRACE (per-wave s_memtime phase sums in the register-stationary coarse kernel) into /tmp and
# runs scripts/dev/k3r_trace.py against it.  EXTRA="-D..." adds experiment switches to the traced build.
set -e
cd dmvsnet_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DDMVS_K3R_TRACE $EXTRA"
/opt/rocm/bin/hipcc $F -c conv3d_coarse.hip -o /tmp/conv3d_coarse_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdmvs_k3rtrace.so layout.o warp_corr.o depth_regress.o conv3d_direct.o conv3d_mfma.o conv3d_wino.o conv2d_c8.o /tmp/conv3d_coarse_trace.o fusion.o
cd ../..
DMVS_LIB=/tmp/libdmvs_k3rtrace.so python scripts/dev/k3r_trace.py "$@"
