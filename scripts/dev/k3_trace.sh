#!/bin/bash
# builds libdmvs with -DDMVS_K3_TRACE (per-workgroup s_memtime phase stamps in the K3 conv / deconv kernels) into /tmp and
# runs scripts/dev/k3_trace.py against it.  EXTRA="-D..." adds experiment switches to the traced build.
set -e
cd dmvsnet_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DDMVS_DEV_BUILD -DDMVS_K3_TRACE $EXTRA"
/opt/rocm/bin/hipcc $F -c conv3d_mfma.hip -o /tmp/conv3d_mfma_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdmvs_k3trace.so $(ls *.o | grep -v conv3d_mfma.o) /tmp/conv3d_mfma_trace.o
cd ../..
DMVS_ALLOW_DEV_BUILD=1 DMVS_LIB=/tmp/libdmvs_k3trace.so python scripts/dev/k3_trace.py "$@"
