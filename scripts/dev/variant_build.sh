#!/bin/bash
# Development: build a variant of ONE source file of the library with extra -D flags next to the product library:
#   scripts/dev/variant_build.sh NAME file.hip -DFLAG=V [-D...]   ->  dmvsnet_amd/csrc/dev/libdmvs_NAME.so
# select it at run time with DMVS_ALLOW_DEV_BUILD=1 DMVS_LIB=dmvsnet_amd/csrc/dev/libdmvs_NAME.so (scripts/dev/ab_bench.sh).
set -e
cd "$(dirname "$0")/../../dmvsnet_amd/csrc"
name=$1; src=$2; shift 2
make -s
mkdir -p dev
obj=${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -w -DDMVS_DEV_BUILD "$@" -c $src -o dev/${src%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dev/libdmvs_$name.so $(ls *.o | grep -v "^$obj$") dev/${src%.hip}_$name.o
echo built dmvsnet_amd/csrc/dev/libdmvs_$name.so
