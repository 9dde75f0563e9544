// Shared helpers for libdmvs_hip.so (gfx950 only; no CUDA / multi-backend paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dmvs.h"

#define DMVS_LAUNCH_CHECK()                         \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        return e__ == hipSuccess ? 0 : (int)e__;    \
    } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// > 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize once per (device, kernel) -- and again when
// a later launch of the same kernel asks for more.  One process-wide, mutex-guarded table (conv3d_mfma.hip).
int dmvs_ensure_dynamic_lds(const void* kernel, size_t lds_bytes);

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// XCD-aware tile order.  The dispatcher deals workgroups round-robin to the 8 XCDs (workgroup id % 8), each with
// its own 4 MB L2; with the natural blockIdx order the 6-26 neighbours whose halo a tile shares all sit on OTHER
// XCDs and every L2 pulls the halo over the fabric again (measured: the tile loads of a 3x3x3 layer run 2-2.6x
// faster with this remap).  Launch a 1-D grid of xcd_grid(n) workgroups; XCD k then walks the k-th contiguous
// eighth of the tile list.  Tile list order: x fastest, then z, then y when `z_fast` (3D kernels: the depth halo
// is the largest one and a (x, z) slab of tiles fits the L2), else x, y, z (per-slice 2D kernels).
static inline unsigned xcd_grid(int ntiles) { return 8u * (unsigned)((ntiles + 7) / 8); }

#ifdef __HIPCC__
__device__ __forceinline__ bool xcd_tile(int nx, int ny, int nz, bool z_fast, int& bx, int& by, int& bz) {
    const int n = nx * ny * nz, per = (n + 7) >> 3;
    const int id = blockIdx.x, t = (id & 7) * per + (id >> 3);
    if (t >= n) return false;
    bx = t % nx;
    const int r = t / nx;
    if (z_fast) { bz = r % nz; by = r / nz; } else { by = r % ny; bz = r / ny; }
    return true;
}
#endif
