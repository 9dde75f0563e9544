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

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
