// Development micro-benchmark: do VALU instructions of one wave issue under the fp32 MFMAs of ANOTHER wave on the same SIMD?
// A 512-thread workgroup = 8 waves = 2 per SIMD; waves 0-3 run `nm` MFMAs (16x16x4 f32 or 32x32x2 f32), waves 4-7 run `nv`
// dependent-free v_add_f32.  Prints the time of each alone and of both together, one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 scripts/dev/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float acc4 __attribute__((ext_vector_type(4)));
typedef float acc16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: 16x16x4 f32, 1: 32x32x2 f32, 2: 32x32x16 bf16, 3: 16x16x32 bf16 (r06: does the bf16 MFMA co-issue with VALU?)
__global__ __launch_bounds__(512) void k(float* out, int nm, int nv, int same_wave) {
    const int wave = threadIdx.x >> 6;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    float s = 0.f;
    const bool do_m = same_wave ? true : wave < 4, do_v = same_wave ? true : wave >= 4;
    if (same_wave == 2) {   // interleaved in ONE instruction stream: 4 VALU per MFMA
        acc4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = (acc4){0.f, 0.f, 0.f, 0.f};
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a + i;
        for (int it = 0; it < nm / 8; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
                asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(b));
            }
        for (int i = 0; i < 8; ++i) s += c[i][0] + v[i];
        if (s == 1234.5f) out[0] = s;
        return;
    }
    if (do_m && nm) {
        if (KIND == 0) {
            acc4 c[8];
            for (int i = 0; i < 8; ++i) c[i] = (acc4){0.f, 0.f, 0.f, 0.f};
            for (int it = 0; it < nm / 8; ++it)
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
            for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
        } else if (KIND == 1) {
            acc16 c[4];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
            for (int it = 0; it < nm / 4; ++it)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
            for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][5];
        } else if (KIND == 2) {
            bf8 x, y;
            for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(a + j); y[j] = (__bf16)(b - j); }
            acc16 c[4];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
            for (int it = 0; it < nm / 4; ++it)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c[i], 0, 0, 0);
            for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][5];
        } else {
            bf8 x, y;
            for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(a + j); y[j] = (__bf16)(b - j); }
            acc4 c[8];
            for (int i = 0; i < 8; ++i) c[i] = (acc4){0.f, 0.f, 0.f, 0.f};
            for (int it = 0; it < nm / 8; ++it)
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c[i], 0, 0, 0);
            for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
        }
    }
    if (do_v && nv) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a + i;
        for (int it = 0; it < nv / 8; ++it)
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(b));
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    if (s == 1234.5f) out[0] = s;
}

template <int KIND>
float run(float* out, int nm, int nv, int same) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, 512>>>(out, nm, nv, same);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<256, 512>>>(out, nm, nv, same);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; hipMalloc(&out, 4);
    const int nm = 80000;
    for (int ratio : {1, 2, 4, 8}) {
        const int nv = nm * ratio;
        printf("16x16x4: %d MFMA (waves 0-3) | %d v_add (waves 4-7): mfma %.3f ms  valu %.3f ms  both %.3f ms\n", nm, nv,
               run<0>(out, nm, 0, 0), run<0>(out, 0, nv, 0), run<0>(out, nm, nv, 0));
        printf("32x32x2: %d MFMA (waves 0-3) | %d v_add (waves 4-7): mfma %.3f ms  valu %.3f ms  both %.3f ms\n", nm / 2, nv,
               run<1>(out, nm / 2, 0, 0), run<1>(out, 0, nv, 0), run<1>(out, nm / 2, nv, 0));
    }
    // r06: the same question for the bf16 MFMAs -- 32x32x16 (8 passes, 32768 FLOP) and 16x16x32 (4 passes,
    // 16384 FLOP).  If `both` = max(mfma, valu) the matrix pipe runs beside the VALU for these opcodes; if it is the sum, as for fp32, a
    // split-bf16 operand path cannot hide its own split / transform VALU work either.
    for (int ratio : {1, 2, 4, 8}) {
        const int nv = nm * ratio;
        printf("32x32x16 bf16: %d MFMA (waves 0-3) | %d v_add (waves 4-7): mfma %.3f ms  valu %.3f ms  both %.3f ms\n", nm / 2, nv,
               run<2>(out, nm / 2, 0, 0), run<2>(out, 0, nv, 0), run<2>(out, nm / 2, nv, 0));
        printf("16x16x32 bf16: %d MFMA (waves 0-3) | %d v_add (waves 4-7): mfma %.3f ms  valu %.3f ms  both %.3f ms\n", nm, nv,
               run<3>(out, nm, 0, 0), run<3>(out, 0, nv, 0), run<3>(out, nm, nv, 0));
    }
    printf("same wave, 16x16x4 with 4 v_add after each MFMA (all 8 waves): %.3f ms;  MFMA only all 8 waves: %.3f ms\n",
           run<0>(out, nm, 0, 2), run<0>(out, nm, 0, 1));
    return 0;
}
