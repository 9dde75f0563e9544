// Issue rate of individual VALU instructions on gfx950 (wave64): cycles per wave-instruction per SIMD.
// Each kernel runs ITER x 32 independent instructions of one kind in 8 waves per SIMD on every CU.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/dev/valu_rate.bin scripts/dev/valu_rate.hip (the binary travels with
// gpurun, it is git-ignored); run: scripts/dev/valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define KERNEL(name, ASM)                                                                 \
    __global__ __launch_bounds__(256) void name(float* out, int iters) {                  \
        float a = threadIdx.x * 1.0f, b = 1.0001f, c = 0.5f, d = 3.0f;                    \
        int ia = threadIdx.x, ib = 3, ic = 5;                                             \
        for (int i = 0; i < iters; ++i) {                                                 \
            REP32(asm volatile(ASM : "+v"(a), "+v"(ia) : "v"(b), "v"(c), "v"(ib), "v"(ic), "v"(d));)         \
        }                                                                                 \
        if (a == 12345.f && ia == 77) out[0] = a;                                         \
    }

KERNEL(k_fma, "v_fma_f32 %0, %2, %3, %0")
KERNEL(k_mul, "v_mul_f32 %0, %2, %0")
KERNEL(k_add, "v_add_f32 %0, %2, %0")
KERNEL(k_max, "v_max_f32 %0, %2, %0")
KERNEL(k_med3f, "v_med3_f32 %0, %0, %2, %3")
KERNEL(k_med3i, "v_med3_i32 %1, %1, %4, %5")
KERNEL(k_addi, "v_add_u32 %1, %4, %1")
KERNEL(k_mul24, "v_mul_u32_u24 %1, %4, %1")
KERNEL(k_mad24, "v_mad_u32_u24 %1, %4, %1, %5")
KERNEL(k_mullo, "v_mul_lo_u32 %1, %4, %1")
KERNEL(k_lshladd, "v_lshl_add_u32 %1, %1, 2, %4")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %2, %0, vcc")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %2, %0")
KERNEL(k_floor, "v_floor_f32 %0, %0")
KERNEL(k_cvt, "v_cvt_i32_f32 %1, %0")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_dppmov, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_mov, "v_mov_b32 %0, %2")
KERNEL(k_fmac, "v_fmac_f32 %0, %2, %3")
KERNEL(k_sub, "v_sub_f32 %0, %0, %2")
KERNEL(k_and, "v_and_b32 %1, %4, %1")

template <typename K> void run(const char* name, K k, float* out) {
    const int iters = 2000, blocks = 256 * 8;  // 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<<<blocks, 256>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // wave-instructions per SIMD: blocks*4 waves / (256 CUs * 4 SIMDs) * iters * 32
    const double per_simd = (double)blocks * 4 / 1024 * iters * 32;
    const double cycles = ms * 1e-3 * 2.4e9;
    fflush(stdout); printf("%-10s %8.3f ms  %6.2f cycles per wave-instruction (at 2.4 GHz)\n", name, ms, cycles / per_simd);
}

int main() {
    float* out; hipMalloc(&out, 4);
#define R(n) run(#n, n, out)
    R(k_fma); R(k_fmac); R(k_mul); R(k_add); R(k_sub); R(k_max); R(k_med3f); R(k_med3i); R(k_addi); R(k_and); R(k_mul24); R(k_mad24);
    R(k_lshladd); R(k_cndmask); R(k_cmp); R(k_floor); R(k_cvt); R(k_rcp); R(k_dppmov); R(k_mov); R(k_mullo);
    return 0;
}
