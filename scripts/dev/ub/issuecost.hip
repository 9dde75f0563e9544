// Development micro-benchmark (r05): what does a tile load cost the wave that issues it, next to fp32 MFMAs?
// 256 workgroups x 8 waves (2 per SIMD, like K3r); every wave runs ITER iterations of NM independent-accumulator
// v_mfma_f32_16x16x4_f32 plus L loads of 1 KiB per wave from an L2-resident buffer, staged into LDS
//   mode 0: no loads (the MFMA stream alone)
//   mode 1: LDS-direct   buffer_load_dwordx4 ... lds      (what K3 / K3w / K3r stage tiles with), counted vmcnt one iteration behind
//   mode 2: registers    buffer_load_dwordx4 -> VGPRs at the top of the iteration, ds_write_b128 at its end (T14 issue-early / write-late)
//   mode 3: registers, loads only (no ds_write): the issue cost of the load alone
// loads are spread between the MFMAs (one after every NM / L MFMAs).  Prints cycles per iteration and the extra cycles per load.
//   hipcc --offload-arch=gfx950 -O3 scripts/dev/ub/issuecost.hip -o scripts/dev/ub/issuecost && scripts/dev/ub/issuecost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE, int L, int NM>
__global__ __launch_bounds__(512, 2) void kern(const float* buf, unsigned foot, int iters, float* sink, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][8 waves][L] KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, (short)0, foot, 0x00020000);
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = (float)lane, b = 1.0f + (float)wave;
    unsigned off = ((blockIdx.x * 8 + wave) * 37u * 1024u) % (foot - 64u * 1024u) & ~1023u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    u4 regs[L > 0 ? L : 1];
    for (int it = 0; it < iters; ++it) {
        float* dst = smem + ((it & 1) * 8 + wave) * (L > 0 ? L : 1) * 256;
        int li = 0;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
            if (L > 0 && (m % (NM / (L > 0 ? L : 1))) == 0 && li < L) {
                const unsigned o = off + (unsigned)li * 1024u + lane * 16u;
                if (MODE == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + li * 256), 16, o, 0, 0, 0);
                else if (MODE >= 2) regs[li] = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
                ++li;
            }
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");   // the previous iteration's loads have landed
        if (MODE == 2) {
#pragma unroll
            for (int l = 0; l < L; ++l) *reinterpret_cast<u4*>(dst + l * 256 + lane * 4) = regs[l];
        }
        if (MODE == 3) {
#pragma unroll
            for (int l = 0; l < L; ++l) a += __builtin_bit_cast(float, regs[l].x) * 1e-30f;
        }
        off += (unsigned)L * 1024u;
        if (off + 64u * 1024u > foot) off = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (acc[0].x + acc[1].x + acc[2].x + acc[3].x + smem[lane] == 1234.5f) sink[0] = 1.f;
}

template <int MODE, int L, int NM> double run(const float* buf, unsigned foot, float* sink, unsigned long long* cyc, double base) {
    const int iters = 2000;
    const size_t lds = 2 * 8 * (L > 0 ? L : 1) * 1024;
    auto k = kern<MODE, L, NM>;
    if (lds > 65536) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k<<<256, 512, lds>>>(buf, foot, iters, sink, cyc);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    k<<<256, 512, lds>>>(buf, foot, iters, sink, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2048]; CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < 2048; ++i) s += (double)h[i];
    const double per = s / 2048 / iters;
    static const char* names[] = {"no loads", "LDS-direct", "regs + ds_write_b128", "regs, load only"};
    printf("%-22s L=%2d NM=%2d: %8.1f ticks / iteration (%.3f ms; pure MFMA pipe time 2 waves x %d x 32 = %d clk)", names[MODE], L, NM, per, ms, NM, 2 * NM * 32);
    if (base > 0 && L > 0) printf("  -> +%.0f ticks per load", (per - base) / L);
    printf("\n");
    return per;
}

int main() {
    const unsigned foot = 2u << 20;   // L2-resident
    float *buf, *sink; unsigned long long* cyc;
    CK(hipMalloc(&buf, foot)); CK(hipMemset(buf, 0, foot)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&cyc, 2048 * 8));
    {
        const double b = run<0, 0, 24>(buf, foot, sink, cyc, 0);
        run<1, 4, 24>(buf, foot, sink, cyc, b); run<2, 4, 24>(buf, foot, sink, cyc, b); run<3, 4, 24>(buf, foot, sink, cyc, b);
        run<1, 8, 24>(buf, foot, sink, cyc, b); run<2, 8, 24>(buf, foot, sink, cyc, b); run<3, 8, 24>(buf, foot, sink, cyc, b);
    }
    {
        const double b = run<0, 0, 48>(buf, foot, sink, cyc, 0);
        run<1, 4, 48>(buf, foot, sink, cyc, b); run<2, 4, 48>(buf, foot, sink, cyc, b); run<3, 4, 48>(buf, foot, sink, cyc, b);
        run<1, 8, 48>(buf, foot, sink, cyc, b); run<2, 8, 48>(buf, foot, sink, cyc, b); run<3, 8, 48>(buf, foot, sink, cyc, b);
    }
    return 0;
}
