// dev micro-benchmark: v_mfma_f32_4x4x1_16b_f32 operand layout, wave_shr / wave_shl DPP on gfx950, issue rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
__global__ void layout(const float* a, const float* b, float* o, int* sh) {
    float4_t acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) o[threadIdx.x * 4 + i] = acc[i];
    int v = threadIdx.x + 100;
    sh[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);        // wave_shr:1
    sh[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
}
template <int NACC>
__global__ __launch_bounds__(256) void rate(float* o, int iters) {
    float4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    o[blockIdx.x * 256 + threadIdx.x] = s;
}
// shader clock under load: s_memtime counts shader cycles, s_memrealtime a constant 100 MHz
template <int KIND>   // 0: 4x4x1 fp32, 1: 16x16x4 fp32, 2: 32x32x2 fp32, 3: v_pk_fma_f32
__global__ __launch_bounds__(256) void clk(float* o, unsigned long long* t, int iters) {
    typedef float float16_t __attribute__((ext_vector_type(16)));
    float4_t a4[4]; float16_t a16[2]; float2_t pk[8];
    for (int i = 0; i < 4; ++i) a4[i] = {0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) a16[i][j] = 0;
    for (int i = 0; i < 8; ++i) pk[i] = {0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float2_t pa = {a, b}, pb = {b, a};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) { for (int i = 0; i < 4; ++i) a4[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, a4[i], 0, 0, 0); }
            if (KIND == 1) { for (int i = 0; i < 4; ++i) a4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a4[i], 0, 0, 0); }
            if (KIND == 2) { for (int i = 0; i < 2; ++i) a16[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a16[i], 0, 0, 0); }
            if (KIND == 3) { for (int i = 0; i < 8; ++i) pk[i] = __builtin_elementwise_fma(pa, pb, pk[i]); }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += a4[i][0];
    for (int i = 0; i < 2; ++i) s += a16[i][0];
    for (int i = 0; i < 8; ++i) s += pk[i].x + pk[i].y;
    o[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = c1 - c0; t[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int KIND> void run_clk(float* d, const char* name, int per_iter, double flop_per_instr) {
    unsigned long long* t; hipMalloc(&t, 1 << 16);
    const int blocks = 512, iters = 20000;
    clk<KIND><<<blocks, 256>>>(d, t, 100);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    clk<KIND><<<blocks, 256>>>(d, t, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2 * 512]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0; for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    const double ghz = cyc / rt * 0.1, n_wave = (double)iters * 8 * per_iter;
    printf("%-12s shader clock %.3f GHz under load; %.2f clk per instruction per SIMD (2 waves); %.1f TFLOP/s (%.3f ms)\n", name, ghz,
           cyc / blocks / n_wave / 2.0, (double)blocks * 4 * n_wave * flop_per_instr / ms * 1e-9, ms);
}
template <int NACC> void run_rate(float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 2;
    rate<NACC><<<blocks, 256>>>(d, 10);
    hipEventRecord(e0);
    rate<NACC><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 * iters * 8 * NACC;   // MFMAs issued (per wave, all waves)
    // 1024 SIMDs, 2 waves each: per-SIMD MFMAs = n / 1024
    printf("NACC=%d: %.3f ms, %.1f clk per MFMA per SIMD at 2.4 GHz, %.1f TFLOP/s\n", NACC, ms, ms * 1e-3 * 2.4e9 / (n / 1024), n * 512 / ms * 1e-9);
}
int main() {
    std::vector<float> a(64), b(64), o(256); std::vector<int> sh(128);
    for (int i = 0; i < 64; ++i) { a[i] = i + 1; b[i] = 1000.f * (i + 1); }
    float *da, *db, *dout; int* ds;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 1 << 22); hipMalloc(&ds, 512);
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    layout<<<1, 64>>>(da, db, dout, ds);
    hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost); hipMemcpy(sh.data(), ds, 512, hipMemcpyDeviceToHost);
    // expected: lane l, vgpr r: D[block l/4][row r][col l%4] = a[4*(l/4) + r] * b[l]
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (o[l * 4 + r] != a[4 * (l / 4) + r] * b[l]) ++bad;
    printf("layout mismatches vs D[l][r] = a[4*(l/4)+r] * b[l]: %d\n", bad);
    if (bad) for (int l = 0; l < 8; ++l) printf("  lane %d: %g %g %g %g\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    printf("wave_shr:1 lanes 0,1,15,16,17,31,32,63: %d %d %d %d %d %d %d %d\n", sh[0], sh[1], sh[15], sh[16], sh[17], sh[31], sh[32], sh[63]);
    printf("wave_shl:1 lanes 0,1,15,16,17,31,32,63: %d %d %d %d %d %d %d %d\n", sh[64], sh[65], sh[79], sh[80], sh[81], sh[95], sh[96], sh[127]);
    run_rate<1>(dout); run_rate<2>(dout); run_rate<4>(dout);
    run_clk<0>(dout, "mfma 4x4x1", 4, 512); run_clk<1>(dout, "mfma 16x16x4", 4, 2048); run_clk<2>(dout, "mfma 32x32x2", 2, 4096);
    run_clk<3>(dout, "v_pk_fma_f32", 8, 256);
    return 0;
}
