// dev micro-benchmark: does HBM streaming slow the fp32 MFMA pipe (or vice versa) when both run on the chip at once?
// A = pure v_mfma_f32_32x32x2 (2 waves per SIMD, ~no memory), B = a 16-byte-per-lane copy.  Alone, then together on two streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_only(float* o, int iters) {
    float16_t a16[2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) a16[i][j] = 0;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
            for (int i = 0; i < 2; ++i) a16[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a16[i], 0, 0, 0);
    o[blockIdx.x * 256 + threadIdx.x] = a16[0][0] + a16[1][0];
}
__global__ __launch_bounds__(256) void copy16(const float4_t* in, float4_t* out, size_t n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
static float g_end_a, g_end_b;   // when each kernel finished, from the common start
static float timed(hipStream_t s0, hipStream_t s1, bool A, bool B, float* d, const float4_t* in, float4_t* out, size_t n, int iters, int reps) {
    hipEvent_t e0, e1, j, ea; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&j); hipEventCreate(&ea);
    hipDeviceSynchronize();
    hipEventRecord(e0, s0);
    hipStreamWaitEvent(s1, e0, 0);
    if (A) mfma_only<<<512, 256, 0, s0>>>(d, iters);
    hipEventRecord(ea, s0);
    if (B) copy16<<<2048, 256, 0, s1>>>(in, out, n, reps);
    hipEventRecord(j, s1);
    hipStreamWaitEvent(s0, j, 0);
    hipEventRecord(e1, s0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventElapsedTime(&g_end_a, e0, ea); hipEventElapsedTime(&g_end_b, e0, j);
    return ms;
}
int main(int argc, char** argv) {
    const size_t n = (size_t)1 << 26;   // 1 GiB of float4 in, 1 GiB out
    float4_t *in, *out; float* d;
    hipMalloc(&in, n * 16); hipMalloc(&out, n * 16); hipMalloc(&d, 1 << 22); hipMemset(in, 0, n * 16);
    hipStream_t s0, s1;
    if (argc > 1) {   // CU-masked streams: argv[1] = mask word of stream 0 (hex, repeated for the 8 XCDs), stream 1 gets the complement
        const uint32_t m = (uint32_t)strtoul(argv[1], nullptr, 16);
        uint32_t ma[8], mb[8];
        for (int i = 0; i < 8; ++i) { ma[i] = m; mb[i] = ~m; }
        printf("cu masks %08x / %08x: %d %d\n", m, ~m, (int)hipExtStreamCreateWithCUMask(&s0, 8, ma), (int)hipExtStreamCreateWithCUMask(&s1, 8, mb));
    } else { hipStreamCreate(&s0); hipStreamCreate(&s1); }
    const int iters = 6000;
    for (int w = 0; w < 2; ++w) { timed(s0, s1, true, false, d, in, out, n, 100, 1); timed(s0, s1, false, true, d, in, out, n, 100, 1); }
    for (int reps : {6, 12}) {
    for (int k = 0; k < 2; ++k) {
        const float a = timed(s0, s1, true, false, d, in, out, n, iters, reps);
        const float b = timed(s0, s1, false, true, d, in, out, n, iters, reps);
        const float ab = timed(s0, s1, true, true, d, in, out, n, iters, reps);
        printf("reps %d: MFMA alone %.3f ms (%.1f TFLOP/s)   copy alone %.3f ms (%.0f GB/s)   together %.3f ms (MFMA done at %.3f, copy done at %.3f)   (max %.3f, sum %.3f)\n", reps, a,
               512.0 * 4 * iters * 16 * 4096 / a * 1e-9, b, 2.0 * n * 16 * reps / b * 1e-6, ab, g_end_a, g_end_b, a > b ? a : b, a + b);
    }
    }
    return 0;
}
