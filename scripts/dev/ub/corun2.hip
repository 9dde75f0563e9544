// dev micro-benchmark: when do kernels of two plain HIP streams overlap?  A = pure fp32 MFMA with `ga` workgroups of 256
// threads, B = the same kernel with `gb` workgroups, both ~T ms alone; launched back to back from the host on two streams with
// NO cross-stream events; host-timed (hipDeviceSynchronize).  Serial = A + B, concurrent = max(A, B) when the chip has room.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef float float16_t __attribute__((ext_vector_type(16)));
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
static double run(hipStream_t s0, hipStream_t s1, int ga, int gb, float* d, int iters) {
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    if (ga) mfma_only<<<ga, 256, 0, s0>>>(d, iters);
    if (gb) mfma_only<<<gb, 256, 0, s1>>>(d + (1 << 20), iters);
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main(int argc, char** argv) {
    float* d; hipMalloc(&d, 1 << 24);
    hipStream_t s0, s1;
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    if (mode == 1) {   // different priorities
        int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
        printf("priority range %d .. %d\n", lo, hi);
        hipStreamCreateWithPriority(&s0, hipStreamNonBlocking, lo); hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi);
    } else if (mode == 2) {   // CU-masked streams, both with every CU
        uint32_t m[8]; for (int i = 0; i < 8; ++i) m[i] = 0xffffffffu;
        hipExtStreamCreateWithCUMask(&s0, 8, m); hipExtStreamCreateWithCUMask(&s1, 8, m);
    } else if (mode == 3) { hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); }
    else { hipStreamCreate(&s0); hipStreamCreate(&s1); }
    printf("mode %d\n", mode);
    const int iters = 3000;
    run(s0, s1, 64, 64, d, 100);
    for (int ga : {32, 256, 1024})
        for (int gb : {32, 256}) {
            const double a = run(s0, s1, ga, 0, d, iters), b = run(s0, s1, 0, gb, d, iters), ab = run(s0, s1, ga, gb, d, iters);
            const double same = [&] { hipDeviceSynchronize(); const auto t0 = std::chrono::steady_clock::now();
                mfma_only<<<ga, 256, 0, s0>>>(d, iters); mfma_only<<<gb, 256, 0, s0>>>(d + (1 << 20), iters); hipDeviceSynchronize();
                return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }();
            printf("A %4d wgs %.3f ms | B %4d wgs %.3f ms | two streams %.3f ms | one stream %.3f ms | sum %.3f max %.3f\n", ga, a, gb, b, ab, same, a + b, a > b ? a : b);
        }
    return 0;
}
