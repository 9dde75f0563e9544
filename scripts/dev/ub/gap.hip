// dev micro-benchmark: the cost of a kernel boundary.  N launches of a tiny kernel back to back on ONE stream (each waits for the
// previous: in-queue barrier + cache write-back / invalidate), alternating on TWO streams (independent), and the same with a
// kernel that leaves `mb` megabytes of dirty lines in the L2s.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(float* o) { if (threadIdx.x == 0) o[blockIdx.x] = 1.0f; }
__global__ __launch_bounds__(256) void dirty(float4* o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = float4{1, 2, 3, 4};
}
template <typename F> static double per_launch_us(int n, F&& launch) {
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) launch(i);
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main() {
    float* d; hipMalloc(&d, 256 << 20);
    hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
    const int n = 2000;
    per_launch_us(100, [&](int) { tiny<<<1, 64, 0, s0>>>(d); });
    printf("tiny kernel, 1 workgroup : one stream %.2f us per launch, two streams alternating %.2f\n",
           per_launch_us(n, [&](int) { tiny<<<1, 64, 0, s0>>>(d); }), per_launch_us(n, [&](int i) { tiny<<<1, 64, 0, (i & 1) ? s1 : s0>>>(d); }));
    printf("tiny kernel, 256 workgroups: one stream %.2f us per launch, two streams alternating %.2f\n",
           per_launch_us(n, [&](int) { tiny<<<256, 64, 0, s0>>>(d); }), per_launch_us(n, [&](int i) { tiny<<<256, 64, 0, (i & 1) ? s1 : s0>>>(d); }));
    for (int mb : {1, 8, 32, 128}) {
        const size_t nv = (size_t)mb << 16;   // float4s
        printf("kernel writing %3d MB     : one stream %.2f us per launch, two streams alternating %.2f  (the write alone at 5 TB/s: %.1f us)\n", mb,
               per_launch_us(500, [&](int) { dirty<<<1024, 256, 0, s0>>>((float4*)d, nv); }),
               per_launch_us(500, [&](int i) { dirty<<<1024, 256, 0, (i & 1) ? s1 : s0>>>((float4*)d + ((i & 1) ? nv : 0), nv); }), mb * 1.048576 / 5.0);
    }
    return 0;
}
