// Development micro-benchmark: sustained fp32 MFMA rate (what "100 %" means for the K3 kernels at real clocks).
//   hipcc --offload-arch=gfx950 -O3 scripts/dev/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float acc4 __attribute__((ext_vector_type(4)));
typedef float acc16 __attribute__((ext_vector_type(16)));
template <int WPS>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    acc4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = (acc4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (s == 1234.5f) out[0] = s;
}
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    acc16 c[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][5];
    if (s == 1234.5f) out[0] = s;
}
int main() {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 1024}) {
        const int iters = 20000;
        k16<1><<<wgs, 256>>>(out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k16<1><<<wgs, 256>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)wgs * 4 * iters * 8 * 2048.0;
        printf("16x16x4  %4d workgroups: %.2f ms  %.1f TFLOP/s\n", wgs, ms, fl / ms / 1e9);
        hipEventRecord(e0);
        k32<<<wgs, 256>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        fl = (double)wgs * 4 * iters * 4 * 4096.0;
        printf("32x32x2  %4d workgroups: %.2f ms  %.1f TFLOP/s\n", wgs, ms, fl / ms / 1e9);
    }
    return 0;
}
