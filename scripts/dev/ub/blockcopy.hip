// Development micro-benchmark (r05): does the LAYOUT of the full-resolution 8-channel activations cap the HBM rate of the
// HBM-bound layers (conv0 out / conv1 in / conv11 skip + out / prob in: 4.2-4.6 TB/s of algorithmic bytes against the 6.3 TB/s
// of a linear copy)?  Every 256-thread workgroup copies one tile of 8 channels x TY rows x TX pixels (16-byte accesses):
//   planar  [C][D][H][W]: a tile is 8 x TY runs of TX * 4 bytes (what the conv kernels read and write today),
//   blocked [D][H/TY][W/TX][C][TY][TX]: the same tile is ONE contiguous run of 8 * TY * TX * 4 bytes.
//   hipcc --offload-arch=gfx950 -O3 scripts/dev/ub/blockcopy.hip -o scripts/dev/ub/blockcopy && scripts/dev/ub/blockcopy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int C = 8, D = 16, H = 592, W = 800;   // stage-2 conv11 output of one branch: 8 x 16 x 592 x 800

template <int TX, int TY, bool BLOCKED>
__global__ __launch_bounds__(256) void tilecopy(const float* in, float* out, int ntx, int nty) {
    // XCD-aware order: XCD k walks the k-th eighth of the tile list (x fastest, then y, then z)
    const int n = ntx * nty * D, per = (n + 7) >> 3;
    const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (t >= n) return;
    const int bx = t % ntx, by = (t / ntx) % nty, z = t / (ntx * nty);
    constexpr int PPR = TX / 4, NP = C * TY * PPR;   // 16-byte pieces per row / per tile
    f4 v[(NP + 255) / 256];
#pragma unroll
    for (int i = 0; i < (NP + 255) / 256; ++i) {
        const int p = i * 256 + threadIdx.x;
        if (p < NP) {
            const int c = p / (TY * PPR), r = (p / PPR) % TY, x4 = p % PPR;
            const size_t off = BLOCKED ? ((size_t)t * NP + p) * 4
                                       : (((size_t)c * D + z) * H + by * TY + r) * W + bx * TX + x4 * 4;
            v[i] = *reinterpret_cast<const f4*>(in + off);
        }
    }
#pragma unroll
    for (int i = 0; i < (NP + 255) / 256; ++i) {
        const int p = i * 256 + threadIdx.x;
        if (p < NP) {
            const int c = p / (TY * PPR), r = (p / PPR) % TY, x4 = p % PPR;
            const size_t off = BLOCKED ? ((size_t)t * NP + p) * 4
                                       : (((size_t)c * D + z) * H + by * TY + r) * W + bx * TX + x4 * 4;
            *reinterpret_cast<f4*>(out + off) = v[i] * 2.0f;
        }
    }
}

__global__ __launch_bounds__(256) void lincopy(const f4* in, f4* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i] * 2.0f;
}

template <int TX, int TY, bool BLOCKED> void run(const float* in, float* out) {
    static_assert(W % TX == 0 && H % TY == 0, "whole tiles");
    const int ntx = W / TX, nty = H / TY, n = ntx * nty * D;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) tilecopy<TX, TY, BLOCKED><<<8 * ((n + 7) / 8), 256>>>(in, out, ntx, nty);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) tilecopy<TX, TY, BLOCKED><<<8 * ((n + 7) / 8), 256>>>(in, out, ntx, nty);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("%-8s tile %3d x %2d (%5d B runs, %2d KB per workgroup): %.3f ms, %.0f GB/s read + write\n", BLOCKED ? "blocked" : "planar", TX, TY,
           BLOCKED ? C * TY * TX * 4 : TX * 4, C * TY * TX * 4 / 1024, ms, 2.0 * C * D * H * W * 4 / ms * 1e-6);
}

int main() {
    const size_t n = (size_t)C * D * H * W;
    float *in, *out; CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMemset(in, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) lincopy<<<256 * 8, 256>>>((const f4*)in, (f4*)out, n / 4);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) lincopy<<<256 * 8, 256>>>((const f4*)in, (f4*)out, n / 4);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("linear copy (grid-stride, 16 B per lane): %.3f ms, %.0f GB/s read + write (%zu MB each way)\n", ms, 2.0 * n * 4 / ms * 1e-6, n * 4 >> 20);
    run<32, 8, false>(in, out); run<32, 8, true>(in, out);
    run<32, 16, false>(in, out); run<32, 16, true>(in, out);
    run<160, 8, false>(in, out); run<160, 8, true>(in, out);
    run<800, 2, false>(in, out); run<800, 2, true>(in, out);
    run<32, 4, false>(in, out); run<32, 4, true>(in, out);
    return 0;
}
