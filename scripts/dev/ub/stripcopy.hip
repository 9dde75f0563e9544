// dev micro-benchmark: the memory pattern of the K3s row sweep alone -- every wave copies a column strip of 8 planes down R
// rows (P pixels per lane: 4 / 8 / 16-byte accesses), NS rows in flight, XCD-aware order.  Prints GB/s (read + write).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatP1 __attribute__((ext_vector_type(1)));
template <int P> struct vecP { typedef float type __attribute__((ext_vector_type(P))); };
template <int P, int NS>
__global__ __launch_bounds__(64) void stripcopy(const float* in, float* out, int V, int H, int W, int R, int nrb, int nstrips) {
    typedef typename vecP<P>::type vec_t;
    const int lane = threadIdx.x;
    const int nwg = nstrips * nrb * V;
    const int t = (int)(blockIdx.x & 7) * ((nwg + 7) >> 3) + (int)(blockIdx.x >> 3);
    if (t >= nwg) return;
    const int strip = t % nstrips, rb = (t / nstrips) % nrb, v = t / (nstrips * nrb);
    const int x = (strip * 64 + lane) * P;
    if (x >= W) return;
    const size_t plane = (size_t)V * H * W;
    const float* ip = in + (size_t)v * H * W + x;
    float* op = out + (size_t)v * H * W + x;
    const int y0 = rb * R, y1 = min(y0 + R, H);
    vec_t rows[NS][8];
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
#pragma unroll
        for (int c = 0; c < 8; ++c) rows[i][c] = *reinterpret_cast<const vec_t*>(ip + c * plane + (size_t)min(y0 + i, H - 1) * W);
    for (int y = y0; y < y1; y += NS) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int c = 0; c < 8; ++c) rows[(s + NS - 1) % NS][c] = *reinterpret_cast<const vec_t*>(ip + c * plane + (size_t)min(y + s + NS - 1, H - 1) * W);
            __builtin_amdgcn_sched_barrier(0);
            if (y + s < y1)
#pragma unroll
                for (int c = 0; c < 8; ++c) *reinterpret_cast<vec_t*>(op + c * plane + (size_t)(y + s) * W) = rows[s % NS][c] * 2.0f;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
template <int P, int NS> void run(const float* in, float* out, int V, int H, int W, int R) {
    const int nstrips = (W + 64 * P - 1) / (64 * P), nrb = (H + R - 1) / R, n = nstrips * nrb * V;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) stripcopy<P, NS><<<8 * ((n + 7) / 8), 64>>>(in, out, V, H, W, R, nrb, nstrips);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) stripcopy<P, NS><<<8 * ((n + 7) / 8), 64>>>(in, out, V, H, W, R, nrb, nstrips);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("P=%d NS=%d R=%d waves=%d: %.3f ms, %.0f GB/s\n", P, NS, R, n, ms, 2.0 * 8 * V * H * W * 4 / ms * 1e-6);
}
int main() {
    const int V = 5, H = 1184, W = 1600;
    float *in, *out; const size_t n = (size_t)8 * V * H * W;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 4); hipMemset(in, 0, n * 4);
    run<1, 4>(in, out, V, H, W, 20); run<1, 6>(in, out, V, H, W, 24); run<1, 4>(in, out, V, H, W, 40);
    run<2, 4>(in, out, V, H, W, 20); run<2, 4>(in, out, V, H, W, 40); run<2, 6>(in, out, V, H, W, 48);
    run<4, 4>(in, out, V, H, W, 40); run<4, 4>(in, out, V, H, W, 80); run<4, 6>(in, out, V, H, W, 48);
    return 0;
}
