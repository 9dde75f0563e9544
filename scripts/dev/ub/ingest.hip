// Development micro-benchmark (r05): how fast does ONE CU ingest bytes that every workgroup of the chip re-reads -- the
// weight slices of the coarse regularisation layers (conv5 / conv6 / conv7: 221-786 KB per layer, re-streamed by every
// workgroup) -- through (a) LDS-direct buffer loads (buffer_load_dwordx4 ... lds, what K3 / K3w stage weights with) and
// (b) plain global_load_dwordx4 into VGPRs (the vector L1 path), as a function of the footprint (L1 / L2 / MALL / HBM
// resident) and of the workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 scripts/dev/ub/ingest.hip -o scripts/dev/ub/ingest && scripts/dev/ub/ingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f4 __attribute__((ext_vector_type(4)));

// every workgroup streams `iters` pieces of 16 KiB (4 waves x 4 instructions x 1 KiB) of the SAME buffer of `foot` bytes,
// DEPTH pieces in flight per wave-group; workgroup b starts at piece (b * 37) so the CUs do not walk in lock-step
template <int DEPTH>
__global__ __launch_bounds__(256) void ingest_lds(const float* buf, unsigned foot, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // DEPTH x 16 KiB ring
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, (short)0, foot, 0x00020000);
    const unsigned npieces = foot / 16384u;
    unsigned p = (blockIdx.x * 37u) % npieces;
    for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + d * 4096 + (wave * 4 + j) * 256), 16,
                                                         p * 16384u + (unsigned)(wave * 4 + j) * 1024u + lane * 16u, 0, 0, 0);
            p = p + 1 == npieces ? 0 : p + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (sink && threadIdx.x == 0) sink[blockIdx.x] = smem[lane];
}

template <int DEPTH>
__global__ __launch_bounds__(256) void ingest_vgpr(const float* buf, unsigned foot, int iters, float* sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned npieces = foot / 16384u;
    unsigned p = (blockIdx.x * 37u) % npieces;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it += DEPTH) {
        f4 v[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[d][j] = *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(buf) + (size_t)p * 16384u + (wave * 4 + j) * 1024u + lane * 16u);
            p = p + 1 == npieces ? 0 : p + 1;
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += v[d][j];
    }
    if (sink && acc.x + acc.y + acc.z + acc.w == 1234.5f) sink[blockIdx.x] = acc.x;
}

// all four waves read the SAME 4 KiB per step (what a 4-wave workgroup does when each wave needs every weight fragment)
template <int DEPTH>
__global__ __launch_bounds__(256) void ingest_vgpr_shared(const float* buf, unsigned foot, int iters, float* sink) {
    const int lane = threadIdx.x & 63;
    const unsigned npieces = foot / 4096u;
    unsigned p = (blockIdx.x * 37u) % npieces;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it += DEPTH) {
        f4 v[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[d][j] = *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(buf) + (size_t)p * 4096u + j * 1024u + lane * 16u);
            p = p + 1 == npieces ? 0 : p + 1;
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += v[d][j];
    }
    if (sink && acc.x + acc.y + acc.z + acc.w == 1234.5f) sink[blockIdx.x] = acc.x;
}

template <typename K>
static double time_kernel(K kernel, int grid, size_t lds, const float* buf, unsigned foot, int iters, float* sink) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (lds > 65536) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kernel<<<grid, 256, lds>>>(buf, foot, iters, sink);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        kernel<<<grid, 256, lds>>>(buf, foot, iters, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    const size_t maxfoot = 512u << 20;
    float *buf, *sink;
    CK(hipMalloc(&buf, maxfoot)); CK(hipMemset(buf, 0, maxfoot)); CK(hipMalloc(&sink, 1 << 20));
    const unsigned foots[] = {32u << 10, 128u << 10, 768u << 10, 3u << 20, 24u << 20, 192u << 20, 512u << 20};
    printf("%-22s %10s %6s %9s %10s %10s\n", "kernel", "footprint", "WG/CU", "ms", "GB/s/CU", "TB/s chip");
    for (unsigned foot : foots)
        for (int wgcu : {1, 2, 4}) {
            const int grid = 256 * wgcu, iters = 2048 / wgcu;   // 32 MiB per CU in all
            const double bytes_lds = (double)grid * iters * 16384.0;
            struct { const char* name; double ms; double bytes; } rows[] = {
                {"lds_dma depth2", time_kernel(ingest_lds<2>, grid, 2 * 16384, buf, foot, iters, sink), bytes_lds},
                {"lds_dma depth4", wgcu <= 2 ? time_kernel(ingest_lds<4>, grid, 4 * 16384, buf, foot, iters, sink) : 0.0, bytes_lds},
                {"vgpr depth2", time_kernel(ingest_vgpr<2>, grid, 0, buf, foot, iters, sink), bytes_lds},
                {"vgpr depth4", time_kernel(ingest_vgpr<4>, grid, 0, buf, foot, iters, sink), bytes_lds},
                {"vgpr 4-waves-same d4", time_kernel(ingest_vgpr_shared<4>, grid, 0, buf, foot, iters, sink), (double)grid * iters * 4096.0 * 4},
            };
            for (auto& r : rows) {
                if (r.ms <= 0.0) continue;
                printf("%-22s %8u K %6d %9.3f %10.1f %10.2f\n", r.name, foot >> 10, wgcu, r.ms, r.bytes / 256.0 / r.ms * 1e-6, r.bytes / r.ms * 1e-9);
            }
        }
    return 0;
}
