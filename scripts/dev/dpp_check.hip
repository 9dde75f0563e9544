// dev check: DPP group broadcast vs __shfl for LPP = 2, 4, 8 and every J; also when the result feeds floorf / sub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
constexpr int dpp_quad(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int kRowShl4 = 0x104, kRowShr4 = 0x114;
__device__ __forceinline__ int quad_bcast_i(int v, int q) {
    switch (q) { case 0: return dpp_i<dpp_quad(0,0,0,0)>(v); case 1: return dpp_i<dpp_quad(1,1,1,1)>(v);
                 case 2: return dpp_i<dpp_quad(2,2,2,2)>(v); default: return dpp_i<dpp_quad(3,3,3,3)>(v); }
}
template <int LPP> __device__ __forceinline__ float grp_bcast_f(float vf, bool hi4, int J) {
    const int v = __builtin_bit_cast(int, vf); int r;
    if constexpr (LPP == 2) r = J == 0 ? dpp_i<dpp_quad(0,0,2,2)>(v) : dpp_i<dpp_quad(1,1,3,3)>(v);
    else if constexpr (LPP == 4) r = quad_bcast_i(v, J);
    else { const int t = quad_bcast_i(v, J % 4);
           if (J < 4) { const int o = dpp_i<kRowShr4>(t); r = hi4 ? o : t; } else { const int o = dpp_i<kRowShl4>(t); r = hi4 ? t : o; } }
    return __builtin_bit_cast(float, r);
}
template <int LPP> __global__ void k(const float* in, float* out_raw, float* out_use, float* ref_raw) {
    const int lane = threadIdx.x; const bool hi4 = lane & 4; const int grp = lane & ~(LPP - 1);
    float v[2] = {in[lane], in[64 + lane]};
#pragma unroll
    for (int j = 0; j < 2 * LPP; ++j) {
        const float b = grp_bcast_f<LPP>(v[j / LPP], hi4, j % LPP);
        const float s = __shfl(v[j / LPP], grp | (j % LPP), 64);
        out_raw[j * 64 + lane] = b; ref_raw[j * 64 + lane] = s;
        const float f = floorf(b); out_use[j * 64 + lane] = (b - f) * 2.f + f;   // consumer pattern of the kernel
    }
}
template <int LPP> int run() {
    float h[128], *d, *o1, *o2, *o3; for (int i = 0; i < 128; ++i) h[i] = i * 1.37f + 0.21f;
    hipMalloc(&d, 512); hipMalloc(&o1, 16*64*4); hipMalloc(&o2, 16*64*4); hipMalloc(&o3, 16*64*4);
    hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
    k<LPP><<<1, 64>>>(d, o1, o2, o3); hipDeviceSynchronize();
    static float a[16*64], b[16*64], c[16*64]; hipMemcpy(a, o1, sizeof a, hipMemcpyDeviceToHost); hipMemcpy(b, o2, sizeof b, hipMemcpyDeviceToHost); hipMemcpy(c, o3, sizeof c, hipMemcpyDeviceToHost);
    int bad = 0, bad2 = 0;
    for (int i = 0; i < 2 * LPP * 64; ++i) { if (a[i] != c[i]) { if (bad < 4) printf("LPP %d j %d lane %d dpp %f shfl %f\n", LPP, i/64, i%64, a[i], c[i]); ++bad; }
        const float f = floorf(c[i]); if (b[i] != (c[i] - f) * 2.f + f) ++bad2; }
    printf("LPP %d: raw mismatches %d, consumer mismatches %d\n", LPP, bad, bad2); return bad + bad2;
}
int main() { return run<2>() + run<4>() + run<8>() ? 1 : 0; }
