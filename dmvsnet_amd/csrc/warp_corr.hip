// K1: fused inverse-homography warp + bilinear gather + 2-group correlation + view sum.
//
// Replaces CostAgg.forward (/root/reference/networks/mvsnet.py:111-153) and homo_warping
// (/root/reference/networks/module.py:212-251).  The reference materialises a [C][D][H][W] warped
// volume per view (0.5-1 GB), multiplies it by the reference feature and reduces it in two more
// passes; here one kernel reads the source features through the cache hierarchy and writes only the
// [2][D][H][W] similarity volume.
//
// Mapping (wave64): features are pixel-major ("HWC") so one bilinear tap of one pixel is C
// contiguous floats.  A pixel is owned by LPP = C/4 adjacent lanes, each holding one float4 of
// channels, so every tap load of a wave is 64/LPP full contiguous C*4-byte runs (128 B for C=32)
// instead of 64 scattered 16-byte pieces.  The two correlation groups are the even / odd channels
// (mvsnet.py:139: view(b,c//2,2,...).mean(1)), i.e. components .x/.z and .y/.w of each float4.
// The per-pixel 2-vector is reduced over the LPP lanes with DPP shuffles after the view loop.
//
// Numerics: coordinates follow the reference's op order exactly (rot*(x,y,1), *depth, +trans,
// z==0 -> +1e-5, /z, normalise to [-1,1], ATen's un-normalise), all in fp32 without contraction, so
// tap positions agree with ATen's grid_sampler to rounding.  Taps outside the image contribute zero
// individually (padding_mode="zeros").  Per tap the channel dot products are formed first and then
// weighted (linear re-association of interpolate-then-multiply; differs by ~1 ulp).
#include "common.h"

#include <cstdlib>
#include <cstring>

struct WarpArgs {
    const float* ref;
    const float* src[DMVS_MAX_SRC_VIEWS];
    const float* proj;   // [nsrc][12]
    const float* depth;  // [D][H][W] hypothesis planes, or NULL: plane d of pixel p is base[p] + d * step[0]
    const float* base;   // [H][W]   (affine hypotheses: linear depth sampling, module.py:476-507 / 560-579)
    const float* step;   // [1] device scalar: the stage's plane spacing (= the interval handed to K4)
    float* sim;          // [2][D][H][W]
    int nsrc, pix_stride, D, H, W, accumulate;
};

// hypothesis plane d at pixel `pix`: from the materialised volume, or base + d * step (mul and add rounded
// separately, like the reference's `lo + d * step`); the affine form drops the D*H*W read
__device__ __forceinline__ float hyp_plane(const WarpArgs& a, int d, size_t plane, size_t pix, float step) {
    return a.depth ? a.depth[(size_t)d * plane + pix] : a.base[pix] + (float)d * step;
}

template <int C, int DCHUNK>
__global__ __launch_bounds__(256) void warp_corr_kernel(WarpArgs a) {
    constexpr int LPP = C / 4;          // lanes per pixel
    constexpr int PPB = 256 / LPP;      // pixels per block
    const int lane_c = threadIdx.x % LPP;
    const int x = blockIdx.x * PPB + threadIdx.x / LPP;
    const int y = blockIdx.y;
    const int d0 = blockIdx.z * DCHUNK;
    const int W = a.W, H = a.H;
    const bool live = x < W;
    const int xc = live ? x : W - 1;  // clamp so that every lane stays in the shuffles

    const float4_t r4 = *reinterpret_cast<const float4_t*>(a.ref + ((size_t)y * W + xc) * a.pix_stride + lane_c * 4);
    const float fx = (float)xc, fy = (float)y;
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    const float half_w = wm1 / 2.0f, half_h = hm1 / 2.0f;  // (width-1)/2, module.py:240-241
    const size_t plane = (size_t)H * W;
    const int dend = min(d0 + DCHUNK, a.D);

    for (int d = d0; d < dend; ++d) {
        const float depth = hyp_plane(a, d, plane, (size_t)y * W + xc, a.depth ? 0.f : a.step[0]);
        float acc0 = 0.f, acc1 = 0.f;
        for (int v = 0; v < a.nsrc; ++v) {
            const float* P = a.proj + v * 12;  // uniform -> scalar loads
            // rot @ (x, y, 1)   (module.py:233)
            const float rx = fmaf(P[1], fy, P[0] * fx) + P[2];
            const float ry = fmaf(P[4], fy, P[3] * fx) + P[5];
            const float rz = fmaf(P[7], fy, P[6] * fx) + P[8];
            // * depth + trans   (module.py:234-236)
            const float px = rx * depth + P[9];
            const float py = ry * depth + P[10];
            float pz = rz * depth + P[11];
            if (pz == 0.0f) pz += 0.00001f;  // module.py:237
            // perspective divide, normalise (module.py:239-241), ATen un-normalise (align_corners=True)
            const float gx = (px / pz) / half_w - 1.0f;
            const float gy = (py / pz) / half_h - 1.0f;
            const float ix = ((gx + 1.0f) / 2.0f) * wm1;
            const float iy = ((gy + 1.0f) / 2.0f) * hm1;
            const float x0f = floorf(ix), y0f = floorf(iy);
            const float tx = ix - x0f, ty = iy - y0f;
            // bounds per tap; out-of-range coordinates are clamped for addressing and get weight 0
            const bool x0in = (x0f >= 0.f) && (x0f <= wm1), x1in = (x0f >= -1.f) && (x0f <= wm1 - 1.f);
            const bool y0in = (y0f >= 0.f) && (y0f <= hm1), y1in = (y0f >= -1.f) && (y0f <= hm1 - 1.f);
            const int x0 = (int)fminf(fmaxf(x0f, 0.f), wm1), x1 = (int)fminf(fmaxf(x0f + 1.f, 0.f), wm1);
            const int y0 = (int)fminf(fmaxf(y0f, 0.f), hm1), y1 = (int)fminf(fmaxf(y0f + 1.f, 0.f), hm1);
            const float w00 = (x0in && y0in) ? (1.f - tx) * (1.f - ty) : 0.f;
            const float w01 = (x1in && y0in) ? tx * (1.f - ty) : 0.f;
            const float w10 = (x0in && y1in) ? (1.f - tx) * ty : 0.f;
            const float w11 = (x1in && y1in) ? tx * ty : 0.f;
            const float* S = a.src[v] + lane_c * 4;
            const float4_t s00 = *reinterpret_cast<const float4_t*>(S + ((size_t)y0 * W + x0) * a.pix_stride);
            const float4_t s01 = *reinterpret_cast<const float4_t*>(S + ((size_t)y0 * W + x1) * a.pix_stride);
            const float4_t s10 = *reinterpret_cast<const float4_t*>(S + ((size_t)y1 * W + x0) * a.pix_stride);
            const float4_t s11 = *reinterpret_cast<const float4_t*>(S + ((size_t)y1 * W + x1) * a.pix_stride);
            // even channels -> group 0, odd channels -> group 1
            const float e00 = fmaf(s00.z, r4.z, s00.x * r4.x), o00 = fmaf(s00.w, r4.w, s00.y * r4.y);
            const float e01 = fmaf(s01.z, r4.z, s01.x * r4.x), o01 = fmaf(s01.w, r4.w, s01.y * r4.y);
            const float e10 = fmaf(s10.z, r4.z, s10.x * r4.x), o10 = fmaf(s10.w, r4.w, s10.y * r4.y);
            const float e11 = fmaf(s11.z, r4.z, s11.x * r4.x), o11 = fmaf(s11.w, r4.w, s11.y * r4.y);
            acc0 = fmaf(w00, e00, fmaf(w01, e01, fmaf(w10, e10, fmaf(w11, e11, acc0))));
            acc1 = fmaf(w00, o00, fmaf(w01, o01, fmaf(w10, o10, fmaf(w11, o11, acc1))));
        }
        // reduce the LPP channel-chunks of this pixel
#pragma unroll
        for (int m = LPP / 2; m >= 1; m >>= 1) {
            acc0 += __shfl_xor(acc0, m, 64);
            acc1 += __shfl_xor(acc1, m, 64);
        }
        if (live && lane_c == 0) {
            const float inv = 2.0f / (float)C;  // mean over C/2 channels of a group
            const size_t o = (size_t)d * plane + (size_t)y * W + x;
            float v0 = acc0 * inv, v1 = acc1 * inv;
            if (a.accumulate) { v0 += a.sim[o]; v1 += a.sim[(size_t)a.D * plane + o]; }
            a.sim[o] = v0;
            a.sim[(size_t)a.D * plane + o] = v1;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// LDS-staged variant (the default).  The kernel above sends every bilinear tap through the vector L1
// (45.6 GB of tap traffic per config-2 depth map against 1.6 GB of compulsory bytes) and is bound by L1
// bandwidth at ~10 TB/s aggregate.  Here a workgroup owns a small reference tile x DC hypothesis planes and,
// per source view,
//   1. every plane's projected coordinate is computed ONCE, by an owner lane of the pixel's lane group
//      (plane j belongs to lane j % LPP), and kept in registers;
//   2. the workgroup reduces the exact bounding box of all taps that fall inside the image (wave shuffles +
//      one LDS exchange) -- no monotonicity assumption about the hypotheses, works for linear / inverse /
//      refine planes alike;
//   3. the box is staged in LDS with asynchronous LDS-direct buffer loads: pixel-major features make a box
//      row one contiguous run, so every wave-instruction moves 256 contiguous bytes and each source pixel is
//      fetched once per (tile, plane chunk) instead of once per tap;
//   4. taps are read from LDS (ds_read_b128, 4x the L1 rate); coordinates are broadcast from the owner lane.
// A box that does not fit the 48 KB window (or a padded pixel stride) falls back to global taps for that
// (view, chunk) only -- a workgroup-uniform branch, same arithmetic.
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Cross-lane helpers on DPP (VALU-rate lane movement folded into the consuming ALU op) and v_readlane, used
// instead of ds_bpermute-based __shfl (an LDS-pipe instruction with a waitcnt behind it) where the pattern is static.
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v))); }
constexpr int dpp_quad(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int kRowShl4 = 0x104, kRowShr4 = 0x114, kRowRor8 = 0x128;
// sum over the LPP lanes of an aligned lane group (LPP in {2,4,8}), result in every lane
template <int LPP> __device__ __forceinline__ float grp_allsum(float v, bool hi4) {
    if constexpr (LPP >= 2) v += dpp_f<dpp_quad(1, 0, 3, 2)>(v);
    if constexpr (LPP >= 4) v += dpp_f<dpp_quad(2, 3, 0, 1)>(v);
    if constexpr (LPP >= 8) { const float a = dpp_f<kRowShl4>(v), b = dpp_f<kRowShr4>(v); v += hi4 ? b : a; }
    return v;
}
// wave-wide min / max as a wave-uniform value: butterfly inside the 16-lane rows, then the 4 row results
template <bool IS_MIN> __device__ __forceinline__ int wave_minmax(int v, bool hi4) {
    auto op = [](int a, int b) { return IS_MIN ? min(a, b) : max(a, b); };
    v = op(v, dpp_i<dpp_quad(1, 0, 3, 2)>(v));
    v = op(v, dpp_i<dpp_quad(2, 3, 0, 1)>(v));
    { const int a = dpp_i<kRowShl4>(v), b = dpp_i<kRowShr4>(v); v = op(v, hi4 ? b : a); }
    v = op(v, dpp_i<kRowRor8>(v));
    return op(op(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
              op(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// a / b given r ~ 1/b to within an ulp: q = a*r, one residual correction -> the correctly rounded quotient for
// normal-range operands (no scaling / denormal fix-up, which the 4 divisions of a projection never need)
__device__ __forceinline__ float fdiv_rn(float a, float b, float r) {
    const float q = a * r;
    return fmaf(fmaf(-b, q, a), r, q);
}

// value of lane K of every aligned LPP-lane group, in all lanes of the group (K, LPP compile-time): one v_mov_dpp
// quad_perm inside a quad, plus one bank-masked row shift to cross the two quads of an 8-lane group.  Replaces a
// ds_bpermute-based __shfl (measured: the two shuffles per sample cost 0.3 of K1's 2.4 ms).  The empty asm keeps
// hipcc (ROCm 7.2) from folding the DPP move into its consumer, which it mis-compiles inside this kernel
// (scripts/dev/dpp_check.hip).
template <int LPP, int K> __device__ __forceinline__ float grp_bcast(float v) {
    if constexpr (LPP == 1) return v;
    int x = __builtin_bit_cast(int, v);
    constexpr int k = K & 3;
    int q;
    if constexpr (LPP == 2) q = __builtin_amdgcn_update_dpp(0, x, dpp_quad(K, K, 2 + K, 2 + K), 0xf, 0xf, true);
    else q = __builtin_amdgcn_update_dpp(0, x, dpp_quad(k, k, k, k), 0xf, 0xf, true);
    if constexpr (LPP == 8) {
        // every quad now holds ITS lane k; the group wants the one of quad K / 4: shift it into the other quad
        // (banks = quads of a 16-lane row: 0b1010 writes quads 1 and 3, 0b0101 quads 0 and 2; others keep q)
        if constexpr (K < 4) q = __builtin_amdgcn_update_dpp(q, q, kRowShr4, 0xf, 0xa, false);
        else q = __builtin_amdgcn_update_dpp(q, q, kRowShl4, 0xf, 0x5, false);
    }
    asm volatile("" : "+v"(q));
    return __builtin_bit_cast(float, q);
}

template <int C>
struct TapMath {
    // weights + integer tap coordinates of one sample; same arithmetic on the weights as the kernel above.
    // x0 / y0 are the UNCLAMPED floor coordinates (saturating float -> int conversion); callers clamp them into
    // the staged window or the image with one v_med3_i32 each.  In-range tests are `med3(v, lo, hi) == v` (two
    // instructions, false for NaN like the reference's comparisons).
    float w00, w01, w10, w11;
    int x0, y0;
    __device__ __forceinline__ void set(float ix, float iy, float wm1, float hm1) {
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float tx = ix - x0f, ty = iy - y0f;
        const bool x0in = __builtin_amdgcn_fmed3f(x0f, 0.f, wm1) == x0f, x1in = __builtin_amdgcn_fmed3f(x0f, -1.f, wm1 - 1.f) == x0f;
        const bool y0in = __builtin_amdgcn_fmed3f(y0f, 0.f, hm1) == y0f, y1in = __builtin_amdgcn_fmed3f(y0f, -1.f, hm1 - 1.f) == y0f;
        x0 = (int)x0f;
        y0 = (int)y0f;
        // per-axis weights, zero outside the image: w = wx * wy is the reference's product when both are in range
        // and exactly 0 otherwise (a NaN coordinate is out of range on its axis: its factor is the selected 0)
        const float wx0 = x0in ? 1.f - tx : 0.f, wx1 = x1in ? tx : 0.f;
        const float wy0 = y0in ? 1.f - ty : 0.f, wy1 = y1in ? ty : 0.f;
        w00 = wx0 * wy0;
        w01 = wx1 * wy0;
        w10 = wx0 * wy1;
        w11 = wx1 * wy1;
    }
};

__device__ __forceinline__ int med3i(int v, int lo, int hi) { return min(max(v, lo), hi); }  // folds to v_med3_i32

__device__ __forceinline__ void corr_taps(const float4_t& s00, const float4_t& s01, const float4_t& s10,
                                          const float4_t& s11, const float4_t& r4, float w00, float w01, float w10,
                                          float w11, float& acc0, float& acc1) {
    const float e00 = fmaf(s00.z, r4.z, s00.x * r4.x), o00 = fmaf(s00.w, r4.w, s00.y * r4.y);
    const float e01 = fmaf(s01.z, r4.z, s01.x * r4.x), o01 = fmaf(s01.w, r4.w, s01.y * r4.y);
    const float e10 = fmaf(s10.z, r4.z, s10.x * r4.x), o10 = fmaf(s10.w, r4.w, s10.y * r4.y);
    const float e11 = fmaf(s11.z, r4.z, s11.x * r4.x), o11 = fmaf(s11.w, r4.w, s11.y * r4.y);
    acc0 = fmaf(w00, e00, fmaf(w01, e01, fmaf(w10, e10, fmaf(w11, e11, acc0))));
    acc1 = fmaf(w00, o00, fmaf(w01, o01, fmaf(w10, o10, fmaf(w11, o11, acc1))));
}

#ifndef DMVS_LBC_SYNC
#define DMVS_LBC_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif
#ifndef DMVS_LBC_WIN
#define DMVS_LBC_WIN 3840
#endif
// LBC ("LDS broadcast"): the owner lane's eight per-sample values (4 weights, 4 window offsets) reach the other lanes
// of its pixel group through a wave-private LDS slot (2 ds_write_b128 per owner and round, 2 broadcast ds_read_b128
// per lane and plane) instead of 8 (C = 8 / 16) or 16 (C = 32) DPP moves, which issue at ~0.6 of the FMA rate
// (scripts/dev/valu_rate.hip); the 8 KB of slots come out of the two staging windows (2 x 15 KB instead of 2 x 16 KB).
template <int C, int DC, bool LBC = false>
__global__ __launch_bounds__(256) void warp_corr_lds_kernel(WarpArgs a) {
    constexpr int LPP = C / 4;                 // lanes per pixel
    constexpr int NPIX = 256 / LPP;            // pixels per workgroup
    constexpr int TW = (C == 8) ? 16 : 8, TH = NPIX / TW;
    constexpr int PPL = (DC + LPP - 1) / LPP;  // planes owned per lane
    constexpr int WIN_F = LBC ? DMVS_LBC_WIN : 4096;   // two 16 (15) KB staging windows (views alternate)
    __shared__ __attribute__((aligned(16))) float box[2 * WIN_F];
    __shared__ int red[2][4][4];
    typedef int int4_t __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float4_t bcw[LBC ? 256 : 1];   // the four tap weights of a lane's plane
    __shared__ __attribute__((aligned(16))) int4_t bco[LBC ? 256 : 1];     // ... and its four window offsets

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane_c = tid % LPP, p = tid / LPP;
    const bool hi4 = (lane & 4) != 0;  // upper quad of an 8-lane group (DPP helpers)
    const int W = a.W, H = a.H;
    const int x = blockIdx.x * TW + p % TW, y = blockIdx.y * TH + p / TW;
    const int d0 = blockIdx.z * DC;
    const bool live = x < W && y < H;
    const int xc = min(x, W - 1), yc = min(y, H - 1);
    const size_t plane = (size_t)H * W;
    const float4_t r4 = *reinterpret_cast<const float4_t*>(a.ref + ((size_t)yc * W + xc) * a.pix_stride + lane_c * 4);
    const float fx = (float)xc, fy = (float)yc;
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    const float half_w = wm1 / 2.0f, half_h = hm1 / 2.0f;
    const float inv_half_w = 1.0f / half_w, inv_half_h = 1.0f / half_h;  // IEEE, once per thread

    float dep[PPL];
#pragma unroll
    for (int s = 0; s < PPL; ++s) {
        const int d = min(d0 + s * LPP + lane_c, a.D - 1);  // planes past the end duplicate the last one
        dep[s] = hyp_plane(a, d, plane, (size_t)yc * W + xc, a.depth ? 0.f : a.step[0]);
    }
    float acc0[DC], acc1[DC];
#pragma unroll
    for (int j = 0; j < DC; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }

    // View pipeline.  Iteration v:  project view v+1 and reduce its bounding box per wave (VALU work that runs
    // while view v's window is still landing)  ->  wait for window v + ONE barrier (which also publishes the
    // per-wave boxes of view v+1)  ->  issue the staging of window v+1 into the other LDS window  ->  sample view v.
    // Nothing a wave waits for is issued right before the wait, and there is one barrier per view instead of two.
    float ix[PPL], iy[PPL];          // projected coordinates of the view being sampled
    float nix[PPL], niy[PPL];        // ... of the next view
    int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1, RS = 0;  // window of the view being sampled
    bool fits = false, empty = true;

    // owner lanes project their planes of view v (same op order as the reference, see the kernel above) and
    // reduce the in-image tap bounding box over the wave (DPP butterflies + readlane: wave-uniform)
    auto project = [&](int v, float* ox, float* oy, int slot) {
        const float* P = a.proj + v * 12;
        const float rx = fmaf(P[1], fy, P[0] * fx) + P[2];
        const float ry = fmaf(P[4], fy, P[3] * fx) + P[5];
        const float rz = fmaf(P[7], fy, P[6] * fx) + P[8];
        int mnx = 0x7fffffff, mxx = -0x7fffffff, mny = 0x7fffffff, mxy = -0x7fffffff;
#pragma unroll
        for (int s = 0; s < PPL; ++s) {
            const float px = rx * dep[s] + P[9];
            const float py = ry * dep[s] + P[10];
            float pz = rz * dep[s] + P[11];
            if (pz == 0.0f) pz += 0.00001f;
            // the reference's four divisions, each as reciprocal + one residual correction (Markstein): the
            // correctly rounded quotient for operands in the normal range, ~15 instructions for the four instead
            // of ~44 for four IEEE sequences with scaling / fix-up (measured: 0.2 of K1's 2.4 ms)
            float rz1 = __builtin_amdgcn_rcpf(pz);
            rz1 = fmaf(fmaf(-pz, rz1, 1.0f), rz1, rz1);
            const float gx = fdiv_rn(fdiv_rn(px, pz, rz1), half_w, inv_half_w) - 1.0f;
            const float gy = fdiv_rn(fdiv_rn(py, pz, rz1), half_h, inv_half_h) - 1.0f;
            ox[s] = ((gx + 1.0f) / 2.0f) * wm1;
            oy[s] = ((gy + 1.0f) / 2.0f) * hm1;
            // in-image part of this sample's 2x2 footprint
            const float x0f = fminf(fmaxf(floorf(ox[s]), -2.f), wm1 + 1.f), y0f = fminf(fmaxf(floorf(oy[s]), -2.f), hm1 + 1.f);
            const int lx = max((int)x0f, 0), hx = min((int)x0f + 1, W - 1);
            const int ly = max((int)y0f, 0), hy = min((int)y0f + 1, H - 1);
            if (lx <= hx && ly <= hy) {
                mnx = min(mnx, lx); mxx = max(mxx, hx);
                mny = min(mny, ly); mxy = max(mxy, hy);
            }
        }
        mnx = wave_minmax<true>(mnx, hi4); mxx = wave_minmax<false>(mxx, hi4);
        mny = wave_minmax<true>(mny, hi4); mxy = wave_minmax<false>(mxy, hi4);
        if (lane == 0) { red[slot][wave][0] = mnx; red[slot][wave][1] = mxx; red[slot][wave][2] = mny; red[slot][wave][3] = mxy; }
    };
    // combine the four per-wave boxes of a view (after a barrier) and start staging its window (asynchronous)
    auto open_window = [&](int v, int slot, int& wx0, int& wx1, int& wy0, int& wy1, int& wrs, bool& wfits, bool& wempty) {
        wx0 = min(min(red[slot][0][0], red[slot][1][0]), min(red[slot][2][0], red[slot][3][0]));
        wx1 = max(max(red[slot][0][1], red[slot][1][1]), max(red[slot][2][1], red[slot][3][1]));
        wy0 = min(min(red[slot][0][2], red[slot][1][2]), min(red[slot][2][2], red[slot][3][2]));
        wy1 = max(max(red[slot][0][3], red[slot][1][3]), max(red[slot][2][3], red[slot][3][3]));
        wempty = wx0 > wx1 || wy0 > wy1;
        const int BW = wx1 - wx0 + 1, BH = wy1 - wy0 + 1;
        wrs = BW * C;  // floats per window row
        wfits = !wempty && a.pix_stride == C && (long)BH * wrs <= WIN_F;
        if (wfits) {
            // the window is dense in LDS (row pitch = RS), so it is one run of 16-byte pieces; a wave-instruction
            // moves 64 of them (1 KiB) wherever the row boundaries fall.  (An LDS-direct load costs the issuing wave
            // 60-100 cycles whatever its width: 16-byte pieces instead of dwords cut the staging instructions 4x.)
            // Piece e lies in window row e / ppr; pixel rows are 16-byte aligned (C >= 8).
            float* win = box + slot * WIN_F;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[v], (short)0, H * W * C * 4, 0x00020000);
            const int ppr = wrs >> 2, npieces = BH * ppr;
            const float inv_ppr = 1.0f / (float)ppr;
            for (int i = wave; i * 64 < npieces; i += 4) {
                const int e = i * 64 + lane;
                int r = (int)((float)e * inv_ppr);
                r += (__mul24(r + 1, ppr) <= e) ? 1 : 0;  // the float quotient is off by at most one
                r -= (__mul24(r, ppr) > e) ? 1 : 0;
                if (e < npieces)  // lanes past the end must not write beyond the window
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(win + i * 256), 16,
                                                             (unsigned)((__mul24(wy0 + r, W) + wx0) * C + (e - __mul24(r, ppr)) * 4) * 4u, 0, 0, 0);
            }
        }
    };

    // coordinate of plane j from its owner lane j % LPP (j is a constant after unrolling: the switch folds)
    auto bcast_plane = [&](const float* c, int j) {
        const float v = c[j / LPP];
        switch (j % LPP) {
            case 0: return grp_bcast<LPP, 0>(v);
            case 1: return grp_bcast<LPP, 1 % LPP>(v);
            case 2: return grp_bcast<LPP, 2 % LPP>(v);
            case 3: return grp_bcast<LPP, 3 % LPP>(v);
            case 4: return grp_bcast<LPP, 4 % LPP>(v);
            case 5: return grp_bcast<LPP, 5 % LPP>(v);
            case 6: return grp_bcast<LPP, 6 % LPP>(v);
            default: return grp_bcast<LPP, 7 % LPP>(v);
        }
    };
    // value v of plane j's owner lane (j % LPP) in every lane of the group
    auto bcast_owner = [&](float v, int j) {
        switch (j % LPP) {
            case 0: return grp_bcast<LPP, 0>(v);
            case 1: return grp_bcast<LPP, 1 % LPP>(v);
            case 2: return grp_bcast<LPP, 2 % LPP>(v);
            case 3: return grp_bcast<LPP, 3 % LPP>(v);
            case 4: return grp_bcast<LPP, 4 % LPP>(v);
            case 5: return grp_bcast<LPP, 5 % LPP>(v);
            case 6: return grp_bcast<LPP, 6 % LPP>(v);
            default: return grp_bcast<LPP, 7 % LPP>(v);
        }
    };
    project(0, ix, iy, 0);
    __syncthreads();
    open_window(0, 0, bx0, bx1, by0, by1, RS, fits, empty);
    for (int v = 0; v < a.nsrc; ++v) {
        const int slot = v & 1;
        const bool more = v + 1 < a.nsrc;
        if (more) project(v + 1, nix, niy, slot ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of window v
        __syncthreads();  // every wave's; every wave is done sampling window v-1; the boxes of view v+1 are visible
        int nx0 = 0, nx1 = -1, ny0 = 0, ny1 = -1, nrs = 0;
        bool nfits = false, nempty = true;
        if (more) open_window(v + 1, slot ^ 1, nx0, nx1, ny0, ny1, nrs, nfits, nempty);
        // sample view v
        if (!empty) {
            if (fits) {
                // Tap math ONCE per (pixel, plane): the owner lane of a plane (the one that projected it) derives the
                // four weights and the four window offsets, the other lanes of the pixel's group receive the 8 values
                // by DPP (1-2 moves each).  Recomputing them in every lane cost ~50 VALU instructions per sample
                // and lane; the kernel is VALU-issue bound (~3.6 cycles per instruction over the whole sampling
                // loop), so the instruction count is the time.
                float tw[PPL][4];
                int to[PPL][4];
#pragma unroll
                for (int sidx = 0; sidx < PPL; ++sidx) {
                    TapMath<C> t;
                    t.set(ix[sidx], iy[sidx], wm1, hm1);
                    // zero-weight taps outside the image may lie outside the window: clamp their address into it
                    const int ax0 = med3i(t.x0, bx0, bx1) - bx0, ax1 = med3i(t.x0 + 1, bx0, bx1) - bx0;
                    const int ay0 = med3i(t.y0, by0, by1) - by0, ay1 = med3i(t.y0 + 1, by0, by1) - by0;
                    // window offsets fit 24 bits: v_mul_u32_u24 is full rate, the 32-bit v_mul_lo_u32 a quarter
                    const int r0 = __mul24(ay0, RS), r1 = __mul24(ay1, RS);
                    to[sidx][0] = r0 + ax0 * C; to[sidx][1] = r0 + ax1 * C; to[sidx][2] = r1 + ax0 * C; to[sidx][3] = r1 + ax1 * C;
                    tw[sidx][0] = t.w00; tw[sidx][1] = t.w01; tw[sidx][2] = t.w10; tw[sidx][3] = t.w11;
                }
                const float* B = box + slot * WIN_F + lane_c * 4;
                if constexpr (LBC) {
#pragma unroll
                    for (int sidx = 0; sidx < PPL; ++sidx) {
                        // (integer offsets in their own int4 array: extracting `bit_cast<int>(v.y)` from a float4 LDS
                        // load is mis-compiled by ROCm 7.2's hipcc into four copies of element x)
                        float4_t wv;
                        int4_t ov;
                        wv.x = tw[sidx][0]; wv.y = tw[sidx][1]; wv.z = tw[sidx][2]; wv.w = tw[sidx][3];
                        ov.x = to[sidx][0]; ov.y = to[sidx][1]; ov.z = to[sidx][2]; ov.w = to[sidx][3];
                        bcw[tid] = wv;
                        bco[tid] = ov;
                        // the other lanes of this WAVE read the slots: order the stores before the loads at wavefront scope
                        DMVS_LBC_SYNC();
#pragma unroll
                        for (int k = 0; k < LPP; ++k) {
                            const int j = sidx * LPP + k;
                            if (j < DC) {
                                const float4_t w4 = bcw[tid - lane_c + k];
                                const int4_t o4 = bco[tid - lane_c + k];
                                const float4_t s00 = *reinterpret_cast<const float4_t*>(B + o4.x);
                                const float4_t s01 = *reinterpret_cast<const float4_t*>(B + o4.y);
                                const float4_t s10 = *reinterpret_cast<const float4_t*>(B + o4.z);
                                const float4_t s11 = *reinterpret_cast<const float4_t*>(B + o4.w);
                                corr_taps(s00, s01, s10, s11, r4, w4.x, w4.y, w4.z, w4.w, acc0[j], acc1[j]);
                            }
                        }
                        DMVS_LBC_SYNC();   // ... and this round's loads before the next round's stores
                    }
                } else
#pragma unroll
                for (int j = 0; j < DC; ++j) {
                    const int sidx = j / LPP;
                    const float w00 = bcast_owner(tw[sidx][0], j), w01 = bcast_owner(tw[sidx][1], j);
                    const float w10 = bcast_owner(tw[sidx][2], j), w11 = bcast_owner(tw[sidx][3], j);
                    const int o00 = __builtin_bit_cast(int, bcast_owner(__builtin_bit_cast(float, to[sidx][0]), j));
                    const int o01 = __builtin_bit_cast(int, bcast_owner(__builtin_bit_cast(float, to[sidx][1]), j));
                    const int o10 = __builtin_bit_cast(int, bcast_owner(__builtin_bit_cast(float, to[sidx][2]), j));
                    const int o11 = __builtin_bit_cast(int, bcast_owner(__builtin_bit_cast(float, to[sidx][3]), j));
                    const float4_t s00 = *reinterpret_cast<const float4_t*>(B + o00);
                    const float4_t s01 = *reinterpret_cast<const float4_t*>(B + o01);
                    const float4_t s10 = *reinterpret_cast<const float4_t*>(B + o10);
                    const float4_t s11 = *reinterpret_cast<const float4_t*>(B + o11);
                    corr_taps(s00, s01, s10, s11, r4, w00, w01, w10, w11, acc0[j], acc1[j]);
                }
            } else {
                const float* G = a.src[v] + lane_c * 4;
#pragma unroll
                for (int j = 0; j < DC; ++j) {
                    const float jx = bcast_plane(ix, j), jy = bcast_plane(iy, j);
                    TapMath<C> t;
                    t.set(jx, jy, wm1, hm1);
                    const int gx0 = med3i(t.x0, 0, W - 1), gx1 = med3i(t.x0 + 1, 0, W - 1);
                    const int gy0 = med3i(t.y0, 0, H - 1), gy1 = med3i(t.y0 + 1, 0, H - 1);
                    const float4_t s00 = *reinterpret_cast<const float4_t*>(G + ((size_t)gy0 * W + gx0) * a.pix_stride);
                    const float4_t s01 = *reinterpret_cast<const float4_t*>(G + ((size_t)gy0 * W + gx1) * a.pix_stride);
                    const float4_t s10 = *reinterpret_cast<const float4_t*>(G + ((size_t)gy1 * W + gx0) * a.pix_stride);
                    const float4_t s11 = *reinterpret_cast<const float4_t*>(G + ((size_t)gy1 * W + gx1) * a.pix_stride);
                    corr_taps(s00, s01, s10, s11, r4, t.w00, t.w01, t.w10, t.w11, acc0[j], acc1[j]);
                }
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < PPL; ++s2) { ix[s2] = nix[s2]; iy[s2] = niy[s2]; }
        bx0 = nx0; bx1 = nx1; by0 = ny0; by1 = ny1; RS = nrs; fits = nfits; empty = nempty;
    }

    // all-reduce the channel chunks of a pixel; lane j of the group then stores plane j
    const float inv = 2.0f / (float)C;
#pragma unroll
    for (int j = 0; j < DC; ++j) {
        acc0[j] = grp_allsum<LPP>(acc0[j], hi4);
        acc1[j] = grp_allsum<LPP>(acc1[j], hi4);
    }
#pragma unroll
    for (int j = 0; j < DC; ++j) {
        if (live && (j % LPP) == lane_c && d0 + j < a.D) {
            const size_t o = (size_t)(d0 + j) * plane + (size_t)y * W + x;
            float v0 = acc0[j] * inv, v1 = acc1[j] * inv;
            if (a.accumulate) { v0 += a.sim[o]; v1 += a.sim[(size_t)a.D * plane + o]; }
            a.sim[o] = v0;
            a.sim[(size_t)a.D * plane + o] = v1;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Pixel-per-lane variant ("px").  The kernel above splits a pixel's channels over C/4 lanes: every lane of the group
// repeats (or receives by DPP) the per-sample tap bookkeeping, so ~half of its VALU instructions are not FMAs, and a
// workgroup covers only 32-128 pixels x 8 planes between two barriers -- too little work to hide the staging latency
// of the next window.  Here a LANE owns a pixel: all C reference channels sit in its registers, the projection, the
// tap weights and the window offsets of a sample are computed exactly once, and the only cross-lane traffic is the
// bounding-box reduction.  A 256-thread workgroup owns a 32 x 8 pixel tile x DC planes (1024 samples per view
// between barriers for DC = 4).
//
// LDS window: pixel-major with a PADDED pixel stride of CW/4 + 1 quads (16-byte pieces), CW = min(C, 16) channels
// per pass (C = 32 runs two channel passes per view, tap math shared).  The stride in quads is odd (5 or 3), so the
// 16 lanes of a ds_read_b128 service group -- 16 consecutive pixels of a tile row, which sample ~consecutive source
// pixels -- start on 16 different bank quads: conflict-free where the r01 kernel's lane groups were not.  The pad
// quad is staged from an out-of-range offset (no memory traffic).  Staging is 16-byte LDS-direct buffer loads as
// before; one window of <= WIN_F floats, 4 workgroups per CU, no intra-workgroup pipelining: the other three
// workgroups of the CU cover a window's flight time.
//
// Bounding box: float min / max of the clamped coordinates (6 instructions per sample instead of ~14 for the exact
// integer in-image box); taps outside the box are exactly the zero-weight taps outside the image, clamped into it.
template <bool IS_MIN> __device__ __forceinline__ float wave_minmax_f(float v, bool hi4) {
    auto op = [](float a, float b) { return IS_MIN ? fminf(a, b) : fmaxf(a, b); };
    v = op(v, dpp_f<dpp_quad(1, 0, 3, 2)>(v));
    v = op(v, dpp_f<dpp_quad(2, 3, 0, 1)>(v));
    { const float a = dpp_f<kRowShl4>(v), b = dpp_f<kRowShr4>(v); v = op(v, hi4 ? b : a); }
    v = op(v, dpp_f<kRowRor8>(v));
    const int i = __builtin_bit_cast(int, v);
    auto rl = [&](int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, l)); };
    return op(op(rl(0), rl(16)), op(rl(32), rl(48)));
}

// floats per staging window (dynamic LDS, chosen per launch): 4 or 3 workgroups per CU.  The larger window holds
// more of the scattered tiles of the later passes (random-weight benchmark, s2.main: 0.78 -> 0.65 ms) at the price of
// a quarter of the latency hiding; the autotuner picks per shape (k1_variant 2 / 3).
constexpr int PX_WIN_F4 = 10096, PX_WIN_F3 = 13400;

// CW = channels per window pass (16, or 8 for C = 8).  C = 32 is two passes over the views (`nhalf` = 2), one per
// channel half: the half's 16 reference channels are (re)loaded, every view is projected / boxed / staged / sampled
// for that half, and the accumulators carry over.  The projection and tap math are repeated for the second half
// (~80 of ~250 instructions per sample) -- the price of a register footprint that holds 4 waves per SIMD (keeping
// all 32 reference channels and both halves' code in one loop body spilled ~190 registers under any bound).
template <int CW, int DC>
__global__ __launch_bounds__(256, 4) void warp_corr_px_kernel(WarpArgs a, int C, int PX_WIN_F) {
    constexpr int QW = CW / 4;              // data quads per window pixel
    constexpr int PSQ = QW + 1;             // pixel stride in quads (odd: 5 or 3)
    constexpr int PSB = PSQ * 16;           // ... in bytes
    constexpr int TW = 32, TH = 8;
    extern __shared__ __attribute__((aligned(16))) float px_smem[];  // [32 floats of reduction scratch][PX_WIN_F window]
    float (*red)[4][4] = reinterpret_cast<float (*)[4][4]>(px_smem);  // per-wave boxes, double buffered by iteration
                                                                       // parity (an empty box skips the other barriers)
    float* const win = px_smem + 32;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool hi4 = (lane & 4) != 0;
    const int W = a.W, H = a.H, PS = a.pix_stride;
    const int x = blockIdx.x * TW + (tid & 31), y = blockIdx.y * TH + (tid >> 5);
    const int d0 = blockIdx.z * DC;
    const bool live = x < W && y < H;
    const int xc = min(x, W - 1), yc = min(y, H - 1);
    const size_t plane = (size_t)H * W;
    const float fx = (float)xc, fy = (float)yc;
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    const float half_w = wm1 / 2.0f, half_h = hm1 / 2.0f;
    const float inv_half_w = 1.0f / half_w, inv_half_h = 1.0f / half_h;
    const float wf = (float)W, hf = (float)H;

    float dep[DC];
#pragma unroll
    for (int j = 0; j < DC; ++j) dep[j] = hyp_plane(a, min(d0 + j, a.D - 1), plane, (size_t)yc * W + xc, a.depth ? 0.f : a.step[0]);
    float acc0[DC], acc1[DC];
#pragma unroll
    for (int j = 0; j < DC; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }

    const int nhalf = C / CW;
    int it = 0;  // (half, view) iteration counter: parity selects the reduction scratch
    for (int h = 0; h < nhalf; ++h) {
    float4_t r4[QW];
    {
        const float4_t* rp = reinterpret_cast<const float4_t*>(a.ref + ((size_t)yc * W + xc) * PS + h * CW);
#pragma unroll
        for (int q = 0; q < QW; ++q) r4[q] = rp[q];
    }
    for (int v = 0; v < a.nsrc; ++v, ++it) {
        const float* P = a.proj + v * 12;
        const float rx = fmaf(P[1], fy, P[0] * fx) + P[2];
        const float ry = fmaf(P[4], fy, P[3] * fx) + P[5];
        const float rz = fmaf(P[7], fy, P[6] * fx) + P[8];
        float ix[DC], iy[DC];
        float mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
#pragma unroll
        for (int j = 0; j < DC; ++j) {
            // the reference's op order (module.py:233-241 + ATen's un-normalise), divisions as in the kernel above
            const float px = rx * dep[j] + P[9];
            const float py = ry * dep[j] + P[10];
            float pz = rz * dep[j] + P[11];
            if (pz == 0.0f) pz += 0.00001f;
            float rz1 = __builtin_amdgcn_rcpf(pz);
            rz1 = fmaf(fmaf(-pz, rz1, 1.0f), rz1, rz1);
            const float gx = fdiv_rn(fdiv_rn(px, pz, rz1), half_w, inv_half_w) - 1.0f;
            const float gy = fdiv_rn(fdiv_rn(py, pz, rz1), half_h, inv_half_h) - 1.0f;
            ix[j] = ((gx + 1.0f) / 2.0f) * wm1;
            iy[j] = ((gy + 1.0f) / 2.0f) * hm1;
            // clamped to one pixel outside the image: everything further out only has zero-weight taps
            const float cx = fminf(fmaxf(ix[j], -1.0f), wf), cy = fminf(fmaxf(iy[j], -1.0f), hf);
            mnx = fminf(mnx, cx); mxx = fmaxf(mxx, cx);
            mny = fminf(mny, cy); mxy = fmaxf(mxy, cy);
        }
        mnx = wave_minmax_f<true>(mnx, hi4); mxx = wave_minmax_f<false>(mxx, hi4);
        mny = wave_minmax_f<true>(mny, hi4); mxy = wave_minmax_f<false>(mxy, hi4);
        float (*rb)[4] = red[it & 1];
        if (lane == 0) { rb[wave][0] = mnx; rb[wave][1] = mxx; rb[wave][2] = mny; rb[wave][3] = mxy; }
        __syncthreads();  // boxes of all waves visible; every wave is done sampling the previous window
        mnx = fminf(fminf(rb[0][0], rb[1][0]), fminf(rb[2][0], rb[3][0]));
        mxx = fmaxf(fmaxf(rb[0][1], rb[1][1]), fmaxf(rb[2][1], rb[3][1]));
        mny = fminf(fminf(rb[0][2], rb[1][2]), fminf(rb[2][2], rb[3][2]));
        mxy = fmaxf(fmaxf(rb[0][3], rb[1][3]), fmaxf(rb[2][3], rb[3][3]));
        // in-image columns / rows touched by a tap with non-zero weight: floor(min) .. floor(max) + 1
        const int bx0 = max((int)floorf(mnx), 0), bx1 = min((int)floorf(mxx) + 1, W - 1);
        const int by0 = max((int)floorf(mny), 0), by1 = min((int)floorf(mxy) + 1, H - 1);
        if (bx0 > bx1 || by0 > by1) continue;  // the whole tile projects outside the image (uniform)
        const int BW = bx1 - bx0 + 1, BH = by1 - by0 + 1, npix = BW * BH;
        const bool fits = npix * PSQ * 4 <= PX_WIN_F;

        if (fits) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[v], (short)0, H * W * PS * 4, 0x00020000);
            const int nslots = npix * PSQ;
            const float inv_bw = 1.0f / (float)BW;
            for (int i = wave; i * 64 < nslots; i += 4) {
                const int e = i * 64 + lane;
                const int k = (PSQ == 5) ? (int)(((unsigned)e * 52429u) >> 18) : (int)(((unsigned)e * 43691u) >> 17);  // e / PSQ
                const int q = e - k * PSQ;
                int r = (int)((float)k * inv_bw);
                r += (__mul24(r + 1, BW) <= k) ? 1 : 0;  // the float quotient is off by at most one
                r -= (__mul24(r, BW) > k) ? 1 : 0;
                const int col = k - __mul24(r, BW);
                // the pad quad of a pixel comes from an out-of-range offset: zero, no memory traffic
                const unsigned off = q < QW ? (unsigned)((__mul24(by0 + r, W) + bx0 + col) * PS + h * CW + q * 4) * 4u : 0x80000000u;
                if (e < nslots)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(win + i * 256), 16, off, 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const char* B = reinterpret_cast<const char*>(win);
#pragma unroll
            for (int j = 0; j < DC; ++j) {
                TapMath<CW> t;
                t.set(ix[j], iy[j], wm1, hm1);
                // zero-weight taps outside the image may lie outside the window: clamp their address into it
                const int ax0 = med3i(t.x0, bx0, bx1) - bx0, ax1 = med3i(t.x0 + 1, bx0, bx1) - bx0;
                const int r0 = __mul24(med3i(t.y0, by0, by1) - by0, BW), r1 = __mul24(med3i(t.y0 + 1, by0, by1) - by0, BW);
                const int to[4] = {__mul24(r0 + ax0, PSB), __mul24(r0 + ax1, PSB), __mul24(r1 + ax0, PSB), __mul24(r1 + ax1, PSB)};
                float e4[4], o4[4];
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    const float4_t* sp = reinterpret_cast<const float4_t*>(B + to[tp]);
                    float e = 0.f, o = 0.f;
#pragma unroll
                    for (int q = 0; q < QW; ++q) {
                        const float4_t s = sp[q];
                        e = fmaf(s.x, r4[q].x, e); o = fmaf(s.y, r4[q].y, o);
                        e = fmaf(s.z, r4[q].z, e); o = fmaf(s.w, r4[q].w, o);
                    }
                    e4[tp] = e; o4[tp] = o;
                }
                acc0[j] = fmaf(t.w00, e4[0], fmaf(t.w01, e4[1], fmaf(t.w10, e4[2], fmaf(t.w11, e4[3], acc0[j]))));
                acc1[j] = fmaf(t.w00, o4[0], fmaf(t.w01, o4[1], fmaf(t.w10, o4[2], fmaf(t.w11, o4[3], acc1[j]))));
            }
        } else {
            // window too large for the LDS: taps straight from global memory, same arithmetic in the same order
#pragma unroll
            for (int j = 0; j < DC; ++j) {
                TapMath<CW> t;
                t.set(ix[j], iy[j], wm1, hm1);
                const int gx0 = med3i(t.x0, 0, W - 1), gx1 = med3i(t.x0 + 1, 0, W - 1);
                const int g0 = med3i(t.y0, 0, H - 1) * W, g1 = med3i(t.y0 + 1, 0, H - 1) * W;
                const int to[4] = {g0 + gx0, g0 + gx1, g1 + gx0, g1 + gx1};
                float e4[4], o4[4];
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    const float4_t* sp = reinterpret_cast<const float4_t*>(a.src[v] + (size_t)to[tp] * PS + h * CW);
                    float e = 0.f, o = 0.f;
#pragma unroll
                    for (int q = 0; q < QW; ++q) {
                        const float4_t s = sp[q];
                        e = fmaf(s.x, r4[q].x, e); o = fmaf(s.y, r4[q].y, o);
                        e = fmaf(s.z, r4[q].z, e); o = fmaf(s.w, r4[q].w, o);
                    }
                    e4[tp] = e; o4[tp] = o;
                }
                acc0[j] = fmaf(t.w00, e4[0], fmaf(t.w01, e4[1], fmaf(t.w10, e4[2], fmaf(t.w11, e4[3], acc0[j]))));
                acc1[j] = fmaf(t.w00, o4[0], fmaf(t.w01, o4[1], fmaf(t.w10, o4[2], fmaf(t.w11, o4[3], acc1[j]))));
            }
        }
    }
    }

    const float inv = 2.0f / (float)C;
#pragma unroll
    for (int j = 0; j < DC; ++j) {
        if (live && d0 + j < a.D) {
            const size_t o = (size_t)(d0 + j) * plane + (size_t)y * W + x;
            float v0 = acc0[j] * inv, v1 = acc1[j] * inv;
            if (a.accumulate) { v0 += a.sim[o]; v1 += a.sim[(size_t)a.D * plane + o]; }
            a.sim[o] = v0;
            a.sim[(size_t)a.D * plane + o] = v1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Quad-planar variant ("q4", the product's kernel since r03).  Features are stored as C/4 planes of 16-byte channel
// quads, [C/4][H][W][4] (FeatureNet's output epilogue writes that directly): a window row of one quad plane is one
// contiguous run in HBM AND in LDS, consecutive pixels are consecutive 16-byte LDS slots (a ds_read_b128 service group of
// 16 lanes = 16 neighbouring pixels hits 16 different bank quads without any padding), and a window may hold only SOME
// of the planes -- which is what makes the window fit whatever the hypotheses look like:
//   * a lane owns a pixel; a workgroup owns a 32 x 8 tile x DC planes and walks the source views;
//   * the window of (tile, chunk, view) is bounded WITHOUT a per-sample reduction: the projection is a ratio of
//     functions that are linear in x, in y and in the depth separately, so over the box [tile] x [dmin, dmax] (dmin /
//     dmax = the tile's hypothesis range, reduced once per workgroup) its extremes sit at the 8 corners as long as
//     the denominator keeps its sign there.  Eight lanes per view project the corners (all views of a launch at once,
//     one 8-lane group each), every wave does so redundantly: no barrier, no cross-wave exchange per view;
//   * mode m = 0..log2(C/4): the window is staged in 2^m channel slabs of (C/4) >> m quad planes, each plane getting
//     2^m times the pixels -- incoherent hypotheses (refine passes of an untrained network: window = tile + depth
//     scatter) cost extra passes over the same taps instead of the global-memory gather of the r02 kernels.  The tap
//     position (two fractions + one LDS offset per plane) is computed once per (view, plane) and kept in registers
//     across the slabs;
//   * coordinates: p(d) = rot (x, y, 1) d + trans with FMAs, ix = px / pz by reciprocal + one residual step
//     (correctly rounded for normal operands), no normalise / un-normalise round trip (the contract is
//     1e-3 rel-L1 on depth, the tap position moves by < 1e-4 px).  Out-of-image taps need no range tests: the
//     coordinate is clamped to [-1, W] x [-1, H] and the window carries the zero border (staged from out-of-range
//     buffer offsets, no memory traffic), so a clamped sample reads zeros exactly where the reference's padding does.
//   * a box with a non-positive denominator at a corner, or a window beyond one quad plane of LDS, takes the exact
//     global-tap path (reference semantics incl. the z == 0 patch) for that (tile, chunk, view).
namespace q4 {
constexpr int TW = 32, TH = 8;
constexpr float kBoxEps = 1.0f / 64.0f;   // slack on the corner bounds (rounding of the corner vs interior projections)

template <bool IS_MIN> __device__ __forceinline__ float grp8_minmax(float v, bool hi4) {
    auto op = [](float a, float b) { return IS_MIN ? fminf(a, b) : fmaxf(a, b); };
    v = op(v, dpp_f<dpp_quad(1, 0, 3, 2)>(v));
    v = op(v, dpp_f<dpp_quad(2, 3, 0, 1)>(v));
    { const float a = dpp_f<kRowShl4>(v), b = dpp_f<kRowShr4>(v); v = op(v, hi4 ? b : a); }
    asm volatile("" : "+v"(v));
    return v;
}

// ray of a pixel through view P (rot @ (x, y, 1), module.py:233) and its point at depth `dep` (module.py:234-239)
__device__ __forceinline__ void ray(const float* P, float fx, float fy, float& rx, float& ry, float& rz) {
    rx = fmaf(P[0], fx, fmaf(P[1], fy, P[2]));
    ry = fmaf(P[3], fx, fmaf(P[4], fy, P[5]));
    rz = fmaf(P[6], fx, fmaf(P[7], fy, P[8]));
}
__device__ __forceinline__ void plane_pt(float rx, float ry, float rz, float t0, float t1, float t2, float dep,
                                         float& ix, float& iy, float& pz) {
    const float px = fmaf(rx, dep, t0), py = fmaf(ry, dep, t1);
    pz = fmaf(rz, dep, t2);
    float r = __builtin_amdgcn_rcpf(pz);
    r = fmaf(fmaf(-pz, r, 1.0f), r, r);
    ix = fdiv_rn(px, pz, r);
    iy = fdiv_rn(py, pz, r);
}

template <int V> struct ic { static constexpr int value = V; };
}  // namespace q4

// NQ = C / 4 quad planes; DC hypothesis planes per workgroup; WINQ = LDS window capacity in 16-byte quads
template <int NQ, int DC, int WINQ, int MINW>
__global__ __launch_bounds__(256, MINW) void warp_corr_q4_kernel(WarpArgs a, int ntx, int nty, int nch) {
    using namespace q4;
    constexpr int C = NQ * 4;
    extern __shared__ __attribute__((aligned(16))) float q4_smem[];   // [WINQ quads of window][8 floats of scratch]
    float* const win = q4_smem;
    float* const red = q4_smem + WINQ * 4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool hi4 = (lane & 4) != 0;
    const int W = a.W, H = a.H;
    // XCD-aware order (common.h): XCD k walks the k-th eighth of the list (chunk fastest, then tile x, tile y), so all
    // chunks of a tile and its neighbours -- whose windows overlap -- share one L2
    const int n = ntx * nty * nch, per = (n + 7) >> 3;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (t >= n) return;
    const int chunk = t % nch, tile = t / nch;
    const int tbx = tile % ntx, tby = tile / ntx;
    const int x = tbx * TW + (tid & 31), y = tby * TH + (tid >> 5);
    const int d0 = chunk * DC;
    const bool live = x < W && y < H;
    const int xc = min(x, W - 1), yc = min(y, H - 1);
    const size_t plane = (size_t)H * W;
    const float fx = (float)xc, fy = (float)yc;
    const float wf = (float)W, hf = (float)H;

    float dep[DC];
    float dmin = INFINITY, dmax = -INFINITY;
    {
        const float step = a.depth ? 0.f : a.step[0];
#pragma unroll
        for (int j = 0; j < DC; ++j) {
            dep[j] = hyp_plane(a, min(d0 + j, a.D - 1), plane, (size_t)yc * W + xc, step);
            dmin = fminf(dmin, dep[j]);
            dmax = fmaxf(dmax, dep[j]);
        }
    }
    float4_t r4[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) r4[q] = *reinterpret_cast<const float4_t*>(a.ref + (((size_t)q * H + yc) * W + xc) * 4);
    float acc0[DC], acc1[DC];
#pragma unroll
    for (int j = 0; j < DC; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }

    // the tile's hypothesis range (once per workgroup)
    dmin = wave_minmax_f<true>(dmin, hi4);
    dmax = wave_minmax_f<false>(dmax, hi4);
    if (lane == 0) { red[wave * 2] = dmin; red[wave * 2 + 1] = dmax; }
    __syncthreads();
    dmin = fminf(fminf(red[0], red[2]), fminf(red[4], red[6]));
    dmax = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
    const float xlo = (float)(tbx * TW), xhi = (float)min(tbx * TW + TW - 1, W - 1);
    const float ylo = (float)(tby * TH), yhi = (float)min(tby * TH + TH - 1, H - 1);

    for (int vg = 0; vg < a.nsrc; vg += 8) {
        // window table of views vg .. vg+7: lane group g = lane / 8 projects the 8 corners of view vg + g
        int tb_x0, tb_x1, tb_y0, tb_y1, tb_ok;
        {
            const float* P = a.proj + min(vg + (lane >> 3), a.nsrc - 1) * 12;
            float rx, ry, rz, ix, iy, pz;
            ray(P, (lane & 1) ? xhi : xlo, (lane & 2) ? yhi : ylo, rx, ry, rz);
            plane_pt(rx, ry, rz, P[9], P[10], P[11], (lane & 4) ? dmax : dmin, ix, iy, pz);
            const float cx = __builtin_amdgcn_fmed3f(ix, -1.0f, wf), cy = __builtin_amdgcn_fmed3f(iy, -1.0f, hf);
            const float mnx = grp8_minmax<true>(cx, hi4), mxx = grp8_minmax<false>(cx, hi4);
            const float mny = grp8_minmax<true>(cy, hi4), mxy = grp8_minmax<false>(cy, hi4);
            const float pzm = grp8_minmax<true>(pz, hi4);
            tb_x0 = max((int)floorf(mnx - kBoxEps), -1);
            tb_x1 = min((int)floorf(mxx + kBoxEps) + 1, W + 1);
            tb_y0 = max((int)floorf(mny - kBoxEps), -1);
            tb_y1 = min((int)floorf(mxy + kBoxEps) + 1, H + 1);
            // the denominator must be positive at every corner (then it is inside the box, and the bounds hold); a NaN fails
            tb_ok = (pzm > 0.0f && mnx <= mxx && mny <= mxy) ? 1 : 0;
        }
        const int vend = min(vg + 8, a.nsrc);
        for (int v = vg; v < vend; ++v) {
            const int sel = (v - vg) * 8;
            const int bx0 = __builtin_amdgcn_readlane(tb_x0, sel), bx1 = __builtin_amdgcn_readlane(tb_x1, sel);
            const int by0 = __builtin_amdgcn_readlane(tb_y0, sel), by1 = __builtin_amdgcn_readlane(tb_y1, sel);
            const int ok = __builtin_amdgcn_readlane(tb_ok, sel);
            const int BW = bx1 - bx0 + 1, BH = by1 - by0 + 1, npix = BW * BH;
            const float* P = a.proj + v * 12;   // uniform: scalar loads
            float rx, ry, rz;
            ray(P, fx, fy, rx, ry, rz);
            const float t0 = P[9], t1 = P[10], t2 = P[11];
            const float* S = a.src[v];

            if (ok && npix <= WINQ) {
                // ---- LDS path
                float tx[DC], ty[DC];
                int off[DC], off1[DC];   // LDS byte offsets of a plane's upper / lower tap row
                const int BW16 = BW * 16;
                auto tap_info = [&]() {
                    const float bw16f = (float)BW16, basef = -(float)((by0 * BW + bx0) * 16);
#pragma unroll
                    for (int j = 0; j < DC; ++j) {
                        float ix, iy, pz;
                        plane_pt(rx, ry, rz, t0, t1, t2, dep[j], ix, iy, pz);
                        const float cx = __builtin_amdgcn_fmed3f(ix, -1.0f, wf), cy = __builtin_amdgcn_fmed3f(iy, -1.0f, hf);
                        const float x0f = floorf(cx), y0f = floorf(cy);
                        tx[j] = cx - x0f;
                        ty[j] = cy - y0f;
                        off[j] = (int)fmaf(y0f, bw16f, fmaf(x0f, 16.0f, basef));   // exact: < 2^24
                        // computed HERE, under the window's flight time, and kept: without the pin the compiler sinks
                        // the whole projection into every slab's sampling code
                        off1[j] = off[j] + BW16;
                        asm volatile("" : "+v"(tx[j]), "+v"(ty[j]), "+v"(off[j]), "+v"(off1[j]));
                    }
                };
                // stage quad planes q0 .. q0 + nqs - 1 of the window, plane pitch `planeq` quads
                auto stage = [&](int q0, int nqs, int planeq) {
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)S, (short)0, NQ * H * W * 16, 0x00020000);
                    const float inv_bw = 1.0f / (float)BW;
                    for (int i = wave; i * 64 < npix; i += 4) {
                        const int e = i * 64 + lane;
                        int r = (int)((float)e * inv_bw);
                        r += (__mul24(r + 1, BW) <= e) ? 1 : 0;  // the float quotient is off by at most one
                        r -= (__mul24(r, BW) > e) ? 1 : 0;
                        const int gx = bx0 + e - __mul24(r, BW), gy = by0 + r;
                        // zero border: columns -1, W, W+1 / rows -1, H, H+1 come from an out-of-range offset
                        const bool in = (unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H;
                        const unsigned o0 = in ? (unsigned)((__mul24(q0, H) + gy) * W + gx) * 16u : 0x80000000u;
                        if (e < npix) {
                            for (int qq = 0; qq < nqs; ++qq)
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(win + (qq * planeq + i * 64) * 4), 16,
                                                                         o0 + (unsigned)qq * (unsigned)(H * W * 16), 0, 0, 0);
                        }
                    }
                };
                // Sampling: a stream of DC * NQS units, a unit = the four taps of one plane in one quad plane (4 x
                // ds_read_b128 + 16 channel FMAs).  The loads run PF units ahead of the FMAs through a ring of
                // PF + 1 register sets; the scheduling barriers keep the compiler from hoisting EVERY load of the
                // chunk to the top (it did: 256 VGPRs + spills).
                auto sample = [&](auto nqs_t, auto planeq_t, auto q0_t) {
                    constexpr int NQS = decltype(nqs_t)::value, PLB = decltype(planeq_t)::value * 16, Q0 = decltype(q0_t)::value;
                    constexpr int NU = DC * NQS, PF = 2;
                    const char* B = reinterpret_cast<const char*>(win);
                    float4_t T[PF + 1][4];
                    auto issue = [&](int u) {   // u is a constant after unrolling
                        const int j = u / NQS, qq = u % NQS;
                        const char* p0 = B + off[j] + qq * PLB;
                        const char* p1 = B + off1[j] + qq * PLB;
                        T[u % (PF + 1)][0] = *reinterpret_cast<const float4_t*>(p0);
                        T[u % (PF + 1)][1] = *reinterpret_cast<const float4_t*>(p0 + 16);
                        T[u % (PF + 1)][2] = *reinterpret_cast<const float4_t*>(p1);
                        T[u % (PF + 1)][3] = *reinterpret_cast<const float4_t*>(p1 + 16);
                    };
#pragma unroll
                    for (int u = 0; u < PF && u < NU; ++u) issue(u);
                    float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int j = u / NQS, qq = u % NQS;
                        if (u + PF < NU) issue(u + PF);
                        const float4_t s00 = T[u % (PF + 1)][0], s01 = T[u % (PF + 1)][1];
                        const float4_t s10 = T[u % (PF + 1)][2], s11 = T[u % (PF + 1)][3];
                        const float4_t r = r4[Q0 + qq];
                        if (qq == 0) {
                            e0 = s00.x * r.x; o0 = s00.y * r.y; e1 = s01.x * r.x; o1 = s01.y * r.y;
                            e2 = s10.x * r.x; o2 = s10.y * r.y; e3 = s11.x * r.x; o3 = s11.y * r.y;
                        } else {
                            e0 = fmaf(s00.x, r.x, e0); o0 = fmaf(s00.y, r.y, o0); e1 = fmaf(s01.x, r.x, e1); o1 = fmaf(s01.y, r.y, o1);
                            e2 = fmaf(s10.x, r.x, e2); o2 = fmaf(s10.y, r.y, o2); e3 = fmaf(s11.x, r.x, e3); o3 = fmaf(s11.y, r.y, o3);
                        }
                        e0 = fmaf(s00.z, r.z, e0); o0 = fmaf(s00.w, r.w, o0); e1 = fmaf(s01.z, r.z, e1); o1 = fmaf(s01.w, r.w, o1);
                        e2 = fmaf(s10.z, r.z, e2); o2 = fmaf(s10.w, r.w, o2); e3 = fmaf(s11.z, r.z, e3); o3 = fmaf(s11.w, r.w, o3);
                        if (qq == NQS - 1) {
                            const float wx0 = 1.0f - tx[j], wy0 = 1.0f - ty[j];
                            const float w00 = wx0 * wy0, w01 = tx[j] * wy0, w10 = wx0 * ty[j], w11 = tx[j] * ty[j];
                            acc0[j] = fmaf(w00, e0, fmaf(w01, e1, fmaf(w10, e2, fmaf(w11, e3, acc0[j]))));
                            acc1[j] = fmaf(w00, o0, fmaf(w01, o1, fmaf(w10, o2, fmaf(w11, o3, acc1[j]))));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // the sums are formed HERE: unpinned, the compiler defers the FMAs of every slab past the next
                    // slabs' barriers and staging loops and keeps all their taps live (256 VGPRs + spills)
#pragma unroll
                    for (int j = 0; j < DC; ++j) asm volatile("" : "+v"(acc0[j]), "+v"(acc1[j]));
                };
                // mode M: 2^M slabs of NQ >> M quad planes, each plane with room for WINQ / (NQ >> M) pixels
                auto run_mode = [&](auto m_t) {
                    constexpr int M = decltype(m_t)::value, NQS = NQ >> M, PLQ = (WINQ / NQS) & ~3;
                    auto slab = [&](auto s_t) {
                        constexpr int SI = decltype(s_t)::value;
                        __syncthreads();   // every wave is done sampling the previous window
                        stage(SI * NQS, NQS, PLQ);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (SI == 0) tap_info();   // VALU work under the window's flight time
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __syncthreads();
                        __builtin_amdgcn_sched_barrier(0);
                        sample(ic<NQS>{}, ic<PLQ>{}, ic<SI * NQS>{});
                    };
                    slab(ic<0>{});
                    if constexpr (M >= 1) slab(ic<1>{});
                    if constexpr (M >= 2) { slab(ic<2>{}); slab(ic<3>{}); }
                    if constexpr (M >= 3) { slab(ic<4>{}); slab(ic<5>{}); slab(ic<6>{}); slab(ic<7>{}); }
                };
                if (npix * NQ <= WINQ) run_mode(ic<0>{});
                else if (NQ == 2 || npix * (NQ / 2) <= WINQ) run_mode(ic<1>{});
                else if constexpr (NQ >= 4) {
                    if (NQ == 4 || npix * (NQ / 4) <= WINQ) run_mode(ic<2>{});
                    else if constexpr (NQ >= 8) run_mode(ic<3>{});
                }
            } else {
                // ---- exact global-tap path (reference semantics: z == 0 patch, per-tap range tests)
#pragma unroll
                for (int j = 0; j < DC; ++j) {
                    if (d0 + j >= a.D) continue;
                    const float px = fmaf(rx, dep[j], t0), py = fmaf(ry, dep[j], t1);
                    float pz = fmaf(rz, dep[j], t2);
                    if (pz == 0.0f) pz += 0.00001f;  // module.py:237
                    TapMath<C> tm;
                    tm.set(px / pz, py / pz, wf - 1.0f, hf - 1.0f);
                    const int gx0 = med3i(tm.x0, 0, W - 1), gx1 = med3i(tm.x0 + 1, 0, W - 1);
                    const int g0 = med3i(tm.y0, 0, H - 1) * W, g1 = med3i(tm.y0 + 1, 0, H - 1) * W;
                    float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const float* Sq = S + (size_t)q * plane * 4;
                        const float4_t s00 = *reinterpret_cast<const float4_t*>(Sq + (size_t)(g0 + gx0) * 4);
                        const float4_t s01 = *reinterpret_cast<const float4_t*>(Sq + (size_t)(g0 + gx1) * 4);
                        const float4_t s10 = *reinterpret_cast<const float4_t*>(Sq + (size_t)(g1 + gx0) * 4);
                        const float4_t s11 = *reinterpret_cast<const float4_t*>(Sq + (size_t)(g1 + gx1) * 4);
                        const float4_t r = r4[q];
                        e0 = fmaf(s00.z, r.z, fmaf(s00.x, r.x, e0)); o0 = fmaf(s00.w, r.w, fmaf(s00.y, r.y, o0));
                        e1 = fmaf(s01.z, r.z, fmaf(s01.x, r.x, e1)); o1 = fmaf(s01.w, r.w, fmaf(s01.y, r.y, o1));
                        e2 = fmaf(s10.z, r.z, fmaf(s10.x, r.x, e2)); o2 = fmaf(s10.w, r.w, fmaf(s10.y, r.y, o2));
                        e3 = fmaf(s11.z, r.z, fmaf(s11.x, r.x, e3)); o3 = fmaf(s11.w, r.w, fmaf(s11.y, r.y, o3));
                    }
                    acc0[j] = fmaf(tm.w00, e0, fmaf(tm.w01, e1, fmaf(tm.w10, e2, fmaf(tm.w11, e3, acc0[j]))));
                    acc1[j] = fmaf(tm.w00, o0, fmaf(tm.w01, o1, fmaf(tm.w10, o2, fmaf(tm.w11, o3, acc1[j]))));
                    __builtin_amdgcn_sched_barrier(0);   // one plane's taps in flight, not the chunk's (registers)
                }
            }
        }
    }

    const float inv = 2.0f / (float)C;
#pragma unroll
    for (int j = 0; j < DC; ++j) {
        if (live && d0 + j < a.D) {
            const size_t o = (size_t)(d0 + j) * plane + (size_t)y * W + x;
            float v0 = acc0[j] * inv, v1 = acc1[j] * inv;
            if (a.accumulate) { v0 += a.sim[o]; v1 += a.sim[(size_t)a.D * plane + o]; }
            a.sim[o] = v0;
            a.sim[(size_t)a.D * plane + o] = v1;
        }
    }
}

// window capacity in quads for `wgs` workgroups per CU (160 KB of LDS; 32 bytes of scratch; a multiple of 8 quads)
constexpr int q4_winq(int wgs) { return (((160 * 1024 / wgs) - 64) / 16) & ~7; }

template <int NQ, int DC, int WGS>
static int launch_q4_v(const WarpArgs& a, hipStream_t st) {
    constexpr int WINQ = q4_winq(WGS);
    const int ntx = ceil_div(a.W, q4::TW), nty = ceil_div(a.H, q4::TH), nch = ceil_div(a.D, DC);
    const size_t lds = (size_t)WINQ * 16 + 32;
    // register budget: 4 waves per SIMD (128 VGPRs) with 4 planes per workgroup, 3 (168) with 8 -- but never more waves
    // than the LDS windows admit
    constexpr int MINW = (DC == 4 && WGS >= 4) ? 4 : (WGS >= 3 ? 3 : 2);
    auto kern = warp_corr_q4_kernel<NQ, DC, WINQ, MINW>;
    if (lds > 48 * 1024) {
        const int rc = dmvs_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
        if (rc) return rc;
    }
    kern<<<xcd_grid(ntx * nty * nch), 256, lds, st>>>(a, ntx, nty, nch);
    DMVS_LAUNCH_CHECK();
}

// variant: 0 default; 1 / 2 / 3: 4 / 3 / 2 workgroups per CU (40 / 53 / 80 KB windows); + 8: 4 planes per workgroup
// even when D > 4 (A/B knobs of scripts/k1_bench.py)
template <int NQ>
static int launch_q4(const WarpArgs& a, hipStream_t st, int variant) {
    const bool dc4 = a.D <= 4 || (variant & 8);
    switch (variant & 7) {
        case 2: return dc4 ? launch_q4_v<NQ, 4, 3>(a, st) : launch_q4_v<NQ, 8, 3>(a, st);
        case 3: return dc4 ? launch_q4_v<NQ, 4, 2>(a, st) : launch_q4_v<NQ, 8, 2>(a, st);
        default: return dc4 ? launch_q4_v<NQ, 4, 4>(a, st) : launch_q4_v<NQ, 8, 4>(a, st);
    }
}


// K1 variant: 0 = automatic, 1 = "lds" (channel-split lanes, small tiles), 2 / 3 = "px" (pixel per lane, 32 x 8 tiles)
// with a 39.5 KB window and 4 workgroups per CU / a 52 KB window and 3 workgroups per CU.
// Set by dmvs_tune("k1_variant", v) or the DMVS_K1 environment variable (lds | px) -- A/B runs and autotuning.
int g_k1_variant = [] {
    const char* e = getenv("DMVS_K1");
    return !e ? 0 : (e[0] == 'l' ? 1 : (e[0] == 'p' ? 2 : 0));
}();

template <int C>
static int launch_warp(const WarpArgs& a, hipStream_t st) {
    constexpr int LPP = C / 4, NPIX = 256 / LPP, TW = (C == 8) ? 16 : 8, TH = NPIX / TW;
    if ((long)a.H * a.W * a.pix_stride < (1L << 29)) {  // buffer-descriptor byte offsets
        if (g_k1_variant == 2 || g_k1_variant == 3) {
            dim3 grid(ceil_div(a.W, 32), ceil_div(a.H, 8), ceil_div(a.D, 4));
            const int win_f = g_k1_variant == 3 ? PX_WIN_F3 : PX_WIN_F4;
            warp_corr_px_kernel<(C > 16 ? 16 : C), 4><<<grid, 256, (win_f + 32) * sizeof(float), st>>>(a, C, win_f);
            DMVS_LAUNCH_CHECK();
        }
        const bool lbc = g_k1_variant == 4;
        if (a.D <= 4) {
            dim3 grid(ceil_div(a.W, TW), ceil_div(a.H, TH), ceil_div(a.D, 4));
            if (lbc) warp_corr_lds_kernel<C, 4, true><<<grid, 256, 0, st>>>(a);
            else warp_corr_lds_kernel<C, 4><<<grid, 256, 0, st>>>(a);
        } else {
            dim3 grid(ceil_div(a.W, TW), ceil_div(a.H, TH), ceil_div(a.D, 8));
            if (lbc) warp_corr_lds_kernel<C, 8, true><<<grid, 256, 0, st>>>(a);
            else warp_corr_lds_kernel<C, 8><<<grid, 256, 0, st>>>(a);
        }
        DMVS_LAUNCH_CHECK();
    }
    constexpr int DCHUNK = 8;
    constexpr int PPB = 256 / (C / 4);
    dim3 grid(ceil_div(a.W, PPB), a.H, ceil_div(a.D, DCHUNK));
    warp_corr_kernel<C, DCHUNK><<<grid, 256, 0, st>>>(a);
    DMVS_LAUNCH_CHECK();
}

static int warp_corr_entry(const float* ref_hwc, const float* const* src_hwc, int nsrc, int pix_stride, const float* proj12,
                           const float* depth_dhw, const float* base_hw, const float* step, float* sim_2dhw, int C, int D,
                           int H, int W, int accumulate, dmvs_stream_t stream) {
    if (!ref_hwc || !src_hwc || !proj12 || !sim_2dhw) return DMVS_EINVAL;
    if (!depth_dhw && (!base_hw || !step)) return DMVS_EINVAL;
    if (nsrc < 1 || nsrc > DMVS_MAX_SRC_VIEWS || D < 1 || H < 1 || W < 1) return DMVS_EINVAL;
    if (pix_stride < C || (pix_stride & 3)) return DMVS_EINVAL;
    WarpArgs a;
    a.ref = ref_hwc;
    for (int v = 0; v < DMVS_MAX_SRC_VIEWS; ++v) a.src[v] = v < nsrc ? src_hwc[v] : nullptr;
    for (int v = 0; v < nsrc; ++v)
        if (!a.src[v]) return DMVS_EINVAL;
    a.proj = proj12; a.depth = depth_dhw; a.base = base_hw; a.step = step; a.sim = sim_2dhw;
    a.nsrc = nsrc; a.pix_stride = pix_stride; a.D = D; a.H = H; a.W = W; a.accumulate = accumulate;
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 8: return launch_warp<8>(a, st);
        case 16: return launch_warp<16>(a, st);
        case 32: return launch_warp<32>(a, st);
        default: return DMVS_EUNSUPPORTED;
    }
}

extern "C" int dmvs_warp_corr(const float* ref_hwc, const float* const* src_hwc, int nsrc, int pix_stride,
                              const float* proj12, const float* depth_dhw, float* sim_2dhw, int C, int D, int H,
                              int W, int accumulate, dmvs_stream_t stream) {
    if (!depth_dhw) return DMVS_EINVAL;
    return warp_corr_entry(ref_hwc, src_hwc, nsrc, pix_stride, proj12, depth_dhw, nullptr, nullptr, sim_2dhw, C, D, H, W,
                           accumulate, stream);
}

extern "C" int dmvs_warp_corr_affine(const float* ref_hwc, const float* const* src_hwc, int nsrc, int pix_stride,
                                     const float* proj12, const float* base_hw, const float* step, float* sim_2dhw,
                                     int C, int D, int H, int W, int accumulate, dmvs_stream_t stream) {
    if (!base_hw || !step) return DMVS_EINVAL;
    return warp_corr_entry(ref_hwc, src_hwc, nsrc, pix_stride, proj12, nullptr, base_hw, step, sim_2dhw, C, D, H, W,
                           accumulate, stream);
}

extern "C" int dmvs_warp_corr_q4(const float* ref_q4, const float* const* src_q4, int nsrc, const float* proj12,
                                 const float* depth_dhw, const float* base_hw, const float* step, float* sim_2dhw,
                                 int C, int D, int H, int W, int accumulate, int variant, dmvs_stream_t stream) {
    if (!ref_q4 || !src_q4 || !proj12 || !sim_2dhw) return DMVS_EINVAL;
    if (!depth_dhw && (!base_hw || !step)) return DMVS_EINVAL;
    if (nsrc < 1 || nsrc > DMVS_MAX_SRC_VIEWS || D < 1 || H < 1 || W < 1) return DMVS_EINVAL;
    if ((long)H * W * C >= (1L << 29)) return DMVS_EUNSUPPORTED;   // buffer-descriptor byte offsets
    WarpArgs a;
    a.ref = ref_q4;
    for (int v = 0; v < DMVS_MAX_SRC_VIEWS; ++v) a.src[v] = v < nsrc ? src_q4[v] : nullptr;
    for (int v = 0; v < nsrc; ++v)
        if (!a.src[v]) return DMVS_EINVAL;
    a.proj = proj12; a.depth = depth_dhw; a.base = depth_dhw ? nullptr : base_hw; a.step = depth_dhw ? nullptr : step;
    a.sim = sim_2dhw;
    a.nsrc = nsrc; a.pix_stride = 4; a.D = D; a.H = H; a.W = W; a.accumulate = accumulate;
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 8: return launch_q4<2>(a, st, variant);
        case 16: return launch_q4<4>(a, st, variant);
        case 32: return launch_q4<8>(a, st, variant);
        default: return DMVS_EUNSUPPORTED;
    }
}
