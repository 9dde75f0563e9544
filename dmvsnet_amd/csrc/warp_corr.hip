// K1: fused inverse-homography warp + bilinear gather + 2-group correlation + view sum.
//
// Replaces CostAgg.forward (/root/reference/networks/mvsnet.py:111-153) and homo_warping
// (/root/reference/networks/module.py:212-251).  The reference materialises a [C][D][H][W] warped
// volume per view (0.5-1 GB), multiplies it by the reference feature and reduces it in two more
// passes; here one kernel reads the source features and writes only the [2][D][H][W] similarity volume.
//
// Two kernels:
//   * warp_corr_q4_kernel (dmvs_warp_corr_q4): the product path.  Quad-planar features [C/4][H][W][4], a lane owns
//     a pixel, per (tile, plane chunk, view) an LDS window bounded from the corners of the tile's (x, y, depth) box and
//     staged with LDS-direct loads in channel slabs; see the comment above the kernel.
//   * warp_corr_kernel (dmvs_warp_corr / dmvs_warp_corr_affine): pixel-major features with any pixel stride, every
//     bilinear tap through the vector L1 -- the generic form (callers that hold [H][W][C] maps, the on-device
//     cross-check of the parity tests).  A pixel is owned by C/4 adjacent lanes (one float4 of channels each), so a
//     tap load of a wave is 64/LPP contiguous C*4-byte runs; coordinates follow the reference's op order exactly
//     (rot*(x,y,1), *depth, +trans, z==0 -> +1e-5, /z, normalise to [-1,1], ATen's un-normalise).
// r02's LDS-window kernels on pixel-major features (channel-split lanes / pixel per lane, 4 autotuned variants) were
// removed in r03: the q4 kernel is 1.45-1.6x faster on every stage-pass of both benchmarks (profiles/r03_*k1*).
//
// The two correlation groups are the even / odd channels (mvsnet.py:139: view(b,c//2,2,...).mean(1)).  Taps outside
// the image contribute zero individually (padding_mode="zeros").  Per tap the channel dot products are formed first
// and then weighted (linear re-association of interpolate-then-multiply; differs by ~1 ulp).
#include "common.h"
#include "dev_guard.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

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
// Helpers of the product kernel below.
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Cross-lane helpers on DPP (VALU-rate lane movement folded into the consuming ALU op) and v_readlane, used
// instead of ds_bpermute-based __shfl (an LDS-pipe instruction with a waitcnt behind it) where the pattern is static.
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v))); }
constexpr int dpp_quad(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int kRowShl4 = 0x104, kRowShr4 = 0x114, kRowRor8 = 0x128;

// a / b given r ~ 1/b to within an ulp: q = a*r, one residual correction -> the correctly rounded quotient for
// normal-range operands (no scaling / denormal fix-up, which the 4 divisions of a projection never need)
__device__ __forceinline__ float fdiv_rn(float a, float b, float r) {
    const float q = a * r;
    return fmaf(fmaf(-b, q, a), r, q);
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

// wave-wide min / max as a wave-uniform value: butterfly inside the 16-lane rows, then the 4 row results
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
constexpr int TW = 32;
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

#ifdef DMVS_Q4_TRACE
// dev build only (scripts/dev/k1_trace.sh): per-workgroup s_memtime stamps of the kernel's phases
__device__ unsigned long long* g_q4_trace;
extern "C" int dmvs_dev_trace(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_q4_trace), &p, sizeof(p)); }
#define Q4_TR(slot) do { if (tid == 0 && g_q4_trace && t < 16384) g_q4_trace[(size_t)t * 32 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
// ... and which window mode every (tile, plane chunk, view) takes: counts[m] for slab mode m = 0..3, counts[4] for the exact
// global-tap path, counts[8 + m] the summed window sizes (16-byte pieces per quad plane) -- scripts/dev/k1_modes.py
__device__ unsigned long long* g_q4_modes;
extern "C" int dmvs_dev_modes(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_q4_modes), &p, sizeof(p)); }
#define Q4_COUNT(m, npix) do { if (tid == 0 && g_q4_modes) { atomicAdd(&g_q4_modes[(m)], 1ull); atomicAdd(&g_q4_modes[8 + (m)], (unsigned long long)(npix)); } } while (0)
#else
#define Q4_TR(slot) do { } while (0)
#define Q4_COUNT(m, npix) do { } while (0)
#endif

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
// one channel quad of a tap against the reference quad: even channels -> e, odd channels -> o
struct RefQuad32 { float4_t r; };
struct RefQuad16 { half2_t e0, e1, o0, o1; };   // (c0, 0), (c2, 0), (0, c1), (0, c3): operands of v_dot2_f32_f16
__device__ __forceinline__ void quad_dot(const float4_t& s, const RefQuad32& q, float& e, float& o) {
    e = fmaf(s.z, q.r.z, fmaf(s.x, q.r.x, e));
    o = fmaf(s.w, q.r.w, fmaf(s.y, q.r.y, o));
}
// (the tap is a typed half4: ROCm 7.2's hipcc mis-compiles bit_cast<half2>(u.y) of a uint2 loaded from LDS into a second
// copy of u.x -- the same bug as the int extraction noted in r02 -- so the halves are taken with a shuffle)
__device__ __forceinline__ void quad_dot(const half4_t& s, const RefQuad16& q, float& e, float& o) {
    const half2_t s0 = __builtin_shufflevector(s, s, 0, 1), s1 = __builtin_shufflevector(s, s, 2, 3);   // (c0, c1), (c2, c3)
    e = __builtin_amdgcn_fdot2(s1, q.e1, __builtin_amdgcn_fdot2(s0, q.e0, e, false), false);
    o = __builtin_amdgcn_fdot2(s1, q.o1, __builtin_amdgcn_fdot2(s0, q.o0, o, false), false);
}

// NQ = C / 4 quad planes; DC hypothesis planes per workgroup; WINQ = LDS window capacity in 16-byte quads; TH = tile
// rows (8: 256 threads).  HF: the features are fp16 (declared extension, BASELINE configs[4] "fp16 features"): 8 bytes
// per pixel quad in HBM and LDS -- half the window bytes, half the LDS read traffic -- products and sums in fp32
// (v_dot2_f32_f16 against the reference quad split into even / odd operands); windows are staged in 16-byte PAIRS of
// pixels, so their columns start even (W must be even).
template <int NQ, int DC, int WINQ, int TH, bool HF>
__global__ __launch_bounds__(32 * TH) void warp_corr_q4_kernel(WarpArgs a, int ntx, int nty, int nch) {
    using namespace q4;
    constexpr int C = NQ * 4, NW = TH / 2;   // waves per workgroup
    constexpr int PXB = HF ? 8 : 16;          // bytes of one pixel's channel quad
    typedef typename std::conditional<HF, half4_t, float4_t>::type tap_t;
    typedef typename std::conditional<HF, RefQuad16, RefQuad32>::type refq_t;
    extern __shared__ __attribute__((aligned(16))) float q4_smem[];   // [WINQ quads of window][16 floats of scratch]
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
    // lane -> pixel of the tile row: a ds_read_b128 is served in 16-lane groups {0-3, 12-15, 20-27} / {4-11, 16-19,
    // 28-31} (+32), conflict-free when the group's 16 quads fall on 16 different bank quads.  With the natural order a
    // group spans 28 pixels, and because the projection scales x by 1 +- a few per cent, quads 16 apart collide
    // (SQ_LDS_BANK_CONFLICT was 0.94x the conflict-free LDS cycles, profiles/r03_d_k1_sq.txt); with this permutation every
    // group owns 16 CONSECUTIVE pixels, whose taps stay within ~17 consecutive quads: 0.71x (r03_g; what remains is the
    // incoherent-hypothesis regime, where neighbouring pixels sample unrelated window positions.  Padding the row pitch to
    // 16 quads, so that a tap's bank would not depend on its row, changed nothing: 0.75x).
    const int h32 = tid & 31;
    const int px32 = h32 < 4 ? h32 : h32 < 12 ? h32 + 12 : h32 < 16 ? h32 - 8 : h32 < 20 ? h32 + 8 : h32 < 28 ? h32 - 12 : h32;
    const int x = tbx * TW + px32, y = tby * TH + (tid >> 5);
    const int d0 = chunk * DC;
    const bool live = x < W && y < H;
    const int xc = min(x, W - 1), yc = min(y, H - 1);
    const size_t plane = (size_t)H * W;
    const float fx = (float)xc, fy = (float)yc;
    const float wf = (float)W, hf = (float)H;

    Q4_TR(0);
#ifdef DMVS_Q4_TRACE
    if (tid == 0 && g_q4_trace && t < 16384) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); g_q4_trace[(size_t)t * 32 + 31] = hw; }
#endif
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
    refq_t r4[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const char* rp = reinterpret_cast<const char*>(a.ref) + (((size_t)q * H + yc) * W + xc) * PXB;
        if constexpr (HF) {
            const uint2_t h = *reinterpret_cast<const uint2_t*>(rp);
            r4[q].e0 = __builtin_bit_cast(half2_t, h.x & 0x0000ffffu); r4[q].o0 = __builtin_bit_cast(half2_t, h.x & 0xffff0000u);
            r4[q].e1 = __builtin_bit_cast(half2_t, h.y & 0x0000ffffu); r4[q].o1 = __builtin_bit_cast(half2_t, h.y & 0xffff0000u);
        } else {
            r4[q].r = *reinterpret_cast<const float4_t*>(rp);
        }
    }
    float acc0[DC], acc1[DC];
#pragma unroll
    for (int j = 0; j < DC; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }

    // the tile's hypothesis range (once per workgroup)
    dmin = wave_minmax_f<true>(dmin, hi4);
    dmax = wave_minmax_f<false>(dmax, hi4);
    if (lane == 0) { red[wave * 2] = dmin; red[wave * 2 + 1] = dmax; }
    __syncthreads();
    Q4_TR(1);
#pragma unroll
    for (int w = 0; w < NW; ++w) { dmin = fminf(dmin, red[2 * w]); dmax = fmaxf(dmax, red[2 * w + 1]); }
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
            const int bx0r = __builtin_amdgcn_readlane(tb_x0, sel), bx1 = __builtin_amdgcn_readlane(tb_x1, sel);
            const int by0 = __builtin_amdgcn_readlane(tb_y0, sel), by1 = __builtin_amdgcn_readlane(tb_y1, sel);
            const int ok = __builtin_amdgcn_readlane(tb_ok, sel);
            const int bx0 = HF ? (bx0r & ~1) : bx0r;                       // fp16: windows are staged in pixel PAIRS
            const int BW = HF ? ((bx1 - bx0 + 2) & ~1) : bx1 - bx0 + 1, BH = by1 - by0 + 1;
            const int npix = HF ? (BW >> 1) * BH : BW * BH;              // window size in 16-byte pieces per quad plane
            const float* P = a.proj + v * 12;   // uniform: scalar loads
            float rx, ry, rz;
            ray(P, fx, fy, rx, ry, rz);
            const float t0 = P[9], t1 = P[10], t2 = P[11];
            const char* S = reinterpret_cast<const char*>(a.src[v]);

            if (ok && npix <= WINQ) {
                // ---- LDS path
                float tx[DC], ty[DC];
                int off[DC];
                const int BW16 = BW * PXB;   // window row pitch in bytes
                auto tap_info = [&]() {
                    const float bw16f = (float)BW16, basef = -(float)((by0 * BW + bx0) * PXB);
#pragma unroll
                    for (int j = 0; j < DC; ++j) {
                        float ix, iy, pz;
                        plane_pt(rx, ry, rz, t0, t1, t2, dep[j], ix, iy, pz);
                        const float cx = __builtin_amdgcn_fmed3f(ix, -1.0f, wf), cy = __builtin_amdgcn_fmed3f(iy, -1.0f, hf);
                        const float x0f = floorf(cx), y0f = floorf(cy);
                        tx[j] = cx - x0f;
                        ty[j] = cy - y0f;
                        off[j] = (int)fmaf(y0f, bw16f, fmaf(x0f, (float)PXB, basef));   // exact: < 2^24
                        // kept across the slabs: unpinned, the compiler re-derives the whole projection in every slab
                        asm volatile("" : "+v"(tx[j]), "+v"(ty[j]), "+v"(off[j]));
                    }
                };
                // stage quad planes q0 .. q0 + nqs - 1 of the window, plane pitch `planeq` quads
                auto stage = [&](int q0, int nqs, int planeq) {
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)S, (short)0, NQ * H * W * PXB, 0x00020000);
                    const int PPR = HF ? BW >> 1 : BW;           // 16-byte pieces per window row (fp16: pixel pairs)
                    const float inv_bw = 1.0f / (float)PPR;
                    for (int i = wave; i * 64 < npix; i += NW) {
                        const int e = i * 64 + lane;
                        int r = (int)((float)e * inv_bw);
                        r += (__mul24(r + 1, PPR) <= e) ? 1 : 0;  // the float quotient is off by at most one
                        r -= (__mul24(r, PPR) > e) ? 1 : 0;
                        const int gx = bx0 + (e - __mul24(r, PPR)) * (HF ? 2 : 1), gy = by0 + r;
                        // zero border: columns -1, W, W+1 / rows -1, H, H+1 come from an out-of-range offset (fp16: gx and
                        // W are even, a pair is inside or outside as a whole)
                        const bool in = (unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H;
                        const unsigned o0 = in ? (unsigned)((__mul24(q0, H) + gy) * W + gx) * (unsigned)PXB : 0x80000000u;
                        if (e < npix) {
                            for (int qq = 0; qq < nqs; ++qq)
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(win + (qq * planeq + i * 64) * 4), 16,
                                                                         o0 + (unsigned)qq * (unsigned)(H * W * PXB), 0, 0, 0);
                        }
                    }
                };
                auto sample = [&](auto nqs_t, auto planeq_t, auto q0_t) {
                    constexpr int NQS = decltype(nqs_t)::value, PLB = decltype(planeq_t)::value * 16, Q0 = decltype(q0_t)::value;
                    const char* B = reinterpret_cast<const char*>(win);
#pragma unroll
                    for (int j = 0; j < DC; ++j) {
                        // (uniform branch; it also keeps the planes apart in the schedule: one plane's taps in flight
                        // per wave and 102-132 VGPRs -- an explicitly software-pipelined version of this loop needed
                        // 140-210 and was slower: occupancy hides the LDS latency better than intra-wave overlap)
                        if (d0 + j >= a.D) continue;
                        const float wx0 = 1.0f - tx[j], wy0 = 1.0f - ty[j];
                        const float w00 = wx0 * wy0, w01 = tx[j] * wy0, w10 = wx0 * ty[j], w11 = tx[j] * ty[j];
                        const char* p0 = B + off[j];
                        const char* p1 = p0 + BW16;
                        float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
                        for (int qq = 0; qq < NQS; ++qq) {
                            const tap_t s00 = *reinterpret_cast<const tap_t*>(p0 + qq * PLB);
                            const tap_t s01 = *reinterpret_cast<const tap_t*>(p0 + qq * PLB + PXB);
                            const tap_t s10 = *reinterpret_cast<const tap_t*>(p1 + qq * PLB);
                            const tap_t s11 = *reinterpret_cast<const tap_t*>(p1 + qq * PLB + PXB);
                            quad_dot(s00, r4[Q0 + qq], e0, o0);
                            quad_dot(s01, r4[Q0 + qq], e1, o1);
                            quad_dot(s10, r4[Q0 + qq], e2, o2);
                            quad_dot(s11, r4[Q0 + qq], e3, o3);
                        }
                        acc0[j] = fmaf(w00, e0, fmaf(w01, e1, fmaf(w10, e2, fmaf(w11, e3, acc0[j]))));
                        acc1[j] = fmaf(w00, o0, fmaf(w01, o1, fmaf(w10, o2, fmaf(w11, o3, acc1[j]))));
                    }
                };
                // mode M: 2^M slabs of NQ >> M quad planes, each plane with room for WINQ / (NQ >> M) pixels
                auto run_mode = [&](auto m_t) {
                    constexpr int M = decltype(m_t)::value, NQS = NQ >> M, PLQ = (WINQ / NQS) & ~3;
                    Q4_COUNT(M, npix);
                    auto slab = [&](auto s_t) {
                        constexpr int SI = decltype(s_t)::value;
                        __syncthreads();   // every wave is done sampling the previous window
                        if (v == 0 && SI == 0) Q4_TR(2);
                        stage(SI * NQS, NQS, PLQ);
                        if (v == 0 && SI == 0) Q4_TR(3);
                        if constexpr (SI == 0) tap_info();   // VALU work under the window's flight time
                        if (v == 0 && SI == 0) Q4_TR(4);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (v == 0 && SI == 0) Q4_TR(5);
                        __syncthreads();
                        if (v == 0 && SI == 0) Q4_TR(6);
                        sample(ic<NQS>{}, ic<PLQ>{}, ic<SI * NQS>{});
                        if (v == 0 && SI == 0) Q4_TR(7);
                        if (v == 1 && SI == 0) Q4_TR(8);
                        if (v == 2 && SI == 0) Q4_TR(9);
                        if (v == 3 && SI == 0) Q4_TR(10);
                    };
                    slab(ic<0>{});
                    if constexpr (M >= 1) slab(ic<1>{});
                    if constexpr (M >= 2) { slab(ic<2>{}); slab(ic<3>{}); }
                    if constexpr (M >= 3) { slab(ic<4>{}); slab(ic<5>{}); slab(ic<6>{}); slab(ic<7>{}); }
                };
                // a mode is chosen by the plane pitch it really uses (PLQ above: rounded DOWN to 4 quads), not by
                // npix * NQS <= WINQ -- windows of PLQ+1 .. WINQ/NQS pieces would overlap the next quad plane;
                // the last mode has one plane per slab, pitch = WINQ >= npix
                constexpr auto plq = [](int m) { return (WINQ / (NQ >> m)) & ~3; };
                if (npix <= plq(0)) run_mode(ic<0>{});
                else if (NQ == 2 || npix <= plq(1)) run_mode(ic<1>{});
                else if constexpr (NQ >= 4) {
                    if (NQ == 4 || npix <= plq(2)) run_mode(ic<2>{});
                    else if constexpr (NQ >= 8) run_mode(ic<3>{});
                }
            } else {
                // ---- exact global-tap path (reference semantics: z == 0 patch, per-tap range tests)
                Q4_COUNT(4, npix);
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
                        const char* Sq = S + (size_t)q * plane * PXB;
                        const tap_t s00 = *reinterpret_cast<const tap_t*>(Sq + (size_t)(g0 + gx0) * PXB);
                        const tap_t s01 = *reinterpret_cast<const tap_t*>(Sq + (size_t)(g0 + gx1) * PXB);
                        const tap_t s10 = *reinterpret_cast<const tap_t*>(Sq + (size_t)(g1 + gx0) * PXB);
                        const tap_t s11 = *reinterpret_cast<const tap_t*>(Sq + (size_t)(g1 + gx1) * PXB);
                        quad_dot(s00, r4[q], e0, o0);
                        quad_dot(s01, r4[q], e1, o1);
                        quad_dot(s10, r4[q], e2, o2);
                        quad_dot(s11, r4[q], e3, o3);
                    }
                    acc0[j] = fmaf(tm.w00, e0, fmaf(tm.w01, e1, fmaf(tm.w10, e2, fmaf(tm.w11, e3, acc0[j]))));
                    acc1[j] = fmaf(tm.w00, o0, fmaf(tm.w01, o1, fmaf(tm.w10, o2, fmaf(tm.w11, o3, acc1[j]))));
                }
            }
        }
    }

    Q4_TR(11);
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
    Q4_TR(12);
}

// window capacity in quads for `wgs` workgroups per CU (160 KB of LDS; 32 bytes of scratch; a multiple of 8 quads)
constexpr int q4_winq(int wgs) { return (((160 * 1024 / wgs) - 64) / 16) & ~7; }

template <int NQ, int DC, int WGS, int TH, bool HF>
static int launch_q4_v(const WarpArgs& a, hipStream_t st) {
    constexpr int WINQ = q4_winq(WGS);
    const int ntx = ceil_div(a.W, q4::TW), nty = ceil_div(a.H, TH), nch = ceil_div(a.D, DC);
    const size_t lds = (size_t)WINQ * 16 + 64;
    auto kern = warp_corr_q4_kernel<NQ, DC, WINQ, TH, HF>;
    if (lds > 48 * 1024) {
        const int rc = dmvs_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
        if (rc) return rc;
    }
    kern<<<xcd_grid(ntx * nty * nch), 32 * TH, lds, st>>>(a, ntx, nty, nch);
    DMVS_LAUNCH_CHECK();
}

// variant: 0 default (4 workgroups per CU, 40 KB windows; 8 planes per workgroup for C = 8, 4 for C >= 16 -- what the
// r03 A/B on MI355X picked, scripts/dev/k1_q4.py); low 3 bits 1 / 2 / 3 / 4: 4 / 3 / 2 / 1 workgroups per CU (40 / 53 /
// 80 / 160 KB windows); + 8: 4 planes per workgroup, + 16: 8 planes (A/B and test knobs)
template <int NQ, bool HF>
static int launch_q4(const WarpArgs& a, hipStream_t st, int variant) {
    const bool dc4 = a.D <= 4 || (variant & 8) || (!(variant & 16) && NQ > 2);
    // default windows: 40 KB (4 workgroups per CU); C = 8 with many source views -- far views on incoherent planes need
    // windows beyond 40 KB and would fall to the global-tap path -- 80 KB (measured at 11 views: s3.main 1.29 -> 1.07 ms,
    // s3.refine 0.98 -> 0.84; the C >= 16 passes lose with larger windows)
    const int wsel = (variant & 7) == 0 ? ((NQ == 2 && a.nsrc > 6) ? 3 : 1) : (variant & 7);
#define Q4_CASE(W_, WGS_) \
    if (wsel == W_) return dc4 ? launch_q4_v<NQ, 4, WGS_, 8, HF>(a, st) : launch_q4_v<NQ, 8, WGS_, 8, HF>(a, st);
    Q4_CASE(1, 4)
    Q4_CASE(2, 3)
    Q4_CASE(3, 2)
    Q4_CASE(4, 1)
#undef Q4_CASE
    return DMVS_EINVAL;
}

// pixel-major features ("HWC", any pixel stride): the generic kernel, taps through the vector L1
template <int C>
static int launch_warp(const WarpArgs& a, hipStream_t st) {
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

static int warp_corr_q4_entry(const void* ref_q4, const void* const* src_q4, int nsrc, const float* proj12,
                              const float* depth_dhw, const float* base_hw, const float* step, float* sim_2dhw,
                              int C, int D, int H, int W, int accumulate, int variant, bool f16, dmvs_stream_t stream) {
    if (!ref_q4 || !src_q4 || !proj12 || !sim_2dhw) return DMVS_EINVAL;
    if (!depth_dhw && (!base_hw || !step)) return DMVS_EINVAL;
    if (nsrc < 1 || nsrc > DMVS_MAX_SRC_VIEWS || D < 1 || H < 1 || W < 1) return DMVS_EINVAL;
    if ((long)H * W * C >= (1L << 29)) return DMVS_EUNSUPPORTED;   // buffer-descriptor byte offsets
    if (f16 && (W & 1)) return DMVS_EUNSUPPORTED;                  // fp16 windows are staged in pixel pairs
    WarpArgs a;
    a.ref = reinterpret_cast<const float*>(ref_q4);
    for (int v = 0; v < DMVS_MAX_SRC_VIEWS; ++v) a.src[v] = v < nsrc ? reinterpret_cast<const float*>(src_q4[v]) : nullptr;
    for (int v = 0; v < nsrc; ++v)
        if (!a.src[v]) return DMVS_EINVAL;
    a.proj = proj12; a.depth = depth_dhw; a.base = depth_dhw ? nullptr : base_hw; a.step = depth_dhw ? nullptr : step;
    a.sim = sim_2dhw;
    a.nsrc = nsrc; a.pix_stride = 4; a.D = D; a.H = H; a.W = W; a.accumulate = accumulate;
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 8: return f16 ? launch_q4<2, true>(a, st, variant) : launch_q4<2, false>(a, st, variant);
        case 16: return f16 ? launch_q4<4, true>(a, st, variant) : launch_q4<4, false>(a, st, variant);
        case 32: return f16 ? launch_q4<8, true>(a, st, variant) : launch_q4<8, false>(a, st, variant);
        default: return DMVS_EUNSUPPORTED;
    }
}

extern "C" int dmvs_warp_corr_q4(const float* ref_q4, const float* const* src_q4, int nsrc, const float* proj12,
                                 const float* depth_dhw, const float* base_hw, const float* step, float* sim_2dhw,
                                 int C, int D, int H, int W, int accumulate, int variant, dmvs_stream_t stream) {
    return warp_corr_q4_entry(ref_q4, reinterpret_cast<const void* const*>(src_q4), nsrc, proj12, depth_dhw, base_hw, step,
                              sim_2dhw, C, D, H, W, accumulate, variant, false, stream);
}

extern "C" int dmvs_warp_corr_q4_f16(const void* ref_q4h, const void* const* src_q4h, int nsrc, const float* proj12,
                                     const float* depth_dhw, const float* base_hw, const float* step, float* sim_2dhw,
                                     int C, int D, int H, int W, int accumulate, int variant, dmvs_stream_t stream) {
    return warp_corr_q4_entry(ref_q4h, src_q4h, nsrc, proj12, depth_dhw, base_hw, step, sim_2dhw, C, D, H, W, accumulate,
                              variant, true, stream);
}
