// K3s -- FeatureNet's two full-resolution layers (module.py:283-286: conv0 = 3 -> 8 -> 8 channels, 3x3, stride 1, BN + ReLU)
// as a register-only row sweep on v_mfma_f32_4x4x1_16b_f32.
//
// Why not K3: with 8 output channels the 16-row MFMA of conv_mfma_kernel runs half empty (conv0.0 / conv0.1 sat at 21-25 %
// of the fp32 MFMA peak, 2.3-2.9x their floors, profiles/r04_n_layer_table.md).  The 4x4x1 instruction is 16 independent
// 4x4 outer products per issue: block b = lane / 4 multiplies 4 rows (the A operand of lanes 4b .. 4b+3) by 4 columns (the
// B operand of the same lanes).  With A = four output channels' weights (the same in every block) and B = the lane's OWN
// pixel, one issue is 4 couts x 64 pixels x 1 (channel, tap) with no empty rows, the accumulator of lane l holds 4 output
// channels of pixel l, and the whole layer needs neither LDS nor a barrier:
//   * one wave = one 64-pixel strip of image columns (62 of them stored: the two edge lanes only feed their neighbours),
//     walked down R rows; every input row is loaded ONCE per strip into registers (lane = pixel, coalesced 256-byte rows),
//     the three rows of the stencil stay in registers, the row after next is in flight under the current row's MFMAs;
//   * the kx = -1 / +1 taps are the neighbour lanes' partial sums: v_mov_b32_dpp wave_shr:1 / wave_shl:1 on the results;
//   * zero padding = the buffer descriptor's range check (an out-of-image lane or row gets offset 2^31 and reads 0);
//   * weights: 2 x 9 x CIN VGPRs per lane (cout = 4h + lane % 4), loaded once per wave.
// fp32 throughout; the MFMA accumulates one exact-product fmaf per (channel, tap) in a fixed order.
#include "common.h"

// dev knock-outs (scripts/dev/c8_ko.sh): 1 = no HBM reads (every row offset out of range), 2 = no stores, 4 = one MFMA pair
// per (channel, row) instead of three (a third of the matrix work), 8 = no neighbour-lane moves
#ifndef DMVS_C8_KO
#define DMVS_C8_KO 0
#endif

namespace {

struct C8Args {
    const float* in;
    float* out;
    const float* w;       // packed by dmvs_pack_conv_weights_c8: [k = (ci, ky, kx)][lane % 4][h] -> cout 4h + lane % 4
    const float* scale;   // folded BN (8 each); nullptr = identity
    const float* shift;
    int V, H, W;
    int in_cs, in_zs;     // element strides of an input channel / a view ([C][V][H][W] planar: V*H*W, H*W; image stack: H*W, 3*H*W)
    int in_elems;         // extent of the input buffer
    int R, nrb, nstrips;  // rows per wave, row blocks, strips
    int relu;
};

// out-of-range markers: a row and a lane marker on different bits, so that marker + marker + in-range offset (< 2^30 bytes,
// checked by the launcher) neither wraps nor lands inside the buffer -- every load and store is unconditional
constexpr unsigned kOob = 0x80000000u, kOobX = 0x40000000u;
constexpr int STRIP = 62;   // stored pixels per wave
#ifndef C8_NS
#define C8_NS 4
#endif
#include "dev_guard.h"
template <int V> struct ic { static constexpr int value = V; };

__device__ __forceinline__ float lane_left(float v) {    // value of lane l - 1 (pixel x - 1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_right(float v) {   // value of lane l + 1 (pixel x + 1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// one output row: 8 channels of the lane's pixel from the three input rows r0 (y - 1), r1 (y), r2 (y + 1).
// The kx = -1 / +1 taps do not shift the INPUT (48 neighbour-lane moves per row, each feeding the next MFMA: measured, they
// do not hide under the other wave's MFMAs) but the OUTPUT: every lane multiplies its own pixel by all three columns of the
// filter into three partial sums P_kx, and out[x] = P_0[x - 1] + P_1[x] + P_2[x + 1] -- 16 moves per row, the MFMA operands
// come straight from the loaded registers, and six independent accumulators take turns.
template <int CIN, typename WK>
__device__ __forceinline__ void c8_row_w(const float (&r0)[CIN], const float (&r1)[CIN], const float (&r2)[CIN],
                                         WK&& wk, float4_t& acc0, float4_t& acc1) {
    float4_t p[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j) { p[j][0] = {0.f, 0.f, 0.f, 0.f}; p[j][1] = {0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float c = ky == 0 ? r0[ci] : ky == 1 ? r1[ci] : r2[ci];
            const int k = (ci * 3 + ky) * 3;
#pragma unroll
            for (int j = 0; j < ((DMVS_C8_KO & 4) ? 1 : 3); ++j) {
                const float2_t wv = wk(k + j);
                p[j][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.x, c, p[j][0], 0, 0, 0);
                p[j][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.y, c, p[j][1], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (DMVS_C8_KO & 8) { acc0[e] = (p[0][0][e] + p[1][0][e]) + p[2][0][e]; acc1[e] = (p[0][1][e] + p[1][1][e]) + p[2][1][e]; continue; }
        acc0[e] = (lane_left(p[0][0][e]) + p[1][0][e]) + lane_right(p[2][0][e]);
        acc1[e] = (lane_left(p[0][1][e]) + p[1][1][e]) + lane_right(p[2][1][e]);
    }
}

template <int CIN>
__device__ __forceinline__ void c8_row(const float (&r0)[CIN], const float (&r1)[CIN], const float (&r2)[CIN],
                                       const float2_t (&w)[CIN * 9], float4_t& acc0, float4_t& acc1) {
    c8_row_w<CIN>(r0, r1, r2, [&](int k) { return w[k]; }, acc0, acc1);
}

template <int CIN>
__global__ __launch_bounds__(64) void conv2d_c8_kernel(C8Args a) {
    const int lane = threadIdx.x;
    // XCD-aware order (common.h): XCD k walks the k-th eighth of the (view, row block, strip) list, strips fastest -- the
    // 256-byte row segments of neighbouring strips straddle cache lines, and so do their 248-byte stores: on ONE L2 the line
    // is fetched once and the partial writes merge before they reach HBM
#ifndef C8_XCD
#define C8_XCD 1
#endif
    const int nwg = a.nstrips * a.nrb * a.V;
    const int t = C8_XCD ? (int)(blockIdx.x & 7) * ((nwg + 7) >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (t >= nwg) return;
    const int strip = t % a.nstrips;
    const int rb = (t / a.nstrips) % a.nrb, v = t / (a.nstrips * a.nrb);
    const int W = a.W, H = a.H;
#ifdef C8_FAKE_ALIGNED   // dev timing experiment only (wrong values at strip edges): 64 aligned pixels per wave, all stored
    const int x = strip * 64 + lane;
#else
    const int x = strip * STRIP - 1 + lane;
#endif
    const bool xin = x >= 0 && x < W;
    const int y0 = rb * a.R, y1 = y0 + a.R;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, a.in_elems * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, (short)0, 8 * a.V * H * W * 4, 0x00020000);

    float2_t w[CIN * 9];
    {
        const float2_t* wp = reinterpret_cast<const float2_t*>(a.w) + (lane & 3);
#pragma unroll
        for (int k = 0; k < CIN * 9; ++k) w[k] = wp[k * 4];
    }
    const unsigned xoff = xin ? (unsigned)(v * a.in_zs + x) * 4u : kOobX;
    const unsigned cs4 = (unsigned)a.in_cs * 4u;
    auto load_row = [&](float (&dst)[CIN], int y) {
        const unsigned off = xoff + ((y >= 0 && y < H && !(DMVS_C8_KO & 1)) ? (unsigned)(y * W) * 4u : kOob);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
            dst[ci] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, off + ci * cs4, 0, 0));
    };
    float sc[8], sh[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { sc[c] = a.scale ? a.scale[c] : 1.0f; sh[c] = a.shift ? a.shift[c] : 0.0f; }

    // NS row slots, rotated statically over a xNS-unrolled loop: rows y - 1, y, y + 1 feed output row y while rows up to
    // y + NS - 2 are in flight.  The loads are issued BEFORE the previous row's stores retire (vmcnt counts both, in order),
    // so waiting for a row never waits for a younger store.  Measured and not kept: 2 or 3 rows in flight (NS = 5, 6), and the
    // row's 8 loads / 8 stores spread through its MFMA stream instead of issued in two bursts -- both neutral.  What the sweep
    // does not reach is the memory system's rate for this pattern: a pure strip copy of the same planes moves 4.8 TB/s (the
    // plain copy 6.3), the 62-pixel strips another 10 % less (their 248-byte rows straddle cache lines)
    constexpr int NS = C8_NS;
    float rows[NS][CIN];
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) load_row(rows[i], y0 - 1 + i);
#ifdef C8_FAKE_ALIGNED
    const bool store_lane = x < W;
#else
    const bool store_lane = lane >= 1 && lane <= STRIP && x < W;
#endif
    const unsigned plane4 = (unsigned)(a.V * H * W) * 4u;
    auto step = [&](auto s_t, int y) {
        constexpr int S = decltype(s_t)::value;
        load_row(rows[(S + NS - 1) % NS], y + NS - 2);
        __builtin_amdgcn_sched_barrier(0);
        float4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        c8_row<CIN>(rows[S % NS], rows[(S + 1) % NS], rows[(S + 2) % NS], w, acc0, acc1);
        const unsigned o = (store_lane && y < H && !((DMVS_C8_KO & 2) && acc0[0] != 1.2345f)) ? (unsigned)((v * H + y) * W + x) * 4u : kOob;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float t = fmaf(c < 4 ? acc0[c] : acc1[c - 4], sc[c], sh[c]);
            if (a.relu) t = fmaxf(t, 0.0f);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), rs_out, o + c * plane4, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int y = y0; y < y1; y += NS) {   // R is a multiple of NS; rows past the image load zeros and store nothing
        step(ic<0>{}, y);
        step(ic<1>{}, y + 1);
        step(ic<2>{}, y + 2);
        step(ic<3>{}, y + 3);
        if constexpr (NS > 4) step(ic<4>{}, y + 4);
        if constexpr (NS > 5) step(ic<5>{}, y + 5);
    }
}

// ---- conv0.0 -> conv0.1 in one sweep: the 8-channel intermediate (303 MB per depth map at config 2) lives in registers.
// Lane l of a strip is pixel x = 60 * strip - 2 + l: the image row is loaded on all 64 lanes, the intermediate is right on lanes
// 1..62 (its edge lanes miss a neighbour), the output on lanes 2..61 -- 60 stored pixels per wave.  Per output row y: the
// intermediate row y + 1 is formed from image rows y .. y + 2 (row y + 3 in flight) and masked to ZERO outside the image (it is
// conv0.1's zero padding, not conv0.0 evaluated out there), then output row y from intermediate rows y - 1 .. y + 1.
struct C8FusedArgs {
    const float* img;     // [V][3][H][W]
    float* out;           // [8][V][H][W]
    const float *w0, *scale0, *shift0;   // dmvs_pack_conv_weights_c8(Cin = 3), folded BN
    const float *w1, *scale1, *shift1;   // (Cin = 8)
    int V, H, W, R, nrb, nstrips;
};
constexpr int STRIPF = 60;

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv0_fused_kernel(C8FusedArgs a) {
    const int lane = threadIdx.x;
    const int nwg = a.nstrips * a.nrb * a.V;
    const int t = (int)(blockIdx.x & 7) * ((nwg + 7) >> 3) + (int)(blockIdx.x >> 3);
    if (t >= nwg) return;
    const int strip = t % a.nstrips;
    const int rb = (t / a.nstrips) % a.nrb, v = t / (a.nstrips * a.nrb);
    const int W = a.W, H = a.H;
    const int x = strip * STRIPF - 2 + lane;
    const bool xin = x >= 0 && x < W;
    const int y0 = rb * a.R, y1 = y0 + a.R;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.img, (short)0, 3 * a.V * H * W * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, (short)0, 8 * a.V * H * W * 4, 0x00020000);

    // conv0.1's weights in registers (144), conv0.0's (54 more would mean one wave per SIMD) in LDS: 27 broadcast reads per row
    __shared__ float2_t w0s[27 * 4];
    float2_t w1[72];
    {
        const float2_t* p0 = reinterpret_cast<const float2_t*>(a.w0);
        const float2_t* p1 = reinterpret_cast<const float2_t*>(a.w1) + (lane & 3);
        w0s[lane] = p0[lane];
        if (lane < 27 * 4 - 64) w0s[64 + lane] = p0[64 + lane];
#pragma unroll
        for (int k = 0; k < 72; ++k) w1[k] = p1[k * 4];
        __syncthreads();
    }
    const float2_t* w0l = w0s + (lane & 3);
    float sc0[8], sh0[8], sc1[8], sh1[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { sc0[c] = a.scale0[c]; sh0[c] = a.shift0[c]; sc1[c] = a.scale1[c]; sh1[c] = a.shift1[c]; }

    const unsigned xoff = xin ? (unsigned)(v * 3 * H * W + x) * 4u : kOobX;
    const unsigned cs4 = (unsigned)(H * W) * 4u;
    float img[4][3], mid[4][8];
    auto load_img = [&](float (&dst)[3], int y) {
        const unsigned off = xoff + ((y >= 0 && y < H) ? (unsigned)(y * W) * 4u : kOob);
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) dst[ci] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, off + ci * cs4, 0, 0));
    };
    // intermediate row r from image rows r - 1, r, r + 1
    auto make_mid = [&](float (&dst)[8], const float (&i0)[3], const float (&i1)[3], const float (&i2)[3], int r) {
        float4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        c8_row_w<3>(i0, i1, i2, [&](int k) { return w0l[k * 4]; }, acc0, acc1);
        const bool ok = xin && r >= 0 && r < H;
#pragma unroll
        for (int c = 0; c < 8; ++c) dst[c] = ok ? fmaxf(fmaf(c < 4 ? acc0[c] : acc1[c - 4], sc0[c], sh0[c]), 0.0f) : 0.0f;
    };
    // image row r lives in slot (r - y0 + 2) & 3, intermediate row r in slot (r - y0 + 1) & 3
    load_img(img[0], y0 - 2);
    load_img(img[1], y0 - 1);
    load_img(img[2], y0);
    load_img(img[3], y0 + 1);
    make_mid(mid[0], img[0], img[1], img[2], y0 - 1);
    load_img(img[0], y0 + 2);
    make_mid(mid[1], img[1], img[2], img[3], y0);
    const bool store_lane = lane >= 2 && lane < 2 + STRIPF && x < W;
    const unsigned plane4 = (unsigned)(a.V * H * W) * 4u;
    auto step = [&](auto s_t, int y) {
        constexpr int S = decltype(s_t)::value;
        load_img(img[(S + 1) & 3], y + 3);
        __builtin_amdgcn_sched_barrier(0);
        make_mid(mid[(S + 2) & 3], img[(S + 2) & 3], img[(S + 3) & 3], img[S & 3], y + 1);
        float4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        c8_row<8>(mid[S & 3], mid[(S + 1) & 3], mid[(S + 2) & 3], w1, acc0, acc1);
        const unsigned o = (store_lane && y < H) ? (unsigned)((v * H + y) * W + x) * 4u : kOob;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float r = fmaxf(fmaf(c < 4 ? acc0[c] : acc1[c - 4], sc1[c], sh1[c]), 0.0f);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), rs_out, o + c * plane4, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int y = y0; y < y1; y += 4) {
        step(ic<0>{}, y);
        step(ic<1>{}, y + 1);
        step(ic<2>{}, y + 2);
        step(ic<3>{}, y + 3);
    }
}

}  // namespace
long g_c8_rows = 0;   // dmvs_tune("c8_rows"): rows per wave, 0 = chosen from the wave count
namespace {

// rows per wave: enough waves for `rounds` full passes over the chip's wave slots (2 per SIMD), 12..32 rows each
void c8_rows(int V, int H, int W, int strip, int& R, int& nrb, int& nstrips) {
    nstrips = ceil_div(W, strip);
    if (g_c8_rows > 0) { R = ceil_div((int)g_c8_rows, C8_NS) * C8_NS; nrb = ceil_div(H, R); return; }
    const int slots = 256 * 4 * 2, sv = nstrips * V;
    R = 32;
    for (int k = 1; k <= 16; ++k) {
        const int n = k * slots / sv;
        if (n < 1) continue;
        const int r = ceil_div(H, n);
        if (r <= 24) { R = r < 8 ? 8 : r; break; }
    }
    R = ceil_div(R, C8_NS) * C8_NS;
    nrb = ceil_div(H, R);
}

}  // namespace

extern "C" long dmvs_conv2d_c8_weight_floats(int Cin) { return (Cin == 3 || Cin == 8) ? (long)Cin * 9 * 8 : 0; }

// w [8][Cin][3][3] (the nn.Conv2d layout) -> [k = (ci, ky, kx)][j = lane % 4][h]: cout = 4h + j
extern "C" int dmvs_pack_conv_weights_c8(const float* w, float* out, int Cin) {
    if (!w || !out || dmvs_conv2d_c8_weight_floats(Cin) == 0) return DMVS_EUNSUPPORTED;
    size_t n = 0;
    for (int ci = 0; ci < Cin; ++ci)
        for (int t = 0; t < 9; ++t)
            for (int j = 0; j < 4; ++j)
                for (int h = 0; h < 2; ++h) out[n++] = w[((size_t)(4 * h + j) * Cin + ci) * 9 + t];
    return 0;
}

extern "C" int dmvs_conv2d_c8(const float* in, float* out, const float* w_packed, const float* scale, const float* shift,
                              int Cin, int V, int H, int W, int flags, dmvs_stream_t stream) {
    if (!in || !out || !w_packed || V < 1 || H < 1 || W < 1) return DMVS_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return DMVS_EINVAL;
    if (flags & ~(DMVS_RELU | DMVS_IN_VIEWS)) return DMVS_EUNSUPPORTED;
    if (Cin != 3 && Cin != 8) return DMVS_EUNSUPPORTED;
    if ((flags & DMVS_IN_VIEWS) && Cin != 3) return DMVS_EUNSUPPORTED;
    if ((long)8 * V * H * W >= (1L << 28)) return DMVS_EUNSUPPORTED;   // byte offsets below the out-of-range markers
    C8Args a = {};
    a.in = in; a.out = out; a.w = w_packed; a.scale = scale; a.shift = shift;
    a.V = V; a.H = H; a.W = W; a.relu = (flags & DMVS_RELU) ? 1 : 0;
    if (flags & DMVS_IN_VIEWS) { a.in_cs = H * W; a.in_zs = 3 * H * W; }
    else { a.in_cs = V * H * W; a.in_zs = H * W; }
    a.in_elems = Cin * V * H * W;
#ifdef C8_FAKE_ALIGNED
    c8_rows(V, H, W, 64, a.R, a.nrb, a.nstrips);
#else
    c8_rows(V, H, W, STRIP, a.R, a.nrb, a.nstrips);
#endif
    const unsigned grid = xcd_grid(a.nstrips * a.nrb * V);
    if (Cin == 3) conv2d_c8_kernel<3><<<grid, 64, 0, (hipStream_t)stream>>>(a);
    else conv2d_c8_kernel<8><<<grid, 64, 0, (hipStream_t)stream>>>(a);
    DMVS_LAUNCH_CHECK();
}

extern "C" int dmvs_featurenet_conv0(const float* imgs, float* out, const float* w0_packed, const float* scale0, const float* shift0,
                                     const float* w1_packed, const float* scale1, const float* shift1, int V, int H, int W,
                                     dmvs_stream_t stream) {
    if (!imgs || !out || !w0_packed || !scale0 || !shift0 || !w1_packed || !scale1 || !shift1 || V < 1 || H < 1 || W < 1) return DMVS_EINVAL;
    if ((long)8 * V * H * W >= (1L << 28)) return DMVS_EUNSUPPORTED;
    C8FusedArgs a = {};
    a.img = imgs; a.out = out; a.w0 = w0_packed; a.scale0 = scale0; a.shift0 = shift0; a.w1 = w1_packed; a.scale1 = scale1; a.shift1 = shift1;
    a.V = V; a.H = H; a.W = W;
    c8_rows(V, H, W, STRIPF, a.R, a.nrb, a.nstrips);
    a.R = ceil_div(a.R, 4) * 4; a.nrb = ceil_div(H, a.R);
    conv0_fused_kernel<<<xcd_grid(a.nstrips * a.nrb * V), 64, 0, (hipStream_t)stream>>>(a);
    DMVS_LAUNCH_CHECK();
}
