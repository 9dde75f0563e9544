// K3w: the stride-1 3x3(x3) convolutions in Winograd F(2x2, 3x3) form on the fp32 matrix cores.
//
// Replaces the same reference code as K3 (/root/reference/networks/module.py:120-157 Conv3d/Conv2d + BatchNorm(eval) +
// ReLU) for the stride-1 3x3 layers: conv2 / conv4 / conv6 of CostRegNet_part(_refine) (module.py:364, 367, 370 and
// 406, 409, 412) and FeatureNet's conv1.1/1.2, conv2.1/2.2, out2 (module.py:291-292, 296-297, 309).  These layers are
// bound by the fp32 MFMA rate (53-71 % of the 157 TF peak in r03's layer table, 10-13 % of HBM), so the lever left is
// the number of multiplies: F(2x2, 3x3) forms each 2x2 output patch from 16 products per (cin, cout, kz) instead of
// 36 -- 2.25x fewer MFMA k-steps -- and the transforms around them are additions the vector ALUs do in the MFMAs' shadow.
// Everything stays fp32: inputs and products are fp32 (v_mfma_f32_16x16x4_f32), the transforms use +, - and the
// exactly representable factors 1/2 and 1/4 folded into the host-transformed weights.  Against the direct form the
// result differs at re-association level (measured in tests/test_gpu_parity.py::test_conv3d_wino: <= 2e-5 of the
// layer's output scale, the tolerance of the direct kernels' own tests).
//
//   Y = A^T [ sum_{ci,kz} (G g G^T) .* (B^T d B) ] A          per 2x2 output patch ("tile"), d = its 4x4 input patch
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// GEMM view: for each of the 16 transform positions xi, M_xi[tile][cout] = sum_{kz, ci} V_xi[tile][kz, ci] * U_xi[kz, ci][cout]
//   -> 16 independent accumulators per (16 tiles x 16 couts) block, v_mfma_f32_16x16x4_f32 with the tiles as rows
//      (A operand: lane = (tile n = lane % 16, channel k = lane / 16)) and the output channels as columns.
//   A lane reads the 4x4 patch of ITS tile and channel from the LDS input tile (12 aligned ds_read_b64), transforms it
//   in registers (32 additions) and feeds the 16 MFMAs of every (kz -> output plane, cout block) that uses the plane;
//   the transformed weights come from LDS as 4 ds_read_b128 per 16 MFMAs, packed by the host in consumption order.
//   In D a lane holds 4 consecutive tiles of one output channel: after the output transform (24 additions per tile)
//   it owns 8 consecutive x of two output rows -> two 16-byte stores per row.
// A 256-thread workgroup owns TZ output planes x TY = 2 * NTR * TRW rows x 32 columns; its 4 waves are NTR tile rows
// (x TRW per wave) times NWM groups of MBW 16-channel blocks.  Input tile and weight slice of a 4*GPC-channel chunk are
// staged with 16-byte LDS-direct loads exactly as in K3 (tile_loader.h), one or two LDS stages.
// Needs W % 4 == 0 and a 16-byte aligned input (the 16-byte tile loader); otherwise DMVS_EUNSUPPORTED and the caller
// runs the direct-form K3 kernel.
#include "common.h"
#include "tile_loader.h"

#include <cmath>
#include <vector>

namespace {

typedef float acc4_t __attribute__((ext_vector_type(4)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));

struct WinoArgs {
    const float* in;
    float* out;
    const float* w;
    const float* scale;
    const float* shift;
    int Cin, Cout, D, H, W, relu;
    int nx, ny, nz;
    int single_buf;
};

template <int KD, int MB, int MBW, int TZ, int TRW, int GPC>
struct WinoGeom {
    static constexpr int NWM = MB / MBW;        // wave groups along the output channels
    static constexpr int NTR = 4 / NWM;         // wave groups along y
    static constexpr int TY = 2 * NTR * TRW;    // output rows of a workgroup
    static constexpr int IZ = KD == 3 ? TZ + 2 : TZ, IY = TY + 2;
    static constexpr int LPR = 10, IXP = 4 * LPR;   // rows start 4 floats left of the first output column (16-byte aligned)
    static constexpr int PS0 = IZ * IY * IXP;
    static constexpr int PS = PS0 + (32 - PS0 % 64 + 64) % 64;   // channel stride = 32 (mod 64) banks: see the patch reads
    static constexpr int CI_CH = 4 * GPC;
    static constexpr int TILE_F = (CI_CH * PS + 63) & ~63;
    static constexpr int WROWS = KD * GPC * MB * 16;   // 64-float rows of transformed weights per chunk
    static constexpr int BUF_F = TILE_F + WROWS * 64;
};

// one chunk's weight slice (NROWS rows of 64 floats, consumption order): 16-byte LDS-direct loads, 1 KiB per
// wave-instruction (same scheme as K3's load_weights)
template <int NROWS>
__device__ __forceinline__ void load_rows64(__amdgpu_buffer_rsrc_t rs_w, float* wl, int chunk, int wave, int lane) {
    constexpr int NI = (NROWS + 3) / 4;
    const unsigned base = (unsigned)chunk * NROWS * 256u + (unsigned)lane * 16u;
#pragma unroll
    for (int r = 0; r < (NI + 3) / 4; ++r) {
        const int j = min(wave + 4 * r, NI - 1);
        if (j * 256 + lane * 4 < NROWS * 64)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(wl + j * 256), 16, base + (unsigned)j * 1024u, 0, 0, 0);
    }
}

template <int KD, int MB, int MBW, int TZ, int TRW, int GPC>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(WinoArgs a) {
    typedef WinoGeom<KD, MB, MBW, TZ, TRW, GPC> G;
    constexpr int IY = G::IY, IZ = G::IZ, IXP = G::IXP, PS = G::PS, NTR = G::NTR;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [1 or 2][BUF_F]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lk = lane >> 4;
    const int trg = wave % NTR, mg = wave / NTR;   // the wave's tile-row group and output-channel group
    int bx, by, bz;
    if (!xcd_tile(a.nx, a.ny, a.nz, KD == 3, bx, by, bz)) return;
    const int ox0 = bx * 32, oy0 = by * G::TY, oz0 = bz * TZ;
    const int ix0a = ox0 - 4, iy0 = oy0 - 1, iz0 = KD == 3 ? oz0 - 1 : oz0;

    // The lane's patch of tile n (output columns 2n, 2n+1) spans tile columns 3 + 2n .. 6 + 2n: read as the three aligned
    // pairs starting at 2 + 2n.  ds_read_b64 is served in two 32-lane groups with bank = dword address mod 64: the 16
    // tiles of one channel cover 32 consecutive banks, the second channel of the group sits PS = 32 (mod 64) further.
    const int pbase = lk * PS + (2 * TRW * trg) * IXP + 2 + 2 * ln;

    acc4_t acc[TZ][TRW][MBW][16];
#pragma unroll
    for (int z = 0; z < TZ; ++z)
#pragma unroll
        for (int t = 0; t < TRW; ++t)
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
                for (int x = 0; x < 16; ++x) acc[z][t][mb][x] = (acc4_t){0.f, 0.f, 0.f, 0.f};

    const int in_vol = a.D * a.H * a.W;
    const int nchunks = a.Cin / G::CI_CH;
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, nchunks * G::WROWS * 256, 0x00020000);
    auto stage = [&](int c, float* dst) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.in + (size_t)(c * G::CI_CH) * in_vol), (short)0, G::CI_CH * in_vol * 4, 0x00020000);
        load_tile4<G::CI_CH, IZ, IY, G::LPR, PS>(a.D, a.H, a.W, rs, dst, iz0, iy0, ix0a, wave, lane);
        load_rows64<G::WROWS>(rs_w, dst + G::TILE_F, c, wave, lane);
    };

    stage(0, smem);
    for (int c = 0; c < nchunks; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float* cur = smem + (a.single_buf ? 0 : (c & 1)) * G::BUF_F;
        if (c + 1 < nchunks && !a.single_buf) stage(c + 1, smem + ((c + 1) & 1) * G::BUF_F);
        const float* tile = cur + pbase;
        const float* wl = cur + G::TILE_F + lane * 4;
#pragma unroll
        for (int g = 0; g < GPC; ++g)
#pragma unroll
            for (int pz = 0; pz < IZ; ++pz)
#pragma unroll
                for (int t = 0; t < TRW; ++t) {
                    const float* p = tile + g * 4 * PS + (pz * IY + 2 * t) * IXP;
                    float d[4][4];
#pragma unroll
                    for (int y = 0; y < 4; ++y) {
                        const float2_t q0 = *reinterpret_cast<const float2_t*>(p + y * IXP);
                        const float2_t q1 = *reinterpret_cast<const float2_t*>(p + y * IXP + 2);
                        const float2_t q2 = *reinterpret_cast<const float2_t*>(p + y * IXP + 4);
                        d[y][0] = q0.y; d[y][1] = q1.x; d[y][2] = q1.y; d[y][3] = q2.x;
                    }
                    float v[16];
#pragma unroll
                    for (int x = 0; x < 4; ++x) {   // B^T d (rows), then (.) B (columns)
                        const float t0 = d[0][x] - d[2][x], t1 = d[1][x] + d[2][x], t2 = d[2][x] - d[1][x], t3 = d[1][x] - d[3][x];
                        d[0][x] = t0; d[1][x] = t1; d[2][x] = t2; d[3][x] = t3;
                    }
#pragma unroll
                    for (int y = 0; y < 4; ++y) {
                        v[4 * y + 0] = d[y][0] - d[y][2];
                        v[4 * y + 1] = d[y][1] + d[y][2];
                        v[4 * y + 2] = d[y][2] - d[y][1];
                        v[4 * y + 3] = d[y][1] - d[y][3];
                    }
#pragma unroll
                    for (int oz = 0; oz < TZ; ++oz) {
                        const int kz = KD == 3 ? pz - oz : 0;
                        if (KD == 3 ? (kz < 0 || kz > 2) : (pz != oz)) continue;
#pragma unroll
                        for (int mb = 0; mb < MBW; ++mb) {
                            const float* wq = wl + (((kz * GPC + g) * MB + mg * MBW + mb) * 4) * 256;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4_t w4 = *reinterpret_cast<const float4_t*>(wq + q * 256);
                                acc[oz][t][mb][4 * q + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[4 * q + 0], w4.x, acc[oz][t][mb][4 * q + 0], 0, 0, 0);
                                acc[oz][t][mb][4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[4 * q + 1], w4.y, acc[oz][t][mb][4 * q + 1], 0, 0, 0);
                                acc[oz][t][mb][4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[4 * q + 2], w4.z, acc[oz][t][mb][4 * q + 2], 0, 0, 0);
                                acc[oz][t][mb][4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[4 * q + 3], w4.w, acc[oz][t][mb][4 * q + 3], 0, 0, 0);
                            }
                        }
                    }
                }
        if (a.single_buf && c + 1 < nchunks) {
            __syncthreads();
            stage(c + 1, smem);
        }
    }

    // epilogue: output transform, BatchNorm scale/shift + ReLU, 16-byte stores (W % 4 == 0: a piece is inside or outside)
    constexpr unsigned kInvalid = 0x80000000u;
    const int out_plane = a.H * a.W, out_vol = a.D * out_plane;
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.out, (short)0, a.Cout * out_vol * 4, 0x00020000);
    const float lo = a.relu ? 0.f : -INFINITY;
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
        const int co = (mg * MBW + mb) * 16 + ln;
        const bool cok = co < a.Cout;
        const float sc = (a.scale && cok) ? a.scale[co] : 1.f;
        const float sh = (a.scale && cok) ? a.shift[co] : 0.f;
#pragma unroll
        for (int oz = 0; oz < TZ; ++oz)
#pragma unroll
            for (int t = 0; t < TRW; ++t) {
                float row[2][8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s0[4], s1[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const float m0 = acc[oz][t][mb][b][r], m1 = acc[oz][t][mb][4 + b][r], m2 = acc[oz][t][mb][8 + b][r], m3 = acc[oz][t][mb][12 + b][r];
                        s0[b] = (m0 + m1) + m2;
                        s1[b] = (m1 - m2) - m3;
                    }
                    row[0][2 * r] = (s0[0] + s0[1]) + s0[2];
                    row[0][2 * r + 1] = (s0[1] - s0[2]) - s0[3];
                    row[1][2 * r] = (s1[0] + s1[1]) + s1[2];
                    row[1][2 * r + 1] = (s1[1] - s1[2]) - s1[3];
                }
                const int oz_g = oz0 + oz, x = ox0 + 8 * lk;
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int oy = oy0 + 2 * (TRW * trg + t) + rr;
                    const bool rok = cok && oz_g < a.D && oy < a.H;
                    const unsigned pos = (unsigned)(co * out_vol + oz_g * out_plane + oy * a.W + x) * 4u;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        v4u_t qv;
                        qv.x = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 0] * sc + sh, lo));
                        qv.y = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 1] * sc + sh, lo));
                        qv.z = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 2] * sc + sh, lo));
                        qv.w = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 3] * sc + sh, lo));
                        __builtin_amdgcn_raw_buffer_store_b128(qv, rs_out, (rok && x + 4 * h < a.W) ? pos + 16u * h : kInvalid, 0, 0);
                    }
                }
            }
    }
}

struct WCfg { int cin, cout, kd, MB, GPC; };
// the layers this kernel is compiled for; (MB = Cout / 16, GPC = 4-channel k-groups per chunk)
const WCfg kWCfgs[] = {
    {16, 16, 3, 1, 1},   // conv2   module.py:364
    {32, 32, 3, 2, 1},   // conv4   module.py:367
    {64, 64, 3, 4, 1},   // conv6   module.py:370
    {64, 64, 1, 4, 1},   // refine conv6 (2D)  module.py:412
    {16, 16, 1, 1, 2},   // FeatureNet conv1.1 / conv1.2
    {32, 32, 1, 2, 2},   // FeatureNet conv2.1 / conv2.2 / out2
};

const WCfg* find_wcfg(int cin, int cout, int kd) {
    for (const WCfg& c : kWCfgs)
        if (c.cin == cin && c.cout == cout && c.kd == kd) return &c;
    return nullptr;
}

template <int KD, int MB, int MBW, int TZ, int TRW, int GPC>
int launch_wino(WinoArgs a, bool single_buf, hipStream_t st) {
    typedef WinoGeom<KD, MB, MBW, TZ, TRW, GPC> G;
    constexpr size_t lds2 = 2 * (size_t)G::BUF_F * sizeof(float);
    a.nx = ceil_div(a.W, 32); a.ny = ceil_div(a.H, G::TY); a.nz = ceil_div(a.D, TZ);
    a.single_buf = (single_buf || lds2 > 160 * 1024) ? 1 : 0;
    const size_t lds = a.single_buf ? lds2 / 2 : lds2;
    static_assert(lds2 / 2 <= 160 * 1024, "one stage must fit the LDS");
    auto kernel = conv_wino_kernel<KD, MB, MBW, TZ, TRW, GPC>;
    if (int e = dmvs_ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds)) return e;
    kernel<<<dim3(xcd_grid(a.nx * a.ny * a.nz)), 256, lds, st>>>(a);
    DMVS_LAUNCH_CHECK();
}

}  // namespace

extern "C" long dmvs_conv3d_wino_weight_floats(int Cin, int Cout, int kdepth) {
    const WCfg* c = find_wcfg(Cin, Cout, kdepth);
    return c ? (long)Cin / 4 * kdepth * c->MB * 16 * 64 : 0;
}

extern "C" int dmvs_pack_conv_weights_wino(const float* w, float* out, int Cin, int Cout, int kdepth) {
    const WCfg* c = find_wcfg(Cin, Cout, kdepth);
    if (!c || !w || !out) return DMVS_EUNSUPPORTED;
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int NT = 9 * kdepth, cich = 4 * c->GPC;
    size_t n = 0;
    // order: chunk, kz, k-group, 16-channel block, quarter q of the 16 transform positions, lane, xi % 4
    for (int ci0 = 0; ci0 < Cin; ci0 += cich)
        for (int kz = 0; kz < kdepth; ++kz)
            for (int g = 0; g < c->GPC; ++g)
                for (int mb = 0; mb < c->MB; ++mb)
                    for (int q = 0; q < 4; ++q)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 4; ++e) {
                                const int xi = 4 * q + e, ya = xi / 4, xb = xi % 4;
                                const int co = mb * 16 + l % 16, ci = ci0 + 4 * g + l / 16;
                                double u = 0.0;   // (G g G^T)[ya][xb], formed in double and rounded once
                                for (int ky = 0; ky < 3; ++ky)
                                    for (int kx = 0; kx < 3; ++kx)
                                        u += Gm[ya][ky] * Gm[xb][kx] * (double)w[((size_t)co * Cin + ci) * NT + (kz * 3 + ky) * 3 + kx];
                                out[n++] = co < Cout ? (float)u : 0.f;
                            }
    return n == (size_t)dmvs_conv3d_wino_weight_floats(Cin, Cout, kdepth) ? 0 : DMVS_EINVAL;
}

extern "C" int dmvs_conv3d_wino(const float* in, float* out, const float* w_packed, const float* scale, const float* shift,
                                int Cin, int Cout, int D, int H, int W, int kdepth, int flags, dmvs_stream_t stream) {
    if (!in || !out || !w_packed || D < 1 || H < 1 || W < 1) return DMVS_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return DMVS_EINVAL;
    if (flags & ~DMVS_RELU) return DMVS_EUNSUPPORTED;   // no residual, no quad-planar output
    const WCfg* c = find_wcfg(Cin, Cout, kdepth);
    if (!c) return DMVS_EUNSUPPORTED;
    if (W % 4 != 0 || ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) != 0) return DMVS_EUNSUPPORTED;
    if ((long)4 * c->GPC * D * H * W >= (1L << 28) || (long)Cout * D * H * W >= (1L << 29)) return DMVS_EINVAL;
    WinoArgs a = {};
    a.in = in; a.out = out; a.w = w_packed; a.scale = scale; a.shift = shift;
    a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W; a.relu = (flags & DMVS_RELU) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const bool flat = kdepth == 1 || D == 1;
    if (kdepth == 3) {
        if (Cout == 16) return flat ? launch_wino<3, 1, 1, 1, 2, 1>(a, true, st) : launch_wino<3, 1, 1, 2, 1, 1>(a, true, st);
        if (Cout == 32) return launch_wino<3, 2, 2, 1, 1, 1>(a, true, st);
        if (Cout == 64) return launch_wino<3, 4, 2, 1, 1, 1>(a, true, st);
    } else {
        if (Cout == 16) return launch_wino<1, 1, 1, 1, 2, 2>(a, false, st);
        if (Cout == 32) return launch_wino<1, 2, 2, 1, 1, 2>(a, false, st);
        if (Cout == 64) return launch_wino<1, 4, 2, 1, 1, 1>(a, true, st);
    }
    return DMVS_EUNSUPPORTED;
}
