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
// Development knock-outs (scripts/dev/wino_ko.sh, -DDMVS_WKO=mask): bit 0 no tile / weight loads, bit 1 no MFMAs (one VALU
// add per MFMA keeps the operands alive), bit 2 no output stores, bit 3 no patch reads / input transform, bit 4 no weight
// reads, bit 5 no barriers (only meaningful with bit 0).  Never set in the product build.
#ifndef DMVS_WKO
#define DMVS_WKO 0
#endif
#ifndef DMVS_WINO_TAU
#define DMVS_WINO_TAU 0   /* tile permutation for 64-byte store runs: measured neutral (same-box A/B), off */
#endif
#include "common.h"
#include "tile_loader.h"
#include "dev_guard.h"

#include <algorithm>
#include <cmath>
#include <vector>

extern long g_wino_stages, g_wino_persistent, g_wino_conv0_grid;
namespace {

typedef float acc4_t __attribute__((ext_vector_type(4)));
#if DMVS_WKO & 2
__device__ __forceinline__ acc4_t wino_mfma(float a, float b, acc4_t c) { c.x += a + b; return c; }
#else
__device__ __forceinline__ acc4_t wino_mfma(float a, float b, acc4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
#endif
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
    // fpn_wino_kernel only (FeatureNet's level-3 merge folded into out3, module.py:333-336)
    const float* lat;    // [Cl][D][H][W]
    const float* td;     // [Cin][D][H/2][W/2]
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

// Q4: the output is written as two quad-planar halves [half][D][Cout/8][H][W][4] (DMVS_OUT_Q4, the layout K1 samples).
//     The MFMA operands are swapped (output channels = rows, tiles = columns), so a lane's 4 accumulator registers are 4
//     CONSECUTIVE channels of ONE tile: a 16-byte piece per output pixel.
template <int KD, int MB, int MBW, int TZ, int TRW, int GPC, bool Q4 = false>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(WinoArgs a) {
    typedef WinoGeom<KD, MB, MBW, TZ, TRW, GPC> G;
    constexpr int IY = G::IY, IZ = G::IZ, IXP = G::IXP, PS = G::PS, NTR = G::NTR;
    constexpr unsigned kInvalid = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [1 or 2][BUF_F]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lk = lane >> 4;
    const int trg = wave % NTR, mg = wave / NTR;   // the wave's tile-row group and output-channel group

    // PERSISTENT workgroups: a workgroup walks the tiles vb = blockIdx.x, + gridDim.x, ... of the XCD-aware tile list
    // (common.h: XCD k owns the k-th contiguous eighth; gridDim.x is a multiple of 8, so vb % 8 stays the workgroup's
    // XCD).  The (tile, chunk) pairs form ONE pipeline: the first chunk of the next tile is staged before the epilogue of
    // the current one, so a tile's first-load latency, the kernel-argument / BatchNorm loads and the store tail of the
    // previous tile are off the MFMA path (r03 knock-outs: with one tile per workgroup these fixed ~5 us per workgroup
    // cost a third of the kernel).
    struct Tile { int ox0, oy0, oz0; };
    const int ntiles = a.nx * a.ny * a.nz, per_xcd = (ntiles + 7) >> 3;
    auto tile_of = [&](int vb, Tile& t) {
        const int q = vb >> 3, id = (vb & 7) * per_xcd + q;
        if (q >= per_xcd || id >= ntiles) return false;
        const int bx = id % a.nx, r = id / a.nx;
        int by, bz;
        if (KD == 3) { bz = r % a.nz; by = r / a.nz; } else { by = r % a.ny; bz = r / a.ny; }
        t.ox0 = bx * 32; t.oy0 = by * G::TY; t.oz0 = bz * TZ;
        return true;
    };
    int vb = blockIdx.x;
    Tile cur, nxt;
    if (!tile_of(vb, cur)) return;

    // The lane's patch of tile n (output columns 2n, 2n+1) spans tile columns 3 + 2n .. 6 + 2n: read as the three aligned
    // pairs starting at 2 + 2n.  ds_read_b64 is served in two 32-lane groups with bank = dword address mod 64: the 16
    // tiles of one channel cover 32 consecutive banks, the second channel of the group sits PS = 32 (mod 64) further.
    // Planar output: MFMA row i is tile tau(i) = (i & 2 ? 8 : 0) + 2 * (i >> 2) + (i & 1), so that the 4 accumulator rows of a
    // lane (i = 4 lk + r) are the tile pairs {2 lk, 2 lk + 1} and {8 + 2 lk, 9 + 2 lk}: its two 16-byte pieces per row, and
    // the pieces of the 4 lk-lanes of a channel are CONTIGUOUS in each store instruction (64-byte runs instead of 16-byte
    // pieces at a 32-byte stride).  A permutation inside the 16-lane group: the patch reads stay conflict-free.
    const int tau = (Q4 || !DMVS_WINO_TAU) ? ln : ((ln & 2) ? 8 : 0) + 2 * (ln >> 2) + (ln & 1);
    const int pbase = lk * PS + (2 * TRW * trg) * IXP + 2 + 2 * tau;

    const int in_vol = a.D * a.H * a.W;
    const int nchunks = a.Cin / G::CI_CH;
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, nchunks * G::WROWS * 256, 0x00020000);
    // stage chunk c of tile t as pipeline step k: input tile + weight slice, asynchronous
    auto stage = [&](const Tile& t, int c, int k, float* dst) {
        if (DMVS_WKO & 1) return;
        const int ix0a = t.ox0 - 4, iy0 = t.oy0 - 1, iz0 = KD == 3 ? t.oz0 - 1 : t.oz0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.in + (size_t)(c * G::CI_CH) * in_vol), (short)0, G::CI_CH * in_vol * 4, 0x00020000);
        load_tile4<G::CI_CH, IZ, IY, G::LPR, PS>(a.D, a.H, a.W, rs, dst, iz0, iy0, ix0a, wave, lane);
        load_rows64<G::WROWS>(rs_w, dst + G::TILE_F, c, wave, lane);
    };

    // once per workgroup: BatchNorm scale / shift of the lane's output channels
    constexpr int NCO = Q4 ? 4 : 1;
    float sc[MBW][NCO], sh[MBW][NCO];
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int r = 0; r < NCO; ++r) {
            const int co = (mg * MBW + mb) * 16 + (Q4 ? 4 * lk + r : ln);
            const bool cok = co < a.Cout;
            sc[mb][r] = (a.scale && cok) ? a.scale[co] : 1.f;
            sh[mb][r] = (a.scale && cok) ? a.shift[co] : 0.f;
        }
    const int out_plane = a.H * a.W, out_vol = a.D * out_plane;
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.out, (short)0, a.Cout * out_vol * 4, 0x00020000);
    const float lo = a.relu ? 0.f : -INFINITY;

    int k = 0;   // pipeline step = chunks done so far over all tiles; LDS stage k & 1 when there are two
    stage(cur, 0, 0, smem);
    for (;;) {
        const bool has_next = tile_of(vb + (int)gridDim.x, nxt);
        const int ox0 = cur.ox0, oy0 = cur.oy0, oz0 = cur.oz0;
        acc4_t acc[TZ][TRW][MBW][16];
#pragma unroll
        for (int z = 0; z < TZ; ++z)
#pragma unroll
            for (int t = 0; t < TRW; ++t)
#pragma unroll
                for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
                    for (int x = 0; x < 16; ++x) acc[z][t][mb][x] = (acc4_t){0.f, 0.f, 0.f, 0.f};

        for (int c = 0; c < nchunks; ++c, ++k) {
            // step k has landed (this wave's share) ... for every wave; and every wave is done with step k - 1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(DMVS_WKO & 32)) __syncthreads();
            const bool last = c + 1 == nchunks;
            float* curb = smem + (a.single_buf ? 0 : (k & 1)) * G::BUF_F;
            if (!a.single_buf) {
                float* nb = smem + ((k + 1) & 1) * G::BUF_F;
                if (!last) stage(cur, c + 1, k + 1, nb);
                else if (has_next) stage(nxt, 0, k + 1, nb);
            }
            const float* tile = curb + pbase;
            const float* wl = curb + G::TILE_F + lane * 4;
#pragma unroll
            for (int g = 0; g < GPC; ++g)
#pragma unroll
                for (int pz = 0; pz < IZ; ++pz)
#pragma unroll
                    for (int t = 0; t < TRW; ++t) {
                        const float* p = tile + g * 4 * PS + (pz * IY + 2 * t) * IXP;
                        float d[4][4];
                        if (DMVS_WKO & 8) {
#pragma unroll
                            for (int y = 0; y < 4; ++y)
#pragma unroll
                                for (int x = 0; x < 4; ++x) d[y][x] = (float)(lane + y * 4 + x + pz);
                        } else
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
                                    const float4_t w4 = (DMVS_WKO & 16) ? (float4_t){1.f + q, 2.f + mb, 3.f + kz, 4.f + g} : *reinterpret_cast<const float4_t*>(wq + q * 256);
                                    const float wv4[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        acc[oz][t][mb][4 * q + e] = Q4 ? wino_mfma(wv4[e], v[4 * q + e], acc[oz][t][mb][4 * q + e])
                                                                       : wino_mfma(v[4 * q + e], wv4[e], acc[oz][t][mb][4 * q + e]);
                                }
                            }
                        }
                    }
            if (a.single_buf && (!last || has_next)) {  // single stage: refill it once every wave is done with step k
                if (!(DMVS_WKO & 32)) __syncthreads();
                if (!last) stage(cur, c + 1, k + 1, smem);
                else stage(nxt, 0, k + 1, smem);
            }
        }

        // epilogue (the next tile's first chunk is already in flight): output transform, BatchNorm scale/shift + ReLU,
        // 16-byte stores (W % 4 == 0: a piece is inside or outside)
        if constexpr (Q4) {
            const int ch = a.Cout >> 1, cq = ch >> 2;
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb) {
                const int co0 = (mg * MBW + mb) * 16 + 4 * lk;   // the lane's 4 channels; its tile is ln
                const bool cok = co0 < a.Cout;
                const int hsel = co0 >= ch ? 1 : 0, cqi = (co0 - hsel * ch) >> 2;
#pragma unroll
                for (int oz = 0; oz < TZ; ++oz)
#pragma unroll
                    for (int t = 0; t < TRW; ++t) {
                        float y[4][2][2];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float s0[4], s1[4];
#pragma unroll
                            for (int b = 0; b < 4; ++b) {
                                const float m0 = acc[oz][t][mb][b][r], m1 = acc[oz][t][mb][4 + b][r], m2 = acc[oz][t][mb][8 + b][r], m3 = acc[oz][t][mb][12 + b][r];
                                s0[b] = (m0 + m1) + m2;
                                s1[b] = (m1 - m2) - m3;
                            }
                            y[r][0][0] = (s0[0] + s0[1]) + s0[2];
                            y[r][0][1] = (s0[1] - s0[2]) - s0[3];
                            y[r][1][0] = (s1[0] + s1[1]) + s1[2];
                            y[r][1][1] = (s1[1] - s1[2]) - s1[3];
                        }
                        const int oz_g = oz0 + oz;
#pragma unroll
                        for (int rr = 0; rr < 2; ++rr) {
                            const int oy = oy0 + 2 * (TRW * trg + t) + rr;
                            const bool rok = cok && oz_g < a.D && oy < a.H;
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int x = ox0 + 2 * ln + e;
                                v4u_t qv;
                                qv.x = __builtin_bit_cast(unsigned, fmaxf(y[0][rr][e] * sc[mb][0] + sh[mb][0], lo));
                                qv.y = __builtin_bit_cast(unsigned, fmaxf(y[1][rr][e] * sc[mb][1 % NCO] + sh[mb][1 % NCO], lo));
                                qv.z = __builtin_bit_cast(unsigned, fmaxf(y[2][rr][e] * sc[mb][2 % NCO] + sh[mb][2 % NCO], lo));
                                qv.w = __builtin_bit_cast(unsigned, fmaxf(y[3][rr][e] * sc[mb][3 % NCO] + sh[mb][3 % NCO], lo));
                                const unsigned off = (unsigned)(((hsel * a.D + oz_g) * cq + cqi) * out_plane + oy * a.W + x) * 16u;
                                __builtin_amdgcn_raw_buffer_store_b128(qv, rs_out, (rok && x < a.W && !((DMVS_WKO & 4) && qv.x != 0x12345678u)) ? off : kInvalid, 0, 0);
                            }
                        }
                    }
            }
        } else {
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb) {
                const int co = (mg * MBW + mb) * 16 + ln;
                const bool cok = co < a.Cout;
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
                        const int oz_g = oz0 + oz;
#pragma unroll
                        for (int rr = 0; rr < 2; ++rr) {
                            const int oy = oy0 + 2 * (TRW * trg + t) + rr;
                            const bool rok = cok && oz_g < a.D && oy < a.H;
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const int x = DMVS_WINO_TAU ? ox0 + 16 * h + 4 * lk : ox0 + 8 * lk + 4 * h;   // rows r = 2h, 2h + 1 of the lane: tiles 8h + 2lk, + 1
                                const unsigned pos = (unsigned)(co * out_vol + oz_g * out_plane + oy * a.W + x) * 4u;
                                v4u_t qv;
                                qv.x = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 0] * sc[mb][0] + sh[mb][0], lo));
                                qv.y = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 1] * sc[mb][0] + sh[mb][0], lo));
                                qv.z = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 2] * sc[mb][0] + sh[mb][0], lo));
                                qv.w = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 3] * sc[mb][0] + sh[mb][0], lo));
                                __builtin_amdgcn_raw_buffer_store_b128(qv, rs_out, (rok && x < a.W && !((DMVS_WKO & 4) && qv.x != 0x12345678u)) ? pos : kInvalid, 0, 0);
                            }
                        }
                    }
            }
        }
        if (!has_next) break;
        cur = nxt;
        vb += (int)gridDim.x;
    }
}

// conv0 of the two regularisation branches (module.py:361 and 403, fused on the host to 2 -> 8 + 8 channels): with Cin = 2
// the MFMA k-group of 4 is (channel, depth tap) PAIRS instead of 4 channels -- lane group lk = (channel lk & 1, depth
// selector lk >> 1).  An output plane o takes two k-steps: step 0 = input planes o + (lk >> 1) (depth taps 0 and 1), step 1 =
// plane o + 2 (depth tap 2) for the lanes with lk < 2 and zero weights for the others: 32 MFMAs per (16 tiles x 16 channels)
// block where the direct form needs 56.  The whole filter bank (8 KB transformed) stays in LDS for the life of the
// persistent workgroup; a tile stage is just the 2-channel input tile, two stages, the next tile's loads under the current
// tile's MFMAs.  Workgroup = 2 output planes x 8 rows x 32 columns, one tile row per wave.
__global__ __launch_bounds__(256, 2) void conv0_wino_kernel(WinoArgs a) {
    constexpr int TZ = 2, TY = 8, IZ = 4, IY = 10, LPR = 10, IXP = 40;
    constexpr int PS0 = IZ * IY * IXP, PS = PS0 + (32 - PS0 % 64 + 64) % 64;
    constexpr int TILE_F = (2 * PS + 63) & ~63, W_F = 2 * 4 * 256;
    constexpr unsigned kInvalid = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [W_F] weights, [2][TILE_F] tiles

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lk = lane >> 4;
    struct Tile { int ox0, oy0, oz0; };
    const int ntiles = a.nx * a.ny * a.nz, per_xcd = (ntiles + 7) >> 3;
    auto tile_of = [&](int vb, Tile& t) {
        const int q = vb >> 3, id = (vb & 7) * per_xcd + q;
        if (q >= per_xcd || id >= ntiles) return false;
        const int bx = id % a.nx, r = id / a.nx;
        t.ox0 = bx * 32; t.oz0 = (r % a.nz) * TZ; t.oy0 = (r / a.nz) * TY;
        return true;
    };
    int vb = blockIdx.x;
    Tile cur, nxt;
    if (!tile_of(vb, cur)) return;

    const int in_vol = a.D * a.H * a.W;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, 2 * in_vol * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, W_F * 4, 0x00020000);
    float* const tiles = smem + W_F;
    auto stage = [&](const Tile& t, float* dst) {
        if (DMVS_WKO & 1) return;
        load_tile4<2, IZ, IY, LPR, PS>(a.D, a.H, a.W, rs_in, dst, t.oz0 - 1, t.oy0 - 1, t.ox0 - 4, wave, lane);
    };
    // patch of tile ln, channel lk & 1, wave's tile row; the plane is picked per k-step
    const int tau = !DMVS_WINO_TAU ? ln : ((ln & 2) ? 8 : 0) + 2 * (ln >> 2) + (ln & 1);   // tile of MFMA row ln (see conv_wino_kernel: 64-byte store runs)
    const int pbase = (lk & 1) * PS + (2 * wave) * IXP + 2 + 2 * tau;
    const int zsel = lk >> 1;
    const float* wl = smem + lane * 4;

    const int co = ln;
    const float sc = a.scale ? a.scale[co] : 1.f, sh = a.scale ? a.shift[co] : 0.f;
    const int out_plane = a.H * a.W, out_vol = a.D * out_plane;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, (short)0, a.Cout * out_vol * 4, 0x00020000);
    const float lo = a.relu ? 0.f : -INFINITY;

    load_rows64<W_F / 64>(rs_w, smem, 0, wave, lane);
    stage(cur, tiles);
    for (int k = 0;; ++k) {
        const bool has_next = tile_of(vb + (int)gridDim.x, nxt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // tile k (and the weights) landed for every wave; every wave is done with tile k - 1
        if (has_next) stage(nxt, tiles + ((k + 1) & 1) * TILE_F);
        const float* tile = tiles + (k & 1) * TILE_F + pbase;
        acc4_t acc[TZ][16];
#pragma unroll
        for (int oz = 0; oz < TZ; ++oz)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const float* p = tile + (oz + (st ? 2 : zsel)) * (IY * IXP);
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
                for (int x = 0; x < 4; ++x) {
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
                for (int q = 0; q < 4; ++q) {
                    const float4_t w4 = *reinterpret_cast<const float4_t*>(wl + (st * 4 + q) * 256);
                    const float wv4[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const acc4_t c0 = st == 0 ? (acc4_t){0.f, 0.f, 0.f, 0.f} : acc[oz][4 * q + e];
                        acc[oz][4 * q + e] = wino_mfma(v[4 * q + e], wv4[e], c0);
                    }
                }
            }
        // epilogue: output transform, BatchNorm + ReLU, 16-byte stores (a lane: channel ln, 8 consecutive x of two rows)
#pragma unroll
        for (int oz = 0; oz < TZ; ++oz) {
            float row[2][8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s0[4], s1[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float m0 = acc[oz][b][r], m1 = acc[oz][4 + b][r], m2 = acc[oz][8 + b][r], m3 = acc[oz][12 + b][r];
                    s0[b] = (m0 + m1) + m2;
                    s1[b] = (m1 - m2) - m3;
                }
                row[0][2 * r] = (s0[0] + s0[1]) + s0[2];
                row[0][2 * r + 1] = (s0[1] - s0[2]) - s0[3];
                row[1][2 * r] = (s1[0] + s1[1]) + s1[2];
                row[1][2 * r + 1] = (s1[1] - s1[2]) - s1[3];
            }
            const int oz_g = cur.oz0 + oz;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int oy = cur.oy0 + 2 * wave + rr;
                const bool rok = oz_g < a.D && oy < a.H;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int x = DMVS_WINO_TAU ? cur.ox0 + 16 * h + 4 * lk : cur.ox0 + 8 * lk + 4 * h;
                    const unsigned pos = (unsigned)(co * out_vol + oz_g * out_plane + oy * a.W + x) * 4u;
                    v4u_t qv;
                    qv.x = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 0] * sc + sh, lo));
                    qv.y = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 1] * sc + sh, lo));
                    qv.z = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 2] * sc + sh, lo));
                    qv.w = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 3] * sc + sh, lo));
                    __builtin_amdgcn_raw_buffer_store_b128(qv, rs_out, (rok && x < a.W && !((DMVS_WKO & 4) && qv.x != 0x12345678u)) ? pos : kInvalid, 0, 0);
                }
            }
        }
        if (!has_next) break;
        cur = nxt;
        vb += (int)gridDim.x;
    }
}

// FeatureNet's level-3 merge + out3 (module.py:333-336) as ONE Winograd convolution:
//     out3( b_lat + W_lat . lat + up2(td) )  =  conv3x3_{W3 o W_lat}(lat) + conv3x3_{W3 . b_lat}(1_image) + conv3x3_{W3}(up2(td))
// * the 1x1 lateral conv is folded into the 3x3 filters on the host (composite 8 -> 16 filters, formed in double), its bias
//   becomes a filter on a constant-one image (zero outside: the border rows / columns of `intra` are zero PADDING, so the
//   bias must not leak there) -- 9 input channels = 3 k-groups of 4 (the 3 spare slots carry zero weights);
// * the nearest x2 upsample makes the 4x4 patch of up2(td) a 3x3 patch T with rows / columns (0, 1, 1, 2): B^T d B then
//   vanishes on transform positions 2 (d2 - d1 = 0) and the rest is (T0 - T1, 2 T1, T1 - T2) per axis: 9 of the 16
//   positions, 12 subtractions instead of 32, the factors 2 folded into the weights -- 9 MFMAs per k-group instead of 16.
// No `intra` tile is built (K3's FPN variant spends as many VALU cycles building it as its MFMAs take, and fp32 MFMA and
// VALU time ADD UP on a SIMD: scripts/dev/mfma_valu_overlap.hip): 48 + 72 MFMAs and ~200 VALU per 32 x 2-pixel tile row.
// 512-thread persistent workgroup = 16 rows x 32 columns of one view, one tile row per wave (2 waves per SIMD); all
// transformed filters (36 KB) stay in LDS; a tile stage = lateral tile (8 planes + the ones plane) + all 32 top-down planes
// (57 KB), two stages: the next tile loads under the current tile's MFMAs.
template <bool Q4>
__global__ __launch_bounds__(512, 2) void fpn_wino_kernel(WinoArgs a, const float* ones) {
    constexpr int IY = 18, IXP = 40, LPR = 10, PS = 736, PPP = PS / 4;          // lateral planes: 184 16-byte pieces (4 pad)
    constexpr int TD_IXP = 24, TD_LPR = 6, TD_PS = 240, TD_PPP = 60;             // top-down planes: 10 rows of 6 pieces, dense
    constexpr int LAT_F = 9 * PS, TD_F = 32 * TD_PS, STAGE_F = LAT_F + TD_F;
    constexpr int WL_F = 3 * 4 * 256, WT_F = 8 * 3 * 256, W_F = WL_F + WT_F;
    constexpr int NI_LAT = (9 * PPP + 63) / 64, NI_TD = (32 * TD_PPP + 63) / 64, NI = NI_LAT + NI_TD;
    static_assert((8 * PPP) % 64 == 0, "the ones plane starts on an instruction boundary");
    constexpr unsigned kInvalid = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [W_F] filters, [2][STAGE_F]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7 = the tile row
    const int ln = lane & 15, lk = lane >> 4;
    struct Tile { int ox0, oy0, z; };
    const int ntiles = a.nx * a.ny * a.nz, per_xcd = (ntiles + 7) >> 3;
    auto tile_of = [&](int vb, Tile& t) {
        const int q = vb >> 3, id = (vb & 7) * per_xcd + q;
        if (q >= per_xcd || id >= ntiles) return false;
        const int bx = id % a.nx, r = id / a.nx;
        t.ox0 = bx * 32; t.oy0 = (r % a.ny) * 16; t.z = r / a.ny;
        return true;
    };
    int vb = blockIdx.x;
    Tile cur, nxt;
    if (!tile_of(vb, cur)) return;

    const int plane = a.H * a.W, in_vol = a.D * plane;
    const int tdH = a.H >> 1, tdW = a.W >> 1, td_plane = tdH * tdW, td_vol = a.D * td_plane;
    const __amdgpu_buffer_rsrc_t rs_lat = __builtin_amdgcn_make_buffer_rsrc((void*)a.lat, (short)0, 8 * in_vol * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_one = __builtin_amdgcn_make_buffer_rsrc((void*)ones, (short)0, plane * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_td = __builtin_amdgcn_make_buffer_rsrc((void*)a.td, (short)0, 32 * td_vol * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, W_F * 4, 0x00020000);
    float* const stages = smem + W_F;
    // One tile stage: the planes are lists of 16-byte pieces in LDS order, 64 pieces (1 KiB) per wave-instruction, the
    // instructions dealt round-robin to the 8 waves; a piece outside the image (or a pad piece) reads zeros.  Which piece a
    // lane moves in its j-th instruction does not depend on the tile: plane offset, row and column are worked out once.
    constexpr int NJ = (NI + 7) / 8;
    int s_rel[NJ], s_rc[NJ];   // element offset relative to the tile origin; row | column << 8 | usable << 16
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = wave + 8 * j;
        if (i < NI_LAT) {
            const int p = i * 64 + lane, pl = p / PPP, r = p - pl * PPP, row = r / LPR, pc = r - row * LPR;
            const bool one = i >= 8 * PPP / 64;
            s_rel[j] = (one ? 0 : pl * in_vol) + row * a.W + 4 * pc;
            s_rc[j] = row | (4 * pc) << 8 | ((pl < 9 && row < IY) ? 1 << 16 : 0) | (p < 9 * PPP ? 1 << 17 : 0);
        } else {
            const int it = i - NI_LAT;
            const int p = it * 64 + lane, pl = p / TD_PPP, r = p - pl * TD_PPP, row = r / TD_LPR, pc = r - row * TD_LPR;
            s_rel[j] = pl * td_vol + row * tdW + 4 * pc;
            s_rc[j] = row | (4 * pc) << 8 | ((i < NI && pl < 32) ? 3 << 16 : 0);
        }
    }
    auto stage = [&](const Tile& t, float* dst) {
        if (DMVS_WKO & 1) return;
        const int lat0 = t.z * plane + (t.oy0 - 1) * a.W + t.ox0 - 4, td0 = t.z * td_plane + ((t.oy0 >> 1) - 1) * tdW + (t.ox0 >> 1) - 4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int i = wave + 8 * j;   // scalar
            if (i >= NI) break;
            const int row = s_rc[j] & 255, col = (s_rc[j] >> 8) & 255;
            if (i < NI_LAT) {
                const int gy = t.oy0 - 1 + row, gx = t.ox0 - 4 + col;
                const bool ok = (s_rc[j] & (1 << 16)) && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const bool one = i >= 8 * PPP / 64;   // scalar: the constant-one plane
                const unsigned off = ok ? (unsigned)(s_rel[j] + (one ? lat0 - t.z * plane : lat0)) * 4u : kInvalid;
                if (s_rc[j] & (1 << 17)) {   // lanes past the last plane are switched off: they would zero the first top-down pieces
                    if (one) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_one, (lds_ptr_t)(dst + i * 256), 16, off, 0, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_lat, (lds_ptr_t)(dst + i * 256), 16, off, 0, 0, 0);
                }
            } else {
                const int gy = (t.oy0 >> 1) - 1 + row, gx = (t.ox0 >> 1) - 4 + col;
                const bool ok = (s_rc[j] & (1 << 16)) && (unsigned)gy < (unsigned)tdH && (unsigned)gx < (unsigned)tdW;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_td, (lds_ptr_t)(dst + LAT_F + (i - NI_LAT) * 256), 16,
                                                         ok ? (unsigned)(s_rel[j] + td0) * 4u : kInvalid, 0, 0, 0);
            }
        }
    };
    // filters, once per workgroup
    for (int j = wave; j < W_F / 256; j += 8)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(smem + j * 256), 16, (unsigned)(j * 1024 + lane * 16), 0, 0, 0);

    constexpr int NCO = Q4 ? 4 : 1;
    float sc[NCO], sh[NCO];
#pragma unroll
    for (int r = 0; r < NCO; ++r) {
        const int co = Q4 ? 4 * lk + r : ln;
        sc[r] = a.scale ? a.scale[co] : 1.f;
        sh[r] = a.scale ? a.shift[co] : 0.f;
    }
    const int out_vol = a.D * plane;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, (short)0, a.Cout * out_vol * 4, 0x00020000);
    const float lo = a.relu ? 0.f : -INFINITY;
    const float* wlat = smem + lane * 4;
    const float* wtd = smem + WL_F + lane * 4;

    stage(cur, stages);
    for (int k = 0;; ++k) {
        const bool has_next = tile_of(vb + (int)gridDim.x, nxt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // stage k landed for every wave; every wave is done with stage k - 1
        if (has_next) stage(nxt, stages + ((k + 1) & 1) * STAGE_F);
        const float* lat_t = stages + (k & 1) * STAGE_F;
        const float* td_t = lat_t + LAT_F;
        acc4_t acc[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) acc[x] = (acc4_t){0.f, 0.f, 0.f, 0.f};
        // 11 k-groups: lateral channels 0-3, 4-7, the ones plane (k-slots 1-3: zero weights, any finite input), then the 8
        // top-down groups (3x3 patch of the half-resolution tile, transform positions {0, 1, 3} x {0, 1, 3}).  The LDS reads
        // of group i + 1 are issued BEFORE the MFMAs of group i (one workgroup of 2 waves per SIMD: nobody else hides them).
        float2_t lq[2][12];
        float4_t lw[2][4];
        float tq[2][9];
        float4_t tw[2][3];
        auto read_lat = [&](int c, int b) {
            const float* p = lat_t + (c < 2 ? 4 * c + lk : 8) * PS + (2 * wave) * IXP + 2 + 2 * ln;
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int h = 0; h < 3; ++h) lq[b][3 * y + h] = *reinterpret_cast<const float2_t*>(p + y * IXP + 2 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) lw[b][q] = *reinterpret_cast<const float4_t*>(wlat + (c * 4 + q) * 256);
        };
        auto read_td = [&](int g, int b) {
            const float* p = td_t + (4 * g + lk) * TD_PS + wave * TD_IXP + 3 + ln;
#pragma unroll
            for (int y = 0; y < 3; ++y)
#pragma unroll
                for (int x = 0; x < 3; ++x) tq[b][3 * y + x] = p[y * TD_IXP + x];
#pragma unroll
            for (int q = 0; q < 3; ++q) tw[b][q] = *reinterpret_cast<const float4_t*>(wtd + (g * 3 + q) * 256);
        };
        auto mma = [&](int xi, float v, float w) { acc[xi] = Q4 ? wino_mfma(w, v, acc[xi]) : wino_mfma(v, w, acc[xi]); };
        auto comp_lat = [&](int b) {
            float d[4][4];
#pragma unroll
            for (int y = 0; y < 4; ++y) { d[y][0] = lq[b][3 * y].y; d[y][1] = lq[b][3 * y + 1].x; d[y][2] = lq[b][3 * y + 1].y; d[y][3] = lq[b][3 * y + 2].x; }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float t0 = d[0][x] - d[2][x], t1 = d[1][x] + d[2][x], t2 = d[2][x] - d[1][x], t3 = d[1][x] - d[3][x];
                d[0][x] = t0; d[1][x] = t1; d[2][x] = t2; d[3][x] = t3;
            }
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                mma(4 * y + 0, d[y][0] - d[y][2], lw[b][y].x);
                mma(4 * y + 1, d[y][1] + d[y][2], lw[b][y].y);
                mma(4 * y + 2, d[y][2] - d[y][1], lw[b][y].z);
                mma(4 * y + 3, d[y][1] - d[y][3], lw[b][y].w);
            }
        };
        auto comp_td = [&](int b) {
            float c[3][3];   // rows (T0 - T1, T1, T1 - T2)
#pragma unroll
            for (int x = 0; x < 3; ++x) { c[0][x] = tq[b][x] - tq[b][3 + x]; c[1][x] = tq[b][3 + x]; c[2][x] = tq[b][3 + x] - tq[b][6 + x]; }
            const float wv[12] = {tw[b][0].x, tw[b][0].y, tw[b][0].z, tw[b][0].w, tw[b][1].x, tw[b][1].y, tw[b][1].z, tw[b][1].w,
                                  tw[b][2].x, tw[b][2].y, tw[b][2].z, tw[b][2].w};
#pragma unroll
            for (int y = 0; y < 3; ++y) {
                constexpr int pos[3] = {0, 1, 3};
                mma(4 * pos[y] + 0, c[y][0] - c[y][1], wv[3 * y]);
                mma(4 * pos[y] + 1, c[y][1], wv[3 * y + 1]);
                mma(4 * pos[y] + 3, c[y][1] - c[y][2], wv[3 * y + 2]);
            }
        };
        // (sched_barrier: the compiler otherwise sinks each group's reads back to just before their first use)
#define DMVS_FENCE() __builtin_amdgcn_sched_barrier(0)
        read_lat(0, 0); DMVS_FENCE();
        read_lat(1, 1); DMVS_FENCE();
        comp_lat(0); DMVS_FENCE();
        read_lat(2, 0); DMVS_FENCE();
        comp_lat(1); DMVS_FENCE();
        read_td(0, 1); DMVS_FENCE();
        comp_lat(0); DMVS_FENCE();
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 1 < 8) { read_td(g + 1, g & 1); DMVS_FENCE(); }
            comp_td((g + 1) & 1); DMVS_FENCE();
        }
#undef DMVS_FENCE
        // epilogue
        const int oz_g = cur.z;
        if constexpr (Q4) {
            const int ch = a.Cout >> 1, cq = ch >> 2, co0 = 4 * lk;
            const int hsel = co0 >= ch ? 1 : 0, cqi = (co0 - hsel * ch) >> 2;
            float y[4][2][2];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s0[4], s1[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float m0 = acc[b][r], m1 = acc[4 + b][r], m2 = acc[8 + b][r], m3 = acc[12 + b][r];
                    s0[b] = (m0 + m1) + m2;
                    s1[b] = (m1 - m2) - m3;
                }
                y[r][0][0] = (s0[0] + s0[1]) + s0[2];
                y[r][0][1] = (s0[1] - s0[2]) - s0[3];
                y[r][1][0] = (s1[0] + s1[1]) + s1[2];
                y[r][1][1] = (s1[1] - s1[2]) - s1[3];
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int oy = cur.oy0 + 2 * wave + rr;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int x = cur.ox0 + 2 * ln + e;
                    v4u_t qv;
                    qv.x = __builtin_bit_cast(unsigned, fmaxf(y[0][rr][e] * sc[0] + sh[0], lo));
                    qv.y = __builtin_bit_cast(unsigned, fmaxf(y[1][rr][e] * sc[1 % NCO] + sh[1 % NCO], lo));
                    qv.z = __builtin_bit_cast(unsigned, fmaxf(y[2][rr][e] * sc[2 % NCO] + sh[2 % NCO], lo));
                    qv.w = __builtin_bit_cast(unsigned, fmaxf(y[3][rr][e] * sc[3 % NCO] + sh[3 % NCO], lo));
                    const unsigned off = (unsigned)(((hsel * a.D + oz_g) * cq + cqi) * plane + oy * a.W + x) * 16u;
                    __builtin_amdgcn_raw_buffer_store_b128(qv, rs_out, (oy < a.H && x < a.W && !((DMVS_WKO & 4) && qv.x != 0x12345678u)) ? off : kInvalid, 0, 0);
                }
            }
        } else {
            float row[2][8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s0[4], s1[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float m0 = acc[b][r], m1 = acc[4 + b][r], m2 = acc[8 + b][r], m3 = acc[12 + b][r];
                    s0[b] = (m0 + m1) + m2;
                    s1[b] = (m1 - m2) - m3;
                }
                row[0][2 * r] = (s0[0] + s0[1]) + s0[2];
                row[0][2 * r + 1] = (s0[1] - s0[2]) - s0[3];
                row[1][2 * r] = (s1[0] + s1[1]) + s1[2];
                row[1][2 * r + 1] = (s1[1] - s1[2]) - s1[3];
            }
            const int co = ln, x = cur.ox0 + 8 * lk;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int oy = cur.oy0 + 2 * wave + rr;
                const unsigned pos = (unsigned)(co * out_vol + oz_g * plane + oy * a.W + x) * 4u;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v4u_t qv;
                    qv.x = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 0] * sc[0] + sh[0], lo));
                    qv.y = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 1] * sc[0] + sh[0], lo));
                    qv.z = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 2] * sc[0] + sh[0], lo));
                    qv.w = __builtin_bit_cast(unsigned, fmaxf(row[rr][4 * h + 3] * sc[0] + sh[0], lo));
                    __builtin_amdgcn_raw_buffer_store_b128(qv, rs_out, (oy < a.H && x + 4 * h < a.W) ? pos + 16u * h : kInvalid, 0, 0);
                }
            }
        }
        if (!has_next) break;
        cur = nxt;
        vb += (int)gridDim.x;
    }
}

}  // namespace
// LDS stages of the 3D layers (dmvs_tune("wino_stages")): 0 = per-layer default, 1 = one, 2 = two wherever they fit
long g_wino_stages = 0;
// 0: one tile per workgroup (dmvs_tune("wino_persistent"), A/B of the persistent tile walk)
long g_wino_persistent = 1;
long g_wino_conv0_grid = 512;   // persistent workgroups of conv0_wino_kernel (dmvs_tune("wino_conv0_grid"), multiple of 8)
namespace {

struct WCfg { int cin, cout, kd, MB, GPC; };
// the layers this kernel is compiled for; (MB = Cout / 16, GPC = 4-channel k-groups per chunk)
const WCfg kWCfgs[] = {
    {16, 16, 3, 1, 1},   // conv2   module.py:364
    {32, 32, 3, 2, 1},   // conv4   module.py:367
    {64, 64, 3, 4, 1},   // conv6   module.py:370
    {64, 64, 1, 4, 1},   // refine conv6 (2D)  module.py:412
    {16, 16, 1, 1, 2},   // FeatureNet conv1.1 / conv1.2
    {32, 32, 1, 2, 2},   // FeatureNet conv2.1 / conv2.2 / out2
    {2, 16, 3, 1, 0},    // conv0 of both branches fused (2 -> 8 + 8), (channel, depth tap) k-groups: conv0_wino_kernel
    {32, 16, 1, 1, 1},   // FeatureNet out3 (alone; with the level-3 merge folded in: fpn_wino_kernel)
};

const WCfg* find_wcfg(int cin, int cout, int kd) {
    for (const WCfg& c : kWCfgs)
        if (c.cin == cin && c.cout == cout && c.kd == kd) return &c;
    return nullptr;
}

template <int KD, int MB, int MBW, int TZ, int TRW, int GPC, bool Q4 = false>
int launch_wino(WinoArgs a, bool single_buf, hipStream_t st) {
    typedef WinoGeom<KD, MB, MBW, TZ, TRW, GPC> G;
    constexpr size_t lds2 = 2 * (size_t)G::BUF_F * sizeof(float);
    static_assert(lds2 / 2 <= 160 * 1024, "one stage must fit the LDS");
    a.nx = ceil_div(a.W, 32); a.ny = ceil_div(a.H, G::TY); a.nz = ceil_div(a.D, TZ);
    if (g_wino_stages) single_buf = g_wino_stages == 1;
    a.single_buf = (single_buf || lds2 > 160 * 1024) ? 1 : 0;
    const size_t lds = a.single_buf ? lds2 / 2 : lds2;
    auto kernel = conv_wino_kernel<KD, MB, MBW, TZ, TRW, GPC, Q4>;
    if (int e = dmvs_ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds)) return e;
    // persistent workgroups: as many as are resident at once (2 per CU by registers, fewer if the LDS stage is large)
    const unsigned resident = 256u * (unsigned)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / lds));
    const unsigned grid = std::min(xcd_grid(a.nx * a.ny * a.nz), g_wino_persistent ? resident : 0xffffffffu);
    kernel<<<dim3(grid), 256, lds, st>>>(a);
    DMVS_LAUNCH_CHECK();
}

int launch_conv0_wino(WinoArgs a, hipStream_t st) {
    constexpr int PS0 = 4 * 10 * 40, PS = PS0 + (32 - PS0 % 64 + 64) % 64;
    constexpr size_t lds = (2 * 4 * 256 + 2 * (size_t)((2 * PS + 63) & ~63)) * sizeof(float);
    a.nx = ceil_div(a.W, 32); a.ny = ceil_div(a.H, 8); a.nz = ceil_div(a.D, 2);
    if (int e = dmvs_ensure_dynamic_lds(reinterpret_cast<const void*>(conv0_wino_kernel), lds)) return e;
    const unsigned grid = std::min(xcd_grid(a.nx * a.ny * a.nz), g_wino_persistent ? (unsigned)g_wino_conv0_grid : 0xffffffffu);
    conv0_wino_kernel<<<dim3(grid), 256, lds, st>>>(a);
    DMVS_LAUNCH_CHECK();
}

// output rows / planes of a workgroup of the variant dispatch() picks (one source of truth for dmvs_conv3d_wino_plan)
void wino_tile(int Cout, int kdepth, int D, int& tz, int& ty) {
    const bool flat = kdepth == 1 || D == 1;
    tz = (kdepth == 3 && Cout == 16 && !flat) ? 2 : 1;   // (conv0: 2 planes x 8 rows whatever the depth)
    ty = Cout == 64 ? 4 : (Cout == 16 && flat) ? 16 : 8;
}

template <bool Q4>
int dispatch(const WinoArgs& a, int kdepth, hipStream_t st) {
    const int Cout = a.Cout;
    const bool flat = kdepth == 1 || a.D == 1;
    if (kdepth == 3) {
        if (a.Cin == 2 && Cout == 16) return Q4 ? DMVS_EUNSUPPORTED : launch_conv0_wino(a, st);
        if (a.Cin == 16 && Cout == 16) return flat ? launch_wino<3, 1, 1, 1, 2, 1, Q4>(a, true, st) : launch_wino<3, 1, 1, 2, 1, 1, Q4>(a, true, st);
        if (a.Cin == 32 && Cout == 32) return launch_wino<3, 2, 2, 1, 1, 1, Q4>(a, true, st);
        if (a.Cin == 64 && Cout == 64) return launch_wino<3, 4, 2, 1, 1, 1, Q4>(a, true, st);
    } else {
        if (a.Cin == 16 && Cout == 16) return launch_wino<1, 1, 1, 1, 2, 2, Q4>(a, false, st);
        if (a.Cin == 32 && Cout == 32) return launch_wino<1, 2, 2, 1, 1, 2, Q4>(a, false, st);
        if (a.Cin == 64 && Cout == 64) return launch_wino<1, 4, 2, 1, 1, 1, Q4>(a, true, st);
        if (a.Cin == 32 && Cout == 16) return launch_wino<1, 1, 1, 1, 2, 1, Q4>(a, false, st);
    }
    return DMVS_EUNSUPPORTED;
}

}  // namespace

extern "C" long dmvs_conv3d_wino_weight_floats(int Cin, int Cout, int kdepth) {
    const WCfg* c = find_wcfg(Cin, Cout, kdepth);
    if (c && Cin == 2) return 2 * 4 * 256;
    return c ? (long)Cin / 4 * kdepth * c->MB * 16 * 64 : 0;
}

extern "C" int dmvs_pack_conv_weights_wino(const float* w, float* out, int Cin, int Cout, int kdepth) {
    const WCfg* c = find_wcfg(Cin, Cout, kdepth);
    if (!c || !w || !out) return DMVS_EUNSUPPORTED;
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int NT = 9 * kdepth, cich = 4 * c->GPC;
    size_t n = 0;
    if (Cin == 2) {   // conv0_wino_kernel: k-step, quarter, lane (cout = l % 16, channel = (l / 16) & 1, depth selector l / 32), xi % 4
        for (int st = 0; st < 2; ++st)
            for (int q = 0; q < 4; ++q)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 4; ++e) {
                        const int xi = 4 * q + e, ya = xi / 4, xb = xi % 4, co = l % 16, ci = (l / 16) & 1, zsel = l / 32;
                        const int kz = st ? 2 : zsel;
                        double u = 0.0;
                        for (int ky = 0; ky < 3; ++ky)
                            for (int kx = 0; kx < 3; ++kx)
                                u += Gm[ya][ky] * Gm[xb][kx] * (double)w[((size_t)co * Cin + ci) * NT + (kz * 3 + ky) * 3 + kx];
                        out[n++] = (st == 1 && zsel == 1) ? 0.f : (float)u;
                    }
        return n == 2048 ? 0 : DMVS_EINVAL;
    }
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

extern "C" int dmvs_conv3d_wino_plan(int Cin, int Cout, int D, int H, int W, int kdepth) {
    if (!find_wcfg(Cin, Cout, kdepth) || D < 1 || H < 1 || W < 1 || W % 4 != 0) return DMVS_EUNSUPPORTED;
    int tz, ty;
    wino_tile(Cout, kdepth, D, tz, ty);
    if (Cin == 2) { tz = 2; ty = 8; }
    const long n = (long)ceil_div(W, 32) * ceil_div(H, ty) * ceil_div(D, tz);
    return n > 0x3fffffff ? 0x3fffffff : (int)n;
}

extern "C" int dmvs_conv3d_wino(const float* in, float* out, const float* w_packed, const float* scale, const float* shift,
                                int Cin, int Cout, int D, int H, int W, int kdepth, int flags, dmvs_stream_t stream) {
    if (!in || !out || !w_packed || D < 1 || H < 1 || W < 1) return DMVS_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return DMVS_EINVAL;
    if (flags & ~(DMVS_RELU | DMVS_OUT_Q4)) return DMVS_EUNSUPPORTED;   // no residual
    const WCfg* c = find_wcfg(Cin, Cout, kdepth);
    if (!c) return DMVS_EUNSUPPORTED;
    if (W % 4 != 0 || ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) != 0) return DMVS_EUNSUPPORTED;
    if ((long)std::max(2, 4 * c->GPC) * D * H * W >= (1L << 28) || (long)Cout * D * H * W >= (1L << 29)) return DMVS_EINVAL;
    WinoArgs a = {};
    a.in = in; a.out = out; a.w = w_packed; a.scale = scale; a.shift = shift;
    a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W; a.relu = (flags & DMVS_RELU) ? 1 : 0;
    return (flags & DMVS_OUT_Q4) ? dispatch<true>(a, kdepth, (hipStream_t)stream) : dispatch<false>(a, kdepth, (hipStream_t)stream);
}

// filters of fpn_wino_kernel: [3 lateral k-groups][4 quarters][64 lanes][4] then [8 top-down k-groups][3 quarters][64 lanes][4]
extern "C" long dmvs_conv3d_wino_fpn_weight_floats(void) { return 3 * 4 * 256 + 8 * 3 * 256; }

extern "C" int dmvs_pack_conv_weights_wino_fpn(const float* w3, const float* w_lat, const float* b_lat, float* out) {
    if (!w3 || !w_lat || !b_lat || !out) return DMVS_EINVAL;
    constexpr int Cout = 16, Cin = 32, Cl = 8;
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    auto U = [&](const double g[9], int ya, int xb) {
        double u = 0.0;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) u += Gm[ya][ky] * Gm[xb][kx] * g[ky * 3 + kx];
        return u;
    };
    size_t n = 0;
    // lateral part: composite filters W3 o W_lat (channels 0-7) and W3 . b_lat (the ones plane, k-slot 0 of group 2)
    for (int c = 0; c < 3; ++c)
        for (int q = 0; q < 4; ++q)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 4; ++e) {
                    const int xi = 4 * q + e, co = l % 16, kslot = l / 16;
                    double g[9];
                    bool zero = false;
                    for (int t = 0; t < 9; ++t) {
                        double acc = 0.0;
                        for (int ci = 0; ci < Cin; ++ci) {
                            const double w = w3[((size_t)co * Cin + ci) * 9 + t];
                            acc += w * (c < 2 ? (double)w_lat[ci * Cl + 4 * c + kslot] : (double)b_lat[ci]);
                        }
                        g[t] = acc;
                    }
                    if (c == 2 && kslot > 0) zero = true;
                    out[n++] = zero ? 0.f : (float)U(g, xi / 4, xi % 4);
                }
    // top-down part: positions {0, 1, 3} x {0, 1, 3}; position 1 carries the factor 2 of B^T d = 2 T1
    static const int pos[3] = {0, 1, 3};
    for (int gk = 0; gk < 8; ++gk)
        for (int q = 0; q < 3; ++q)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 4; ++e) {
                    const int j = 4 * q + e, co = l % 16, ci = 4 * gk + l / 16;
                    if (j >= 9) { out[n++] = 0.f; continue; }
                    const int ya = pos[j / 3], xb = pos[j % 3];
                    double g[9];
                    for (int t = 0; t < 9; ++t) g[t] = w3[((size_t)co * Cin + ci) * 9 + t];
                    out[n++] = (float)(U(g, ya, xb) * (ya == 1 ? 2.0 : 1.0) * (xb == 1 ? 2.0 : 1.0));
                }
    (void)Cout;
    return n == (size_t)dmvs_conv3d_wino_fpn_weight_floats() ? 0 : DMVS_EINVAL;
}

extern "C" int dmvs_conv3d_wino_fpn2(const float* lat, const float* td, const float* ones_hw, float* out, const float* w_packed,
                                     const float* scale, const float* shift, int D, int H, int W, int flags,
                                     dmvs_stream_t stream) {
    if (!lat || !td || !ones_hw || !out || !w_packed || D < 1 || H < 2 || W < 8) return DMVS_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return DMVS_EINVAL;
    if (flags & ~(DMVS_RELU | DMVS_OUT_Q4)) return DMVS_EUNSUPPORTED;
    if ((H & 1) || (W & 7)) return DMVS_EUNSUPPORTED;
    if (((reinterpret_cast<uintptr_t>(lat) | reinterpret_cast<uintptr_t>(td) | reinterpret_cast<uintptr_t>(out) |
          reinterpret_cast<uintptr_t>(ones_hw)) & 15) != 0) return DMVS_EUNSUPPORTED;
    if ((long)16 * D * H * W >= (1L << 29)) return DMVS_EUNSUPPORTED;   // lat (8 ch), td (32 ch at 1/4) and out (16 ch) offsets
    WinoArgs a = {};
    a.lat = lat; a.td = td; a.out = out; a.w = w_packed; a.scale = scale; a.shift = shift;
    a.Cin = 32; a.Cout = 16; a.D = D; a.H = H; a.W = W; a.relu = (flags & DMVS_RELU) ? 1 : 0;
    a.nx = ceil_div(W, 32); a.ny = ceil_div(H, 16); a.nz = D;
    constexpr size_t lds = ((3 * 4 + 8 * 3) * 256 + 2 * (size_t)(9 * 736 + 32 * 240)) * sizeof(float);
    static_assert(lds <= 160 * 1024, "filters + two stages must fit the LDS");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = std::min(xcd_grid(a.nx * a.ny * a.nz), 256u);
    if (flags & DMVS_OUT_Q4) {
        if (int e = dmvs_ensure_dynamic_lds(reinterpret_cast<const void*>(fpn_wino_kernel<true>), lds)) return e;
        fpn_wino_kernel<true><<<dim3(grid), 512, lds, st>>>(a, ones_hw);
    } else {
        if (int e = dmvs_ensure_dynamic_lds(reinterpret_cast<const void*>(fpn_wino_kernel<false>), lds)) return e;
        fpn_wino_kernel<false><<<dim3(grid), 512, lds, st>>>(a, ones_hw);
    }
    DMVS_LAUNCH_CHECK();
}

