// K3 tail: conv11 (transposed conv 16 -> 8, BN + ReLU) + skip add + `prob` (conv 8 -> 2) in ONE kernel.
//
// Replaces the last three steps of CostRegNet_part(.forward) -- /root/reference/networks/module.py:376 (conv11),
// :396 / :434 (x = conv0 + conv11(x)) and :379 / :397 / :435 (prob) -- for both the 3D and the refine net (whose conv11
// and prob are 3D too, module.py:418-421).  Unfused (r01 / r02) the full-resolution 8-channel tensor t = conv0 +
// conv11(...) is written by the transposed-conv kernel (32 B per voxel and branch) and read again, with a halo, by
// the `prob` kernel: 64 of the ~112 B per voxel and branch the tail moves.  Here t only ever exists in LDS.
//
// A 3D tile with a one-voxel halo would recompute 3x the transposed-conv MFMA work at these channel counts; the
// workable form MARCHES ALONG DEPTH:
//   * a workgroup (4 waves) owns a footprint of 4 input rows x 32 input columns = an 8 x 64 tile of t, walks the
//     input planes gz = 0 .. Di-1 and keeps two input planes (ring) + all conv11 weights + the two t planes of the
//     current step in LDS (80.6 KB: two workgroups per CU, so one's MFMA phase overlaps the other's VALU phase);
//   * MFMA phase (v_mfma_f32_16x16x4_f32, the y-parity-merged layout of deconv_mfma_kernel: rows 0-7 / 8-15 of the A
//     operand = output parity py 0 / 1 of the 8 channels): wave w turns input rows iy0 + w, iy0 + w + 1 of planes gz,
//     gz + 1 into t rows 2w, 2w + 1 of output planes 2 gz, 2 gz + 1; epilogue = BN scale / shift, ReLU, + conv0 skip
//     (read from HBM), ZERO outside the volume (t is `prob`'s zero padding there), written to LDS;
//   * VALU phase: `prob` as a running sum over depth -- a thread owns 2 outputs x 2 channels of a tile row and three
//     accumulator sets (output planes z'-1, z', z'+1); every arriving t plane z' adds its 3x3 (ky, kx) window through
//     the kz = 2, 1, 0 weights, the oldest set is then complete and stored.  No t plane is ever kept beyond its step;
//   * tiles overlap by one input row / column (2 rows / columns of t): the inner 6 x 62 outputs of a tile are valid,
//     the MFMA phase recomputes 8/6 x 64/62 = 1.38x of conv11 (instead of 3x), the z direction nothing.
// Numerics: the MFMA k-order equals deconv_mfma_kernel's (same packed weights), `prob` sums taps in (ci, ky, kz, kx)
// order -- fp32 re-association against the unfused kernels (tests: 2e-5 abs on unit-scale data).
#include "common.h"
#include "tile_loader.h"
// development knock-outs (scripts/dev/tail_ko.sh): 1 no VALU phase, 2 no MFMAs, 4 no skip loads, 8 no input plane loads
#ifndef DMVS_TAIL_KO
#define DMVS_TAIL_KO 0
#endif

namespace {

typedef float acc4_t __attribute__((ext_vector_type(4)));

struct TailArgs {
    const float* in;     // [16][Di][Hi][Wi]   conv9 output (+ its skip)
    const float* skip;   // [8][Do][Ho][Wo]    this branch's slice of conv0's output
    const float* w11;    // conv11 weights, dmvs_pack_conv_weights_mfma(16, 8, DECONV_S2, 3) order: [4 chunks][18][64]
    const float* scale;  // [8] folded BatchNorm of conv11
    const float* shift;  // [8]
    const float* wprob;  // [27][8][2]  (dmvs_conv3d_direct layout)
    float* out;          // [2][Do][Ho][Wo]
    int Di, Hi, Wi, nx, ny, nzs, zlen;   // tile grid; nzs depth segments of zlen input planes
};

constexpr int CIN = 16, CMID = 8;
constexpr int IXP = 40, IY = 5, PS = 208;         // input tile: 5 rows x 40 floats per channel, channel stride 208 (bank spread)
constexpr int PLANE_F = CIN * PS;                 // one input plane of the ring
constexpr int W11_F = 4 * 18 * 64;                // all four channel chunks of conv11's packed weights
constexpr int TP = 66, T_CH = 8 * TP, T_PL = CMID * T_CH;   // t tile: [pz][ch][8 rows][66]
constexpr int WP_F = 27 * CMID * 2;
constexpr int LDS_F = 2 * PLANE_F + W11_F + 2 * T_PL + WP_F;

__global__ __launch_bounds__(256, 2) void reg_tail_kernel(TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const ring = smem;                          // [2][PLANE_F]
    float* const w11 = ring + 2 * PLANE_F;             // [72][64]
    float* const tb = w11 + W11_F;                     // [2][8][8][TP]
    float* const wp = tb + 2 * T_PL;                   // [8][27][2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lk = lane >> 4;
    int bx, by, bz;
    if (!xcd_tile(a.nx, a.ny, a.nzs, true, bx, by, bz)) return;
    const int Do = 2 * a.Di, Ho = 2 * a.Hi, Wo = 2 * a.Wi;
    const int ix0 = 31 * bx - 1, iy0 = 3 * by - 1;     // first input column / row of the footprint
    const int xa = ix0 & ~3, xoff = ix0 - xa;          // 16-byte aligned start of the staged rows
    // depth segment: outputs [2 za, 2 zb); the march starts one input plane early and ends one late (t halo planes)
    const int za = bz * a.zlen, zb = min(za + a.zlen, a.Di);
    const int g0 = max(za - 1, 0), g1 = min(zb + 1, a.Di);

    const int in_vol = a.Di * a.Hi * a.Wi;
    auto load_plane = [&](int gz, float* dst) {   // all 16 channels of input plane gz: wave w stages channels 4w .. 4w+3
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.in + (size_t)(4 * wave) * in_vol), (short)0, 4 * in_vol * 4, 0x00020000);
        load_tile4<4, 1, IY, IXP / 4, PS>(a.Di, a.Hi, a.Wi, rs, dst + 4 * wave * PS, gz, iy0, xa, 0, lane);
    };
    load_plane(g0, ring + (g0 & 1) * PLANE_F);
    load_plane(g0 + 1, ring + ((g0 + 1) & 1) * PLANE_F);
    for (int i = tid; i < W11_F; i += 256) w11[i] = a.w11[i];
    for (int i = tid; i < WP_F; i += 256) {   // [tap][ci][co] -> [ci][tap][co]: a channel's 27 pairs are one run
        const int co = i & 1, q = i >> 1, ci = q % CMID, t = q / CMID;
        wp[(ci * 27 + t) * 2 + co] = a.wprob[i];
    }

    // MFMA-phase constants of this lane: its 4 accumulator rows are channels ch0 .. ch0+3 of y parity py
    const int ch0 = (lk & 1) * 4, py = lk >> 1;
    float sc[4], sh[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) { sc[rr] = a.scale[ch0 + rr]; sh[rr] = a.shift[ch0 + rr]; }
    const int oy_m = 2 * (iy0 + wave) + py;                    // output row this lane's t values belong to
    const int trow = 2 * wave + py;                            // ... its row in the t tile
    const bool oy_ok = (unsigned)oy_m < (unsigned)Ho;
    const size_t out_plane = (size_t)Ho * Wo, out_vol = (size_t)Do * out_plane;
    // VALU-phase constants: thread (r, p) owns outputs at tile columns 2p+1, 2p+2 of tile row r
    const int r = tid >> 5, p = tid & 31;
    const int rm = max(r - 1, 0), rp = min(r + 1, 7);          // (rows 0 / 7 hold no valid output: clamped reads)
    const int oy_v = 2 * iy0 + r, ox_v = 2 * ix0 + 2 * p + 1;
    const bool row_valid = r >= 1 && r <= 6 && (unsigned)oy_v < (unsigned)Ho;
    const bool v0 = row_valid && p <= 30 && (unsigned)ox_v < (unsigned)Wo;
    const bool v1 = row_valid && p <= 30 && (unsigned)(ox_v + 1) < (unsigned)Wo;
    float2_t accA[2], accB[2], accC[2];   // outputs z'-1, z', z'+1 (x = channel 0, y = channel 1) of the 2 columns
#pragma unroll
    for (int k = 0; k < 2; ++k) { accA[k] = accB[k] = accC[k] = (float2_t){0.f, 0.f}; }

    auto store_plane = [&](const float2_t (&acc)[2], int z) {   // finished output plane z (uniform validity in z)
        if (z < 2 * za || z >= 2 * zb) return;
        float* o = a.out + (size_t)z * out_plane + (size_t)oy_v * Wo + ox_v;
        if (v0) { o[0] = acc[0].x; o[out_vol] = acc[0].y; }
        if (v1) { o[1] = acc[1].x; o[out_vol + 1] = acc[1].y; }
    };

    for (int gz = g0; gz < g1; ++gz) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // plane gz + 1 (and, first step, gz) landed: this wave's share
        __syncthreads();                                    // ... everyone's; the previous VALU phase is done with tb
        const float* pl0 = ring + (gz & 1) * PLANE_F;
        const float* pl1 = ring + ((gz + 1) & 1) * PLANE_F;

        // ---------------- MFMA phase: t planes 2 gz, 2 gz + 1 of this wave's two tile rows
        // the conv0 skip values of the lane's 32 t positions first: their HBM latency runs under the 144 MFMAs
        float2_t sk[2][2][4];
        bool okx[2];
#pragma unroll
        for (int xb = 0; xb < 2; ++xb) {
            const int ox = 2 * (ix0 + xb * 16 + ln);
            okx[xb] = oy_ok && (unsigned)ox < (unsigned)Wo;   // Wo is even: both x parities share the test
#pragma unroll
            for (int pz = 0; pz < 2; ++pz) {
                const float* sp = a.skip + (size_t)(2 * gz + pz) * out_plane + (size_t)oy_m * Wo + ox;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    sk[pz][xb][rr] = (okx[xb] && !(DMVS_TAIL_KO & 4)) ? *reinterpret_cast<const float2_t*>(sp + (size_t)(ch0 + rr) * out_vol) : (float2_t){0.f, 0.f};
            }
        }
        acc4_t acc[2][2][2];   // [pz][px][xb]
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) acc[i >> 2][(i >> 1) & 1][i & 1][rr] = 0.f;
        const int boff = lk * PS + wave * IXP + ln + xoff;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* wl = w11 + c * 18 * 64 + lane;
            int step = 0;
#pragma unroll
            for (int oz = 0; oz < 2; ++oz) {
                const float* pl = (oz ? pl1 : pl0) + c * 4 * PS + boff;
#pragma unroll
                for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                    for (int ox = 0; ox < 2; ++ox) {
                        float bv[2];
#pragma unroll
                        for (int xb = 0; xb < 2; ++xb) bv[xb] = pl[oy * IXP + xb * 16 + ox];
#pragma unroll
                        for (int pz = oz; pz < 2; ++pz)
#pragma unroll
                            for (int px = ox; px < 2; ++px) {
                                const float av = wl[step * 64];
                                ++step;
#pragma unroll
                                for (int xb = 0; xb < 2; ++xb) {
                                    if (DMVS_TAIL_KO & 2) acc[pz][px][xb][0] = fmaf(av, bv[xb], acc[pz][px][xb][0]);
                                    else acc[pz][px][xb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[xb], acc[pz][px][xb], 0, 0, 0);
                                }
                            }
                    }
            }
        }
        // epilogue: BN + ReLU + conv0 skip, zero outside the volume, into the t tile
#pragma unroll
        for (int pz = 0; pz < 2; ++pz) {
#pragma unroll
            for (int xb = 0; xb < 2; ++xb) {
                float* tp = tb + pz * T_PL + trow * TP + 2 * (xb * 16 + ln);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    float2_t v;
                    v.x = okx[xb] ? fmaxf(acc[pz][0][xb][rr] * sc[rr] + sh[rr], 0.f) + sk[pz][xb][rr].x : 0.f;
                    v.y = okx[xb] ? fmaxf(acc[pz][1][xb][rr] * sc[rr] + sh[rr], 0.f) + sk[pz][xb][rr].y : 0.f;
                    *reinterpret_cast<float2_t*>(tp + (ch0 + rr) * T_CH) = v;
                }
            }
        }
        __syncthreads();   // t complete; every wave is done with input plane gz
        if (gz + 1 < g1 && !(DMVS_TAIL_KO & 8)) load_plane(gz + 2, ring + (gz & 1) * PLANE_F);   // (plane Di: zeros, via the range check)

        // ---------------- VALU phase: the two new t planes feed the running `prob` sums
#pragma unroll
        for (int pz = 0; pz < 2; ++pz) {
            const float* tpl = tb + pz * T_PL + 2 * p;
#pragma unroll 1
            for (int ci = 0; ci < ((DMVS_TAIL_KO & 1) ? 1 : CMID); ++ci) {
                float2_t wreg[27];
#pragma unroll
                for (int t = 0; t < 27; ++t) wreg[t] = *reinterpret_cast<const float2_t*>(wp + (ci * 27 + t) * 2);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float* row = tpl + ci * T_CH + (ky == 0 ? rm : (ky == 1 ? r : rp)) * TP;
                    const float2_t x01 = *reinterpret_cast<const float2_t*>(row);
                    const float2_t x23 = *reinterpret_cast<const float2_t*>(row + 2);
                    const float x[4] = {x01.x, x01.y, x23.x, x23.y};
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float2_t w2 = wreg[(2 * 3 + ky) * 3 + kx], w1 = wreg[(1 * 3 + ky) * 3 + kx], w0 = wreg[ky * 3 + kx];
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const float2_t xv = (float2_t){x[kx + k], x[kx + k]};
                            accA[k] = __builtin_elementwise_fma(w2, xv, accA[k]);   // out z'-1 <- kz = 2
                            accB[k] = __builtin_elementwise_fma(w1, xv, accB[k]);   // out z'   <- kz = 1
                            accC[k] = __builtin_elementwise_fma(w0, xv, accC[k]);   // out z'+1 <- kz = 0
                        }
                    }
                }
            }
            store_plane(accA, 2 * gz + pz - 1);
#pragma unroll
            for (int k = 0; k < 2; ++k) { accA[k] = accB[k]; accB[k] = accC[k]; accC[k] = (float2_t){0.f, 0.f}; }
        }
    }
    store_plane(accA, 2 * g1 - 1);   // the last plane of the march (t beyond it is zero padding or another segment's)
}

}  // namespace

extern "C" int dmvs_reg_tail(const float* in16, const float* skip8, const float* w11_packed, const float* scale,
                             const float* shift, const float* w_prob, float* out2, int Di, int Hi, int Wi,
                             dmvs_stream_t stream) {
    if (!in16 || !skip8 || !w11_packed || !scale || !shift || !w_prob || !out2 || Di < 1 || Hi < 1 || Wi < 1) return DMVS_EINVAL;
    if ((Wi & 3) || (reinterpret_cast<uintptr_t>(in16) & 15) || (reinterpret_cast<uintptr_t>(skip8) & 7))
        return DMVS_EUNSUPPORTED;   // 16-byte input rows, 8-byte skip pairs: the caller runs the two layers separately
    if ((long)4 * Di * Hi * Wi >= (1L << 28) || (long)64 * Di * Hi * Wi >= (1L << 31)) return DMVS_EUNSUPPORTED;
    TailArgs a;
    a.in = in16; a.skip = skip8; a.w11 = w11_packed; a.scale = scale; a.shift = shift; a.wprob = w_prob; a.out = out2;
    a.Di = Di; a.Hi = Hi; a.Wi = Wi;
    a.nx = ceil_div(2 * Wi + 1, 62);
    a.ny = ceil_div(2 * Hi + 1, 6);
    // depth segments: enough workgroups for 2 per CU x 256 CUs x ~3 rounds, but never shorter than 4 input planes
    // (every segment re-marches 2 halo planes)
    int nzs = 1;
    while (nzs < 8 && (long)a.nx * a.ny * nzs < 1536 && ceil_div(Di, nzs * 2) >= 4) nzs *= 2;
    a.zlen = ceil_div(Di, nzs);
    a.nzs = ceil_div(Di, a.zlen);
    const size_t lds = (size_t)LDS_F * sizeof(float);
    if (int e = dmvs_ensure_dynamic_lds(reinterpret_cast<const void*>(reg_tail_kernel), lds)) return e;
    reg_tail_kernel<<<dim3(xcd_grid(a.nx * a.ny * a.nzs)), 256, lds, (hipStream_t)stream>>>(a);
    DMVS_LAUNCH_CHECK();
}
