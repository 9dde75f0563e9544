// K2: direct LDS-tiled 3D convolution / transposed convolution (fp32 VALU) with fused
// BatchNorm(eval) scale/shift + ReLU + residual-add epilogue.
//
// Replaces the Conv3d / Deconv3d wrappers of /root/reference/networks/module.py:120-208 as used by
// CostRegNet_part (module.py:358-398) and CostRegNet_part_refine (module.py:400-436): in the reference
// every layer is conv -> batch_norm -> relu (-> add) = 3-4 passes over the activation; here one.
//
// Layout: planar fp32 [C][D][H][W] (B=1), x fastest.  A 256-thread block owns an output tile, stages the
// input tile (+halo) of CIN_B channels in LDS, and every thread accumulates PX consecutive x outputs for
// COUT_B output channels in registers.  Weights are packed [tap][Cin][Cout]; their index is wave-uniform
// so they arrive through the scalar cache (s_load) and feed v_fma as SGPR operands.
//
// Transposed conv (k3 s2 p1 output_padding 1) is done in gather form on the INPUT grid: input position i
// produces outputs 2i (tap k=1 of i) and 2i+1 (tap k=2 of i, tap k=0 of i+1) per axis, so one thread
// owning input position (z,y,x) produces the 2x2x2 output block and needs the 2x2x2 input neighbourhood
// (SURVEY.md section 9).  ConvTranspose weights are [Cin][Cout][k][k][k]; packing only re-indexes them.
//
// kdepth=1 runs a 1x3x3 kernel per depth slice (depth stride 1): the 2D bottleneck of the refine net.
#include "common.h"
#include "tile_loader.h"

struct ConvArgs {
    const float* in;
    float* out;
    const float* w;      // [taps][Cin][Cout]
    const float* scale;  // [Cout] or null
    const float* shift;  // [Cout] or null
    const float* skip;   // like out, or null
    int Cin, Cout, D, H, W, Do, Ho, Wo, relu;
    int nx, ny, nz;      // tile grid of the 1-D XCD-ordered launch (conv_cout2_kernel only)
    // conv_cout2_kernel<..., NZB > 0> only (prob + the branch's half of K4, dmvs_prob_regress):
    const float* hyp;       // [D][H][W] hypothesis planes, or null: plane d = hyp_base[pix] + d * hyp_step[0]
    const float* hyp_base;  // [H][W]
    const float* hyp_step;  // [1]
    float alpha;            // logits are scaled by it before the softmax (DepthNet.refine: 5)
    float* dsp;             // [2][H][W]: the branch's two depth expectations
};

__device__ __forceinline__ float epilogue(const ConvArgs& a, float v, int co, size_t oidx) {
    if (a.scale) v = v * a.scale[co] + a.shift[co];
    if (a.relu) v = fmaxf(v, 0.f);
    if (a.skip) v += a.skip[oidx];
    return v;
}

// ------------------------------------------------------------------------- conv, stride 1 or 2
template <int STRIDE, int KD, int CIN_B, int COUT_B, int TZ, int TY, int TXT, int PX>
__global__ __launch_bounds__(TZ* TY* TXT) void conv_direct_kernel(ConvArgs a) {
    constexpr int NT = TZ * TY * TXT;
    constexpr int TX = TXT * PX;
    constexpr int SZ = (KD == 3) ? STRIDE : 1;  // depth stride
    constexpr int IZ = (KD == 3) ? (TZ - 1) * STRIDE + 3 : TZ;
    constexpr int IY = (TY - 1) * STRIDE + 3;
    constexpr int IX = (TX - 1) * STRIDE + 3;
    constexpr int IXP = (IX + 3) & ~3;
    constexpr int NR = (PX - 1) * STRIDE + 3;  // input row values a thread needs
    __shared__ float tile[CIN_B * IZ * IY * IXP];

    const int ntz = (a.Do + TZ - 1) / TZ;
    const int bz = blockIdx.z % ntz, co0 = (blockIdx.z / ntz) * COUT_B;
    const int oz0 = bz * TZ, oy0 = blockIdx.y * TY, ox0 = blockIdx.x * TX;
    const int iz0 = (KD == 3) ? oz0 * STRIDE - 1 : oz0, iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;
    const int tid = threadIdx.x;
    const int tx = tid % TXT, ty = (tid / TXT) % TY, tz = tid / (TXT * TY);

    float acc[PX][COUT_B];
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int c = 0; c < COUT_B; ++c) acc[p][c] = 0.f;

    const size_t in_plane = (size_t)a.H * a.W;
    for (int ci0 = 0; ci0 < a.Cin; ci0 += CIN_B) {
        for (int idx = tid; idx < CIN_B * IZ * IY * IX; idx += NT) {
            const int x = idx % IX, y = (idx / IX) % IY, z = (idx / (IX * IY)) % IZ, c = idx / (IX * IY * IZ);
            const int gz = iz0 + z, gy = iy0 + y, gx = ix0 + x;
            // unconditional load from a clamped address + select: keeps the loads of the unrolled loop in flight
            const bool ok = gz >= 0 && gz < a.D && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const int gzc = min(max(gz, 0), a.D - 1), gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
            const float v = a.in[((size_t)(ci0 + c) * a.D + gzc) * in_plane + (size_t)gyc * a.W + gxc];
            tile[((c * IZ + z) * IY + y) * IXP + x] = ok ? v : 0.f;
        }
        __syncthreads();
#pragma unroll 1
        for (int c = 0; c < CIN_B; ++c) {
#pragma unroll
            for (int kz = 0; kz < KD; ++kz) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float* row = &tile[((c * IZ + tz * SZ + kz) * IY + ty * STRIDE + ky) * IXP + tx * PX * STRIDE];
                    float r[NR];
#pragma unroll
                    for (int i = 0; i < NR; ++i) r[i] = row[i];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float* wp = a.w + ((size_t)((kz * 3 + ky) * 3 + kx) * a.Cin + ci0 + c) * a.Cout + co0;
#pragma unroll
                        for (int co = 0; co < COUT_B; ++co) {
                            const float wv = wp[co];
#pragma unroll
                            for (int p = 0; p < PX; ++p) acc[p][co] = fmaf(wv, r[p * STRIDE + kx], acc[p][co]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    const int oz = oz0 + tz, oy = oy0 + ty;
    if (oz >= a.Do || oy >= a.Ho) return;
    const size_t out_plane = (size_t)a.Ho * a.Wo;
#pragma unroll
    for (int co = 0; co < COUT_B; ++co) {
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const int ox = ox0 + tx * PX + p;
            if (ox < a.Wo) {
                const size_t o = ((size_t)(co0 + co) * a.Do + oz) * out_plane + (size_t)oy * a.Wo + ox;
                a.out[o] = epilogue(a, acc[p][co], co0 + co, o);
            }
        }
    }
}

// ------------------------------------------------------------------------- transposed conv, stride 2
// tap index along one axis for (output parity p, input offset o): p0/o0 -> 1, p1/o0 -> 2, p1/o1 -> 0.
template <int KD, int CIN_B, int COUT_B, int TZ, int TY, int TX>
__global__ __launch_bounds__(TZ* TY* TX) void deconv_direct_kernel(ConvArgs a) {
    constexpr int NT = TZ * TY * TX;
    constexpr int IZ = (KD == 3) ? TZ + 1 : TZ, IY = TY + 1, IX = TX + 1;
    constexpr int NPZ = (KD == 3) ? 2 : 1;  // output parities along depth
    __shared__ float tile[CIN_B * IZ * IY * IX];

    const int ntz = (a.D + TZ - 1) / TZ;
    const int bz = blockIdx.z % ntz, co0 = (blockIdx.z / ntz) * COUT_B;
    const int iz0 = bz * TZ, iy0 = blockIdx.y * TY, ix0 = blockIdx.x * TX;
    const int tid = threadIdx.x;
    const int tx = tid % TX, ty = (tid / TX) % TY, tz = tid / (TX * TY);

    float acc[NPZ][2][2][COUT_B];
#pragma unroll
    for (int pz = 0; pz < NPZ; ++pz)
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int c = 0; c < COUT_B; ++c) acc[pz][py][px][c] = 0.f;

    const size_t in_plane = (size_t)a.H * a.W;
    for (int ci0 = 0; ci0 < a.Cin; ci0 += CIN_B) {
        for (int idx = tid; idx < CIN_B * IZ * IY * IX; idx += NT) {
            const int x = idx % IX, y = (idx / IX) % IY, z = (idx / (IX * IY)) % IZ, c = idx / (IX * IY * IZ);
            const int gz = iz0 + z, gy = iy0 + y, gx = ix0 + x;
            const bool ok = gz < a.D && gy < a.H && gx < a.W;
            const float v = a.in[((size_t)(ci0 + c) * a.D + min(gz, a.D - 1)) * in_plane + (size_t)min(gy, a.H - 1) * a.W + min(gx, a.W - 1)];
            tile[((c * IZ + z) * IY + y) * IX + x] = ok ? v : 0.f;
        }
        __syncthreads();
#pragma unroll 1
        for (int c = 0; c < CIN_B; ++c) {
#pragma unroll
            for (int oz = 0; oz < NPZ; ++oz)
#pragma unroll
                for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                    for (int ox = 0; ox < 2; ++ox) {
                        const float v = tile[((c * IZ + tz + oz) * IY + ty + oy) * IX + tx + ox];
#pragma unroll
                        for (int pz = oz; pz < NPZ; ++pz)  // parity 0 only pairs with offset 0
#pragma unroll
                            for (int py = oy; py < 2; ++py)
#pragma unroll
                                for (int px = ox; px < 2; ++px) {
                                    const int kz = (KD == 3) ? (pz == 0 ? 1 : (oz == 0 ? 2 : 0)) : 0;
                                    const int ky = py == 0 ? 1 : (oy == 0 ? 2 : 0);
                                    const int kx = px == 0 ? 1 : (ox == 0 ? 2 : 0);
                                    const float* wp = a.w + ((size_t)((kz * 3 + ky) * 3 + kx) * a.Cin + ci0 + c) * a.Cout + co0;
#pragma unroll
                                    for (int co = 0; co < COUT_B; ++co)
                                        acc[pz][py][px][co] = fmaf(wp[co], v, acc[pz][py][px][co]);
                                }
                    }
        }
        __syncthreads();
    }

    const int iz = iz0 + tz, iy = iy0 + ty, ix = ix0 + tx;
    if (iz >= a.D || iy >= a.H || ix >= a.W) return;
    const size_t out_plane = (size_t)a.Ho * a.Wo;
#pragma unroll
    for (int co = 0; co < COUT_B; ++co)
#pragma unroll
        for (int pz = 0; pz < NPZ; ++pz)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int oz = (KD == 3) ? 2 * iz + pz : iz;
                const size_t o = ((size_t)(co0 + co) * a.Do + oz) * out_plane + (size_t)(2 * iy + py) * a.Wo + 2 * ix;
                float2_t v;
                v.x = epilogue(a, acc[pz][py][0][co], co0 + co, o);
                v.y = epilogue(a, acc[pz][py][1][co], co0 + co, o + 1);
                *reinterpret_cast<float2_t*>(a.out + o) = v;
            }
}

// ------------------------------------------------------------------------- Cout = 2 ("prob"), stride 1
// The `prob` head (nn.Conv3d(8, 2, 3, padding=1, bias=False), module.py:379,421) has too few output channels
// for the matrix cores (M = 2 of 16 rows) and runs at full resolution on both branches of every stage-pass, so
// it gets its own VALU kernel.  No BN / ReLU / residual.
//  * a thread owns PX = 8 consecutive x outputs of both channels as 8 PACKED accumulators (v_pk_fma_f32: the weight
//    pair (co 0, co 1) of a (tap, ci) is one operand, the input value is broadcast through op_sel); per (plane, ky)
//    it reads one 10-float row segment from LDS (three ds_read_b128) and issues 24 packed FMAs; the channel's 27
//    weight pairs are read once per channel from LDS ([Cin][tap][2], broadcast reads with immediate offsets) into
//    registers -- not from the scalar cache: s_load and ds_read share lgkmcnt, mixing them drains the counter;
//  * lane -> (x segment, row) is permuted so that every ds_read_b128 is bank-conflict free (see below);
//  * stages of CIN_B = 1 channel, double buffered with asynchronous LDS-direct loads (small stages = 5 workgroups
//    per CU; 3- / 4-stage rings with counted vmcnt, NS > 2, were slower), XCD-aware tile order;
//  * V4: 16-byte tile loads (load_tile4, 9 instead of 27 load instructions per wave and stage).  The 36-float row
//    pitch that makes the reads conflict-free holds the 34 floats of a 32-wide tile only if the row starts at the
//    tile's left halo, and a 16-byte load needs the row to start at a multiple of 4: so V4 tiles are SHIFTED by one
//    voxel -- tile bx owns outputs 32 bx - 31 .. 32 bx, its rows start at 32 bx - 32 -- at the price of one extra
//    tile column and 4-byte-aligned output stores.  Otherwise (W % 4 != 0) the dword row loader;
//  * PZ: depth outputs per thread (PZ = 2: 4 input planes feed 2 output planes, fewer LDS reads per FMA; measured
//    slower than PZ = 1 because of the larger tile / lower occupancy, kept as a template parameter).
// Loads and compute overlap almost completely and are balanced (knock-outs in DESIGN.md): the layer runs within
// ~25 % of its memory floor.
//  * NZB > 0 (r06, VERDICT r05 item 6: `prob` -> K4): the workgroup owns ALL D = NZB * TZ planes of its (y, x) tile -- NZB z blocks one
//    after the other in the same chunk pipeline, the finished block's accumulators parked in registers --, the four waves exchange
//    their logits through the (then idle) LDS stages and every thread regresses four (pixel, channel) columns: softmax over D and
//    the depth expectation, the same operations in the same order as depth_regress_kernel (bit-identical).  Written: the branch's
//    [2][H][W] expectations instead of its [2][D][H][W] logits; K4's remainder is dmvs_depth_select on [4][H][W].
template <int CIN_B, int TZ, int TY, bool V4, int NS, int PZ, int NZB = 0>
__global__ __launch_bounds__(256) void conv_cout2_kernel(ConvArgs a) {
    constexpr int PX = 8, TXT = 4, TX = PX * TXT;  // 32 outputs in x per block
    constexpr int NB = NZB > 0 ? NZB : 1;          // z blocks walked by the workgroup
    static_assert(NZB == 0 || (PZ == 1 && CIN_B == 1 && NS == 2 && TY == 16 && TZ == 4), "fused regression: the product tile only");
    constexpr int IZ = TZ * PZ + 2, IY = TY + 2, IX = TX + 2;
    constexpr int IXP = 36;
    constexpr int PS = IZ * IY * IXP;
    constexpr int BUF_F = (CIN_B * PS + 63) & ~63;
    static_assert(TZ * TY * TXT == 256, "tile must map onto 256 threads");
    static_assert(V4 || NS == 2, "the counted wait needs load_tile4's uniform loads per wave");
    constexpr int LPW = CIN_B * ((IZ * IY + (64 / (IXP / 4)) - 1) / (64 / (IXP / 4)) + 3) / 4;  // loads per wave per stage
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [NS][BUF_F] tiles, then the weights (<= 27*16*2 floats)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // lane -> (tx, ty): a ds_read_b128 is served in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} and
    // the same + 32, each group touching all 64 banks once when its 16 pieces tile 64 consecutive banks.  With a
    // 36-float row pitch rows k, k+1, k+8, k+9 start at banks 0, 36, 32, 4: together with the four 8-float thread
    // strides they tile the banks exactly, so group g takes rows {2g, 2g+1, 2g+8, 2g+9} (the natural lane / 4 row
    // order gives every group a 2-way conflict: measured half of the kernel's LDS cycles).
    int tx, ty;
    if constexpr (TY == 16) {
        const int h = lane & 31;                                   // position inside the 32-lane half
        const bool g1 = (h >= 4 && h < 12) || (h >= 16 && h < 20) || h >= 28;
        const int j = g1 ? (h < 12 ? h - 4 : h < 20 ? h - 8 : h - 16)   // rank inside the group
                         : (h < 4 ? h : h < 16 ? h - 8 : h - 12);
        const int g = (lane >> 5) * 2 + (g1 ? 1 : 0), q = j >> 2;
        tx = j & 3;
        ty = 2 * g + (q & 1) + 8 * (q >> 1);
    } else {
        tx = tid % TXT;
        ty = (tid / TXT) % TY;
    }
    const int tz = tid / (TXT * TY);
    int bx, by, bz;
    if (!xcd_tile(a.nx, a.ny, a.nz, true, bx, by, bz)) return;
    const int ox0 = V4 ? bx * TX - (TX - 1) : bx * TX, oy0 = by * TY, oz0 = bz * TZ * PZ;  // first output voxel

    float2_t acc[PZ][PX];
#pragma unroll
    for (int j = 0; j < PZ; ++j)
#pragma unroll
        for (int p = 0; p < PX; ++p) acc[j][p] = (float2_t){0.f, 0.f};
    float2_t held[NB > 1 ? PX : 1];   // NZB = 2: the first z block's logits while the second one accumulates

    const int in_vol = a.D * a.H * a.W;
    const int cpb = a.Cin / CIN_B;          // chunks per z block
    const int nchunks = cpb * NB;
    auto stage = [&](int cc, float* dst) {  // chunk cc = (z block, channels [c * CIN_B, (c + 1) * CIN_B)) of the tile, asynchronous
        const int zb = NB > 1 ? cc / cpb : 0, c = NB > 1 ? cc - zb * cpb : cc;
        const int oz = oz0 + zb * TZ * PZ;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.in + (size_t)(c * CIN_B) * in_vol), (short)0, CIN_B * in_vol * 4, 0x00020000);
        if constexpr (V4)
            load_tile4<CIN_B, IZ, IY, IXP / 4, PS>(a.D, a.H, a.W, rs, dst, oz - 1, oy0 - 1, ox0 - 1, wave, lane);
        else
            load_tile<CIN_B, IZ, IY, IX, IXP, PS, true>(a.D, a.H, a.W, rs, dst, c * CIN_B, oz - 1, oy0 - 1, ox0 - 1, wave, lane);
    };
    // Weights go through LDS, not the scalar cache: s_load and ds_read share the lgkmcnt counter and scalar loads
    // return out of order, so a loop that mixes them drains to lgkmcnt(0) at every weight use.  Layout
    // every lane reads the same address (LDS broadcast, conflict-free); global layout [tap][Cin][2] ...
    float* wl = smem + NS * BUF_F;
    // ... re-ordered to [Cin][tap][2] so that a channel's 27 pairs are one run read with immediate offsets
    for (int i = tid; i < 27 * a.Cin * 2; i += 256) {
        const int co = i & 1, q = i >> 1, ci = q % a.Cin, t = q / a.Cin;
        wl[(ci * 27 + t) * 2 + co] = a.w[i];
    }
#pragma unroll
    for (int c = 0; c < NS - 1; ++c)
        if (c < nchunks) stage(c, smem + c * BUF_F);
    for (int c = 0; c < nchunks; ++c) {
        // stage c has landed (this wave's share); up to NS - 2 younger stages stay in flight
        const int younger = min(NS - 2, nchunks - 1 - c);
        if (NS >= 4 && younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPW) : "memory");
        else if (NS >= 3 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + NS - 1 < nchunks) stage(c + NS - 1, smem + ((c + NS - 1) % NS) * BUF_F);
        const float* tile = smem + (c % NS) * BUF_F + (tz * PZ * IY + ty) * IXP + tx * PX;
        if (NB > 1 && c == cpb) {   // the second z block starts: park the first one's logits
#pragma unroll
            for (int p = 0; p < PX; ++p) { held[p] = acc[0][p]; acc[0][p] = (float2_t){0.f, 0.f}; }
        }
        const int cw = NB > 1 ? (c >= cpb ? c - cpb : c) : c;   // the chunk's channel block
#pragma unroll
        for (int ci = 0; ci < CIN_B; ++ci) {
            const float* wc = wl + (cw * CIN_B + ci) * 54;
            float2_t wreg[27];
#pragma unroll
            for (int t = 0; t < 27; ++t) wreg[t] = *reinterpret_cast<const float2_t*>(wc + t * 2);
#pragma unroll
            for (int q = 0; q < PZ + 2; ++q)  // input plane q of the thread's column feeds outputs j = q - kz
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float* row = tile + ci * PS + (q * IY + ky) * IXP;
                    const float4_t r0 = *reinterpret_cast<const float4_t*>(row);
                    const float4_t r1 = *reinterpret_cast<const float4_t*>(row + 4);
                    const float4_t r2 = *reinterpret_cast<const float4_t*>(row + 8);  // .z/.w unused: a b128 tiles the banks, a b64 would not
                    const float r[PX + 2] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y};
#pragma unroll
                    for (int j = 0; j < PZ; ++j) {
                        const int kz = q - j;
                        if (kz < 0 || kz > 2) continue;
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float2_t wt = wreg[(kz * 3 + ky) * 3 + kx];
#pragma unroll
                            for (int p = 0; p < PX; ++p) {
                                const float x = r[p + kx];
                                acc[j][p] = __builtin_elementwise_fma(wt, (float2_t){x, x}, acc[j][p]);
                            }
                        }
                    }
                }
        }
    }

    const size_t plane = (size_t)a.H * a.W;
    if constexpr (NZB > 0) {
        constexpr int DD = NZB * TZ;                 // all planes of the volume
        // the hypothesis planes of the thread's two pixels (rows r0, r0 + 8 of the tile, column tid & 31): issued before the
        // exchange so that they fly under it
        const int ex_x = tid & 31, ex_r = tid >> 5;
        const int gx = ox0 + ex_x;
        float dep[2][DD];
        bool live[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int gy = oy0 + ex_r + 8 * k;
            live[k] = gx >= 0 && gx < a.W && gy < a.H;
            const size_t pix = (size_t)min(gy, a.H - 1) * a.W + min(max(gx, 0), a.W - 1);
#pragma unroll
            for (int d = 0; d < DD; ++d)
                dep[k][d] = a.hyp ? a.hyp[(size_t)d * plane + pix] : a.hyp_base[pix] + (float)d * a.hyp_step[0];
        }
        __syncthreads();   // every wave is done with the tiles and the weights: the LDS becomes the exchange [2][DD][16][32]
        float* ex = smem;
        static_assert((size_t)2 * DD * TY * 32 <= (size_t)NS * BUF_F + 27 * 16 * 2, "the logits of a tile fit the idle LDS");
#pragma unroll
        for (int zb = 0; zb < NB; ++zb) {
            const int pl = zb * TZ + tz;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4_t v0, v1;
                const float2_t* src = (NB > 1 && zb == 0) ? held : acc[0];
                v0.x = src[4 * h].x; v0.y = src[4 * h + 1].x; v0.z = src[4 * h + 2].x; v0.w = src[4 * h + 3].x;
                v1.x = src[4 * h].y; v1.y = src[4 * h + 1].y; v1.z = src[4 * h + 2].y; v1.w = src[4 * h + 3].y;
                *reinterpret_cast<float4_t*>(ex + ((0 * DD + pl) * TY + ty) * 32 + tx * PX + 4 * h) = v0;
                *reinterpret_cast<float4_t*>(ex + ((1 * DD + pl) * TY + ty) * 32 + tx * PX + 4 * h) = v1;
            }
        }
        __syncthreads();
        // softmax over the planes + expectation, operation for operation what depth_regress_kernel<false, DD> does for a channel
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                float v[DD];
#pragma unroll
                for (int d = 0; d < DD; ++d) v[d] = ex[((ch * DD + d) * TY + ex_r + 8 * k) * 32 + ex_x] * a.alpha;
                float m = -INFINITY, sum = 0.f, e = 0.f;
#pragma unroll
                for (int d = 0; d < DD; ++d) m = fmaxf(m, v[d]);
#pragma unroll
                for (int d = 0; d < DD; ++d) { v[d] = expf(v[d] - m); sum += v[d]; }
#pragma unroll
                for (int d = 0; d < DD; ++d) { const float pr = v[d] / sum; e += pr * dep[k][d]; }
                if (live[k]) a.dsp[(size_t)ch * plane + (size_t)(oy0 + ex_r + 8 * k) * a.W + gx] = e;
            }
        }
        return;
    }
    const int oy = oy0 + ty, ox = ox0 + tx * PX;
    typedef float float4u_t __attribute__((ext_vector_type(4), aligned(4)));  // shifted tiles: 4-byte aligned runs
#pragma unroll
    for (int j = 0; j < PZ; ++j) {
        const int oz = oz0 + tz * PZ + j;
        if (oz >= a.D || oy >= a.H || ox >= a.W || ox + PX <= 0) continue;
        float* o0 = a.out + (size_t)oz * plane + (size_t)oy * a.W + ox;
        float* o1 = o0 + (size_t)a.D * plane;
        if (ox >= 0 && ox + PX <= a.W) {
            float4u_t v;
            v.x = acc[j][0].x; v.y = acc[j][1].x; v.z = acc[j][2].x; v.w = acc[j][3].x; *reinterpret_cast<float4u_t*>(o0) = v;
            v.x = acc[j][4].x; v.y = acc[j][5].x; v.z = acc[j][6].x; v.w = acc[j][7].x; *reinterpret_cast<float4u_t*>(o0 + 4) = v;
            v.x = acc[j][0].y; v.y = acc[j][1].y; v.z = acc[j][2].y; v.w = acc[j][3].y; *reinterpret_cast<float4u_t*>(o1) = v;
            v.x = acc[j][4].y; v.y = acc[j][5].y; v.z = acc[j][6].y; v.w = acc[j][7].y; *reinterpret_cast<float4u_t*>(o1 + 4) = v;
        } else {
#pragma unroll
            for (int p = 0; p < PX; ++p)
                if (ox + p >= 0 && ox + p < a.W) { o0[p] = acc[j][p].x; o1[p] = acc[j][p].y; }
        }
    }
}

template <int CIN_B, int TZ, int TY, bool V4, int NS, int PZ>
static int launch_cout2_v(ConvArgs a, hipStream_t st) {
    static_assert(NS <= 4, "the counted wait handles up to 2 younger stages");
    constexpr int PS = (TZ * PZ + 2) * (TY + 2) * 36;
    constexpr size_t lds = (NS * (size_t)((CIN_B * PS + 63) & ~63) + 27 * 16 * 2) * sizeof(float);
    if (a.Cin > 16) return DMVS_EUNSUPPORTED;
    static_assert(lds <= 160 * 1024, "two pipeline stages must fit the 160 KB LDS");
    if (int e = dmvs_ensure_dynamic_lds(reinterpret_cast<const void*>(conv_cout2_kernel<CIN_B, TZ, TY, V4, NS, PZ>), lds)) return e;
    a.nx = V4 ? ceil_div(a.W - 1, 32) + 1 : ceil_div(a.W, 32);  // V4 tiles are shifted by one voxel
    a.ny = ceil_div(a.H, TY); a.nz = ceil_div(a.D, TZ * PZ);
    conv_cout2_kernel<CIN_B, TZ, TY, V4, NS, PZ><<<dim3(xcd_grid(a.nx * a.ny * a.nz)), 256, lds, st>>>(a);
    DMVS_LAUNCH_CHECK();
}

// prob + the branch's half of K4 (conv_cout2_kernel<..., NZB>): one workgroup per (y, x) tile walks all D = 4 * NZB planes
template <int NZB>
static int launch_cout2_fused(ConvArgs a, hipStream_t st) {
    constexpr int PS = 6 * 18 * 36;
    constexpr size_t lds = (2 * (size_t)((PS + 63) & ~63) + 27 * 16 * 2) * sizeof(float);
    auto kernel = conv_cout2_kernel<1, 4, 16, true, 2, 1, NZB>;
    if (int e = dmvs_ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds)) return e;
    a.nx = ceil_div(a.W - 1, 32) + 1; a.ny = ceil_div(a.H, 16); a.nz = 1;   // (V4: tiles shifted by one voxel)
    kernel<<<dim3(xcd_grid(a.nx * a.ny)), 256, lds, st>>>(a);
    DMVS_LAUNCH_CHECK();
}

extern "C" int dmvs_prob_regress(const float* in, const float* w_packed, int Cin, int D, int H, int W, const float* hyp_dhw,
                                 const float* base_hw, const float* step, float alpha, float* dsp_2hw, dmvs_stream_t stream) {
    if (!in || !w_packed || !dsp_2hw || Cin < 2 || D < 1 || H < 1 || W < 1) return DMVS_EINVAL;
    if (!hyp_dhw && (!base_hw || !step)) return DMVS_EINVAL;
    // the shapes the fused form is built for: a workgroup holds all planes (4 or 8), 16-byte tile loads; everything else stays on
    // dmvs_conv3d_direct + dmvs_depth_regress
    if ((D != 4 && D != 8) || Cin % 2 || Cin > 16 || W % 4 || (reinterpret_cast<uintptr_t>(in) & 15) || (long)2 * D * H * W >= (1L << 28))
        return DMVS_EUNSUPPORTED;
    ConvArgs a = {};
    a.in = in; a.w = w_packed; a.Cin = Cin; a.Cout = 2; a.D = D; a.H = H; a.W = W; a.Do = D; a.Ho = H; a.Wo = W;
    a.hyp = hyp_dhw; a.hyp_base = hyp_dhw ? nullptr : base_hw; a.hyp_step = hyp_dhw ? nullptr : step; a.alpha = alpha; a.dsp = dsp_2hw;
    return D == 4 ? launch_cout2_fused<1>(a, (hipStream_t)stream) : launch_cout2_fused<2>(a, (hipStream_t)stream);
}

template <int CIN_B, int TZ, int TY>
static int launch_cout2(const ConvArgs& a, hipStream_t st) {
    const bool v4 = a.W % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0;
    // Measured on config 2 (ms per call at 32 x 592 x 800): dword loader + 36-float pitch 0.21; 16-byte loader with a
    // 40-float pitch (cannot be made conflict-free, 13-float windows) 0.27; 3- / 4-stage rings 0.32 / 0.34; PZ = 2 0.235.
    return v4 ? launch_cout2_v<CIN_B, TZ, TY, true, 2, 1>(a, st) : launch_cout2_v<CIN_B, TZ, TY, false, 2, 1>(a, st);
}

// ------------------------------------------------------------------------- dispatch
template <int STRIDE, int KD, int CIN_B, int COUT_B, int TZ, int TY, int TXT, int PX>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
    if (a.Cin % CIN_B || a.Cout % COUT_B) return DMVS_EUNSUPPORTED;
    dim3 grid(ceil_div(a.Wo, TXT * PX), ceil_div(a.Ho, TY), ceil_div(a.Do, TZ) * (a.Cout / COUT_B));
    conv_direct_kernel<STRIDE, KD, CIN_B, COUT_B, TZ, TY, TXT, PX><<<grid, TZ * TY * TXT, 0, st>>>(a);
    DMVS_LAUNCH_CHECK();
}

template <int KD, int CIN_B, int COUT_B, int TZ, int TY, int TX>
static int launch_deconv(const ConvArgs& a, hipStream_t st) {
    if (a.Cin % CIN_B || a.Cout % COUT_B) return DMVS_EUNSUPPORTED;
    dim3 grid(ceil_div(a.W, TX), ceil_div(a.H, TY), ceil_div(a.D, TZ) * (a.Cout / COUT_B));
    deconv_direct_kernel<KD, CIN_B, COUT_B, TZ, TY, TX><<<grid, TZ * TY * TX, 0, st>>>(a);
    DMVS_LAUNCH_CHECK();
}

template <int STRIDE, int KD, int TZ, int TY, int TXT, int PX, int CIN_B>
static int conv_by_cout(const ConvArgs& a, hipStream_t st) {
    if (a.Cout % 16 == 0) return launch_conv<STRIDE, KD, CIN_B, 16, TZ, TY, TXT, PX>(a, st);
    if (a.Cout % 8 == 0) return launch_conv<STRIDE, KD, CIN_B, 8, TZ, TY, TXT, PX>(a, st);
    if (a.Cout == 2) return launch_conv<STRIDE, KD, CIN_B, 2, TZ, TY, TXT, PX>(a, st);
    return DMVS_EUNSUPPORTED;
}

extern "C" int dmvs_conv3d_direct(const float* in, float* out, const float* w_packed, const float* scale,
                                  const float* shift, const float* skip, int Cin, int Cout, int D, int H, int W,
                                  int mode, int kdepth, int flags, dmvs_stream_t stream) {
    if (!in || !out || !w_packed || Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1) return DMVS_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return DMVS_EINVAL;
    if (kdepth != 1 && kdepth != 3) return DMVS_EINVAL;
    if (flags & ~DMVS_RELU) return DMVS_EUNSUPPORTED;   // no upsampled residual / quad-planar output here; retired bit 4
    ConvArgs a;
    a.in = in; a.out = out; a.w = w_packed; a.scale = scale; a.shift = shift; a.skip = skip;
    a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W; a.relu = (flags & DMVS_RELU) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const bool k3 = kdepth == 3;
    if (mode == DMVS_CONV_S1) {
        a.Do = D; a.Ho = H; a.Wo = W;
        if (Cout == 2 && Cin % 2 == 0 && k3 && !scale && !skip && !(flags & DMVS_RELU) &&
            (long)2 * D * H * W < (1L << 28))  // the prob head
            return launch_cout2<1, 4, 16>(a, st);
        if (Cin == 2) return k3 ? conv_by_cout<1, 3, 4, 8, 8, 4, 2>(a, st) : DMVS_EUNSUPPORTED;
        return k3 ? conv_by_cout<1, 3, 4, 8, 8, 4, 4>(a, st) : conv_by_cout<1, 1, 1, 16, 16, 2, 8>(a, st);
    }
    if (mode == DMVS_CONV_S2) {
        a.Do = k3 ? (D + 1) / 2 : D; a.Ho = (H + 1) / 2; a.Wo = (W + 1) / 2;
        return k3 ? conv_by_cout<2, 3, 2, 8, 16, 2, 2>(a, st) : conv_by_cout<2, 1, 1, 16, 16, 2, 4>(a, st);
    }
    if (mode == DMVS_DECONV_S2) {
        a.Do = k3 ? 2 * D : D; a.Ho = 2 * H; a.Wo = 2 * W;
        if (Cout % 8) return DMVS_EUNSUPPORTED;
        return k3 ? launch_deconv<3, 8, 8, 2, 8, 16>(a, st) : launch_deconv<1, 8, 8, 1, 16, 16>(a, st);
    }
    return DMVS_EINVAL;
}
