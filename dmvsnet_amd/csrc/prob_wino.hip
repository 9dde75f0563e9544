// K2w: the `prob` heads (Conv3d 8 -> 2, k3 s1 p1, no BatchNorm / ReLU / bias) in Winograd F(2x2,3x3) form on the VECTOR
// ALUs, marching along depth.
//
// Replaces the same reference code as conv_cout2_kernel (/root/reference/networks/module.py:379,397: `self.prob`, applied
// at 397 / 435) -- the layer has 2 output channels, 2 of the 16 rows of an MFMA, so it is a VALU kernel, and the r04
// counters show the direct form bound by VALU issue (72 % of the cycles, 78 % of the instructions useful packed FMAs:
// profiles/r04_e_*prob*).  What is left is the number of multiplies:
//   * in-plane F(2x2,3x3): Y = A^T [ sum_{ci,kz} (G g G^T) .* (B^T d B) ] A -- 16 products per 2x2 outputs, input
//     plane and (ci, kz) instead of 36; the two output channels are the two halves of a packed fp32 FMA (v_pk_fma_f32:
//     the transformed filter pair is one operand, the transformed input value is broadcast);
//   * a thread owns one 2x2 output patch and MARCHES along depth: the transformed patch V of input plane p feeds the three
//     output planes p+1, p, p-1 (kz = 0, 1, 2), so its 32 transform additions are paid once per 48 packed FMAs, and the
//     accumulators stay in the Winograd domain (3 planes x 16 positions x 2 channels = 96 registers); when plane p is done,
//     output plane p-1 is complete: output transform (A^T . A), 8-byte stores, the register set becomes plane p+2's;
//   * per output voxel 8 x (48 + 32) / 4 = 160 VALU instructions instead of 278 (216 packed FMAs + address / select overhead).
// Inputs, products and sums are fp32; the filter transform is formed on the host in double and rounded once (like K3w).
// Against the direct form the result moves at re-association level (tests: 2e-5 against ATen, the bound of every conv layer).
//
// A 256-thread workgroup owns a 64 x 16 output tile (32 x 8 patches; a half-wave = 32 patches of one row: its ds_read_b64
// of a patch row is one contiguous 256-byte run, conflict-free at any pitch) and a depth segment [z0, z1); one pipeline
// stage = CB channels of ONE input plane (18 rows x 72 floats each, 16-byte LDS-direct pieces from x0 - 4), a ring of NS
// stages with counted vmcnt.  Depth segments exist so that small grids still fill the chip (they cost two halo planes of
// loads + transforms each; the FMAs of a halo plane are issued only for the output planes inside the segment).
#include "common.h"
#include "tile_loader.h"

#include <algorithm>
#include <type_traits>

#ifndef DMVS_PWKO
#define DMVS_PWKO 0   // dev knock-outs: 1 no tile loads, 2 no FMAs, 4 no stores
#endif

namespace {

struct ProbArgs {
    const float* in;   // [8][D][H][W]
    const float* w;    // packed by dmvs_pack_prob_weights_wino: [8 ci][16 pos][4 (kz 0,1,2, pad)][2 co]
    float* out;        // [2][D][H][W]
    int D, H, W;
    int nx, ny, nseg, seg;   // tile grid (x, y), depth segments, planes per segment
};

constexpr int CIN = 8;
constexpr int TW = 64, TH = 16;             // output tile
constexpr int IY = TH + 2, IXP = 72, LPR = IXP / 4;   // staged rows / pitch (floats) / 16-byte pieces per row
constexpr int PS = IY * IXP;                // channel stride inside a stage
constexpr int WFLOATS = CIN * 16 * 4 * 2;   // transformed filters

template <int CB, int NS>
__global__ __launch_bounds__(256, 2) void prob_wino_kernel(ProbArgs a) {
    constexpr int NCH = CIN / CB;                       // stages per input plane
    constexpr int STG = (CB * PS + 63) & ~63;           // floats per stage
    constexpr int G = (IY + (64 / LPR) - 1) / (64 / LPR);
    constexpr int LPW = CB * ((G + 3) / 4);             // LDS-direct loads per wave and stage (load_tile4: uniform)
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [NS][STG] stages, then the filters
    float* const wl = smem + NS * STG;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = lane & 31, ty = 2 * wave + (lane >> 5);         // the thread's patch inside the tile
    // XCD-aware order (common.h): depth segment fastest, then x, then y
    const int n = a.nx * a.ny * a.nseg, per = (n + 7) >> 3;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (t >= n) return;
    const int sg = t % a.nseg, bx = (t / a.nseg) % a.nx, by = t / (a.nseg * a.nx);
    const int x0 = bx * TW, y0 = by * TH, z0 = sg * a.seg, z1 = min(z0 + a.seg, a.D);

    for (int i = tid; i < WFLOATS; i += 256) wl[i] = a.w[i];       // (visible after the first barrier of the pipeline)

    const int vol = a.D * a.H * a.W;
    // input planes z0 - 1 .. z1, without the zero-padding planes outside the volume (they add nothing)
    const int pfirst = max(z0 - 1, 0), plast = min(z1, a.D - 1);
    const int nstages = (plast - pfirst + 1) * NCH;
    auto stage = [&](int k) {
        if (DMVS_PWKO & 1) return;
        const int p = pfirst + k / NCH, c = k % NCH;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.in + (size_t)(c * CB) * vol), (short)0, CB * vol * 4, 0x00020000);
        load_tile4<CB, 1, IY, LPR, PS>(a.D, a.H, a.W, rs, smem + (k % NS) * STG, p, y0 - 1, x0 - 4, wave, lane);
    };

    float2_t acc[3][16];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[s][i] = (float2_t){0.f, 0.f};

    // patch rows 2 ty .. 2 ty + 3 of the staged plane, columns 2 tx + 2 .. 2 tx + 7 (three aligned pairs; the patch is
    // columns 2 tx + 3 .. 2 tx + 6: the tile starts 4 floats left of x0, the halo column is index 3)
    const int pbase = (2 * ty) * IXP + 2 * tx + 2;

    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, (short)0, 2 * vol * 4, 0x00020000);
    const int ox = x0 + 2 * tx, oy = y0 + 2 * ty;
    const bool oin = ox < a.W && oy < a.H;      // W, H even: a patch is inside or outside as a whole

#pragma unroll
    for (int k = 0; k < NS - 1; ++k)
        if (k < nstages) stage(k);

    // ONE loop over the pipeline stages k = (input plane p, channel chunk c); the roles of the three accumulator sets are
    // STATIC -- acc[2]: output plane p + 1 (kz = 0 filters), acc[1]: plane p (kz = 1), acc[0]: plane p - 1 (kz = 2), which is
    // complete after plane p's last chunk -- and the sets are moved down one place per plane (32 register-pair moves per
    // ~700 VALU instructions).  r04 tried rotating the ROLES instead (three compile-time bodies, and compile-time bodies
    // for the edge planes that feed fewer than three output planes): the register allocator then keeps two copies of the
    // 48 accumulator pairs and moves them at every join (53-81 moves per 48 FMAs, 250 spills with run-time flags).
    // Edge planes simply run the full body: what they add to an output plane outside [z0, z1) is never stored.
    auto finish = [&](int z) {   // output plane z from acc[0]: Y = A^T M A, A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]
        float2_t s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s0[j] = acc[0][0 * 4 + j] + acc[0][1 * 4 + j] + acc[0][2 * 4 + j];
            s1[j] = acc[0][1 * 4 + j] - acc[0][2 * 4 + j] - acc[0][3 * 4 + j];
        }
        const float2_t y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
        const float2_t y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
        if (!(DMVS_PWKO & 4)) {
            typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
            const unsigned pos = oin ? (unsigned)((z * a.H + oy) * a.W + ox) * 4u : 0x80000000u;
            const unsigned rowb = (unsigned)a.W * 4u, chb = (unsigned)vol * 4u;
            auto st2 = [&](float lo, float hi, unsigned off) {
                v2u_t v; v.x = __builtin_bit_cast(unsigned, lo); v.y = __builtin_bit_cast(unsigned, hi);
                __builtin_amdgcn_raw_buffer_store_b64(v, rs_out, off, 0, 0);
            };
            const bool ok = !(pos & 0x80000000u);
            st2(y00.x, y01.x, pos);
            st2(y10.x, y11.x, ok ? pos + rowb : pos);
            st2(y00.y, y01.y, ok ? pos + chb : pos);
            st2(y10.y, y11.y, ok ? pos + chb + rowb : pos);
        }
    };

    int p = pfirst, c = 0;
#pragma unroll 1
    for (int k = 0; k < nstages; ++k) {
        const int younger = min(NS - 2, nstages - 1 - k);
        if (NS >= 4 && younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPW) : "memory");
        else if (NS >= 3 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (k + NS - 1 < nstages) stage(k + NS - 1);
        const float* tile = smem + (k % NS) * STG + pbase;
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
            // rows of the patch -> row transform d B per row: (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
            float tr[4][4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float* row = tile + cc * PS + rr * IXP;
                const float2_t q0 = *reinterpret_cast<const float2_t*>(row);
                const float2_t q1 = *reinterpret_cast<const float2_t*>(row + 2);
                const float2_t q2 = *reinterpret_cast<const float2_t*>(row + 4);
                const float d0 = q0.y, d1 = q1.x, d2 = q1.y, d3 = q2.x;
                tr[rr][0] = d0 - d2; tr[rr][1] = d1 + d2; tr[rr][2] = d2 - d1; tr[rr][3] = d1 - d3;
            }
            // column transform B^T (.): V[i][j] over the rows i
            float V[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                V[0 * 4 + j] = tr[0][j] - tr[2][j];
                V[1 * 4 + j] = tr[1][j] + tr[2][j];
                V[2 * 4 + j] = tr[2][j] - tr[1][j];
                V[3 * 4 + j] = tr[1][j] - tr[3][j];
            }
            const float* wc = wl + ((c * CB + cc) * 16) * 8;
            if (DMVS_PWKO & 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[0][i].x += V[i];
                continue;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float4_t w01 = *reinterpret_cast<const float4_t*>(wc + i * 8);        // kz 0, kz 1 pairs
                const float2_t w2 = *reinterpret_cast<const float2_t*>(wc + i * 8 + 4);     // kz 2 pair
                const float2_t v = (float2_t){V[i], V[i]};
                acc[2][i] = __builtin_elementwise_fma((float2_t){w01.x, w01.y}, v, acc[2][i]);
                acc[1][i] = __builtin_elementwise_fma((float2_t){w01.z, w01.w}, v, acc[1][i]);
                acc[0][i] = __builtin_elementwise_fma(w2, v, acc[0][i]);
                // the filter pairs of 4 positions in flight at a time: unfenced, the scheduler hoists all 96 filter registers
                // of a channel above the FMAs
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (++c == NCH) {   // the plane is done: output plane p - 1 is complete
            if (p - 1 >= z0) finish(p - 1);
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[0][i] = acc[1][i]; acc[1][i] = acc[2][i]; acc[2][i] = (float2_t){0.f, 0.f}; }
            c = 0; ++p;
        }
    }
    // the last input plane of the VOLUME: the plane behind it is zero padding, output plane D - 1 is complete too
    if (plast == a.D - 1 && z1 == a.D && plast >= z0) finish(plast);
}

template <int CB, int NS>
int launch_prob(ProbArgs a, hipStream_t st) {
    constexpr size_t lds = ((size_t)NS * ((CB * PS + 63) & ~63) + WFLOATS) * sizeof(float);
    static_assert(lds <= 160 * 1024, "stages + filters must fit the LDS");
    auto kern = prob_wino_kernel<CB, NS>;
    if (lds > 48 * 1024)
        if (int e = dmvs_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return e;
    kern<<<dim3(xcd_grid(a.nx * a.ny * a.nseg)), 256, lds, st>>>(a);
    DMVS_LAUNCH_CHECK();
}

long g_prob_min_blocks = 768;   // depth segments are added until the grid has at least this many workgroups

void plan(int D, int H, int W, ProbArgs& a) {
    a.nx = ceil_div(W, TW); a.ny = ceil_div(H, TH);
    const long xy = (long)a.nx * a.ny;
    int nseg = (int)std::min<long>((g_prob_min_blocks + xy - 1) / xy, std::max(1, D / 4));   // >= 4 planes per segment
    nseg = std::max(1, std::min(nseg, D));
    a.seg = ceil_div(D, nseg);
    a.nseg = ceil_div(D, a.seg);
}

}  // namespace

extern "C" long dmvs_prob_wino_weight_floats(void) { return WFLOATS; }

// w [2][8][3][3][3] (PyTorch Conv3d weight of `prob`, module.py:379) -> [ci][pos = 4 i + j][kz (3 + pad)][co]: U = G g G^T
extern "C" int dmvs_pack_prob_weights_wino(const float* w, float* out) {
    if (!w || !out) return DMVS_EINVAL;
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int i = 0; i < WFLOATS; ++i) out[i] = 0.f;
    for (int co = 0; co < 2; ++co)
        for (int ci = 0; ci < CIN; ++ci)
            for (int kz = 0; kz < 3; ++kz) {
                const float* g = w + ((size_t)(co * CIN + ci) * 3 + kz) * 9;
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) {
                        double u = 0.0;
                        for (int r = 0; r < 3; ++r)
                            for (int c = 0; c < 3; ++c) u += Gm[i][r] * (double)g[r * 3 + c] * Gm[j][c];
                        out[((ci * 16 + i * 4 + j) * 4 + kz) * 2 + co] = (float)u;
                    }
            }
    return 0;
}

extern "C" int dmvs_prob_wino_plan(int D, int H, int W) {
    if (D < 1 || H < 2 || W < 4 || (H & 1) || (W & 3)) return DMVS_EUNSUPPORTED;
    ProbArgs a{};
    plan(D, H, W, a);
    return a.nx * a.ny * a.nseg;
}

extern "C" int dmvs_prob_wino(const float* in, float* out, const float* w_packed, int D, int H, int W, dmvs_stream_t stream) {
    if (!in || !out || !w_packed || D < 1 || H < 1 || W < 1) return DMVS_EINVAL;
    // 16-byte pieces of whole rows, patches inside or outside as a whole, byte offsets below 2 GB
    if ((H & 1) || (W & 3) || (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 7)) return DMVS_EUNSUPPORTED;
    if ((long)CIN * D * H * W >= (1L << 29)) return DMVS_EUNSUPPORTED;
    ProbArgs a{};
    a.in = in; a.out = out; a.w = w_packed; a.D = D; a.H = H; a.W = W;
    plan(D, H, W, a);
    return launch_prob<2, 3>(a, (hipStream_t)stream);
}
