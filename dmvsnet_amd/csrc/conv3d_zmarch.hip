// K3z: conv2 of the regularisation U-Nets (16 -> 16, stride 1, 3x3x3; /root/reference/networks/module.py:364 and 406, operator
// module.py:120-157 Conv3d + BatchNorm(eval) + ReLU) in Winograd F(2x2, 3x3) form with REGISTER-STATIONARY filters, marching
// along z -- K3r's pipeline (csrc/conv3d_coarse.hip) carried to a layer that fills the chip.
//
// Why not K3r as it is.  conv2 is 10 % of the GPU time, thousands of work units per launch -- the tail that made K3r neutral end to
// end is noise here -- but with Cin = Cout = 16 a K3r unit would be ONE stage of 24 MFMAs per wave between two barriers, with the
// input transform (2 VALU per MFMA) and the output exchange paid per stage, and every input plane staged three times (once per
// depth tap).  On gfx950 fp32 MFMA and VALU share an issue port (scripts/dev/mfma_valu_overlap.hip: their times ADD), so what
// decides the kernel is instructions per MFMA.  K3z therefore MARCHES ALONG Z:
//   * a 256-thread workgroup = 4 waves = the 4 Winograd transform rows i.  Wave i keeps U[4i + p][kz][ci][co] (G g G^T, formed in
//     double on the host, rounded once) for its 4 positions p, all 3 depth taps and all 16 x 16 channel pairs in REGISTERS as
//     MFMA B operands: 12 float4 = 48 VGPRs.  No filter bytes through LDS, no ds_read for B;
//   * a work COLUMN = an 8 x 8 group of outputs (4 x 4 Winograd tiles = the 16 rows of v_mfma_f32_16x16x4_f32) x a z segment of
//     `zs` planes x all 16 output channels.  One pipeline STAGE = one input plane of the column (16 channels x 10 x 20 floats,
//     14 KB, 16-byte LDS-direct loads, lane-linear as in K3r): the wave transforms ITS row of every 4x4 patch once (8 VALU + 6
//     ds_read_b64 per 4-channel group) and feeds up to 12 MFMAs with it -- the plane is depth tap 0 of output plane pz + 1, tap 1
//     of pz and tap 2 of pz - 1 -- into three accumulator sets (3 x 4 float4) that shift through the MFMAs' C operand: 0.67 VALU per MFMA on the input side
//     instead of K3w's 1.33 and K3r's 2, every input plane staged once per segment instead of three times;
//   * when plane pz is done, output plane pz - 1 is complete: the wave reduces its 4 positions to the two output columns
//     (M[i][:] A), the 4 partial results meet in LDS (8 KB, two alternating buffers) and after the NEXT stage's barrier every wave
//     finishes a quarter of the plane (row sum over i = A^T, BatchNorm, ReLU, one 16-byte store per lane) -- the finish runs
//     behind the next plane's first MFMAs, a stage costs ONE barrier;
//   * the loader costs no VALU per stage: a lane's byte offsets are formed once per COLUMN (range check of (y, x) against the
//     image, invalid pieces at offset 2^31 = zero fill = the convolution's padding), the plane is selected by the SCALAR offset of
//     the buffer load, planes outside the volume by a descriptor of zero records;
//   * persistent 256-thread workgroups, several per CU: workgroup b works on XCD b % 8, takes whole columns of that XCD's eighth
//     of the column list round-robin (neighbouring columns run at the same time: halo shared in L2) and an equal share of the last
//     partial round's planes.  Independent workgroups on a SIMD fill each other's barrier and LDS-latency gaps -- what the
//     whole-CU K3r workgroup cannot do.
// Arithmetic: fp32 throughout; per output the products are accumulated in the fixed order (depth tap 0, 1, 2) x (channel group
// 0..3) whatever the segment length, so the result does not depend on `zs`, the grid or the slab a volume was cut into.  Against
// K3w / K3 the result moves at re-association level (tests: 2e-5 of the output scale against ATen).
// Needs W % 4 == 0 and 16-byte aligned tensors (as K3w); otherwise DMVS_EUNSUPPORTED and the caller runs K3w / K3.
#include "common.h"
#include "tile_loader.h"

#include <algorithm>
#include <type_traits>

#ifndef DMVS_K3Z_RING
#define DMVS_K3Z_RING 2   /* LDS plane slots (the barrier-in-the-middle pipeline needs exactly two) */
#endif
#ifndef DMVS_ZKO
#define DMVS_ZKO 0   /* development knock-outs (scripts/dev/variant_build.sh): 1 no tile loads, 2 no output stores, 4 no MFMAs, 8 no barrier, 16 no finish (exchange reads + output transform rows) */
#endif
#include "dev_guard.h"   // after the defaults of this file's development switches

// persistent workgroups of a K3z launch (dmvs_tune("k3z_grid"), a multiple of 8); 0 = as many as are resident (3 or 2 per CU)
long g_k3z_grid = 0;
// cap of a z segment's length (dmvs_tune("k3z_zs")); 0 = none: a segment is a whole column, or what a workgroup's plane range cuts out of one
long g_k3z_zs = 0;
// ring of 3 only: 1 = counted vmcnt at the stage wait, 0 = vmcnt(0) (dmvs_tune("k3z_counted_wait"), bit-identical: the gate)
long g_k3z_counted_wait = 1;

namespace {

typedef float acc4_t __attribute__((ext_vector_type(4)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
#if DMVS_ZKO & 4
__device__ __forceinline__ acc4_t z_mfma(float a, float b, acc4_t c) { c.x += a; c.y += b; return c; }   // keeps the operands alive
#else
__device__ __forceinline__ acc4_t z_mfma(float a, float b, acc4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
#endif

struct ZArgs {
    const float* in;
    float* out;
    const float* w;
    const float* scale;
    const float* shift;
    int D, H, W, relu;
    int ngx, ngy, zs;         // 8 x 8-output groups along x / y; cap of a z segment's length (>= D: none)
    int counted;
};

template <int RING>
struct ZGeom {
    static constexpr int CIN = 16;
    static constexpr int IXP = 20, IY = 10, PLANE = IXP * IY;    // rows ox0 - 4 .. ox0 + 15, oy0 - 1 .. oy0 + 8
    static constexpr int PS = PLANE + (32 - PLANE % 64 + 64) % 64;   // channel stride = 32 (mod 64) banks (K3r's patch-read layout)
    static constexpr int NI = (CIN * PS / 4 + 63) / 64;          // 16-byte load instructions per stage (14)
    static constexpr int NS = (NI + 3) / 4;                      // ... per wave (4): slots past NI go to the trash area
    static constexpr int STAGE_F = NI * 256;
    static constexpr int TRASH_F = (4 * NS - NI) * 256;
    static constexpr int EX1_F = 4 * 4 * 64 * 2;                 // exchange: [wave][r][lane][2]
    static constexpr size_t LDS = (size_t)(RING * STAGE_F + TRASH_F + 2 * EX1_F) * sizeof(float);
    static constexpr int WPS = 2;                               // waves per SIMD the register budget is set for (2 x 256 VGPRs)
    static_assert(PS % 4 == 0 && PS % 64 == 32 && CIN * PS == NI * 256, "stage layout");
};

template <int RING>
__global__ __launch_bounds__(256, ZGeom<RING>::WPS) void zmarch_kernel(ZArgs a) {
    typedef ZGeom<RING> G;
    constexpr int IXP = G::IXP, PS = G::PS, NS = G::NS;
    constexpr unsigned kInvalid = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [RING][STAGE_F] planes, [TRASH_F], [2][EX1_F] exchange
    float* const ex = smem + RING * G::STAGE_F + G::TRASH_F;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = the Winograd transform row i
    const int ln = lane & 15, lk = lane >> 4, tx = ln & 3, ty = ln >> 2;

    // ---- work assignment.  XCD b % 8 owns the (b % 8)-th contiguous eighth of the column list (8 x 8-output groups, x fastest, then
    // y); its nslots workgroups take whole columns round-robin -- column c of round r goes to slot c % nslots, so the workgroups
    // of an XCD work on NEIGHBOURING columns at the same time and share their halo in that XCD's L2 (giving every workgroup one
    // long contiguous range instead measured 6-12 % slower: the halo of a column was evicted before its neighbour came round) --
    // and only the LAST, partial round is cut finer: its columns' planes form one list that the nslots workgroups split into equal
    // contiguous ranges (a static round-robin over whole columns left 4-18 % of the slots idle in the last round: 3700 units on
    // 512 slots = 8 rounds for 7.2).  A workgroup walks its planes as z SEGMENTS: a whole column is one segment of D planes (the
    // fewest halo stages possible), a range that starts or ends inside a column starts / ends a segment there.  a.zs caps the
    // segment length (tests: the result does not depend on how the planes are cut).
    const int xcd = blockIdx.x & 7, slot = (int)(blockIdx.x >> 3), nslots = (int)(gridDim.x >> 3);
    const int ncols = a.ngx * a.ngy, per = (ncols + 7) >> 3;
    const int mine = min(per, ncols - xcd * per);
    if (mine <= 0) return;
    const int c0 = xcd * per;
    const int rounds = mine / nslots, tail_c = rounds * nslots;         // whole rounds; first column of the partial round
    // (32-bit: fewer than nslots columns x D planes x nslots < 2^31 for any grid the launcher makes; a 64-bit division is expanded on
    // the VALUs and everything derived from it -- down to the buffer descriptors -- would be treated as divergent)
    const int tail_planes = (mine - tail_c) * a.D;
    const int tp0 = __builtin_amdgcn_readfirstlane(tail_planes * slot / nslots);
    const int tp1 = __builtin_amdgcn_readfirstlane(tail_planes * (slot + 1) / nslots);
    const int vfull = rounds * a.D, p1 = vfull + (tp1 - tp0);            // the workgroup's virtual plane list [0, p1)
    if (p1 <= 0) return;
    // segment that starts at virtual plane v: origin of its column, first plane, length
    auto coords = [&](int v, int& ox0, int& oy0, int& z0, int& zse) {
        int c, lim;
        if (v < vfull) {
            const int r = v / a.D;
            z0 = v - r * a.D;
            c = r * nslots + slot;
            lim = a.D - z0;
        } else {
            const int q = tp0 + (v - vfull), cc = q / a.D;
            z0 = q - cc * a.D;
            c = tail_c + cc;
            lim = min(a.D - z0, p1 - v);
        }
        const int g = c0 + c, gy = g / a.ngx;
        oy0 = 8 * gy;
        ox0 = 8 * (g - gy * a.ngx);
        zse = min(lim, a.zs);
    };

    // ---- the wave's filters: [wave][k-group][kz][lane][4 positions], loaded once
    float4_t w[12];
    {
        const float4_t* wp = reinterpret_cast<const float4_t*>(a.w) + (size_t)wave * 12 * 64 + lane;
#pragma unroll
        for (int n = 0; n < 12; ++n) w[n] = wp[n * 64];
    }
    // ---- the finishing role of this wave: output row parity and x half; the lane's channel is ln
    const int frr = wave & 1, fxh = wave >> 1;
    const float bsc = a.scale ? a.scale[ln] : 1.f, bsh = a.scale ? a.shift[ln] : 0.f;
    const float lo = a.relu ? 0.f : -INFINITY;

    // ---- loader: the stage is lane-linear in LDS; piece (wave + 4 sl) * 64 + lane of every stage is the same (channel, row, x)
    const int plane = a.H * a.W, vol = a.D * plane;
    int roff[NS];
    unsigned yx[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        const int qi = wave + 4 * sl, f = (qi * 64 + lane) * 4;
        const int c = f / PS, rem = f - c * PS;
        const bool okp = qi < G::NI && rem < G::PLANE;
        const int row = rem / IXP, x = rem - row * IXP;
        roff[sl] = c * vol + row * a.W + x;
        yx[sl] = okp ? (unsigned)(row | (x << 8)) : 0x3f3fu;   // a pad piece fails every range test below
    }
    // the issue stream runs RING - 1 stages ahead of the compute stream: its own (column, plane) counters and the lane's byte
    // offsets of the column it is in
    int pq = 0, tq = 0, qz0 = 0, qnt = 0, qzse = 0;
    unsigned voff[NS];
    bool q_valid = false;
    int q_soff = 0, ring_q = 0, q_dsto = 0;
    auto issue_begin = [&]() {
        const bool on = pq < p1;
        if (on && tq == 0) {
            int ox0, oy0;
            coords(pq, ox0, oy0, qz0, qzse);
            qnt = qzse + 2;
            const int yb = oy0 - 1, xb = ox0 - 4;
            // valid tile rows / columns of this column, [lo, hi] (uniform); the lane's (y, x) bytes are range-checked together:
            // with the guard bit 7 set, a byte-wise subtraction keeps the guard iff it did not borrow (K3r's test)
            const unsigned yl = max(0, -yb), xl = max(0, -xb);
            const unsigned yh = min(G::IY, a.H - yb) - 1, xh = min(IXP, a.W - xb) - 1;
            const unsigned LO = yl | (xl << 8), HG = (yh | (xh << 8)) | 0x8080u;
            const int base = yb * a.W + xb;
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const unsigned ge = (yx[sl] | 0x8080u) - LO, le = HG - yx[sl];
                const bool ok = (ge & le & 0x8080u) == 0x8080u;
                voff[sl] = ok ? (unsigned)(roff[sl] + base) * 4u : kInvalid;
            }
        }
        const int pz = qz0 - 1 + tq;
        q_valid = on && pz >= 0 && pz < a.D;
        q_soff = q_valid ? pz * plane * 4 : 0;
        q_dsto = ring_q * G::STAGE_F;
        ring_q = ring_q + 1 == RING ? 0 : ring_q + 1;
        if (on && ++tq == qnt) { tq = 0; pq += qzse; }
    };
    // every stage issues exactly NS loads per wave (past the last stage / outside the volume: a descriptor of zero records --
    // no traffic, zeros into a slot nobody reads or into the padding plane): what the counted vmcnt of the ring of 3 relies on
    auto issue_slot = [&](int sl) {
        const int qi = wave + 4 * sl;   // scalar; the LDS address of an LDS-direct load is wave-uniform (M0)
        const int dsto = __builtin_amdgcn_readfirstlane(qi < G::NI ? q_dsto + qi * 256 : RING * G::STAGE_F + (qi - G::NI) * 256);
        // (locals: hipcc 7.2 silently drops the kernel's HOST stub when a by-reference capture is passed to this builtin directly)
        const unsigned vo = voff[sl];
        const int so = __builtin_amdgcn_readfirstlane(q_soff);
        // ONE load instruction per slot whatever the plane: a plane outside the volume (or past the last stage) gets a descriptor of
        // zero records -- every lane out of range, no traffic, zeros into LDS -- and q_soff = 0
        const int nrec = __builtin_amdgcn_readfirstlane(q_valid ? G::CIN * vol * 4 : 0);   // (a descriptor in VGPRs costs a waterfall loop per load)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, nrec, 0x00020000);
        if (!(DMVS_ZKO & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + dsto), 16, vo, so, 0, 0);
    };

    // ---- patch reads (K3r's layout): the lane's tile (tx, ty), channel lk of a k-group; row i of B^T d = d[ra] + sg * d[rb]
    const int ti = wave;
    const int ra = ti == 0 ? 0 : (ti == 2 ? 2 : 1), rb = ti == 0 ? 2 : (ti == 1 ? 2 : (ti == 2 ? 1 : 3));
    const float sg = ti == 1 ? 1.f : -1.f;
    const int lbase = lk * PS + 2 * ty * IXP + 2 + 2 * tx;
    const int baseA = lbase + ra * IXP, baseB = lbase + rb * IXP;

    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.out, (short)0, 16 * vol * 4, 0x00020000);
    // the finish of an output plane: row sum over the transform rows i (A^T), BatchNorm, ReLU, 16-byte store
    float2_t P[4][2];
    auto finish_read = [&](int eb) {
        const float2_t* exr = reinterpret_cast<const float2_t*>(ex + eb * G::EX1_F) + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int rs = 0; rs < 2; ++rs) P[i][rs] = exr[(i * 4 + 2 * fxh + rs) * 64];
    };
    auto finish_store = [&](int ox0, int oy0, int oz) {
        float y[4];
#pragma unroll
        for (int rs = 0; rs < 2; ++rs) {
            if (frr == 0) {
                y[2 * rs] = (P[0][rs].x + P[1][rs].x) + P[2][rs].x;
                y[2 * rs + 1] = (P[0][rs].y + P[1][rs].y) + P[2][rs].y;
            } else {
                y[2 * rs] = (P[1][rs].x - P[2][rs].x) - P[3][rs].x;
                y[2 * rs + 1] = (P[1][rs].y - P[2][rs].y) - P[3][rs].y;
            }
        }
        const int x = ox0 + 4 * fxh, yy = oy0 + 2 * lk + frr;
        const unsigned pos = (unsigned)(ln * vol + oz * plane + yy * a.W + x) * 4u;
        v4u_t qv;
        qv.x = __builtin_bit_cast(unsigned, fmaxf(y[0] * bsc + bsh, lo));
        qv.y = __builtin_bit_cast(unsigned, fmaxf(y[1] * bsc + bsh, lo));
        qv.z = __builtin_bit_cast(unsigned, fmaxf(y[2] * bsc + bsh, lo));
        qv.w = __builtin_bit_cast(unsigned, fmaxf(y[3] * bsc + bsh, lo));
        __builtin_amdgcn_raw_buffer_store_b128(qv, rs_out, (yy < a.H && x < a.W && !((DMVS_ZKO & 2) && qv.x != 0x12345678u)) ? pos : kInvalid, 0, 0);
    };

    // ---- pipeline.  Knock-outs of the first build (barrier at the top of a stage: patch reads -> transform -> 48 MFMAs -> exchange;
    // profiles/r06_f_k3z_knockouts.txt) showed its phases ADD (MFMA 0.085 + everything else 0.044 + tile loads 0.016 = the 0.145 ms
    // of conv2 at stage 2): the waves of a SIMD contend for the one issue port fp32 MFMA and VALU share, which phase-LOCKS them --
    // all in their MFMA block, then all in their LDS round trips with the pipe idle.  So a stage is now three blocks of 16 MFMAs
    // with the other work BETWEEN them, and the stage barrier sits behind the first block:
    //     A  tap-2 block (completes output plane z0 + t - 2) on this plane's transformed patches v
    //     B  its output transform columns -> exchange area
    //     C  wait + barrier: the NEXT plane has landed for every wave, every wave's partial sums are written, every wave has read
    //        this plane (during the previous stage) -> its ring slot is free
    //     D  issue the loads of the plane after next into that slot (ring of 2); patch reads of the next plane; exchange reads
    //     E  tap-1 block            (the LDS latency of D flies under it)
    //     F  transform of the next plane -> vn; finish of the completed plane: row sum, BatchNorm, ReLU, store
    //     G  tap-0 block
    // -- no wave ever leaves the matrix pipe alone for longer than ~50 VALU instructions.
    float v[4][4];
    acc4_t acc[2][4];
    int ring_c = 0, k = 0;
    int ox0 = 0, oy0 = 0, z0 = 0, zse = 0, t = 0;
    float2_t rd[4][6];
    auto patch_read = [&](int slot) {
        const float* pa = smem + slot * G::STAGE_F + baseA;
        const float* pb = smem + slot * G::STAGE_F + baseB;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            const int o = kg * 4 * PS;
            rd[kg][0] = *reinterpret_cast<const float2_t*>(pa + o);
            rd[kg][1] = *reinterpret_cast<const float2_t*>(pa + o + 2);
            rd[kg][2] = *reinterpret_cast<const float2_t*>(pa + o + 4);
            rd[kg][3] = *reinterpret_cast<const float2_t*>(pb + o);
            rd[kg][4] = *reinterpret_cast<const float2_t*>(pb + o + 2);
            rd[kg][5] = *reinterpret_cast<const float2_t*>(pb + o + 4);
        }
    };
    auto patch_xform = [&](float (&vo)[4][4]) {
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            const float t0 = fmaf(sg, rd[kg][3].y, rd[kg][0].y), t1 = fmaf(sg, rd[kg][4].x, rd[kg][1].x),
                        t2 = fmaf(sg, rd[kg][4].y, rd[kg][1].y), t3 = fmaf(sg, rd[kg][5].x, rd[kg][2].x);
            vo[kg][0] = t0 - t2; vo[kg][1] = t1 + t2; vo[kg][2] = t2 - t1; vo[kg][3] = t1 - t3;
        }
    };
    // prologue: the first plane, alone; then the second one flies while the first is read and transformed
    issue_begin();
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) issue_slot(sl);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (also the filter loads)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_begin();
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) issue_slot(sl);
    patch_read(0);
    patch_xform(v);
    ring_c = 1;
    static_assert(RING == 2, "the barrier-in-the-middle pipeline needs exactly two plane slots");
    // one stage = one input plane: depth tap 0 / 1 / 2 of the output planes in accumulator sets 0 / 1 / (transient).  WHICH taps a
    // plane serves (bit kz of MASK) is a compile-time property of the stage's place in the column -- t = 0: tap 0 only, t = 1:
    // taps 0 and 1, 2 <= t < zs: all three, t = zs: taps 1 and 2, t = zs + 1: tap 2 -- so no accumulator is updated conditionally
    // (a conditional update is a PHI the register allocator pays 16-32 v_mov per stage for: the first build, 3.4 VALU per MFMA).
    // The accumulator sets do not rotate through the code: the FIRST MFMA of a block reads the set one plane younger as its C
    // operand and writes its own (D != C costs nothing).
    auto stage = [&](auto mask_t) {
        constexpr int MASK = decltype(mask_t)::value;
        constexpr bool do0 = MASK & 1, do1 = MASK & 2, do2 = MASK & 4;
        const int pz = z0 - 1 + t;
        // a plane outside the volume is zero padding: pz = -1 only ever meets MASK 1 (the output plane that starts there starts from
        // zero), pz = D only MASK 4 (the plane it would complete is complete as it stands in set 1); the others skip the test
        const bool pv = (MASK == 1 || MASK == 4) ? (pz >= 0 && pz < a.D) : true;
        // ---- A: tap 2
        acc4_t done[4];
        if constexpr (do2) {
            if (pv) {
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
                    const float4_t wv = w[kg * 3 + 2];
                    const float wq[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                    for (int p = 0; p < 4; ++p) done[p] = z_mfma(v[kg][p], wq[p], kg == 0 ? acc[1][p] : done[p]);
                }
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) done[p] = acc[1][p];
            }
            // ---- B: this wave's share of the output transform, M[i][0..3] A -> the two output columns of each tile (register r =
            // tile (tx = r, ty = lk) of channel ln), handed to the finishing waves through LDS
            float2_t* const exw = reinterpret_cast<float2_t*>(ex + (k & 1) * G::EX1_F) + (size_t)wave * 4 * 64 + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m0 = done[0][r], m1 = done[1][r], m2 = done[2][r], m3 = done[3][r];
                float2_t sv;
                sv.x = (m0 + m1) + m2;
                sv.y = (m1 - m2) - m3;
                exw[r * 64] = sv;
            }
        }
        // ---- C
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(DMVS_ZKO & 8)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- D
        issue_begin();   // the plane after next, into the slot of the plane every wave has finished reading
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) issue_slot(sl);
        patch_read(ring_c);
        ring_c ^= 1;
        if (do2 && !(DMVS_ZKO & 16)) finish_read(k & 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- E: tap 1
        if constexpr (do1) {
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                const float4_t wv = w[kg * 3 + 1];
                const float wq[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[1][p] = z_mfma(v[kg][p], wq[p], kg == 0 ? acc[0][p] : acc[1][p]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- F
        float vn[4][4];
        patch_xform(vn);
        if (do2 && !(DMVS_ZKO & 16)) finish_store(ox0, oy0, z0 + t - 2);
        __builtin_amdgcn_sched_barrier(0);
        // ---- G: tap 0
        if constexpr (do0) {
            if (pv) {
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
                    const float4_t wv = w[kg * 3 + 0];
                    const float wq[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        acc[0][p] = z_mfma(v[kg][p], wq[p], kg == 0 ? (acc4_t){0.f, 0.f, 0.f, 0.f} : acc[0][p]);
                }
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[0][p] = (acc4_t){0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
            for (int p = 0; p < 4; ++p) v[kg][p] = vn[kg][p];
        ++k;
        ++t;
    };
    for (int p = 0; p < p1; p += zse) {
        coords(p, ox0, oy0, z0, zse);
        t = 0;
        stage(std::integral_constant<int, 1>{});
        if (zse == 1) {
            stage(std::integral_constant<int, 2>{});
        } else {
            stage(std::integral_constant<int, 3>{});
            while (t < zse) stage(std::integral_constant<int, 7>{});
            stage(std::integral_constant<int, 6>{});
        }
        stage(std::integral_constant<int, 4>{});
    }
    // the dummy loads of the stages past the end still write (zeros) into this workgroup's LDS: they must have landed before the
    // LDS can be handed to another workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int RING>
int launch_zmarch(ZArgs a, hipStream_t st) {
    typedef ZGeom<RING> G;
    auto kernel = zmarch_kernel<RING>;
    if (dmvs_ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), G::LDS)) { (void)hipGetLastError(); return DMVS_EUNSUPPORTED; }
    const unsigned resident = 256u * (unsigned)std::min<size_t>(G::WPS, (160 * 1024) / G::LDS);
    unsigned grid = g_k3z_grid ? (unsigned)g_k3z_grid : resident;
    a.zs = g_k3z_zs ? (int)g_k3z_zs : a.D;
    // no more workgroups than output planes per XCD (a workgroup with an empty range exits at once)
    grid = std::min(grid, xcd_grid((int)std::min<long>((long)a.ngx * a.ngy * a.D, 1L << 30)));
    kernel<<<dim3(grid), 256, G::LDS, st>>>(a);
    DMVS_LAUNCH_CHECK();
}

bool zmarch_shape(int Cin, int Cout, int kdepth) { return Cin == 16 && Cout == 16 && kdepth == 3; }

// (r06 also built this pipeline for conv0 of the two branches -- 2 -> 16 channels, (channel, plane) PAIRS as the MFMA k-group, a
// pair of input planes per stage, producer / finisher waves, 128-byte store runs: parity-green, 0.354 / 0.367 / 0.238 ms at the
// three main-pass shapes where K3w's conv0 kernel takes 0.370 / 0.353 / 0.213 -- not a win, removed; profiles/r06_l_conv0_layers.txt,
// source in git history, commit 60f402f.)

}  // namespace

extern "C" long dmvs_conv3d_zmarch_weight_floats(int Cin, int Cout, int kdepth) {
    return zmarch_shape(Cin, Cout, kdepth) ? 4L * 12 * 256 : 0;
}

extern "C" int dmvs_pack_conv_weights_zmarch(const float* w, float* out, int Cin, int Cout, int kdepth) {
    if (!w || !out || !zmarch_shape(Cin, Cout, kdepth)) return DMVS_EUNSUPPORTED;
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    size_t n = 0;
    // order: wave (= transform row i), k-group, kz, lane (cout = l % 16, channel = 4 kg + l / 16), position p
    for (int i = 0; i < 4; ++i)
        for (int kg = 0; kg < 4; ++kg)
            for (int kz = 0; kz < 3; ++kz)
                for (int l = 0; l < 64; ++l)
                    for (int p = 0; p < 4; ++p) {
                        const int ci = 4 * kg + l / 16, co = l % 16;
                        double u = 0.0;   // (G g G^T)[i][p], formed in double and rounded once
                        for (int ky = 0; ky < 3; ++ky)
                            for (int kx = 0; kx < 3; ++kx)
                                u += Gm[i][ky] * Gm[p][kx] * (double)w[((size_t)co * Cin + ci) * 27 + (kz * 3 + ky) * 3 + kx];
                        out[n++] = (float)u;
                    }
    return n == (size_t)dmvs_conv3d_zmarch_weight_floats(Cin, Cout, kdepth) ? 0 : DMVS_EINVAL;
}

extern "C" int dmvs_conv3d_zmarch(const float* in, float* out, const float* w_packed, const float* scale, const float* shift,
                                  int Cin, int Cout, int D, int H, int W, int kdepth, int flags, dmvs_stream_t stream) {
    if (!in || !out || !w_packed || D < 1 || H < 1 || W < 1) return DMVS_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return DMVS_EINVAL;
    if (flags & ~DMVS_RELU) return DMVS_EUNSUPPORTED;   // no residual, planar output only
    if (!zmarch_shape(Cin, Cout, kdepth)) return DMVS_EUNSUPPORTED;
    if (W % 4 != 0 || ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) != 0) return DMVS_EUNSUPPORTED;
    if ((long)16 * D * H * W >= (1L << 29) || D > 4096) return DMVS_EUNSUPPORTED;   // one descriptor per tensor: byte offsets < 2^31
    ZArgs a = {};
    a.in = in; a.out = out; a.w = w_packed; a.scale = scale; a.shift = shift;
    a.D = D; a.H = H; a.W = W; a.relu = (flags & DMVS_RELU) ? 1 : 0;
    a.ngx = ceil_div(W, 8); a.ngy = ceil_div(H, 8);
    a.counted = g_k3z_counted_wait ? 1 : 0;
    return launch_zmarch<DMVS_K3Z_RING>(a, (hipStream_t)stream);
}
