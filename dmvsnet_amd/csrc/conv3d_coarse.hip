// K3r: the stride-1 3x3(x3) layers of the COARSE levels of the regularisation U-Nets -- conv4 (32 -> 32, 1/4 scale) and
// conv6 (64 -> 64, 1/8 scale) of CostRegNet_part / _part_refine (/root/reference/networks/module.py:367, 370, 409, 412;
// Conv3d / Conv2d + BatchNorm(eval) + ReLU, module.py:120-157) -- in Winograd F(2x2, 3x3) form with REGISTER-STATIONARY weights.
//
// Why another kernel for layers K3w already covers.  At 1/4 and 1/8 scale a layer is a few hundred K3w workgroups, each of
// which walks 8-16 channel chunks of (weight slice + input tile) through ONE LDS stage: load, wait, barrier, 96 MFMAs, barrier.
// With one or two such workgroups per CU nothing overlaps the waits, the 49 KB weight slice of every chunk costs ~50
// LDS-DMA instructions to issue, and a grid of 160-600 workgroups quantises badly on 256 CUs: r04's layer table has these
// layers at 3.2-7.4x (main passes) and 16-60x (refine passes) their floors.  It is NOT the weight bandwidth (VERDICT r04's
// reading): scripts/dev/ub/ingest.hip measures 125-150 GB/s per CU = 52-62 B/clk/CU for an L2-resident buffer through LDS-DMA
// (profiles/r05_a_ub_ingest.txt), so 786 KB of conv6 filters per workgroup is ~6 us of streaming; it is the serial chunk
// pipeline.  Here the weights never move after the prologue:
//   * a 512-thread workgroup = 8 waves = (transform row i = 0..3) x (channel half h = 0..1).  Wave (i, h) keeps the
//     transformed filters U[4i + p][kz][ci][co] of ITS 4 transform positions p, ITS half of the input channels and NCB
//     16-channel output blocks in REGISTERS (64-96 VGPRs), as MFMA B operands -- the 8 waves of a CU hold the whole
//     filter bank of a cout group (conv4: all 32 output channels; conv6: 16 of 64, so 4 cout groups share the volume);
//   * a work unit = one 4 x 4 group of 2x2 output tiles (8 x 8 outputs of one plane) x the cout group.  Row i of the input
//     transform B^T d needs only TWO rows of the 4x4 patch, so the 8 waves together do exactly the 32 additions per patch
//     K3w does, no redundancy; each wave then runs 4 MFMAs (its 4 positions) per (4-channel group, depth tap);
//   * the output transform is split the same way: wave (i, h) reduces its 4 positions to the two output COLUMNS
//     (M[i][:] A), the 8 partial results meet in LDS (16 KB per 16-channel block) and 4 waves per block finish the ROW sum
//     over i (A^T), BatchNorm, ReLU and store 16 bytes per lane;
//   * input tiles are staged per 16 input channels (3D; the whole unit for the 2D layers) with 16-byte LDS-direct loads
//     issued by all 8 waves, lane-linear over the stage, THREE stages in a ring with a counted vmcnt and raw s_barrier (for
//     every instantiation, the 2D layers included: a ring of two measured neutral, r05): the loads of stage k + 2 are issued
//     between the MFMA steps of stage k, so a tile has two stages of MFMA time to land;
//   * persistent: 256 workgroups, workgroup b works for cout group (b / 8) % ncg on XCD b % 8 and walks the units of its
//     XCD's contiguous eighth of the group list; the finish of unit j (partial sums of the other waves) is read after the
//     first barrier of unit j + 1, so a unit costs no barrier of its own (3D).
// Everything is fp32 (v_mfma_f32_16x16x4_f32, transforms with + and -, G g G^T formed in double on the host and rounded
// once); against K3w / K3 the result moves at re-association level (tests: 2e-5 of the output scale against ATen).
// W % 4 != 0 (the 37 x 50 volumes of config 2's stage 1): dword LDS-direct loads (4x the load instructions).
#include "common.h"
#include "tile_loader.h"

#include <algorithm>

#ifndef DMVS_K3R_LW
#define DMVS_K3R_LW 8   /* waves that issue the tile loads of the 16-byte path: 8 = all; 4 = the channel-half-1 waves only (one per SIMD) */
#endif
#ifndef DMVS_K3R_RING
#define DMVS_K3R_RING 3   /* LDS stages: 3 = the loads of stage k + 2 fly during stage k; 2 = one stage ahead, 40 KB less LDS */
#endif
#include "dev_guard.h"   // after the defaults of this file's development switches
#ifdef DMVS_K3R_TRACE
// dev build only (scripts/dev/k3r_trace.sh): per (workgroup, wave) sums of s_memtime ticks spent in the phases of a stage:
// 0 wait + barrier, 1 tile-load issue, 2 finish of the previous unit, 3 patch reads + transforms + MFMAs, 4 partial output transform,
// 5 whole kernel, 6 prologue (filters, loader slots, first issues), 7 stages
__device__ unsigned long long* g_k3r_trace;
extern "C" int dmvs_dev_trace_k3r(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_k3r_trace), &p, sizeof(p)); }
#define R_NOW() __builtin_amdgcn_s_memtime()
#define R_ACC(slot) do { const unsigned long long n__ = R_NOW(); tr[slot] += n__ - tr_t; tr_t = n__; } while (0)
#else
#define R_NOW() 0ull
#define R_ACC(slot) do { } while (0)
#endif

// persistent workgroups of a K3r launch (dmvs_tune("k3r_grid"), a multiple of 32: 8 XCDs x up to 4 cout groups); 256 = one per CU
long g_k3r_grid = 256;
// 1 (default): the stage wait is the COUNTED s_waitcnt vmcnt(NS); 0: vmcnt(0) -- the conservative form the counted one must agree with
// bit for bit (dmvs_tune("k3r_counted_wait"); tests/test_gpu_parity.py::test_conv3d_coarse_counted_wait_is_bit_identical)
long g_k3r_counted_wait = 1;

namespace {

typedef float acc4_t __attribute__((ext_vector_type(4)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));

struct CoarseArgs {
    const float* in;
    float* out;
    const float* w;
    const float* scale;
    const float* shift;
    int Cin, Cout, D, H, W, relu;
    int ngx, ngy;   // 8 x 8-output groups along x / y
    int st4;        // 16-byte output stores allowed (W % 4 == 0, aligned base)
    int counted;    // counted vmcnt at the stage wait (g_k3r_counted_wait)
};

template <int KD, int CIN, int NCB, bool V4>
struct CoarseGeom {
    static constexpr int CPS = KD == 3 ? 16 : 32;    // input channels per LDS stage
    static constexpr int NST = CIN / CPS;            // stages per unit
    static constexpr int GPH = CPS / 8;              // 4-channel k-groups per wave and stage (the two wave halves split a stage)
    static constexpr int RING = DMVS_K3R_RING;
    static constexpr int IXP = 20, IY = 10, PLANE = IXP * IY;   // rows ox0 - 4 .. ox0 + 15, oy0 - 1 .. oy0 + 8
    static constexpr int PS0 = KD * PLANE;
    static constexpr int PS = PS0 + (32 - PS0 % 64 + 64) % 64;   // channel stride = 32 (mod 64) banks: see the patch reads
    static constexpr int PF = V4 ? 4 : 1;                        // floats per LDS-direct piece
    static constexpr int NI = (CPS * PS / PF + 63) / 64;         // load instructions per stage
    // LOADER waves: all eight.  An LDS-direct load costs the issuing wave ~150 ticks whoever issues it; giving the 40 loads of a
    // stage to the four waves of channel half 1 only (one per SIMD, DMVS_K3R_LW = 4: the partner wave keeps the matrix pipe fed)
    // makes those four the critical path -- measured 1.10 ms against 1.01 ms for the twelve conv4 / conv6 shapes of config 2
    // (profiles/r05_e_k3r_trace_loader_waves.txt against r05_c_k3r_trace.txt)
    static constexpr int LW = V4 ? DMVS_K3R_LW : 8;
    static constexpr int NS = (NI + LW - 1) / LW;                // ... per loader wave
    static constexpr int STAGE_F = NS * LW * 64 * PF;
    static constexpr int EXB = NST == 1 ? 2 : 1;                 // one-stage units alternate between two exchange buffers
    static constexpr int EX1_F = 8 * NCB * 4 * 64 * 2;           // exchange: [wave][block][r][lane][2]
    static constexpr int EX_F = EXB * EX1_F;
    static constexpr int NW = NST * GPH * KD * NCB;              // float4 weight registers per lane
    static constexpr size_t LDS = (size_t)(RING * STAGE_F + EX_F) * sizeof(float);
    static_assert(PS % 4 == 0 && PS % 64 == 32, "channel stride");
    static_assert(LDS <= 160 * 1024, "ring + exchange must fit the LDS");
};

template <int KD, int CIN, int NCB, bool V4>
__global__ __launch_bounds__(512, 2) void coarse_kernel(CoarseArgs a) {
    typedef CoarseGeom<KD, CIN, NCB, V4> G;
    constexpr int NST = G::NST, GPH = G::GPH, RING = G::RING, IXP = G::IXP, PS = G::PS, PF = G::PF, NS = G::NS, LW = G::LW;
    constexpr unsigned kInvalid = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [RING][STAGE_F] tiles, [EX_F] exchange
    float* const ex = smem + RING * G::STAGE_F;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, lk = lane >> 4, tx = ln & 3, ty = ln >> 2;
    const int ti = wave & 3, th = wave >> 2;   // transform row, channel half
    [[maybe_unused]] unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_t = R_NOW(), tr_0 = tr_t;

    // ---- work assignment: XCD b % 8 owns the (b % 8)-th contiguous eighth of the group list (x fastest, then z, then y)
    const int ncg = a.Cout / (16 * NCB);
    const int xcd = blockIdx.x & 7, q = (int)(blockIdx.x >> 3);
    const int cg = q % ncg, slot = q / ncg, nslots = (int)(gridDim.x >> 3) / ncg;
    const int ngroups = a.ngx * a.ngy * a.D, per = (ngroups + 7) >> 3;
    const int mine = min(per, ngroups - xcd * per);   // groups of this XCD
    if (slot >= mine) return;
    const int nunits = (mine - slot + nslots - 1) / nslots;
    const int g0 = xcd * per + slot;
    auto coords = [&](int j, int& ox0, int& oy0, int& oz) {
        const int g = g0 + j * nslots;
        const int gx = g % a.ngx, r = g / a.ngx;
        oz = r % a.D;
        oy0 = 8 * (r / a.D);
        ox0 = 8 * gx;
    };

    // ---- the wave's filters: [cout group][wave][stage][k-group][kz][block][lane][4 positions], loaded once
    float4_t w[G::NW];
    {
        const float4_t* wp = reinterpret_cast<const float4_t*>(a.w) + (size_t)(cg * 8 + wave) * G::NW * 64 + lane;
#pragma unroll
        for (int n = 0; n < G::NW; ++n) w[n] = wp[n * 64];
    }
    // ---- the finishing role of this wave: output row parity, x half and 16-channel block
    const bool fin = wave < 4 * NCB;
    const int frr = wave & 1, fxh = (wave >> 1) & 1, fncb = fin ? (wave >> 2) : 0;
    const int co_f = (cg * NCB + fncb) * 16 + ln;
    const float bsc = a.scale ? a.scale[co_f] : 1.f, bsh = a.scale ? a.shift[co_f] : 0.f;
    const float lo = a.relu ? 0.f : -INFINITY;

    // ---- loader: the stage is lane-linear in LDS; piece (wave + 8 sl) * 64 + lane of every stage is the same (channel, plane, row,
    // x) for this lane, decoded once
    const int plane = a.H * a.W, vol = a.D * plane;
    int roff[NS];
    unsigned zyx[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        const int qi = (wave & (LW - 1)) + LW * sl, f = (qi * 64 + lane) * PF;
        const int c = f / PS, rem = f - c * PS;
        const bool okp = qi < G::NI && c < G::CPS && rem < KD * G::PLANE;
        const int row = rem / IXP, x = rem - row * IXP, z = row / G::IY, y = row - z * G::IY;
        roff[sl] = c * vol + z * plane + y * a.W + x;
        zyx[sl] = okp ? (unsigned)(z | (y << 8) | (x << 16)) : 0x3f3f3fu;   // a pad piece fails every range test below
    }
    // one descriptor for the whole tensor; an invalid piece gets offset 2^31 (+- the unit's base): out of range either way
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, CIN * vol * 4, 0x00020000);
    int jq = 0, sq = 0, qox = 0, qoy = 0, qoz = 0;   // the issue stream: next (unit, channel block) and its unit's origin
    unsigned q_ubase4 = 0, q_LO = 0, q_HG = 0;
    float* q_dst = smem;
    int ring_q = 0;
    // the next stage of this workgroup goes to ring slot ring_q: issue_begin sets up its uniform state, issue_slot(sl) sends the
    // wave's sl-th piece.  Past the last stage the pieces are sent out of range (no traffic, zeros into a slot nobody reads):
    // every stage issues exactly NS loads per wave, which is what the counted vmcnt below relies on.
    auto issue_begin = [&]() {
        const bool on = jq < nunits;
        if (on && sq == 0) coords(jq, qox, qoy, qoz);
        const int s = sq, ox0 = qox, oy0 = qoy, oz = qoz;
        if (++sq == NST) { sq = 0; ++jq; }
        const int zb = oz - (KD == 3 ? 1 : 0), yb = oy0 - 1, xb = ox0 - 4;
        q_ubase4 = (unsigned)(s * G::CPS * vol + zb * plane + yb * a.W + xb) * 4u;
        // valid tile coordinates of this unit, [lo, hi] per axis (uniform); the lane's (z, y, x) bytes are range-checked together:
        // with the guard bit 7 set, a byte-wise subtraction keeps the guard iff it did not borrow
        const unsigned zl = max(0, -zb), yl = max(0, -yb), xl = max(0, -xb);
        const unsigned zh = min(KD, a.D - zb) - 1, yh = min(G::IY, a.H - yb) - 1, xh = min(IXP, a.W - xb) - 1;
        q_LO = on ? (zl | (yl << 8) | (xl << 16)) : 0x7f7f7fu;   // off: no coordinate is >= 127
        q_HG = (zh | (yh << 8) | (xh << 16)) | 0x808080u;
        q_dst = smem + ring_q * G::STAGE_F + (wave & (LW - 1)) * 64 * PF;
        ring_q = ring_q + 1 == RING ? 0 : ring_q + 1;
    };
    const bool loader = LW == 8 || th == 1;
    auto issue_slot = [&](int sl) {
        if (!loader) return;
        const unsigned ge = (zyx[sl] | 0x808080u) - q_LO, le = q_HG - zyx[sl];
        const bool ok = (ge & le & 0x808080u) == 0x808080u;
        const unsigned off = ok ? q_ubase4 + (unsigned)roff[sl] * 4u : kInvalid;
        if constexpr (V4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(q_dst + sl * LW * 64 * 4), 16, off, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(q_dst + sl * LW * 64), 4, off, 0, 0, 0);
    };

    // ---- patch reads: the lane's tile (tx, ty), channel lk of a k-group; row i of B^T d = d[ra] + sg * d[rb].  Columns 3 + 2 tx ..
    // 6 + 2 tx of the tile row are read as the three aligned pairs from 2 + 2 tx.  ds_read_b64: two 32-lane groups, bank = dword
    // address mod 64; the 4 x 4 tiles of one channel cover banks {0-7, 40-47, 16-23, 56-63} (+ a common offset), the second
    // channel of the group sits PS = 32 (mod 64) further: conflict-free.
    const int ra = ti == 0 ? 0 : (ti == 2 ? 2 : 1), rb = ti == 0 ? 2 : (ti == 1 ? 2 : (ti == 2 ? 1 : 3));
    const float sg = ti == 1 ? 1.f : -1.f;
    const int lbase = (th * GPH * 4 + lk) * PS + 2 * ty * IXP + 2 + 2 * tx;
    const int baseA = lbase + ra * IXP, baseB = lbase + rb * IXP;

    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.out, (short)0, a.Cout * vol * 4, 0x00020000);
    // the finish of a unit: row sum over the transform rows i (A^T), BatchNorm, ReLU, 16-byte store -- its LDS reads are issued
    // right behind the barrier, the arithmetic and the store follow a few MFMA steps later
    float2_t P[4][2];
    auto finish_read = [&](int eb) {
        if (!fin) return;
        const float2_t* exr = reinterpret_cast<const float2_t*>(ex + eb * G::EX1_F) + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int rs = 0; rs < 2; ++rs) {
                const int r = 2 * fxh + rs;
                const float2_t u = exr[((i * NCB + fncb) * 4 + r) * 64], v = exr[(((i + 4) * NCB + fncb) * 4 + r) * 64];
                P[i][rs].x = u.x + v.x;
                P[i][rs].y = u.y + v.y;
            }
    };
    auto finish_store = [&](int ox0, int oy0, int oz) {
        if (!fin) return;
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
        const bool rok = yy < a.H;
        const unsigned pos = (unsigned)(co_f * vol + oz * plane + yy * a.W + x) * 4u;
        if (a.st4) {
            v4u_t qv;
            qv.x = __builtin_bit_cast(unsigned, fmaxf(y[0] * bsc + bsh, lo));
            qv.y = __builtin_bit_cast(unsigned, fmaxf(y[1] * bsc + bsh, lo));
            qv.z = __builtin_bit_cast(unsigned, fmaxf(y[2] * bsc + bsh, lo));
            qv.w = __builtin_bit_cast(unsigned, fmaxf(y[3] * bsc + bsh, lo));
            __builtin_amdgcn_raw_buffer_store_b128(qv, rs_out, (rok && x < a.W) ? pos : kInvalid, 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(y[e] * bsc + bsh, lo)), rs_out,
                                                      (rok && x + e < a.W) ? pos + 4u * e : kInvalid, 0, 0);
        }
    };

    // ---- pipeline.  Stage k computes out of ring slot k % 3 while the loads of stage k + 2 are issued BETWEEN its MFMA steps
    // (all 8 waves issuing a stage's 40 KB at once right behind the barrier cost 800-1700 ticks per stage with the matrix pipe
    // idle: profiles/r05_b_k3r_trace.txt); at its end the wave waits for its share of stage k + 1 with a COUNTED vmcnt -- the NS
    // newest loads (stage k + 2) may stay in flight, loads retire in order, stores only make the wait more conservative.
    constexpr int NSTEP = GPH * KD;                      // (k-group, depth tap) steps of a stage
    constexpr int FIN_AT = NSTEP >= 4 ? 1 : 0;           // the previous unit's finish runs behind this step's MFMAs
#pragma unroll
    for (int pre = 0; pre < RING - 1; ++pre) {
        issue_begin();
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) issue_slot(sl);
    }
    R_ACC(6);
    int ring_c = 0;
    int pox = 0, poy = 0, poz = 0;
    for (int j = 0; j < nunits; ++j) {
        int ox0, oy0, oz;
        coords(j, ox0, oy0, oz);
        acc4_t acc[NCB][4];
#pragma unroll
        for (int nb = 0; nb < NCB; ++nb)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[nb][p] = (acc4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            // this stage has landed (this wave's share) ... for every wave; and every wave is done with the previous stage and has
            // written its partial sums of the previous unit
            // TOOLCHAIN ASSUMPTION of the counted form (checked for ROCm 7.2.0 hipcc / gfx950): issue_slot compiles to
            // exactly ONE buffer_load ... lds per call, so a loader wave has issued exactly NS VMEM loads for stage k + 2 since its
            // loads for stage k + 1, nothing else loads in the loop, loads retire in order and the stores of the finish only make
            // the wait more conservative.  Two gates stand behind it: tests/test_static_isa.py disassembles the code object and
            // counts the LDS-DMA loads between the counted waits of every coarse_kernel instantiation, and the GPU suite compares
            // this form bit for bit with the vmcnt(0) form (a.counted = 0).
            if (loader && RING == 3 && a.counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (only its own stores and the prologue's filter loads)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            R_ACC(0);
            issue_begin();   // the stage after next, into the slot the previous stage used
            const bool fin_now = s == 0 && j > 0;
            if (fin_now) finish_read(G::EXB == 2 ? ((j - 1) & 1) : 0);
            const float* pa = smem + ring_c * G::STAGE_F + baseA;
            const float* pb = smem + ring_c * G::STAGE_F + baseB;
            ring_c = ring_c + 1 == RING ? 0 : ring_c + 1;
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                const int gg = st / KD, kz = st - gg * KD;
                const int o = gg * 4 * PS + kz * G::PLANE;
                const float2_t a0 = *reinterpret_cast<const float2_t*>(pa + o), a1 = *reinterpret_cast<const float2_t*>(pa + o + 2),
                               a2 = *reinterpret_cast<const float2_t*>(pa + o + 4);
                const float2_t b0 = *reinterpret_cast<const float2_t*>(pb + o), b1 = *reinterpret_cast<const float2_t*>(pb + o + 2),
                               b2 = *reinterpret_cast<const float2_t*>(pb + o + 4);
                const float t0 = fmaf(sg, b0.y, a0.y), t1 = fmaf(sg, b1.x, a1.x), t2 = fmaf(sg, b1.y, a1.y), t3 = fmaf(sg, b2.x, a2.x);
                const float v0 = t0 - t2, v1 = t1 + t2, v2 = t2 - t1, v3 = t1 - t3;
#pragma unroll
                for (int nb = 0; nb < NCB; ++nb) {
                    const float4_t wv = w[((s * GPH + gg) * KD + kz) * NCB + nb];
                    acc[nb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0, wv.x, acc[nb][0], 0, 0, 0);
                    acc[nb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, wv.y, acc[nb][1], 0, 0, 0);
                    acc[nb][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v2, wv.z, acc[nb][2], 0, 0, 0);
                    acc[nb][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v3, wv.w, acc[nb][3], 0, 0, 0);
                }
                if (st == FIN_AT && fin_now) finish_store(pox, poy, poz);
#pragma unroll
                for (int sl = 0; sl < NS; ++sl)
                    if (sl * NSTEP / NS == st) issue_slot(sl);
            }
#ifdef DMVS_K3R_TRACE
            asm volatile("s_nop 0" ::: "memory");
            tr[7] += 1;
#endif
            R_ACC(3);
        }
        // this wave's share of the output transform: M[i][0..3] A -> the two output columns of each tile (register r = tile
        // (tx = r, ty = lk) of channel ln), handed to the finishing waves through LDS
        float2_t* const exw = reinterpret_cast<float2_t*>(ex + (G::EXB == 2 ? (j & 1) : 0) * G::EX1_F) + (size_t)wave * NCB * 4 * 64 + lane;
#pragma unroll
        for (int nb = 0; nb < NCB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m0 = acc[nb][0][r], m1 = acc[nb][1][r], m2 = acc[nb][2][r], m3 = acc[nb][3][r];
                float2_t sv;
                sv.x = (m0 + m1) + m2;
                sv.y = (m1 - m2) - m3;
                exw[(nb * 4 + r) * 64] = sv;
            }
        R_ACC(4);
        pox = ox0; poy = oy0; poz = oz;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    R_ACC(0);
    finish_read(G::EXB == 2 ? ((nunits - 1) & 1) : 0);
    finish_store(pox, poy, poz);
    R_ACC(2);
    // the out-of-range loads of the two stages past the end still write (zeros) into this workgroup's LDS: they must have
    // landed before the LDS can be handed to another workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef DMVS_K3R_TRACE
    tr[5] = R_NOW() - tr_0;
    if (lane == 0 && g_k3r_trace && blockIdx.x < 4096)
        for (int i = 0; i < 8; ++i) g_k3r_trace[((size_t)blockIdx.x * 8 + wave) * 8 + i] = tr[i];
#endif
}

struct RCfg { int cin, cout, kd, ncb; };
// the layers this kernel is compiled for; ncb = 16-channel output blocks per workgroup (Cout / (16 ncb) cout groups)
const RCfg kRCfgs[] = {
    {32, 32, 3, 2},   // conv4            module.py:367
    {64, 64, 3, 1},   // conv6            module.py:370
    {32, 32, 1, 2},   // conv4 on a depth-1 volume (refine passes): the middle 3x3 slice as a 2D layer
    {64, 64, 1, 2},   // refine conv6 (2D) module.py:412, and conv6 on a depth-1 volume
};
const RCfg* find_rcfg(int cin, int cout, int kd) {
    for (const RCfg& c : kRCfgs)
        if (c.cin == cin && c.cout == cout && c.kd == kd) return &c;
    return nullptr;
}
int nw_of(const RCfg& c) { return (c.cin / 8) * c.kd * c.ncb; }   // NST * GPH = Cin / 8 whatever the stage size

template <int KD, int CIN, int NCB, bool V4>
int launch_coarse(CoarseArgs a, hipStream_t st) {
    typedef CoarseGeom<KD, CIN, NCB, V4> G;
    auto kernel = coarse_kernel<KD, CIN, NCB, V4>;
    // a device that cannot give the kernel its ring + exchange (<= 160 KB on gfx950, with no headroom for the 64 -> 64 2D form) is
    // "shape not covered here", not a hard error: `auto` then falls back to K3w
    if (dmvs_ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), G::LDS)) { (void)hipGetLastError(); return DMVS_EUNSUPPORTED; }
    kernel<<<dim3((unsigned)g_k3r_grid), 512, G::LDS, st>>>(a);
    DMVS_LAUNCH_CHECK();
}

}  // namespace

extern "C" long dmvs_conv3d_coarse_weight_floats(int Cin, int Cout, int kdepth) {
    const RCfg* c = find_rcfg(Cin, Cout, kdepth);
    return c ? (long)(Cout / (16 * c->ncb)) * 8 * nw_of(*c) * 256 : 0;
}

extern "C" int dmvs_pack_conv_weights_coarse(const float* w, float* out, int Cin, int Cout, int kdepth) {
    const RCfg* c = find_rcfg(Cin, Cout, kdepth);
    if (!c || !w || !out) return DMVS_EUNSUPPORTED;
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int NT = 9 * kdepth, cps = kdepth == 3 ? 16 : 32, nst = Cin / cps, gph = cps / 8, ncg = Cout / (16 * c->ncb);
    size_t n = 0;
    // order: cout group, wave (i = transform row, h = channel half of a stage), stage, k-group, kz, block, lane, position p
    for (int cg = 0; cg < ncg; ++cg)
        for (int wave = 0; wave < 8; ++wave)
            for (int s = 0; s < nst; ++s)
                for (int gg = 0; gg < gph; ++gg)
                    for (int kz = 0; kz < kdepth; ++kz)
                        for (int nb = 0; nb < c->ncb; ++nb)
                            for (int l = 0; l < 64; ++l)
                                for (int p = 0; p < 4; ++p) {
                                    const int i = wave & 3, h = wave >> 2;
                                    const int ci = s * cps + (h * gph + gg) * 4 + l / 16, co = (cg * c->ncb + nb) * 16 + l % 16;
                                    double u = 0.0;   // (G g G^T)[i][p], formed in double and rounded once
                                    for (int ky = 0; ky < 3; ++ky)
                                        for (int kx = 0; kx < 3; ++kx)
                                            u += Gm[i][ky] * Gm[p][kx] * (double)w[((size_t)co * Cin + ci) * NT + (kz * 3 + ky) * 3 + kx];
                                    out[n++] = (float)u;
                                }
    return n == (size_t)dmvs_conv3d_coarse_weight_floats(Cin, Cout, kdepth) ? 0 : DMVS_EINVAL;
}

extern "C" int dmvs_conv3d_coarse(const float* in, float* out, const float* w_packed, const float* scale, const float* shift,
                                  int Cin, int Cout, int D, int H, int W, int kdepth, int flags, dmvs_stream_t stream) {
    if (!in || !out || !w_packed || D < 1 || H < 1 || W < 1) return DMVS_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return DMVS_EINVAL;
    if (flags & ~DMVS_RELU) return DMVS_EUNSUPPORTED;   // no residual, planar output only
    const RCfg* c = find_rcfg(Cin, Cout, kdepth);
    if (!c) return DMVS_EUNSUPPORTED;
    if ((long)std::max(Cin, Cout) * D * H * W >= (1L << 29)) return DMVS_EUNSUPPORTED;   // one descriptor per tensor: byte offsets < 2^31
    CoarseArgs a = {};
    a.in = in; a.out = out; a.w = w_packed; a.scale = scale; a.shift = shift;
    a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W; a.relu = (flags & DMVS_RELU) ? 1 : 0;
    a.ngx = ceil_div(W, 8); a.ngy = ceil_div(H, 8);
    a.st4 = (W % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) ? 1 : 0;
    a.counted = g_k3r_counted_wait ? 1 : 0;
    const bool v4 = W % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    hipStream_t st = (hipStream_t)stream;
    if (kdepth == 3) {
        if (Cin == 32) return v4 ? launch_coarse<3, 32, 2, true>(a, st) : launch_coarse<3, 32, 2, false>(a, st);
        return v4 ? launch_coarse<3, 64, 1, true>(a, st) : launch_coarse<3, 64, 1, false>(a, st);
    }
    if (Cin == 32) return v4 ? launch_coarse<1, 32, 2, true>(a, st) : launch_coarse<1, 32, 2, false>(a, st);
    return v4 ? launch_coarse<1, 64, 2, true>(a, st) : launch_coarse<1, 64, 2, false>(a, st);
}
