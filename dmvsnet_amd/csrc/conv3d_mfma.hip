// K3: 3D convolution / transposed convolution as an implicit GEMM on the fp32 matrix cores, with the same
// fused BatchNorm(eval) + ReLU + residual epilogue as K2.
//
// Replaces the same reference code as K2 (/root/reference/networks/module.py:120-208, used at 358-436) for the
// layers with Cin >= 8 of CostRegNet_part / _part_refine: conv1..conv6 (module.py:363-370, 405-412) and the
// three deconvolutions (module.py:372-376, 414-418).  fp32-input MFMA (v_mfma_f32_32x32x2_f32 /
// v_mfma_f32_16x16x4_f32) is exact fp32 -- a k-ordered fmaf chain -- so parity with the fp32 reference is at
// re-association level, at the same 157 TF peak as the vector ALUs but with one operand register per
// 2048/4096 FLOP instead of per 128.
//
// GEMM view:  D[cout][voxel] = sum_{tap, ci} W[cout][tap, ci] * X[tap, ci][voxel]
//   weights  packed on the host in EXACT consumption order; each channel chunk's slice (4-28 KB) is staged in LDS
//            next to the input tile and shared by the 4 waves;
//   inputs   32 consecutive x positions of one (z, y) row, read from an LDS tile [ci][z][y][x] (+halo) with one
//            ds_read_b32 per MFMA; taps are just LDS address offsets;
//   M = 32   (Cout >= 32, v_mfma_f32_32x32x2): weights are the A operand, inputs the B operand; in D a lane is a
//            voxel and its registers are channels -> a store instruction writes 128 contiguous bytes of 2 planes;
//   M = 16   (Cout <= 16, v_mfma_f32_16x16x4): the roles are SWAPPED (voxels = MFMA rows, channels = columns), so a
//            lane holds 4 consecutive voxels of one channel and the epilogue moves 16 bytes per lane.
// A 256-thread workgroup (4 waves) owns TZ x TY rows of 32 voxels; each wave owns ROWS of them and all output
// channels (MB blocks of M), so one weight fragment feeds ROWS*XB MFMAs and one input fragment feeds MB.
// Small (low-resolution) volumes use the ROWS=1 tiles so the grid still covers the 256 CUs.
//
// Pipeline: input tile and weight slice of chunk c+1 are fetched with asynchronous 16-byte LDS-direct buffer loads
// (no VGPRs, no ds_write; load_tile4, dword variant when W % 4 != 0) into the second LDS buffer while the MFMAs
// consume chunk c; one barrier per chunk.  The chunk size is picked per layer for occupancy (2 channels = packed-K
// on the 16-row MFMA: 2 taps x 2 channels per k-group).  The launch is 1-D with an XCD-aware tile order (common.h).
//
// Transposed conv (k3 s2 p1 output_padding 1): gather form on the INPUT grid.  A wave owns one input row; the
// 2x2x2 output parities are 8 accumulator sets; parity p pairs with input offset o along an axis through tap
// (p,o) = (0,0)->1, (1,0)->2, (1,1)->0, 27 (parity, offset) combinations in total.
// The two x-parities of a voxel are stored together as one float2 -> stores stay fully coalesced.
//
// KD = 1: 1x3x3 kernel per depth slice, no depth stride / upsampling -- the 2D bottleneck (conv5/6/7) of the
// refine net run on [C][1][H][W].
// Development knock-outs (scripts/ko_build.sh): bit 0 no tile loads, bit 1 MFMA -> one VALU fma, bit 2 no
// epilogue memory traffic, bit 3 no B-operand LDS reads, bit 4 tile loads issued but all out of range (no memory
// traffic, same instruction stream).  Never set in the product build.
#ifndef DMVS_KO
#define DMVS_KO 0
#endif
#ifndef DMVS_X   // dev A/B switches of the r04 changes (scripts/dev): bit 0 BN constants of the 16-row conv in the epilogue,
#define DMVS_X 0 // bit 1 SCALAR BN loads of the 32-row epilogues (slower end to end), bit 2 deconv BN constants in the epilogue
#endif
#include "common.h"
#include "tile_loader.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>


#include <map>
#include <type_traits>
#include <utility>

int dmvs_ensure_dynamic_lds(const void* kernel, size_t lds_bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> configured;   // (device, kernel) -> largest size set
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_pair(dev, kernel);
    auto it = configured.find(key);
    if (it == configured.end() || it->second < lds_bytes) {
        hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        configured[key] = lds_bytes;
    }
    return 0;
}

// 3D layers with fewer workgroups than this keep two LDS stages (dmvs_tune("k3_single_buf_min_blocks")).  Measured on
// config 2 (r02): 0 (every 3D layer single-staged) 72.0, 256-2048 71.5, all double-staged 69.9 depth-maps/s.
long g_single_buf_min_blocks = 0;
// big tiles need at least this many workgroups; two-block layers below the second number split their M blocks
// (dmvs_tune("k3_min_blocks" / "k3_split_blocks"))
long g_min_blocks = 768, g_split_blocks = 1024;
// transposed convs with a residual prefetch it under the MFMAs (dmvs_tune("k3_deconv_prefetch"), see deconv_mfma_kernel)
long g_deconv_prefetch = 1;

#ifdef DMVS_K3_TRACE
// dev build only (scripts/dev/k3_trace.sh): per-workgroup s_memtime stamps of the kernel's phases.  Slots: 0 start, 1 first
// chunk landed (wait + barrier passed), 2 chunk loop done, 3 epilogue loads landed (deconv), 4 stores retired (end);
// 8 = cycles spent at the chunk waits + barriers (sum over chunks), 9 = cycles inside the MFMA sections, 15 = HW_ID.
__device__ unsigned long long* g_k3_trace;
extern "C" int dmvs_dev_trace_k3(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_k3_trace), &p, sizeof(p)); }
#define K3_NOW() __builtin_amdgcn_s_memtime()
#define K3_TR(slot, val) do { if (threadIdx.x == 0 && g_k3_trace && blockIdx.x < 65536) g_k3_trace[(size_t)blockIdx.x * 16 + (slot)] = (val); } while (0)
#define K3_TR_HW() do { if (threadIdx.x == 0 && g_k3_trace && blockIdx.x < 65536) { unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); g_k3_trace[(size_t)blockIdx.x * 16 + 15] = hw; g_k3_trace[(size_t)blockIdx.x * 16 + 14] = xcc; } } while (0)
#else
#define K3_NOW() 0ull
#define K3_TR(slot, val) do { } while (0)
#define K3_TR_HW() do { } while (0)
#endif

namespace {

struct ConvArgs {
    const float* in;
    float* out;
    const float* w;
    const float* scale;
    const float* shift;
    const float* skip;
    int Cin, Cout, D, H, W, Do, Ho, Wo, relu;
    int skip_up2;  // residual is at half resolution in H and W: read skip[co][z][y/2][x/2] (FPN top-down add)
    int nx, ny, nz;  // tile grid (the launch is 1-D, see xcd_tile)
    int st4;         // output rows are whole 16-byte pieces (Wo % 4 == 0, aligned base): 16-byte stores allowed
    int in_cs, in_zs, in_elems;  // DMVS_IN_VIEWS: channel / slice strides of `in` and its length (floats); 0 = planar [C][D][H][W]
    int single_buf;  // one LDS stage instead of two (see launch_conv_tile_v)
    int outq4;        // output = two quad-planar tensors [Do][Cout/8][Ho][Wo][4] (channels [0, Cout/2) then the rest): DMVS_OUT_Q4
};

typedef float acc16_t __attribute__((ext_vector_type(16)));
typedef float acc4_t __attribute__((ext_vector_type(4)));

template <int M> struct Frag;
template <> struct Frag<32> {
    static constexpr int KK = 2, NV = 32, ACC = 16;
    typedef acc16_t acc_t;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    // accumulator register r of lane -> output row (channel) inside the M block
    static __device__ __forceinline__ int row(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }
};
template <> struct Frag<16> {
    static constexpr int KK = 4, NV = 16, ACC = 4;
    typedef acc4_t acc_t;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int r, int lk) { return lk * 4 + r; }
};

// Stage one chunk's weight slice (NROWS rows of 64 floats, already in consumption order) in LDS: 16-byte
// LDS-direct loads, 4 rows (1 KiB) per wave-instruction; the lanes past the slice's end are switched off.
template <int NROWS>
__device__ __forceinline__ void load_weights(__amdgpu_buffer_rsrc_t rs_w, float* wl, int chunk, int wave, int lane) {
    constexpr int NI = (NROWS + 3) / 4;  // instructions per slice
    const unsigned base = (unsigned)chunk * NROWS * 256u + (unsigned)lane * 16u;
#pragma unroll
    for (int r = 0; r < (NI + 3) / 4; ++r) {
        const int j = min(wave + 4 * r, NI - 1);  // slot past the end re-loads the last piece
        if (j * 256 + lane * 4 < NROWS * 64)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(wl + j * 256), 16, base + (unsigned)j * 1024u, 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------ conv
// V4: tile staged with 16-byte LDS-direct loads (load_tile4): rows start XOFF floats left of the first tap at a
// 16-byte aligned x and are dense; the channel stride is padded so that the lk groups of a B read (4 x 16 lanes for
// M = 16, 2 x 32 for M = 32) fall on different banks.
template <int M, int STRIDE, int KD, int KS, int CI_CH, int TZ, int TY, bool V4>
struct ConvGeom {
    static constexpr int KK = Frag<M>::KK;
    static constexpr int NT = KS * KS * KD;  // taps; KS = in-plane kernel size (1, 3 or 5), pad KS/2
    static constexpr int IZ = KD == 3 ? (TZ - 1) * STRIDE + 3 : TZ, IY = (TY - 1) * STRIDE + KS, IX = 31 * STRIDE + KS;
    static constexpr int XOFF = V4 && KS > 1 ? 4 - KS / 2 : 0;
    static constexpr int IXP = V4 ? (XOFF + IX + 3) / 4 * 4 : IX + 1;
    static constexpr int LPR = IXP / 4;
    static constexpr int PS0 = IZ * IY * IXP;
    static constexpr int BANK = M == 16 ? 16 : 32;
    static constexpr int PS = V4 ? PS0 + (BANK - PS0 % 64 + 64) % 64 : PS0;
    static constexpr int GPC = CI_CH / KK;                 // k-groups per tap (0 in packed-K mode)
    static constexpr int TPG = CI_CH < KK ? KK / CI_CH : 1;  // taps per k-group (packed-K: Cin=2 -> 2 taps x 2 ch)
    static constexpr int NSTEPS = CI_CH < KK ? (NT + TPG - 1) / TPG : NT * GPC;  // MFMA k-steps per chunk
    static constexpr int TILE_F = (CI_CH * PS + 63) & ~63;
};

// min-waves hint: the M = 16 instantiations are the HBM-bound full-resolution layers -- ask for 3 waves/SIMD
// (<= 168 registers) so three workgroups per CU overlap one another's load / MFMA / store phases.
// MBS ("M-block split", MB = 2 layers on small volumes): waves 0/1 and 2/3 each share a row group and take ONE of the
// two 32-channel blocks each, so a workgroup owns half as many rows and the layer yields twice as many workgroups of
// half the MFMA work -- the 1/8-scale bottleneck layers (conv5 / conv6: a few hundred workgroups for 1024 SIMDs)
// otherwise leave most of the chip idle in their last round.
template <int M, int MB, int STRIDE, int KD, int KS, int CI_CH, int TZ, int TY, int ROWS, bool V4, bool MBS = false>
__global__ __launch_bounds__(256, (M == 16 ? 3 : 1)) void conv_mfma_kernel(ConvArgs a) {
    typedef Frag<M> F;
    typedef typename F::acc_t acc_t;
    typedef ConvGeom<M, STRIDE, KD, KS, CI_CH, TZ, TY, V4> G;
    constexpr int PAD = KS / 2;
    constexpr int XB = 32 / F::NV;
    constexpr int SZ = KD == 3 ? STRIDE : 1;
    constexpr int IZ = G::IZ, IY = G::IY, IX = G::IX, IXP = G::IXP, PS = G::PS, GPC = G::GPC;
    constexpr int WROWS = G::NSTEPS * MB;           // weight rows (64 floats each) per chunk
    constexpr int BUF_F = G::TILE_F + WROWS * 64;   // one pipeline stage: tile + weight slice
    constexpr bool PACKED = CI_CH < F::KK;          // several taps share one MFMA k-group (conv0, Cin = 2)
    // M = 16: the operand roles are swapped (voxels are the MFMA rows, channels the columns), so a lane ends up
    // with 4 CONSECUTIVE voxels of one channel and the epilogue moves 16 bytes per lane: a quarter of the store
    // instructions, and the 64-byte runs of a dword-per-lane 16-voxel store (4.3 TB/s measured) become 5.2 TB/s.
    constexpr bool TR = M == 16;
    static_assert(TZ * TY == (MBS ? 2 : 4) * ROWS, "tile rows must equal the row-owning waves x ROWS");
    static_assert(!MBS || MB == 2, "M-block split: two-block layers only");
    constexpr int MBL = MBS ? 1 : MB;  // M blocks per wave
    static_assert(PACKED ? (F::KK % CI_CH == 0) : (CI_CH % F::KK == 0), "channel chunk vs MFMA k-group");
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][BUF_F]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane % F::NV, lk = lane / F::NV;
    const int rwave = MBS ? (wave >> 1) : wave;  // which rows of the tile this wave owns
    const int mb0 = MBS ? (wave & 1) : 0;        // ... and its first M block
    int bx, by, bz;
    if (!xcd_tile(a.nx, a.ny, a.nz, KD == 3, bx, by, bz)) return;
    const int ox0 = bx * 32, oy0 = by * TY, oz0 = bz * TZ;
    const int ix0 = ox0 * STRIDE - PAD, iy0 = oy0 * STRIDE - PAD, iz0 = KD == 3 ? oz0 * STRIDE - 1 : oz0;
    if (DMVS_X & 8) {   // dev experiment: co-resident workgroups at different issue priorities (breaks phase lock-step?)
        switch ((blockIdx.x >> 3) & 3) {
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            case 3: __builtin_amdgcn_s_setprio(3); break;
            default: break;
        }
    }

    int boff[ROWS][XB];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const int r = rwave * ROWS + i, tz = r / TY, ty = r % TY;
#pragma unroll
        for (int xb = 0; xb < XB; ++xb)
            boff[i][xb] = (PACKED ? lk % CI_CH : lk) * PS + (tz * SZ * IY + ty * STRIDE) * IXP + (xb * F::NV + ln) * STRIDE + G::XOFF;
    }

    // packed-K: tile offset of the lane's (tap, channel) in every k-step, relative to the tile base: boff[0][0] + the tap
    // this lane's k index selects (rows i > 0 / x blocks add compile-time constants: ROWS divides TY, so a wave's rows
    // differ in y only)
    static_assert(TY % ROWS == 0 || !PACKED, "a wave's rows must differ in y only");
    int ptap[PACKED ? G::NSTEPS : 1];
    if constexpr (PACKED) {
        constexpr int TPG = G::TPG, NT = G::NT;
        const int tsel = lk / CI_CH;
#pragma unroll
        for (int st = 0; st < G::NSTEPS; ++st) {
            int toff = 0;
#pragma unroll
            for (int q = 0; q < TPG; ++q) {
                const int t = st * TPG + q < NT ? st * TPG + q : NT - 1;  // padded taps carry zero weights
                const int o = ((t / (KS * KS)) * IY + (t / KS) % KS) * IXP + t % KS;
                toff = (tsel == q) ? o : toff;
            }
            ptap[st] = boff[0][0] + toff;
            asm volatile("" : "+v"(ptap[st]));   // keep it a register: the compiler otherwise re-derives the select chain
        }
    }

    acc_t acc[MBL][ROWS][XB];
#pragma unroll
    for (int mb = 0; mb < MBL; ++mb)
#pragma unroll
        for (int i = 0; i < ROWS; ++i)
#pragma unroll
            for (int xb = 0; xb < XB; ++xb)
#pragma unroll
                for (int r = 0; r < F::ACC; ++r) acc[mb][i][xb][r] = 0.f;

    const int in_vol = a.D * a.H * a.W;
    auto chunk_rsrc = [&](int ci0, int nch) {  // descriptor of the channels [ci0, ci0 + nch) only
        if (a.in_cs)   // DMVS_IN_VIEWS: the chunk starts at channel ci0 of slice 0 and ends with the tensor (see the flag)
            return __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)ci0 * a.in_cs), (short)0, (a.in_elems - ci0 * a.in_cs) * 4, 0x00020000);
        return __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)ci0 * in_vol), (short)0, nch * in_vol * 4, 0x00020000);
    };
    const int nchunks = a.Cin / CI_CH;
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, nchunks * WROWS * 256, 0x00020000);

    auto stage = [&](int c, float* dst) {  // chunk c: input tile + weight slice, asynchronous
        if (!(DMVS_KO & 1)) {
            if constexpr (V4)
                load_tile4<CI_CH, IZ, IY, G::LPR, PS>(a.D, a.H, a.W, chunk_rsrc(c * CI_CH, CI_CH), dst, iz0, iy0, ix0 - G::XOFF, wave, lane, a.in_cs, a.in_zs);
            else
                load_tile<CI_CH, IZ, IY, IX, IXP, PS, true>(a.D, a.H, a.W, chunk_rsrc(c * CI_CH, CI_CH), dst, c * CI_CH, iz0, iy0, ix0, wave, lane, a.in_cs, a.in_zs);
        }
        load_weights<WROWS>(rs_w, dst + G::TILE_F, c, wave, lane);
    };
    // BatchNorm constants of the lane's channel(s) (M = 16: one channel per M block): loaded FIRST, under the first tile --
    // in the epilogue the load is an exposed round trip of several thousand cycles at the end of every workgroup's life
    // (r04 phase trace: 10 k of conv1's 54 k ticks were spent between the last MFMA and the retirement of the stores)
    float sc_tr[TR ? MBL : 1], sh_tr[TR ? MBL : 1];
    if constexpr (TR && !(DMVS_X & 1)) {
#pragma unroll
        for (int mb = 0; mb < MBL; ++mb) {
            const int co = (mb0 + mb) * M + ln;
            const bool cok = co < a.Cout;
            sc_tr[mb] = (a.scale && cok) ? a.scale[co] : 1.f;
            sh_tr[mb] = (a.scale && cok) ? a.shift[co] : 0.f;
        }
    }
    K3_TR(0, K3_NOW());
    K3_TR_HW();
    [[maybe_unused]] unsigned long long tr_wait = 0, tr_mfma = 0, tr_t = K3_NOW();
    stage(0, smem);
    for (int c = 0; c < nchunks; ++c) {
        // chunk c has landed (this wave's share) ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... for every wave; and every wave is done reading the other buffer (chunk c-1)
        __syncthreads();
#ifdef DMVS_K3_TRACE
        { const unsigned long long n = K3_NOW(); tr_wait += n - tr_t; tr_t = n; if (c == 0) K3_TR(1, n); }
#endif
        float* cur = smem + (a.single_buf ? 0 : (c & 1)) * BUF_F;
        if (c + 1 < nchunks && !a.single_buf) {
            stage(c + 1, smem + ((c + 1) & 1) * BUF_F);
        }
        const float* tile = cur;
        const float* wl = cur + G::TILE_F + lane;
        if constexpr (PACKED) {
            // k index of a lane inside a step: lk = tap_select * CI_CH + ci.  The lane's tap offset of every step is a
            // per-lane constant of the whole kernel (ptap[], set up before the chunk loop): one address add per step,
            // the (row, x block) of an MFMA is an immediate offset (r04: the select chain that formed the offset inside
            // the loop cost 1.6 VALU instructions per MFMA -- paid in full next to fp32 MFMAs)
#pragma unroll
            for (int st = 0; st < G::NSTEPS; ++st) {
                float av[MBL];
#pragma unroll
                for (int mb = 0; mb < MBL; ++mb) av[mb] = wl[(st * MB + mb0 + mb) * 64];
                const float* ps = tile + ptap[st];
#pragma unroll
                for (int i = 0; i < ROWS; ++i)
#pragma unroll
                    for (int xb = 0; xb < XB; ++xb) {
                        const float bv = (DMVS_KO & 8) ? (float)lane : ps[i * STRIDE * IXP + xb * F::NV * STRIDE];
#pragma unroll
                        for (int mb = 0; mb < MBL; ++mb) {
                            if (DMVS_KO & 2) acc[mb][i][xb][0] = fmaf(av[mb], bv, acc[mb][i][xb][0]);
                            else acc[mb][i][xb] = TR ? F::mfma(bv, av[mb], acc[mb][i][xb]) : F::mfma(av[mb], bv, acc[mb][i][xb]);
                        }
                    }
            }
        } else
#pragma unroll
        for (int kz = 0; kz < KD; ++kz)
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int toff = (kz * IY + ky) * IXP + kx;
                    const int t = (kz * KS + ky) * KS + kx;
#pragma unroll
                    for (int g = 0; g < GPC; ++g) {
                        float av[MBL];
#pragma unroll
                        for (int mb = 0; mb < MBL; ++mb) av[mb] = wl[((t * GPC + g) * MB + mb0 + mb) * 64];
#pragma unroll
                        for (int i = 0; i < ROWS; ++i)
#pragma unroll
                            for (int xb = 0; xb < XB; ++xb) {
                                const float bv = (DMVS_KO & 8) ? (float)lane : tile[boff[i][xb] + toff + g * F::KK * PS];
#pragma unroll
                                for (int mb = 0; mb < MBL; ++mb) {
                                    if (DMVS_KO & 2) acc[mb][i][xb][0] = fmaf(av[mb], bv, acc[mb][i][xb][0]);
                                    else acc[mb][i][xb] = TR ? F::mfma(bv, av[mb], acc[mb][i][xb]) : F::mfma(av[mb], bv, acc[mb][i][xb]);
                                }
                            }
                    }
                }
#ifdef DMVS_K3_TRACE
        { asm volatile("s_nop 0" ::: "memory"); const unsigned long long n = K3_NOW(); tr_mfma += n - tr_t; tr_t = n; }
#endif
        if (a.single_buf && c + 1 < nchunks) {  // single stage: refill it once every wave is done with chunk c
            __syncthreads();
            stage(c + 1, smem);
        }
    }
    K3_TR(2, K3_NOW());
    K3_TR(8, tr_wait);
    K3_TR(9, tr_mfma);

    // epilogue: BN scale/shift + ReLU + residual; 128-byte runs per channel plane.  Branch-free: residual
    // loads and stores go through range-checked buffer descriptors, an element that must not be touched
    // (tile overhang, padded channel) gets an out-of-range offset (load returns 0 / store is dropped); with no
    // residual the descriptor has zero records.  All residual loads of a lane are in flight together.
    constexpr unsigned kInvalid = 0x80000000u;
    const int out_plane = a.Ho * a.Wo, out_vol = a.Do * out_plane;
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.out, (short)0, a.Cout * out_vol * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_skip = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.skip ? a.skip : a.out), (short)0, a.skip ? (a.skip_up2 ? a.Cout * out_vol : a.Cout * out_vol * 4) : 0, 0x00020000);
    const float lo = a.relu ? 0.f : -INFINITY;
    if constexpr (TR) {
        typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int mb = 0; mb < MBL; ++mb) {
            const int co = (mb0 + mb) * M + ln;  // the lane's channel; its registers are voxels lk * 4 + r of each 16-block
            const bool cok = co < a.Cout;
            const float sc = (DMVS_X & 1) ? ((a.scale && cok) ? a.scale[co] : 1.f) : sc_tr[mb];
            const float sh = (DMVS_X & 1) ? ((a.scale && cok) ? a.shift[co] : 0.f) : sh_tr[mb];
            const unsigned cooff = cok ? (unsigned)(co * out_vol) * 4u : kInvalid;
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                const int r = rwave * ROWS + i, oz = oz0 + r / TY, oy = oy0 + r % TY;
                const bool rok = oz < a.Do && oy < a.Ho;
#pragma unroll
                for (int xb = 0; xb < XB; ++xb) {
                    const int ox = ox0 + xb * F::NV + lk * 4;
                    const unsigned rowpos = (unsigned)(oz * out_plane + oy * a.Wo + ox) * 4u;
                    float v[4];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        float sk = 0.f;
                        if (a.skip) {  // uniform; no M = 16 layer of the networks has a residual: plain dword loads
                            const bool ok = rok && ox + rr < a.Wo && cok;
                            const unsigned spos = !a.skip_up2 ? rowpos + 4u * rr
                                : (unsigned)((oz * (a.Ho >> 1) + (oy >> 1)) * (a.Wo >> 1) + ((ox + rr) >> 1)) * 4u;
                            sk = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_skip, ok ? spos + (a.skip_up2 ? cooff >> 2 : cooff) : kInvalid, 0, 0));
                        }
                        v[rr] = fmaxf(acc[mb][i][xb][rr] * sc + sh, lo) + sk;
                    }
                    if ((DMVS_KO & 4) && v[0] + v[1] + v[2] + v[3] != 1234.56789f) continue;
                    if (a.outq4) {  // quad-planar halves [half][Do][Cout/8][Ho][Wo][4]: the 4 lanes of a channel quad write one 16-byte piece per voxel
                        const int ch = a.Cout >> 1, hsel = co >= ch ? 1 : 0, cq = ch >> 2, cl = co - hsel * ch;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const unsigned off = (rok && ox + rr < a.Wo && cok)
                                ? ((unsigned)(((hsel * a.Do + oz) * cq + (cl >> 2)) * out_plane + oy * a.Wo + ox + rr) * 4u + (unsigned)(cl & 3)) * 4u : kInvalid;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[rr]), rs_out, off, 0, 0);
                        }
                    } else if (a.st4) {  // Wo % 4 == 0 and 16-byte aligned planes: the 4 voxels are inside or outside together
                        const unsigned off = (rok && ox < a.Wo && cok) ? rowpos + cooff : kInvalid;
                        v4u_t q;
                        q.x = __builtin_bit_cast(unsigned, v[0]); q.y = __builtin_bit_cast(unsigned, v[1]);
                        q.z = __builtin_bit_cast(unsigned, v[2]); q.w = __builtin_bit_cast(unsigned, v[3]);
                        __builtin_amdgcn_raw_buffer_store_b128(q, rs_out, off, 0, 0);
                    } else {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const unsigned off = (rok && ox + rr < a.Wo && cok) ? rowpos + 4u * rr + cooff : kInvalid;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[rr]), rs_out, off, 0, 0);
                        }
                    }
                }
            }
        }
    } else {
#pragma unroll
    for (int mb = 0; mb < MBL; ++mb) {
        float sc[F::ACC], sh[F::ACC];
        unsigned cooff[F::ACC];
#pragma unroll
        for (int rr = 0; rr < F::ACC; ++rr) {
            const int co = (mb0 + mb) * M + F::row(rr, lk);
            const bool cok = co < a.Cout;
            const int coc = cok ? co : 0;
            if (M == 32 && (a.Cout & 31) == 0 && (DMVS_X & 2)) {   // dev switch, OFF: measured -1.5 % end to end (r04)
                // the lane's channel is (wave-uniform block) + row(rr, 0) + 4 * lk: both candidates through the SCALAR
                // cache and one select, instead of 2 x 16 vector loads per lane whose round trip ends every workgroup
                const int c0 = (mb0 + mb) * M + F::row(rr, 0);
                const float s0 = a.scale ? a.scale[c0] : 1.f, s1 = a.scale ? a.scale[c0 + 4] : 1.f;
                const float h0 = a.scale ? a.shift[c0] : 0.f, h1 = a.scale ? a.shift[c0 + 4] : 0.f;
                sc[rr] = lk ? s1 : s0;
                sh[rr] = lk ? h1 : h0;
            } else {
                sc[rr] = a.scale ? a.scale[coc] : 1.f;
                sh[rr] = a.scale ? a.shift[coc] : 0.f;
            }
            cooff[rr] = cok ? (unsigned)(co * out_vol) * 4u : kInvalid;
        }
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int r = rwave * ROWS + i, oz = oz0 + r / TY, oy = oy0 + r % TY;
            const bool rok = oz < a.Do && oy < a.Ho;
#pragma unroll
            for (int xb = 0; xb < XB; ++xb) {
                const int ox = ox0 + xb * F::NV + ln;
                const unsigned pos = (rok && ox < a.Wo) ? (unsigned)(oz * out_plane + oy * a.Wo + ox) * 4u : kInvalid;
                const unsigned spos = !a.skip_up2 ? pos
                    : (rok && ox < a.Wo) ? (unsigned)((oz * (a.Ho >> 1) + (oy >> 1)) * (a.Wo >> 1) + (ox >> 1)) * 4u : kInvalid;
                float sk[F::ACC];
#pragma unroll
                for (int rr = 0; rr < F::ACC; ++rr) {
                    const unsigned so = a.skip_up2 ? cooff[rr] >> 2 : cooff[rr];  // channel stride is 1/4 at half res
                    sk[rr] = (DMVS_KO & 4) ? 0.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_skip, ((spos | cooff[rr]) & kInvalid) ? kInvalid : spos + so, 0, 0));
                }
                if (a.outq4) {
                    // quad-planar halves [half][Do][Cout/8][Ho][Wo][4] (the layout K1 samples): registers 4q .. 4q+3 of a
                    // lane are 4 CONSECUTIVE channels of its voxel -> one 16-byte piece; the 32 lanes of a row write 512
                    // contiguous bytes of a quad plane
                    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
                    const int ch = a.Cout >> 1, cq = ch >> 2;
#pragma unroll
                    for (int q = 0; q < F::ACC / 4; ++q) {
                        const int co0 = (mb0 + mb) * M + F::row(4 * q, lk), hsel = co0 >= ch ? 1 : 0;
                        float w4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) w4[e] = fmaxf(acc[mb][i][xb][4 * q + e] * sc[4 * q + e] + sh[4 * q + e], lo) + sk[4 * q + e];
                        const unsigned off = (!(pos & kInvalid) && co0 < a.Cout)
                            ? (unsigned)(((hsel * a.Do + oz) * cq + ((co0 - hsel * ch) >> 2)) * out_plane + oy * a.Wo + ox) * 16u : kInvalid;
                        v4u_t qv;
                        qv.x = __builtin_bit_cast(unsigned, w4[0]); qv.y = __builtin_bit_cast(unsigned, w4[1]);
                        qv.z = __builtin_bit_cast(unsigned, w4[2]); qv.w = __builtin_bit_cast(unsigned, w4[3]);
                        __builtin_amdgcn_raw_buffer_store_b128(qv, rs_out, off, 0, 0);
                    }
                } else
#pragma unroll
                for (int rr = 0; rr < F::ACC; ++rr) {
                    const unsigned off = (pos | cooff[rr]) & kInvalid ? kInvalid : pos + cooff[rr];
                    const float v = fmaxf(acc[mb][i][xb][rr] * sc[rr] + sh[rr], lo) + sk[rr];
                    if ((DMVS_KO & 4) && v != 1234.56789f) continue;  // keeps the accumulators alive, never stores
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, off, 0, 0);
                }
            }
        }
    }
    }
#ifdef DMVS_K3_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    K3_TR(4, K3_NOW());
#endif
}

// ------------------------------------------------------------------------------------------------ deconv
// PYM ("y-parity merged", Cout = 8 on the 16-row MFMA): rows 0-7 of the A operand are output parity py = 0 and
// rows 8-15 py = 1 of the SAME input fragment, so the 16 rows carry 2 x 8 real channels instead of 8 + 8 zeros:
// input offset oy = 0 feeds [W(ky=1) ; W(ky=2)], oy = 1 feeds [0 ; W(ky=0)] -> 18 instead of 27 k-steps per group.
template <int M, int KD, int CI_CH, int TZ, int TY, bool PYM = false, int NMB = 1>
struct DeconvGeom {
    static constexpr int IZ = KD == 3 ? TZ + 1 : TZ, IY = TY + 1, IX = 33, IXP = 34;
    static constexpr int PS = IZ * IY * IXP;
    static constexpr int GPC = CI_CH / Frag<M>::KK;
    static constexpr int TILE_F = (CI_CH * PS + 63) & ~63;
    static constexpr int WROWS = (PYM ? (KD == 3 ? 18 : 6) : (KD == 3 ? 27 : 9)) * GPC * NMB;
    static constexpr int BUF_F = TILE_F + WROWS * 64;
};

// PREF: the residual ("skip") values of the epilogue are PREFETCHED under the MFMAs.  The epilogue moves 12x the bytes of
// the input tiles (conv11: 32 KB of residual + 32 KB of output per workgroup against 5 KB of input), and without the
// prefetch its loads are only issued after the last MFMA, four at a time with a full memory round trip each -- a
// workgroup then alternates between a phase that only computes and a phase that only waits on HBM.  With PREF all 16
// residual loads of the wave (4 groups of ACC = 4 eight-byte loads: 8 KB per wave in flight) are issued right after the tile
// loads of chunk 1, in the first iteration; the wait of chunk 1 is a COUNTED vmcnt(16) (loads retire in order: everything but
// the 16 newest -- the residuals -- must have landed) and the later waits find them long done, so the residuals fly under
// the MFMAs and the epilogue only scales, adds and stores.  (First version: one group per chunk, vmcnt(4) each -- 4 % slower
// alone, equal end to end.)
// The counted wait assumes that the compiler emits exactly NGRP * ACC = 16 VMEM loads for the prefetch and keeps them the NEWEST
// in the queue: validated with ROCm 7.2.0 hipcc (AMD clang 22); the build gate is
// tests/test_gpu_parity.py::test_deconv_residual_prefetch_is_bit_identical (PREF vs the epilogue-load form, dmvs_tune
// "k3_deconv_prefetch" = 0) -- re-run it after any toolchain change; a mis-count shows up there as a bit difference.
// The chunk loop of a PREF instantiation has a COMPILE-TIME trip count (NCH = Cin / CI_CH) and is fully unrolled: every
// residual group is then issued exactly once in straight-line code and owns its registers (with a runtime loop the
// compiler sees several possible issue points per group and guards them with vmcnt(0) -- which would also drain the tile
// loads that are meant to fly under the MFMAs).
// NMB = 2 ("M-block split", conv7: 64 -> 32 on the 16-row MFMA): waves 0/1 and 2/3 share an input row and take one 16-channel
// block each, so a workgroup owns two input rows instead of four and a wave's accumulators are 64 registers instead of the
// 128 of the 32-row form (356 registers, one wave per SIMD): twice the workgroups of half the MFMA chain each, 4 waves per
// SIMD -- conv7 only ever runs on the 1/8-scale grids of 74-520 workgroups, where the 32-row form left most SIMDs with one
// wave or none (r04 layer table: 3.6-4.8x its floor; 0.62 -> 0.51 ms per depth map, +0.4 % end to end).  Choosing the block
// by WORKGROUP instead (four rows, half the weight bytes per workgroup) measured 0.49 ms alone but 89.7 vs 90.5 end to end.
template <int M, int KD, int CI_CH, int TZ, int TY, bool PYM, bool PREF, int NCH = 0, int NMB = 1>
__global__ __launch_bounds__(256, (M == 16 ? 3 : 1)) void deconv_mfma_kernel(ConvArgs a) {
    typedef Frag<M> F;
    typedef typename F::acc_t acc_t;
    typedef DeconvGeom<M, KD, CI_CH, TZ, TY, PYM, NMB> G;
    static_assert(NMB == 1 || (NMB == 2 && M == 16 && !PYM && !PREF), "M-block split: two 16-channel blocks");
    static_assert(!PYM || M == 16, "y-parity merge is the Cout = 8 layout of the 16-row MFMA");
    constexpr int NPY = PYM ? 1 : 2;  // accumulator sets along y (merged: both parities live in one set's rows)
    constexpr int XB = 32 / F::NV;
    constexpr int NPZ = KD == 3 ? 2 : 1;
    constexpr int IZ = G::IZ, IY = G::IY, IX = G::IX, IXP = G::IXP, PS = G::PS, GPC = G::GPC;
    constexpr int WROWS = G::WROWS, BUF_F = G::BUF_F;
    static_assert(TZ * TY * NMB == 4, "one (input row, M block) per wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][BUF_F]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane % F::NV, lk = lane / F::NV;
    const int rwave = NMB == 2 ? (wave >> 1) : wave, mb = NMB == 2 ? (wave & 1) : 0;   // the wave's input row and M block
    const int tz = rwave / TY, ty = rwave % TY;
    int bx, by, bz;
    if (!xcd_tile(a.nx, a.ny, a.nz, KD == 3, bx, by, bz)) return;
    const int ix0 = bx * 32, iy0 = by * TY, iz0 = bz * TZ;

    acc_t acc[NPZ][NPY][2][XB];
#pragma unroll
    for (int pz = 0; pz < NPZ; ++pz)
#pragma unroll
        for (int p = 0; p < 2 * NPY; ++p)
#pragma unroll
            for (int xb = 0; xb < XB; ++xb)
#pragma unroll
                for (int r = 0; r < F::ACC; ++r) acc[pz][p >> 1][p & 1][xb][r] = 0.f;

    const int in_vol = a.D * a.H * a.W;
    auto chunk_rsrc = [&](int ci0, int nch) {  // descriptor of the channels [ci0, ci0 + nch) only
        return __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)ci0 * in_vol), (short)0, nch * in_vol * 4, 0x00020000);
    };
    const int nchunks = PREF ? NCH : a.Cin / CI_CH;
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, nchunks * WROWS * 256, 0x00020000);

    // residual prefetch (PREF): group g = (xb, pz, pyi) of the epilogue below, ACC 8-byte loads each
    constexpr unsigned kInvalid = 0x80000000u;
    constexpr int NGRP = XB * NPZ * NPY;
    const int out_plane = a.Ho * a.Wo, out_vol = a.Do * out_plane;
    const bool has_skip = a.skip != nullptr;
    const char* skp = reinterpret_cast<const char*>(has_skip ? a.skip : a.out);  // byte-addressed
    float2_t skv[PREF ? NGRP : 1][F::ACC];
    auto group_pos = [&](int g) -> unsigned {   // byte offset of the group's float2 inside a channel volume, or kInvalid
        const int xb = g / (NPZ * NPY), pz = (g / NPY) % NPZ, pyi = g % NPY;
        const int iz = iz0 + tz, iy = iy0 + ty, ix = ix0 + xb * F::NV + ln;
        const bool ok = iz < a.D && iy < a.H && ix < a.W;
        const int py = PYM ? (lk >> 1) : pyi;  // merged: the lane's rows are all of one y parity
        const int oz = KD == 3 ? 2 * iz + pz : iz;
        return ok ? (unsigned)(oz * out_plane + (2 * iy + py) * a.Wo + 2 * ix) * 4u : kInvalid;
    };
    auto chan_off = [&](int rr) -> unsigned {
        const int co = PYM ? (F::row(rr, lk) & 7) : mb * M + F::row(rr, lk);
        return co < a.Cout ? (unsigned)(co * out_vol) * 4u : kInvalid;
    };
    auto prefetch_group = [&](auto g_t) {
        constexpr int g = decltype(g_t)::value;
        if constexpr (PREF) {
            const unsigned pos = group_pos(g);
#pragma unroll
            for (int rr = 0; rr < F::ACC; ++rr) {
                const unsigned co = chan_off(rr);
                const bool inr = !((pos | co) & kInvalid) && has_skip;
                // unconditional load from a clamped (always valid) address; the select happens at the use
                skv[g][rr] = *reinterpret_cast<const float2_t*>(skp + (inr ? (pos + co) : 0u));
            }
        }
    };
    // every group is issued in iteration 0, after the tile loads of chunk 1
    auto prefetch_at = [&](int c) {
        if constexpr (PREF) {
            asm volatile("" ::: "memory");   // after this chunk's tile / weight loads, in program order
            static_assert(NGRP <= 8, "prefetch groups");
#define DMVS_PF(G_) if constexpr (G_ < NGRP) { if (c == 0) prefetch_group(std::integral_constant<int, G_>{}); }
            DMVS_PF(0) DMVS_PF(1) DMVS_PF(2) DMVS_PF(3) DMVS_PF(4) DMVS_PF(5) DMVS_PF(6) DMVS_PF(7)
#undef DMVS_PF
            asm volatile("" ::: "memory");
        }
    };

    // BatchNorm constants of the lane's channels: loaded FIRST (PREF: a load in the epilogue is an exposed round trip)
    float sc[F::ACC], sh[F::ACC];
    unsigned cooff[F::ACC];
    auto load_bn = [&]() {
#pragma unroll
    for (int rr = 0; rr < F::ACC; ++rr) {
        const int co = PYM ? (F::row(rr, lk) & 7) : mb * M + F::row(rr, lk);
        const bool cok = co < a.Cout;
        const int coc = cok ? co : 0;
        if (M == 32 && a.Cout == 32 && (DMVS_X & 2)) {   // scalar-cache loads + select: dev switch, OFF (see conv_mfma_kernel)
            const int c0 = F::row(rr, 0);
            const float s0 = a.scale ? a.scale[c0] : 1.f, s1 = a.scale ? a.scale[c0 + 4] : 1.f;
            const float h0 = a.scale ? a.shift[c0] : 0.f, h1 = a.scale ? a.shift[c0 + 4] : 0.f;
            sc[rr] = lk ? s1 : s0;
            sh[rr] = lk ? h1 : h0;
        } else {
            sc[rr] = a.scale ? a.scale[coc] : 1.f;
            sh[rr] = a.scale ? a.shift[coc] : 0.f;
        }
        cooff[rr] = cok ? (unsigned)(co * out_vol) * 4u : kInvalid;
    }
    };
    constexpr bool BN_FIRST = PREF || !(DMVS_X & 4);
    if constexpr (BN_FIRST) load_bn();
    if constexpr (PREF) asm volatile("" ::: "memory");
    K3_TR(0, K3_NOW());
    K3_TR_HW();
    [[maybe_unused]] unsigned long long tr_wait = 0, tr_mfma = 0, tr_t = K3_NOW();
    load_tile<CI_CH, IZ, IY, IX, IXP, PS, false>(a.D, a.H, a.W, chunk_rsrc(0, CI_CH), smem, 0, iz0, iy0, ix0, wave, lane);
    load_weights<WROWS>(rs_w, smem + G::TILE_F, 0, wave, lane);
#pragma unroll(PREF ? NCH : 1)
    for (int c = 0; c < (PREF ? NCH : nchunks); ++c) {
        // chunk c has landed.  PREF, c = 1: the 16 residual loads issued in iteration 0 are the NEWEST in the queue and may stay
        // in flight; loads retire in order, so "at most 16 outstanding" means the tile and weight loads are done
        if (PREF && c == 1) {
            static_assert(!PREF || NGRP * F::ACC == 16, "counted wait of the prefetch");
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
#ifdef DMVS_K3_TRACE
        { const unsigned long long n = K3_NOW(); tr_wait += n - tr_t; tr_t = n; if (c == 0) K3_TR(1, n); }
#endif
        float* cur = smem + (c & 1) * BUF_F;
        if (c + 1 < nchunks) {
            float* nxt = smem + ((c + 1) & 1) * BUF_F;
            load_tile<CI_CH, IZ, IY, IX, IXP, PS, false>(a.D, a.H, a.W, chunk_rsrc((c + 1) * CI_CH, CI_CH), nxt, (c + 1) * CI_CH, iz0, iy0, ix0, wave, lane);
            load_weights<WROWS>(rs_w, nxt + G::TILE_F, c + 1, wave, lane);
        }
        prefetch_at(c);
        const float* tile = cur;
        const float* wl = cur + G::TILE_F + lane;
        int step = 0;
#pragma unroll
        for (int oz = 0; oz < NPZ; ++oz)
#pragma unroll
            for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                for (int ox = 0; ox < 2; ++ox)
#pragma unroll
                    for (int g = 0; g < GPC; ++g) {
                        float bv[XB];
#pragma unroll
                        for (int xb = 0; xb < XB; ++xb)
                            bv[xb] = tile[(g * F::KK + lk) * PS + ((tz + oz) * IY + ty + oy) * IXP + xb * F::NV + ln + ox];
#pragma unroll
                        for (int pz = oz; pz < NPZ; ++pz)
#pragma unroll
                            for (int py = PYM ? 0 : oy; py < NPY; ++py)
#pragma unroll
                                for (int px = ox; px < 2; ++px) {
                                    const float av = wl[(step * NMB + mb) * 64];
                                    ++step;
#pragma unroll
                                    for (int xb = 0; xb < XB; ++xb)
                                        acc[pz][py][px][xb] = F::mfma(av, bv[xb], acc[pz][py][px][xb]);
                                }
                    }
#ifdef DMVS_K3_TRACE
        { asm volatile("s_nop 0" ::: "memory"); const unsigned long long n = K3_NOW(); tr_mfma += n - tr_t; tr_t = n; }
#endif
    }
    K3_TR(2, K3_NOW());
    K3_TR(8, tr_wait);
    K3_TR(9, tr_mfma);

    // epilogue (branch-free, see conv_mfma_kernel): the two x-parities of a voxel form one float2
    typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
    const int iz = iz0 + tz, iy = iy0 + ty;
    const bool rok = iz < a.D && iy < a.H;
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.out, (short)0, a.Cout * out_vol * 4, 0x00020000);
    const float lo = a.relu ? 0.f : -INFINITY;
    if constexpr (!BN_FIRST) load_bn();
#pragma unroll
    for (int xb = 0; xb < XB; ++xb) {
        const int ix = ix0 + xb * F::NV + ln;
        const bool ok = rok && ix < a.W;
#pragma unroll
        for (int pz = 0; pz < NPZ; ++pz)
#pragma unroll
            for (int pyi = 0; pyi < NPY; ++pyi) {
                const int py = PYM ? (lk >> 1) : pyi;  // merged: the lane's rows are all of one y parity
                const int oz = KD == 3 ? 2 * iz + pz : iz;
                const unsigned pos = ok ? (unsigned)(oz * out_plane + (2 * iy + py) * a.Wo + 2 * ix) * 4u : kInvalid;
                // residual: unconditional float2 loads from a clamped (always valid) address, zeroed by a select.
                // (ROCm 7.2 clang mis-compiles element extraction from __builtin_amdgcn_raw_buffer_load_b64 into a
                // single-dword load broadcast to both halves, so the 8-byte residual read is a plain global load.)
                float2_t sk[F::ACC];
#pragma unroll
                for (int rr = 0; rr < F::ACC; ++rr) {
                    const bool inr = !((pos | cooff[rr]) & kInvalid) && has_skip;
                    float2_t t;
                    if constexpr (PREF) t = skv[(xb * NPZ + pz) * NPY + pyi][rr];
                    else t = *reinterpret_cast<const float2_t*>(skp + (inr ? (pos + cooff[rr]) : 0u));
                    sk[rr].x = inr ? t.x : 0.f;
                    sk[rr].y = inr ? t.y : 0.f;
                }
#pragma unroll
                for (int rr = 0; rr < F::ACC; ++rr) {
                    const unsigned off = (pos | cooff[rr]) & kInvalid ? kInvalid : pos + cooff[rr];
                    const float vx = fmaxf(acc[pz][pyi][0][xb][rr] * sc[rr] + sh[rr], lo) + sk[rr].x;
                    const float vy = fmaxf(acc[pz][pyi][1][xb][rr] * sc[rr] + sh[rr], lo) + sk[rr].y;
                    v2u_t v;
                    v.x = __builtin_bit_cast(unsigned, vx);
                    v.y = __builtin_bit_cast(unsigned, vy);
                    __builtin_amdgcn_raw_buffer_store_b64(v, rs_out, off, 0, 0);
                }
            }
    }
#ifdef DMVS_K3_TRACE
    K3_TR(3, K3_NOW());   // every store issued (the residual loads they depend on have landed)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    K3_TR(4, K3_NOW());
#endif
}

// ------------------------------------------------------------------------------------------------ configs
// One row per supported (Cin, Cout, mode, kdepth): MFMA shape M, number of M blocks, channel chunk per pipeline
// stage.  The chunk is chosen so that two stages (input tile + weight slice each) fit the 160 KB LDS for every
// tile variant of the layer.  The packer and the launcher both read this table, so the weight stream always
// matches the kernel.
// Channel chunk per pipeline stage, one constant per layer family (used by the table AND the dispatcher).  Small
// chunks = small LDS stages = more workgroups per CU, which is what hides a workgroup's load / store latency
// (conv1: 1 -> 2 workgroups per CU, 0.29 -> 0.23 ms); 2 channels on the 16-row MFMA means packed-K (2 taps x 2).
// Layers that already hold >= 2 workgroups per CU with 4-channel chunks gain nothing from 2 (measured per layer).
#ifndef DMVS_CONV1_CI
#define DMVS_CONV1_CI 1
#endif
constexpr int CI_CONV1 = DMVS_CONV1_CI;   // 1: (4 taps x 1 channel) per k-group, a 14 KB LDS stage: 8 workgroups per CU (r04: conv1 -6 % vs 2 = (2 taps x 2 channels), 28 KB stages)
#include "dev_guard.h"   // after the default of the LAST development switch of this file
constexpr int CI_CONV2 = 4, CI_CONV4 = 2, CI_CONV6 = 4, CI_CONV6_2D = 4;
constexpr int CI_F00 = 2, CI_F01 = 4, CI_F1 = 4, CI_F2 = 4, CI_FO3 = 4, CI_K5A = 2, CI_K5B = 2;
constexpr int CI_K1 = 4;
// transposed convs conv9 / conv11: 4-channel chunks halve the LDS stage (co-residency of the two branch streams:
// regularisation 7.88 -> 7.62 ms); conv7 keeps 8
constexpr int DCI11 = 4, DCI9 = 4;
struct Cfg { int cin, cout, mode, kd, M, MB, ci_ch, pym; };  // pym: y-parity-merged deconv (Cout = 8)
const Cfg kCfgs[] = {
    {2, 16, DMVS_CONV_S1, 3, 16, 1, 2},     // conv0 of both branches fused (2 -> 8+8), packed-K  module.py:361
    {8, 16, DMVS_CONV_S2, 3, 16, 1, CI_CONV1},     // conv1   module.py:363 (packed-K)
    {16, 16, DMVS_CONV_S1, 3, 16, 1, CI_CONV2},    // conv2   module.py:364
    {16, 32, DMVS_CONV_S2, 3, 32, 1, 2},    // conv3   module.py:366
    {32, 32, DMVS_CONV_S1, 3, 32, 1, CI_CONV4},    // conv4   module.py:367
    {32, 64, DMVS_CONV_S2, 3, 32, 2, 2},    // conv5   module.py:369
    {64, 64, DMVS_CONV_S1, 3, 32, 2, CI_CONV6},    // conv6   module.py:370
    {64, 32, DMVS_DECONV_S2, 3, 16, 2, 8},  // conv7   module.py:372 (two 16-channel blocks split over the wave pairs)
    {32, 16, DMVS_DECONV_S2, 3, 16, 1, DCI9},  // conv9   module.py:374
    {16, 8, DMVS_DECONV_S2, 3, 16, 1, DCI11, 1},// conv11  module.py:376 (rows 0-7 / 8-15 = the two y parities)
    {32, 64, DMVS_CONV_S2, 1, 32, 2, 2},    // refine conv5 (2D)  module.py:411
    {64, 64, DMVS_CONV_S1, 1, 32, 2, CI_CONV6_2D},    // refine conv6 (2D)  module.py:412
    {64, 32, DMVS_DECONV_S2, 1, 16, 2, 8},  // refine conv7 (2D)  module.py:414
    // FeatureNet (module.py:283-311) on [C][V][H][W]: the V views are kdepth = 1 slices
    {4, 8, DMVS_CONV_S1, 1, 16, 1, CI_F00},      // conv0.0 (RGB + one zero channel)
    {8, 8, DMVS_CONV_S1, 1, 16, 1, CI_F01},      // conv0.1
    {8, 16, DMVS_CONV2D_K5S2, 1, 16, 1, CI_K5A}, // conv1.0
    {16, 16, DMVS_CONV_S1, 1, 16, 1, CI_F1},    // conv1.1, conv1.2
    {16, 32, DMVS_CONV2D_K5S2, 1, 32, 1, CI_K5B},// conv2.0
    {32, 32, DMVS_CONV_S1, 1, 32, 1, CI_F2},    // conv2.1, conv2.2, out2
    {32, 16, DMVS_CONV_S1, 1, 16, 1, CI_FO3},    // out3
    {32, 64, DMVS_CONV2D_K1, 1, 32, 2, CI_K1},  // out1
    {16, 32, DMVS_CONV2D_K1, 1, 32, 1, CI_K1},  // inner1
    {8, 32, DMVS_CONV2D_K1, 1, 32, 1, CI_K1},   // inner2
};

int taps_of(int mode, int kdepth) {
    return mode == DMVS_CONV2D_K5S2 ? 25 : mode == DMVS_CONV2D_K1 ? 1 : 9 * kdepth;
}

const Cfg* find_cfg(int cin, int cout, int mode, int kdepth) {
    for (const Cfg& c : kCfgs)
        if (c.cin == cin && c.cout == cout && c.mode == mode && c.kd == kdepth) return &c;
    return nullptr;
}

int tap_of(int p, int o) { return p == 0 ? 1 : (o == 0 ? 2 : 0); }

// Pick the tile by how many workgroups it yields: big tiles amortise the halo and the weight slice, but the
// low-resolution layers (1/4, 1/8 scale) would leave most of the 256 CUs idle with them.
#define kMinBlocks g_min_blocks

template <typename K>
int launch_with_lds(K kernel, dim3 tiles, size_t lds_bytes, ConvArgs a, hipStream_t st) {
    a.nx = tiles.x; a.ny = tiles.y; a.nz = tiles.z;
    a.st4 = a.Wo % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
    const dim3 grid(xcd_grid(tiles.x * tiles.y * tiles.z));
    if (int e = dmvs_ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds_bytes)) return e;
    kernel<<<grid, 256, lds_bytes, st>>>(a);
    DMVS_LAUNCH_CHECK();
}

template <int M, int MB, int STRIDE, int KD, int KS, int CI_CH, int TZ, int TY, bool V4, bool MBS = false>
int launch_conv_tile_v(const ConvArgs& a, hipStream_t st) {
    typedef ConvGeom<M, STRIDE, KD, KS, CI_CH, TZ, TY, V4> G;
    constexpr int ROWS = TZ * TY / (MBS ? 2 : 4);
    constexpr size_t lds2 = 2 * (size_t)(G::TILE_F + G::NSTEPS * MB * 64) * sizeof(float);
    static_assert(lds2 <= 160 * 1024, "two pipeline stages must fit the 160 KB LDS");
    // The 3D (regularisation) layers run with ONE LDS stage: the load of chunk c+1 is no longer overlapped with the
    // MFMAs of chunk c inside a workgroup, but the halved footprint doubles the workgroups per CU and lets the
    // kernels of the two branch streams share a CU -- measured: conv1 0.215 -> 0.184 ms alone, the regularisation
    // 8.34 -> 8.05 ms per depth map.  The 2D (FeatureNet) layers are slightly faster double buffered.
    // (keeping two stages for the small grids of the 1/4- and 1/8-scale layers, which have nobody to share a CU with,
    // measured neutral to slightly negative: g_single_buf_min_blocks.)
    ConvArgs b = a;
    dim3 grid(ceil_div(a.Wo, 32), ceil_div(a.Ho, TY), ceil_div(a.Do, TZ));
    b.single_buf = (KD == 3 && (long)grid.x * grid.y * grid.z >= g_single_buf_min_blocks) ? 1 : 0;
    return launch_with_lds(conv_mfma_kernel<M, MB, STRIDE, KD, KS, CI_CH, TZ, TY, ROWS, V4, MBS>, grid, b.single_buf ? lds2 / 2 : lds2, b, st);
}

// 16-byte tile loads need whole pieces inside a row and aligned rows
inline bool can_v4(const ConvArgs& a) { return a.W % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0; }

template <int M, int MB, int STRIDE, int KD, int KS, int CI_CH, int TZ, int TY, bool MBS = false>
int launch_conv_tile(const ConvArgs& a, hipStream_t st) {
    return can_v4(a) ? launch_conv_tile_v<M, MB, STRIDE, KD, KS, CI_CH, TZ, TY, true, MBS>(a, st)
                     : launch_conv_tile_v<M, MB, STRIDE, KD, KS, CI_CH, TZ, TY, false, MBS>(a, st);
}

// Tile choice of a conv layer, one source of truth for the launcher and dmvs_conv3d_mfma_plan: returns TZ * 256 + TY,
// bit 17 set for the M-block-split variant.  Flat layers (kdepth 1, or a 3D layer whose output has depth 1) use TZ = 1.
#define kSplitBlocks g_split_blocks   // two-block layers with fewer small-tile workgroups than this split their M blocks
constexpr int kPlanMBS = 1 << 17;
inline int conv_tile_choice(int stride, int kd, int Do, int Ho, int Wo, int MB) {
    const bool flat = (kd == 1) || Do == 1;
    const int big_ty_flat = (stride == 1) ? 16 : 8, big_ty = (stride == 1) ? 8 : 4;
    const long big_blocks = flat ? (long)ceil_div(Wo, 32) * ceil_div(Ho, big_ty_flat) * Do
                                 : (long)ceil_div(Wo, 32) * ceil_div(Ho, big_ty) * ceil_div(Do, 2);
    const bool big = big_blocks >= kMinBlocks;
    if (!big && MB == 2) {
        const long small_blocks = flat ? (long)ceil_div(Wo, 32) * ceil_div(Ho, 4) * Do
                                       : (long)ceil_div(Wo, 32) * ceil_div(Ho, 2) * ceil_div(Do, 2);
        if (small_blocks < kSplitBlocks) return kPlanMBS | (flat ? 256 + 2 : 2 * 256 + 1);
    }
    if (flat) return 256 + (big ? big_ty_flat : 4);
    return 2 * 256 + (big ? big_ty : 2);
}

template <int M, int MB, int STRIDE, int KD, int CI_CH, int KS = 3>
int launch_conv(const ConvArgs& a, hipStream_t st) {
    constexpr int BIG_TY_FLAT = (STRIDE == 1) ? 16 : 8, BIG_TY = (STRIDE == 1) ? 8 : 4;
    const int choice = conv_tile_choice(STRIDE, KD, a.Do, a.Ho, a.Wo, MB);
    const int tz = (choice >> 8) & 255, ty = choice & 255;
    if constexpr (MB == 2) {
        if (choice & kPlanMBS) {
            if (tz == 1) return launch_conv_tile<M, MB, STRIDE, KD, KS, CI_CH, 1, 2, true>(a, st);
            if (KD == 3) return launch_conv_tile<M, MB, STRIDE, 3, 3, CI_CH, 2, 1, true>(a, st);
            return DMVS_EUNSUPPORTED;
        }
    }
    if (tz == 1) {
        if (ty == BIG_TY_FLAT) return launch_conv_tile<M, MB, STRIDE, KD, KS, CI_CH, 1, BIG_TY_FLAT>(a, st);
        return launch_conv_tile<M, MB, STRIDE, KD, KS, CI_CH, 1, 4>(a, st);
    }
    if (KD == 3) {
        if (ty == BIG_TY) return launch_conv_tile<M, MB, STRIDE, 3, 3, CI_CH, 2, BIG_TY>(a, st);
        return launch_conv_tile<M, MB, STRIDE, 3, 3, CI_CH, 2, 2>(a, st);
    }
    return DMVS_EUNSUPPORTED;
}

template <int M, int KD, int CI_CH, int TZ, int TY, bool PYM, int NMB = 1>
int launch_deconv_tile(const ConvArgs& a, hipStream_t st) {
    typedef DeconvGeom<M, KD, CI_CH, TZ, TY, PYM, NMB> G;
    constexpr size_t lds = 2 * (size_t)G::BUF_F * sizeof(float);
    static_assert(lds <= 160 * 1024, "two pipeline stages must fit the 160 KB LDS");
    dim3 grid(ceil_div(a.W, 32), ceil_div(a.H, TY), ceil_div(a.D, TZ));
    // residual prefetch: conv11; conv9's 64 and conv7's 128 residual registers spill under the occupancy bound
    if constexpr (PYM)   // conv11 (16 -> 8): 4 chunks, 4 residual groups of 4 loads = 32 registers
        if (a.skip && g_deconv_prefetch && a.Cin == 4 * CI_CH)
            return launch_with_lds(deconv_mfma_kernel<M, KD, CI_CH, TZ, TY, PYM, true, 4>, grid, lds, a, st);
    return launch_with_lds(deconv_mfma_kernel<M, KD, CI_CH, TZ, TY, PYM, false, 0, NMB>, grid, lds, a, st);
}

template <int M, int KD, int CI_CH, bool PYM = false, int NMB = 1>
int launch_deconv(const ConvArgs& a, hipStream_t st) {
    if constexpr (NMB == 2) {   // two input rows per workgroup
        if (KD == 1 || a.D == 1) return launch_deconv_tile<M, KD, CI_CH, 1, 2, PYM, 2>(a, st);
        return launch_deconv_tile<M, 3, CI_CH, 2, 1, PYM, 2>(a, st);
    } else {
        if (KD == 1 || a.D == 1) return launch_deconv_tile<M, KD, CI_CH, 1, 4, PYM>(a, st);
        return launch_deconv_tile<M, 3, CI_CH, 2, 2, PYM>(a, st);
    }
}

}  // namespace

extern "C" long dmvs_conv3d_mfma_weight_floats(int Cin, int Cout, int mode, int kdepth) {
    const Cfg* c = find_cfg(Cin, Cout, mode, kdepth);
    if (!c) return 0;
    const int KK = c->M == 32 ? 2 : 4, nt = taps_of(mode, kdepth);
    if (c->ci_ch < KK) {  // packed-K: TPG taps per k-step
        const int tpg = KK / c->ci_ch;
        return (long)(Cin / c->ci_ch) * ((nt + tpg - 1) / tpg) * c->MB * 64;
    }
    if (c->pym) return (long)(nt / 3 * 2) * (Cin / KK) * 64;  // 18 (kd 3) or 6 (kd 1) k-steps per group
    return (long)nt * (Cin / KK) * c->MB * 64;
}

extern "C" int dmvs_pack_conv_weights_mfma(const float* w, float* out, int Cin, int Cout, int mode, int kdepth) {
    const Cfg* c = find_cfg(Cin, Cout, mode, kdepth);
    if (!c || !w || !out) return DMVS_EUNSUPPORTED;
    const int M = c->M, KK = (M == 32 ? 2 : 4), GPC = c->ci_ch / KK, NT = taps_of(mode, kdepth);
    size_t n = 0;
    for (int ci0 = 0; ci0 < Cin; ci0 += c->ci_ch) {
        if (mode != DMVS_DECONV_S2 && c->ci_ch < KK) {
            // packed-K conv: k-step st covers taps st*TPG .. st*TPG+TPG-1, lane k = tap_select*ci_ch + ci
            const int tpg = KK / c->ci_ch, nsteps = (NT + tpg - 1) / tpg;
            for (int st = 0; st < nsteps; ++st)
                for (int mb = 0; mb < c->MB; ++mb)
                    for (int l = 0; l < 64; ++l) {
                        const int co = mb * M + l % M, k = l / M, ci = ci0 + k % c->ci_ch, t = st * tpg + k / c->ci_ch;
                        out[n++] = (co < Cout && t < NT) ? w[((size_t)co * Cin + ci) * NT + t] : 0.f;
                    }
        } else if (mode != DMVS_DECONV_S2) {
            // conv weight [Cout][Cin][kd][3][3]; order: chunk, tap, k-group, M block, lane
            for (int t = 0; t < NT; ++t)
                for (int g = 0; g < GPC; ++g)
                    for (int mb = 0; mb < c->MB; ++mb)
                        for (int l = 0; l < 64; ++l) {
                            const int co = mb * M + l % M, ci = ci0 + g * KK + l / M;
                            out[n++] = co < Cout ? w[((size_t)co * Cin + ci) * NT + t] : 0.f;
                        }
        } else {
            // ConvTranspose weight [Cin][Cout][kd][3][3]; order: chunk, input offset, k-group, valid parities, M block, lane
            const int npz = kdepth == 3 ? 2 : 1;
            for (int oz = 0; oz < npz; ++oz)
                for (int oy = 0; oy < 2; ++oy)
                    for (int ox = 0; ox < 2; ++ox)
                        for (int g = 0; g < GPC; ++g)
                            for (int pz = oz; pz < npz; ++pz)
                                for (int py = c->pym ? 1 : oy; py < 2; ++py)
                                    for (int px = ox; px < 2; ++px)
                                      for (int mb = 0; mb < c->MB; ++mb) {
                                        const int kz = kdepth == 3 ? tap_of(pz, oz) : 0;
                                        for (int l = 0; l < 64; ++l) {
                                            // merged: row r of the fragment is channel r % 8 of y parity r / 8
                                            const int row = l % M, co = c->pym ? row % 8 : mb * M + row, ci = ci0 + g * KK + l / M;
                                            const int pyl = c->pym ? row / 8 : py;
                                            const int t = (kz * 3 + tap_of(pyl, oy)) * 3 + tap_of(px, ox);
                                            out[n++] = (co < Cout && pyl >= oy) ? w[((size_t)ci * Cout + co) * NT + t] : 0.f;
                                        }
                                    }
        }
    }
    return n == (size_t)dmvs_conv3d_mfma_weight_floats(Cin, Cout, mode, kdepth) ? 0 : DMVS_EINVAL;
}

extern "C" int dmvs_conv3d_mfma(const float* in, float* out, const float* w_packed, const float* scale,
                                const float* shift, const float* skip, int Cin, int Cout, int D, int H, int W,
                                int mode, int kdepth, int flags, dmvs_stream_t stream) {
    if (!in || !out || !w_packed || D < 1 || H < 1 || W < 1) return DMVS_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return DMVS_EINVAL;
    if (flags & ~(DMVS_RELU | DMVS_SKIP_UP2 | DMVS_OUT_Q4 | DMVS_IN_VIEWS)) return DMVS_EUNSUPPORTED;   // incl. the retired bit 4
    if ((flags & DMVS_IN_VIEWS) && !(Cin == 4 && kdepth == 1 && mode == DMVS_CONV_S1 && (long)3 * D * H * W < (1L << 29))) return DMVS_EUNSUPPORTED;
    const Cfg* c = find_cfg(Cin, Cout, mode, kdepth);
    if (!c) return DMVS_EUNSUPPORTED;
    if ((long)c->ci_ch * D * H * W >= (1L << 28)) return DMVS_EINVAL;  // one channel chunk < 1 GB (descriptor offsets)
    ConvArgs a;
    a.in = in; a.out = out; a.w = w_packed; a.scale = scale; a.shift = shift; a.skip = skip;
    a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W; a.relu = (flags & DMVS_RELU) ? 1 : 0;
    a.skip_up2 = (flags & DMVS_SKIP_UP2) ? 1 : 0;
    if (a.skip_up2 && (!skip || mode == DMVS_DECONV_S2)) return DMVS_EINVAL;
    a.outq4 = (flags & DMVS_OUT_Q4) ? 1 : 0;
    if (a.outq4 && (mode == DMVS_DECONV_S2 || skip || (Cout & 7))) return DMVS_EINVAL;
    a.in_cs = a.in_zs = a.in_elems = 0;
    if (flags & DMVS_IN_VIEWS) { a.in_cs = H * W; a.in_zs = 3 * H * W; a.in_elems = 3 * D * H * W; }
    hipStream_t st = (hipStream_t)stream;
    const bool k3 = kdepth == 3;
    {   // output < 2 GB (the epilogue's range-checked byte offsets)
        const long vox = (mode == DMVS_CONV_S1 || mode == DMVS_CONV2D_K1) ? (long)D * H * W
                       : mode == DMVS_CONV2D_K5S2 ? (long)D * ((H + 1) / 2) * ((W + 1) / 2)
                       : mode == DMVS_CONV_S2 ? (long)(k3 ? (D + 1) / 2 : D) * ((H + 1) / 2) * ((W + 1) / 2)
                                              : (long)(k3 ? 2 * D : D) * 2 * H * 2 * W;
        if (Cout * vox >= (1L << 29)) return DMVS_EINVAL;
    }
    if (mode == DMVS_CONV_S1) {
        a.Do = D; a.Ho = H; a.Wo = W;
        if (Cin == 2 && Cout == 16 && k3) return launch_conv<16, 1, 1, 3, 2>(a, st);
        if (Cin == 16 && Cout == 16 && k3) return launch_conv<16, 1, 1, 3, CI_CONV2>(a, st);
        if (Cin == 32 && Cout == 32 && k3) return launch_conv<32, 1, 1, 3, CI_CONV4>(a, st);
        if (Cin == 64 && Cout == 64) return k3 ? launch_conv<32, 2, 1, 3, CI_CONV6>(a, st) : launch_conv<32, 2, 1, 1, CI_CONV6_2D>(a, st);
        if (!k3) {  // FeatureNet 3x3 layers
            if (Cin == 4 && Cout == 8) return launch_conv<16, 1, 1, 1, CI_F00>(a, st);
            if (Cin == 8 && Cout == 8) return launch_conv<16, 1, 1, 1, CI_F01>(a, st);
            if (Cin == 16 && Cout == 16) return launch_conv<16, 1, 1, 1, CI_F1>(a, st);
            if (Cin == 32 && Cout == 32) return launch_conv<32, 1, 1, 1, CI_F2>(a, st);
            if (Cin == 32 && Cout == 16) return launch_conv<16, 1, 1, 1, CI_FO3>(a, st);
        }
    } else if (mode == DMVS_CONV2D_K1 && !k3) {
        a.Do = D; a.Ho = H; a.Wo = W;
        if (Cin == 32 && Cout == 64) return launch_conv<32, 2, 1, 1, CI_K1, 1>(a, st);
        if (Cin == 16 && Cout == 32) return launch_conv<32, 1, 1, 1, CI_K1, 1>(a, st);
        if (Cin == 8 && Cout == 32) return launch_conv<32, 1, 1, 1, CI_K1, 1>(a, st);
    } else if (mode == DMVS_CONV2D_K5S2 && !k3) {
        a.Do = D; a.Ho = (H + 1) / 2; a.Wo = (W + 1) / 2;
        if (Cin == 8 && Cout == 16) return launch_conv<16, 1, 2, 1, CI_K5A, 5>(a, st);
        if (Cin == 16 && Cout == 32) return launch_conv<32, 1, 2, 1, CI_K5B, 5>(a, st);
    } else if (mode == DMVS_CONV_S2) {
        a.Do = k3 ? (D + 1) / 2 : D; a.Ho = (H + 1) / 2; a.Wo = (W + 1) / 2;
        if (Cin == 8 && Cout == 16 && k3) return launch_conv<16, 1, 2, 3, CI_CONV1>(a, st);
        if (Cin == 16 && Cout == 32 && k3) return launch_conv<32, 1, 2, 3, 2>(a, st);
        if (Cin == 32 && Cout == 64) return k3 ? launch_conv<32, 2, 2, 3, 2>(a, st) : launch_conv<32, 2, 2, 1, 2>(a, st);
    } else if (mode == DMVS_DECONV_S2) {
        a.Do = k3 ? 2 * D : D; a.Ho = 2 * H; a.Wo = 2 * W;
        if (Cin == 64 && Cout == 32) return k3 ? launch_deconv<16, 3, 8, false, 2>(a, st) : launch_deconv<16, 1, 8, false, 2>(a, st);
        if (Cin == 32 && Cout == 16 && k3) return launch_deconv<16, 3, DCI9>(a, st);
        if (Cin == 16 && Cout == 8 && k3) return launch_deconv<16, 3, DCI11, true>(a, st);
    }
    return DMVS_EUNSUPPORTED;
}

extern "C" int dmvs_conv3d_mfma_plan(int Cin, int Cout, int D, int H, int W, int mode, int kdepth) {
    const Cfg* c = find_cfg(Cin, Cout, mode, kdepth);
    if (!c || D < 1 || H < 1 || W < 1) return DMVS_EUNSUPPORTED;
    const bool k3 = kdepth == 3;
    if (mode == DMVS_DECONV_S2) {   // launch_deconv
        const bool flat = kdepth == 1 || D == 1;
        return c->MB == 2 ? (flat ? 256 + 2 : 2 * 256 + 1) : (flat ? 256 + 4 : 2 * 256 + 2);
    }
    const int stride = (mode == DMVS_CONV_S2 || mode == DMVS_CONV2D_K5S2) ? 2 : 1;
    const int Do = (mode == DMVS_CONV_S2 && k3) ? (D + 1) / 2 : D;
    const int Ho = stride == 2 ? (H + 1) / 2 : H, Wo = stride == 2 ? (W + 1) / 2 : W;
    return conv_tile_choice(stride, kdepth, Do, Ho, Wo, c->MB) | ((W % 4 == 0) ? 0x10000 : 0);
}

