// K4: dual-depth regression -- softmax over D, depth expectation, (small,huge) min/max, checkerboard
// selection and photometric confidence in one pass.
//
// Replaces DepthNet.forward (/root/reference/networks/mvsnet.py:15-66, mode 0) and DepthNet.refine
// (mvsnet.py:67-100, mode 1) plus depth_regression (module.py:454-460): ~60 elementwise launches and a
// materialised [4][D][H][W] softmax volume in the reference.  HBM-bound: reads 4*D + D floats per pixel
// once, writes 6..9 floats per pixel (plus the optional softmax volume, which eval never reads).
//
// One thread per pixel, x fastest (all plane reads/writes coalesced).  Softmax is the usual
// max-subtracted form; the max is found in a first sweep over the 4*D logits of the pixel and the
// exponentials in a second sweep (second read hits L2: the 4*D*256 B working set of a wave is tiny).
#include "common.h"

// DREG > 0: D == DREG is a compile-time constant (the 4-plane refine passes, the 8-plane stage-3 pass): the 4*D logits
// of a pixel stay in registers, each is read and exponentiated ONCE; same operations in the same order as the
// three-sweep form, so the results are bit-identical.  D = 32 / 64 take the channel-split kernel below.
// The part of DepthNet.forward / .refine behind the four expectations (mvsnet.py:22-61, 72-97): population spread -> confidence,
// (small, huge) min / max pairs, checkerboard selection.  One copy for the three kernels below.
__device__ __forceinline__ void regress_tail(const float (&e4)[4], float interval, int mode, int x, int y, size_t plane, size_t pix,
                                             float* __restrict__ sel, float* __restrict__ conf) {
    // population std of the four depths (var(1, unbiased=False).sqrt(), mvsnet.py:61,96)
    const float mean = (e4[0] + e4[1] + e4[2] + e4[3]) / 4.0f;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) var += (e4[c] - mean) * (e4[c] - mean);
    var /= 4.0f;
    const float z = interval / (sqrtf(var) + 1e-5f);
    conf[pix] = 2.0f * (1.0f / (1.0f + expf(-z)) - 0.5f);

    const float sm = fminf(e4[0], e4[1]), sM = fmaxf(e4[0], e4[1]);
    const float hm = fminf(e4[2], e4[3]), hM = fmaxf(e4[2], e4[3]);
    if (mode == 1) {
        // (row%2, col%2): (0,0) small_min, (0,1) small_max, (1,0) huge_max, (1,1) huge_min  mvsnet.py:88-91
        const int r = y & 1, c = x & 1;
        sel[pix] = r == 0 ? (c == 0 ? sm : sM) : (c == 0 ? hM : hm);
        return;
    }
    // mode 0: four refine hypotheses, mvsnet.py:27-56
    const int q = y & 3;
    float lo = (q & 1) ? hm : sm, hi = (q & 1) ? hM : sM;
    if (q >= 2) { const float l2 = 2.f * lo - hi, h2 = 2.f * hi - lo; lo = l2; hi = h2; }  // *_d variants
    // six-stack (3m-2M, 2m-M, m, M, 2M-m, 3M-2m); window [0:4] or [2:6]
    const float st[6] = {3.f * lo - 2.f * hi, 2.f * lo - hi, lo, hi, 2.f * hi - lo, 3.f * hi - 2.f * lo};
    const int off = ((y + x) & 1) ? 2 : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) sel[k * plane + pix] = off ? st[k + 2] : st[k];
}

// K4's remainder when the expectations come from dmvs_prob_regress (the `prob` heads regress their own two channels): reads the
// [4][H][W] expectations, writes the selection and the confidence
__global__ __launch_bounds__(256) void depth_select_kernel(const float* __restrict__ dsp, const float* __restrict__ interval_p,
                                                           int mode, int H, int W, float* __restrict__ sel, float* __restrict__ conf) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const size_t plane = (size_t)H * W, pix = (size_t)y * W + x;
    const float e4[4] = {dsp[pix], dsp[plane + pix], dsp[2 * plane + pix], dsp[3 * plane + pix]};
    regress_tail(e4, interval_p[0], mode, x, y, plane, pix, sel, conf);
}

template <bool WRITE_PROB, int DREG>
__global__ __launch_bounds__(256) void depth_regress_kernel(const float* __restrict__ logits,
                                                            const float* __restrict__ depth,
                                                            const float* __restrict__ interval_p, float alpha,
                                                            int mode, int D, int H, int W, float* __restrict__ dsp,
                                                            float* __restrict__ sel, float* __restrict__ conf,
                                                            float* __restrict__ prob, const float* __restrict__ base) {
    // base != NULL: affine hypotheses, plane d of a pixel = base[pix] + d * interval (see hyp_plane in warp_corr.hip)
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const size_t plane = (size_t)H * W;
    const size_t pix = (size_t)y * W + x;
    const size_t cstride = (size_t)D * plane;

    float e4[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (DREG > 0) {
        float v[4][DREG];
#pragma unroll
        for (int d = 0; d < DREG; ++d)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c][d] = logits[c * cstride + d * plane + pix] * alpha;
        float dep[DREG];
#pragma unroll
        for (int d = 0; d < DREG; ++d) dep[d] = base ? base[pix] + (float)d * interval_p[0] : depth[d * plane + pix];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float m = -INFINITY, s = 0.f;
#pragma unroll
            for (int d = 0; d < DREG; ++d) m = fmaxf(m, v[c][d]);
#pragma unroll
            for (int d = 0; d < DREG; ++d) { v[c][d] = expf(v[c][d] - m); s += v[c][d]; }
#pragma unroll
            for (int d = 0; d < DREG; ++d) {
                const float p = v[c][d] / s;
                if (WRITE_PROB) prob[c * cstride + d * plane + pix] = p;
                e4[c] += p * dep[d];
            }
        }
    } else {
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int c = 0; c < 4; ++c) m[c] = fmaxf(m[c], logits[c * cstride + d * plane + pix] * alpha);
    }
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] += expf(logits[c * cstride + d * plane + pix] * alpha - m[c]);
    }
    // expectation: sum_d softmax * depth  (p = e / s rounded first, as softmax then mul then sum)
    for (int d = 0; d < D; ++d) {
        const float dep = base ? base[pix] + (float)d * interval_p[0] : depth[d * plane + pix];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float p = expf(logits[c * cstride + d * plane + pix] * alpha - m[c]) / s[c];
            if (WRITE_PROB) prob[c * cstride + d * plane + pix] = p;
            e4[c] += p * dep;
        }
    }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) dsp[c * plane + pix] = e4[c];
    regress_tail(e4, interval_p[0], mode, x, y, plane, pix, sel, conf);
}

// Channel-split form for the large-D main passes (D = 32 / 64: stage 2 / stage 1).  One thread per pixel keeps 4 * D
// logits in 210-256 registers (1-2 waves per SIMD) and stage 1 has only 118 k pixels = 1.8 workgroups per CU: the
// kernel was latency-bound at 1.1-2.3 TB/s.  Here a workgroup owns 64 pixels and wave c regresses channel c of them
// (D logits in registers: 8 waves per SIMD, 4x the workgroups, every load still a 256-byte run); the four depth
// estimates of a pixel meet in LDS and wave 0 does the pixel's selection / confidence.  Same operations per channel in
// the same order as the one-thread form: bit-identical results.  0.39 -> 0.28 ms per depth map (0.36 -> 0.51 of 8 TB/s);
// for D = 8 / 4 (1.9 M pixels: grid and registers are no issue there) the split form measured the same as one thread.
template <int DREG>
__global__ __launch_bounds__(256) void depth_regress_split_kernel(const float* __restrict__ logits,
                                                                  const float* __restrict__ depth,
                                                                  const float* __restrict__ interval_p, float alpha,
                                                                  int mode, int H, int W, float* __restrict__ dsp,
                                                                  float* __restrict__ sel, float* __restrict__ conf,
                                                                  const float* __restrict__ base) {
    __shared__ float e_lds[4][64];
    const int lane = threadIdx.x & 63, c = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane, y = blockIdx.y;
    const bool live = x < W;
    const int xc = live ? x : W - 1;
    const size_t plane = (size_t)H * W;
    const size_t pix = (size_t)y * W + xc;
    const size_t cstride = (size_t)DREG * plane;
    float v[DREG];
#pragma unroll
    for (int d = 0; d < DREG; ++d) v[d] = logits[c * cstride + d * plane + pix] * alpha;
    float m = -INFINITY, s = 0.f, e = 0.f;
#pragma unroll
    for (int d = 0; d < DREG; ++d) m = fmaxf(m, v[d]);
#pragma unroll
    for (int d = 0; d < DREG; ++d) { v[d] = expf(v[d] - m); s += v[d]; }
#pragma unroll
    for (int d = 0; d < DREG; ++d) {
        const float dep = base ? base[pix] + (float)d * interval_p[0] : depth[d * plane + pix];
        e += (v[d] / s) * dep;
    }
    if (live) dsp[c * plane + pix] = e;
    e_lds[c][lane] = e;
    __syncthreads();
    if (c != 0 || !live) return;
    const float e4[4] = {e_lds[0][lane], e_lds[1][lane], e_lds[2][lane], e_lds[3][lane]};
    regress_tail(e4, interval_p[0], mode, x, y, plane, pix, sel, conf);
}

static int depth_regress_entry(const float* logits, const float* depth, const float* base, const float* interval, float alpha,
                               int mode, int D, int H, int W, float* dsp, float* sel, float* conf, float* prob,
                               dmvs_stream_t stream) {
    if (!logits || (!depth && !base) || !interval || !dsp || !sel || !conf) return DMVS_EINVAL;
    if (D < 1 || H < 1 || W < 1 || (mode != 0 && mode != 1)) return DMVS_EINVAL;
    dim3 grid(ceil_div(W, 256), H);
    hipStream_t st = (hipStream_t)stream;
    if (prob)
        depth_regress_kernel<true, 0><<<grid, 256, 0, st>>>(logits, depth, interval, alpha, mode, D, H, W, dsp, sel, conf, prob, base);
    else if (D == 4)
        depth_regress_kernel<false, 4><<<grid, 256, 0, st>>>(logits, depth, interval, alpha, mode, D, H, W, dsp, sel, conf, nullptr, base);
    else if (D == 8)
        depth_regress_kernel<false, 8><<<grid, 256, 0, st>>>(logits, depth, interval, alpha, mode, D, H, W, dsp, sel, conf, nullptr, base);
    else if (D == 32)
        depth_regress_split_kernel<32><<<dim3(ceil_div(W, 64), H), 256, 0, st>>>(logits, depth, interval, alpha, mode, H, W, dsp, sel, conf, base);
    else if (D == 64)
        depth_regress_split_kernel<64><<<dim3(ceil_div(W, 64), H), 256, 0, st>>>(logits, depth, interval, alpha, mode, H, W, dsp, sel, conf, base);
    else
        depth_regress_kernel<false, 0><<<grid, 256, 0, st>>>(logits, depth, interval, alpha, mode, D, H, W, dsp, sel, conf, nullptr, base);
    DMVS_LAUNCH_CHECK();
}

extern "C" int dmvs_depth_regress(const float* logits, const float* depth, const float* interval, float alpha,
                                  int mode, int D, int H, int W, float* dsp, float* sel, float* conf, float* prob,
                                  dmvs_stream_t stream) {
    if (!depth) return DMVS_EINVAL;
    return depth_regress_entry(logits, depth, nullptr, interval, alpha, mode, D, H, W, dsp, sel, conf, prob, stream);
}

extern "C" int dmvs_depth_regress_affine(const float* logits, const float* base_hw, const float* interval, float alpha,
                                         int mode, int D, int H, int W, float* dsp, float* sel, float* conf, float* prob,
                                         dmvs_stream_t stream) {
    if (!base_hw) return DMVS_EINVAL;
    return depth_regress_entry(logits, nullptr, base_hw, interval, alpha, mode, D, H, W, dsp, sel, conf, prob, stream);
}

extern "C" int dmvs_depth_select(const float* dsp_4hw, const float* interval, int mode, int H, int W, float* sel, float* conf,
                                 dmvs_stream_t stream) {
    if (!dsp_4hw || !interval || !sel || !conf || H < 1 || W < 1 || (mode != 0 && mode != 1)) return DMVS_EINVAL;
    depth_select_kernel<<<dim3(ceil_div(W, 256), H), 256, 0, (hipStream_t)stream>>>(dsp_4hw, interval, mode, H, W, sel, conf);
    DMVS_LAUNCH_CHECK();
}
