// Layout glue, relative projections and hypothesis-plane sampling.
// None of these is hot (a few MB..100 MB, pure streaming); they exist so that the whole stage loop
// stays on the device with no host synchronisation (the reference never syncs inside
// MVSNet.forward) and no ~25-launch elementwise chains (module.py:556-649).
#include "common.h"

#include <cstring>

// ------------------------------------------------------------------ NCHW slice -> HWC
// 256 pixels per block.  Reads are coalesced per channel plane, the transpose goes through LDS
// (row pad 257 -> conflict-free column reads), writes are one contiguous 256*C-float run.
template <int C>
__global__ __launch_bounds__(256) void nchw_to_hwc_kernel(const float* __restrict__ src, long chan_stride, int c0, int HW,
                                                          float* __restrict__ dst) {
    __shared__ float tile[C * 257];
    const int p0 = blockIdx.x * 256;
    const int t = threadIdx.x;
    const int np = min(256, HW - p0);
    if (t < np) {
#pragma unroll
        for (int c = 0; c < C; ++c) tile[c * 257 + t] = src[(size_t)(c0 + c) * chan_stride + p0 + t];
    }
    __syncthreads();
    const int total = np * C;
    float* out = dst + (size_t)p0 * C;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const int i = k * 256 + t;
        if (i < total) out[i] = tile[(i % C) * 257 + (i / C)];
    }
}

extern "C" int dmvs_planar_to_hwc(const float* src, long chan_stride, int c0, int C, int H, int W, float* dst,
                                  dmvs_stream_t s) {
    if (!src || !dst || H <= 0 || W <= 0 || c0 < 0 || chan_stride < (long)H * W) return DMVS_EINVAL;
    const int HW = H * W;
    dim3 grid(ceil_div(HW, 256));
    hipStream_t st = (hipStream_t)s;
    switch (C) {
        case 8: nchw_to_hwc_kernel<8><<<grid, 256, 0, st>>>(src, chan_stride, c0, HW, dst); break;
        case 16: nchw_to_hwc_kernel<16><<<grid, 256, 0, st>>>(src, chan_stride, c0, HW, dst); break;
        case 32: nchw_to_hwc_kernel<32><<<grid, 256, 0, st>>>(src, chan_stride, c0, HW, dst); break;
        default: return DMVS_EUNSUPPORTED;
    }
    DMVS_LAUNCH_CHECK();
}

extern "C" int dmvs_nchw_to_hwc(const float* src, int c0, int C, int H, int W, float* dst, dmvs_stream_t s) {
    return dmvs_planar_to_hwc(src, (long)H * W, c0, C, H, W, dst, s);
}

// ------------------------------------------------------------------ relative projections
// One thread per source view.  Composition K*E is fp32 (as mvsnet.py:134,136); the 4x4 inverse and
// the product are done in fp64 with partial pivoting and rounded once to fp32.  torch.inverse is an
// fp32 LU whose own rounding error on these matrices moves projected pixels by <= 2e-4 px
// (measured, DESIGN.md "projection"), the same order as fp32 coordinate rounding, so bit parity with
// it is neither possible nor needed.
__device__ static void compose_fp32(const float* pair, double M[4][4]) {
    const float* E = pair;       // [4][4]
    const float* K = pair + 16;  // [4][4], only [:3][:3] used
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            float acc = 0.f;
            for (int k = 0; k < 3; ++k) acc = fmaf(K[r * 4 + k], E[k * 4 + c], acc);
            M[r][c] = (double)acc;
        }
    for (int c = 0; c < 4; ++c) M[3][c] = (double)E[12 + c];
}

__global__ void relative_proj_kernel(const float* __restrict__ pairs, int V, float* __restrict__ out) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (v >= V) return;
    double R[4][4], S[4][4], A[4][8];
    compose_fp32(pairs, R);
    compose_fp32(pairs + (size_t)v * 32, S);
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            A[r][c] = R[r][c];
            A[r][4 + c] = (r == c) ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; ++col) {  // Gauss-Jordan, partial pivoting
        int piv = col;
        double best = fabs(A[col][col]);
        for (int r = col + 1; r < 4; ++r)
            if (fabs(A[r][col]) > best) { best = fabs(A[r][col]); piv = r; }
        if (piv != col)
            for (int c = 0; c < 8; ++c) { double t = A[col][c]; A[col][c] = A[piv][c]; A[piv][c] = t; }
        const double inv = 1.0 / A[col][col];
        for (int c = 0; c < 8; ++c) A[col][c] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = A[r][col];
            for (int c = 0; c < 8; ++c) A[r][c] -= f * A[col][c];
        }
    }
    float* o = out + (size_t)(v - 1) * 12;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 4; ++c) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc += S[r][k] * A[k][4 + c];
            if (c < 3) o[r * 3 + c] = (float)acc; else o[9 + r] = (float)acc;
        }
    }
}

extern "C" int dmvs_relative_proj(const float* pairs, int V, float* out12, dmvs_stream_t s) {
    if (!pairs || !out12 || V < 2) return DMVS_EINVAL;
    relative_proj_kernel<<<1, 64, 0, (hipStream_t)s>>>(pairs, V, out12);
    DMVS_LAUNCH_CHECK();
}

// ------------------------------------------------------------------ hypothesis planes
// torch.linspace(a, b, n): step = (b-a)/(n-1); i < n/2 ? a + step*i : b - step*(n-1-i).
__device__ __forceinline__ float linspace_at(float a, float b, int n, int i) {
    const float step = (b - a) / (float)(n - 1);
    return (i < n / 2) ? a + step * (float)i : b - step * (float)(n - 1 - i);
}

// First stage: planes depend only on d and on the (row,col) parity (module.py:560-579, 598-634).
__global__ __launch_bounds__(256) void hyp_first_kernel(const float* __restrict__ dv, int n, int D, int H, int W,
                                                        int inverse, float* __restrict__ out,
                                                        float* __restrict__ out_itv) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const int d = blockIdx.z;
    const float dmin = dv[0], dmax = dv[n - 1];
    float itv = (dmax - dmin) / (float)(D - 1);
    const bool same = ((y & 1) == (x & 1));
    float val;
    if (!inverse) {
        const float plane = dmin + (float)d * itv;
        val = same ? plane - itv : plane + itv;
    } else {
        // module.py:606-626: the interval is recomputed from the shifted ends before each variant
        float lo = dmin - itv, hi = dmax - itv;
        itv = (hi - lo) / (float)(D - 1);
        const float vn = 1.0f / linspace_at(1.0f / lo, 1.0f / hi, D, d);
        lo = dmin + itv; hi = dmax + itv;
        itv = (hi - lo) / (float)(D - 1);
        const float vp = 1.0f / linspace_at(1.0f / lo, 1.0f / hi, D, d);
        val = same ? vn : vp;
    }
    if (x < W) out[((size_t)d * H + y) * W + x] = val;
    if (x == 0 && y == 0 && d == 0) out_itv[0] = itv;
}

extern "C" int dmvs_hypotheses_first(const float* dv, int n, int D, int H, int W, int inverse, float* out,
                                     float* out_itv, dmvs_stream_t s) {
    if (!dv || !out || !out_itv || n < 2 || D < 2 || H <= 0 || W <= 0) return DMVS_EINVAL;
    dim3 grid(ceil_div(W, 256), H, D);
    hyp_first_kernel<<<grid, 256, 0, (hipStream_t)s>>>(dv, n, D, H, W, inverse, out, out_itv);
    DMVS_LAUNCH_CHECK();
}

// Later stages.  Plane d at coarse pixel (r,c) (module.py:476-507 / 525-554):
//   variant n (r%2==c%2): lo = last-(D+2)/2*pix, hi = last+(D-2)/2*pix
//   variant p           : lo = last-(D-2)/2*pix, hi = last+(D+2)/2*pix
//   linear : lo + d*((hi-lo)/(D-1));   inverse: 1/(1/lo + d*((1/hi-1/lo)/(D-1)))
// then the x2 bilinear upsample of mvsnet.py:233 (align_corners=False: src=(dst+.5)/2-.5 clamped at 0).
__device__ __forceinline__ float hyp_sample(float last, bool same, float pix, int D, int d, int inverse) {
    const float a = (float)(D + 2) / 2.0f * pix, b = (float)(D - 2) / 2.0f * pix;
    const float lo = same ? last - a : last - b;
    const float hi = same ? last + b : last + a;
    if (!inverse) return lo + (float)d * ((hi - lo) / (float)(D - 1));
    const float ilo = 1.0f / lo, ihi = 1.0f / hi;
    return 1.0f / (ilo + (float)d * ((ihi - ilo) / (float)(D - 1)));
}

__global__ __launch_bounds__(256) void hyp_next_kernel(const float* __restrict__ last, int h, int w,
                                                       const float* __restrict__ dv, int n, float ratio, int D,
                                                       int inverse, float* __restrict__ out,
                                                       float* __restrict__ out_itv, int nplanes, int up) {
    const int W = up * w, H = up * h;
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const float depth_interval = (dv[n - 1] - dv[0]) / (float)n;  // mvsnet.py:196
    const float pix = ratio * depth_interval;                      // mvsnet.py:226-227
    if (x == 0 && y == 0) out_itv[0] = ((float)D * pix) / (float)(D - 1);  // module.py:491
    if (x >= W) return;
    if (up == 1) {   // same-resolution transition (stages beyond the reference's three: the resize is the identity)
        const float l = last[y * w + x];
        const bool sm = ((y & 1) == (x & 1));
        for (int d = 0; d < nplanes; ++d) out[((size_t)d * H + y) * W + x] = hyp_sample(l, sm, pix, D, d, inverse);
        return;
    }
    float sy = ((float)y + 0.5f) * 0.5f - 0.5f; sy = sy < 0.f ? 0.f : sy;
    float sx = ((float)x + 0.5f) * 0.5f - 0.5f; sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float l00 = last[y0 * w + x0], l01 = last[y0 * w + x1], l10 = last[y1 * w + x0], l11 = last[y1 * w + x1];
    const bool s00 = ((y0 & 1) == (x0 & 1)), s01 = ((y0 & 1) == (x1 & 1));
    const bool s10 = ((y1 & 1) == (x0 & 1)), s11 = ((y1 & 1) == (x1 & 1));
    for (int d = 0; d < nplanes; ++d) {   // nplanes = D, or 1: only the base plane of the affine form
        const float v00 = hyp_sample(l00, s00, pix, D, d, inverse), v01 = hyp_sample(l01, s01, pix, D, d, inverse);
        const float v10 = hyp_sample(l10, s10, pix, D, d, inverse), v11 = hyp_sample(l11, s11, pix, D, d, inverse);
        out[((size_t)d * H + y) * W + x] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    }
}

static int hyp_next_launch(const float* last, int h, int w, int up, const float* dv, int n, float ratio, int D, int inverse,
                           float* out, float* out_itv, int nplanes, dmvs_stream_t s) {
    if (!last || !dv || !out || !out_itv || h <= 0 || w <= 0 || n < 2 || D < 2 || (up != 1 && up != 2)) return DMVS_EINVAL;
    dim3 grid(ceil_div(up * w, 256), up * h);
    hyp_next_kernel<<<grid, 256, 0, (hipStream_t)s>>>(last, h, w, dv, n, ratio, D, inverse, out, out_itv, nplanes, up);
    DMVS_LAUNCH_CHECK();
}

extern "C" int dmvs_hypotheses_next(const float* last, int h, int w, const float* dv, int n, float ratio, int D,
                                    int inverse, float* out, float* out_itv, dmvs_stream_t s) {
    return hyp_next_launch(last, h, w, 2, dv, n, ratio, D, inverse, out, out_itv, D, s);
}

extern "C" int dmvs_hypotheses_next_up(const float* last, int h, int w, int up, const float* dv, int n, float ratio, int D,
                                       int inverse, int base_only, float* out, float* out_itv, dmvs_stream_t s) {
    if (base_only && inverse) return DMVS_EINVAL;   // inverse-depth sampling is not affine in d
    return hyp_next_launch(last, h, w, up, dv, n, ratio, D, inverse, out, out_itv, base_only ? 1 : D, s);
}

// Affine form of the linear-depth hypotheses: plane d = base + d * interval, so only plane 0 (the
// checkerboarded lower end, upsampled for the later stages) is written: [H][W] instead of [D][H][W].
extern "C" int dmvs_hypothesis_base_first(const float* dv, int n, int D, int H, int W, float* base_hw, float* out_itv,
                                          dmvs_stream_t s) {
    if (!dv || !base_hw || !out_itv || n < 2 || D < 2 || H <= 0 || W <= 0) return DMVS_EINVAL;
    dim3 grid(ceil_div(W, 256), H, 1);
    hyp_first_kernel<<<grid, 256, 0, (hipStream_t)s>>>(dv, n, D, H, W, 0, base_hw, out_itv);
    DMVS_LAUNCH_CHECK();
}

extern "C" int dmvs_hypothesis_base_next(const float* last, int h, int w, const float* dv, int n, float ratio, int D,
                                         float* base_hw, float* out_itv, dmvs_stream_t s) {
    return hyp_next_launch(last, h, w, 2, dv, n, ratio, D, 0, base_hw, out_itv, 1, s);
}

// ------------------------------------------------------------------ misc
extern "C" int dmvs_version(void) { return DMVS_VERSION; }

extern long g_single_buf_min_blocks;   // conv3d_mfma.hip
extern long g_min_blocks, g_split_blocks, g_wino_stages, g_wino_persistent, g_wino_conv0_grid, g_deconv_prefetch;
extern long g_c8_rows;   // conv2d_c8.hip
extern long g_k3r_grid, g_k3r_counted_wait;   // conv3d_coarse.hip
extern long g_k3z_grid, g_k3z_zs, g_k3z_counted_wait;   // conv3d_zmarch.hip

extern "C" int dmvs_tune(const char* name, int value) {
    if (!name) return DMVS_EINVAL;
    if (!strcmp(name, "k3_single_buf_min_blocks")) { if (value < 0) return DMVS_EINVAL; g_single_buf_min_blocks = value; return 0; }
    if (!strcmp(name, "k3_min_blocks")) { if (value < 0) return DMVS_EINVAL; g_min_blocks = value; return 0; }
    if (!strcmp(name, "k3_split_blocks")) { if (value < 0) return DMVS_EINVAL; g_split_blocks = value; return 0; }
    if (!strcmp(name, "wino_conv0_grid")) { if (value < 8 || value % 8) return DMVS_EINVAL; g_wino_conv0_grid = value; return 0; }
    if (!strcmp(name, "k3_deconv_prefetch")) { g_deconv_prefetch = value ? 1 : 0; return 0; }
    if (!strcmp(name, "c8_rows")) { if (value < 0 || value > 4096) return DMVS_EINVAL; g_c8_rows = value; return 0; }
    if (!strcmp(name, "wino_persistent")) { g_wino_persistent = value ? 1 : 0; return 0; }
    if (!strcmp(name, "k3r_grid")) { if (value < 32 || value > 1024 || value % 32) return DMVS_EINVAL; g_k3r_grid = value; return 0; }
    if (!strcmp(name, "k3z_grid")) { if (value < 0 || value > 4096 || value % 8) return DMVS_EINVAL; g_k3z_grid = value; return 0; }
    if (!strcmp(name, "k3z_zs")) { if (value < 0 || value > 64) return DMVS_EINVAL; g_k3z_zs = value; return 0; }
    if (!strcmp(name, "k3z_counted_wait")) { g_k3z_counted_wait = value ? 1 : 0; return 0; }
    if (!strcmp(name, "k3r_counted_wait")) { g_k3r_counted_wait = value ? 1 : 0; return 0; }
    if (!strcmp(name, "wino_stages")) { if (value < 0 || value > 2) return DMVS_EINVAL; g_wino_stages = value; return 0; }
    return DMVS_EUNSUPPORTED;
}

extern "C" const char* dmvs_error_string(int code) {
    if (code == 0) return "ok";
    if (code == DMVS_EINVAL) return "dmvs: invalid argument";
    if (code == DMVS_EUNSUPPORTED) return "dmvs: unsupported channel count or mode";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "dmvs: unknown error";
}
