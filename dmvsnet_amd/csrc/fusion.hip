// N4: geometric-consistency check of one (reference, source) depth-map pair, fused into one kernel.
//
// Replaces reproject_with_depth_pytorch + check_geometric_consistency(_pytorch)
// (/root/reference/filter/pcd.py:151-242): project every reference pixel with its depth into the source view,
// bilinearly sample the source depth map there (grid_sample bilinear / zeros / align_corners=True, i.e. at the
// projected pixel coordinate), lift the sample back into the reference view, and keep the pixel when the
// reprojection lands within `dist_thresh` px and the depths agree to `rel_thresh` (1 px / 1 % in the reference).
// The reference materialises ~15 [H*W]-sized temporaries per pair; here: one pass, HBM-bound
// (reads 2 depth maps, writes mask + reprojected depth, accumulates the per-pixel vote and depth sums that
// filter_depth (pcd.py:283-300) builds with numpy).
//
// The chained 3x3 / 4x4 products of the reference are folded on the host (fp64, rounded once) into
//   P[0..11]  A1 (3x3), b1 (3):  K_src * xyz_src            = A1 * (x, y, 1) * d_ref + b1
//   P[12..23] A2 (3x3), t2 (3):  xyz_reprojected (ref cam)  = A2 * (xs, ys, 1) * d_sampled + t2
//   P[24..32] K_ref (3x3)
#include "common.h"

// LADDER: the dynamic-threshold variant of the Tanks&Temples filter (filter/dypcd_tanks.py:164-184): nine gates
// (dist < i * dist_base and rel < i * rel_base, i = 2..10) are evaluated on the same reprojection; level_votes[i-2]
// accumulates gate i per pixel, the last gate (i = 10) plays the role of the single gate for mask / depth / sums.
// That variant does not patch zero reference depths (rel = |d_reproj - d| / d is inf / NaN there: all gates false).
template <bool LADDER>
__global__ __launch_bounds__(256) void geo_consistency_kernel(const float* __restrict__ depth_ref,
                                                              const float* __restrict__ depth_src,
                                                              const float* __restrict__ P, int H, int W,
                                                              float dist_thresh, float rel_thresh,
                                                              unsigned char* __restrict__ mask,
                                                              float* __restrict__ depth_reproj,
                                                              int* __restrict__ vote_sum, float* __restrict__ depth_sum,
                                                              int* __restrict__ level_votes) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const size_t pix = (size_t)y * W + x;
    const float fx = (float)x, fy = (float)y;
    const float d = depth_ref[pix];
    // reference pixel -> source image
    const float hx = (P[0] * fx + P[1] * fy + P[2]) * d + P[9];
    const float hy = (P[3] * fx + P[4] * fy + P[5]) * d + P[10];
    const float hz = (P[6] * fx + P[7] * fy + P[8]) * d + P[11];
    const float xs = hx / hz, ys = hy / hz;
    // bilinear sample of the source depth at (xs, ys); taps outside the image contribute zero
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    const float x0f = floorf(xs), y0f = floorf(ys);
    const float tx = xs - x0f, ty = ys - y0f;
    const bool x0in = (x0f >= 0.f) && (x0f <= wm1), x1in = (x0f >= -1.f) && (x0f <= wm1 - 1.f);
    const bool y0in = (y0f >= 0.f) && (y0f <= hm1), y1in = (y0f >= -1.f) && (y0f <= hm1 - 1.f);
    const int x0 = (int)fminf(fmaxf(x0f, 0.f), wm1), x1 = (int)fminf(fmaxf(x0f + 1.f, 0.f), wm1);
    const int y0 = (int)fminf(fmaxf(y0f, 0.f), hm1), y1 = (int)fminf(fmaxf(y0f + 1.f, 0.f), hm1);
    const float s00 = depth_src[(size_t)y0 * W + x0], s01 = depth_src[(size_t)y0 * W + x1];
    const float s10 = depth_src[(size_t)y1 * W + x0], s11 = depth_src[(size_t)y1 * W + x1];
    const float sd = ((x0in && y0in) ? (1.f - tx) * (1.f - ty) * s00 : 0.f) + ((x1in && y0in) ? tx * (1.f - ty) * s01 : 0.f) +
                     ((x0in && y1in) ? (1.f - tx) * ty * s10 : 0.f) + ((x1in && y1in) ? tx * ty * s11 : 0.f);
    // sampled source point -> reference camera
    const float rx = (P[12] * xs + P[13] * ys + P[14]) * sd + P[21];
    const float ry = (P[15] * xs + P[16] * ys + P[17]) * sd + P[22];
    const float rz = (P[18] * xs + P[19] * ys + P[20]) * sd + P[23];
    const float kx = P[24] * rx + P[25] * ry + P[26] * rz;
    const float ky = P[27] * rx + P[28] * ry + P[29] * rz;
    float kz = P[30] * rx + P[31] * ry + P[32] * rz;
    if (kz == 0.f) kz += 0.00001f;  // pcd.py:194
    const float xr = kx / kz, yr = ky / kz;
    const float dist = sqrtf((xr - fx) * (xr - fx) + (yr - fy) * (yr - fy));
    const float dref = (!LADDER && d == 0.f) ? 1e-4f : d;  // pcd.py:219 (the ladder variant divides by the raw depth)
    const float rel = fabsf(rz - dref) / dref;
    bool ok;
    if constexpr (LADDER) {
        // dist_thresh / rel_thresh carry the BASES; gate i compares with i * base (dypcd_tanks.py:179-181)
        ok = false;
#pragma unroll
        for (int i = 2; i <= 10; ++i) {
            ok = dist < (float)i * dist_thresh && rel < (float)i * rel_thresh;
            if (level_votes) level_votes[(size_t)(i - 2) * H * W + pix] += ok ? 1 : 0;
        }
    } else
        ok = dist < dist_thresh && rel < rel_thresh;  // NaN compares false, as in the reference
    if (mask) mask[pix] = ok ? 1 : 0;
    if (depth_reproj) depth_reproj[pix] = ok ? rz : 0.f;
    if (vote_sum) vote_sum[pix] += ok ? 1 : 0;
    if (depth_sum) depth_sum[pix] += ok ? rz : 0.f;
}

extern "C" int dmvs_geo_consistency(const float* depth_ref, const float* depth_src, const float* proj33, int H, int W,
                                    float dist_thresh, float rel_thresh, unsigned char* mask, float* depth_reproj,
                                    int* vote_sum, float* depth_sum, dmvs_stream_t stream) {
    if (!depth_ref || !depth_src || !proj33 || H < 1 || W < 1) return DMVS_EINVAL;
    dim3 grid(ceil_div(W, 256), H);
    geo_consistency_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(depth_ref, depth_src, proj33, H, W, dist_thresh,
                                                                         rel_thresh, mask, depth_reproj, vote_sum, depth_sum, nullptr);
    DMVS_LAUNCH_CHECK();
}

extern "C" int dmvs_geo_consistency_ladder(const float* depth_ref, const float* depth_src, const float* proj33, int H,
                                           int W, float dist_base, float rel_base, int* level_votes,
                                           unsigned char* mask, float* depth_reproj, int* vote_sum, float* depth_sum,
                                           dmvs_stream_t stream) {
    if (!depth_ref || !depth_src || !proj33 || H < 1 || W < 1) return DMVS_EINVAL;
    dim3 grid(ceil_div(W, 256), H);
    geo_consistency_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(depth_ref, depth_src, proj33, H, W, dist_base,
                                                                        rel_base, mask, depth_reproj, vote_sum, depth_sum, level_votes);
    DMVS_LAUNCH_CHECK();
}
