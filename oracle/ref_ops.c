/* ORACLE -- test infrastructure only; never linked into or called by the product.
 *
 * Plain-C restatement of the published algorithms of the PyTorch ATen ops that carry the arithmetic of
 * DMVSNet's hot path (the reference itself contains no arithmetic of its own for them):
 *   grid_sample(bilinear, zeros, align_corners=True)   used at /root/reference/networks/module.py:247
 *   Conv3d k3 p1 s{1,2} / ConvTranspose3d k3 s2 p1 op1  module.py:142,187
 *   BatchNorm eval + ReLU                               module.py:151-157
 *   softmax over D + expectation + selection            /root/reference/networks/mvsnet.py:15-100
 * Scalar loops, double accumulation where it is free; sized for tiny test shapes only.
 * Cross-checked against oracle/dmvs_oracle.py (ATen) and the golden vectors in tests/test_oracle.py.
 */
#include <math.h>
#include <stddef.h>

/* a2+a3: homography warp + 2-group correlation, summed over views.  NCHW features (as the reference). */
void ref_warp_corr(const float* ref, const float* const* src, int nsrc, const float* proj12 /*[nsrc][12]*/,
                   const float* depth /*[D][H][W]*/, float* sim /*[2][D][H][W]*/, int C, int D, int H, int W) {
    const size_t HW = (size_t)H * W;
    for (size_t i = 0; i < 2 * (size_t)D * HW; ++i) sim[i] = 0.f;
    for (int v = 0; v < nsrc; ++v) {
        const float* P = proj12 + 12 * v;
        const float* S = src[v];
        for (int d = 0; d < D; ++d)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const float dep = depth[(size_t)d * HW + (size_t)y * W + x];
                    const float rx = P[0] * x + P[1] * y + P[2], ry = P[3] * x + P[4] * y + P[5];
                    const float rz = P[6] * x + P[7] * y + P[8];
                    const float px = rx * dep + P[9], py = ry * dep + P[10];
                    float pz = rz * dep + P[11];
                    if (pz == 0.f) pz += 0.00001f; /* module.py:237 */
                    const float gx = px / pz / ((W - 1) / 2.0f) - 1.f, gy = py / pz / ((H - 1) / 2.0f) - 1.f;
                    const float ix = (gx + 1.f) / 2.f * (W - 1), iy = (gy + 1.f) / 2.f * (H - 1);
                    const float x0 = floorf(ix), y0 = floorf(iy);
                    double g[2] = {0.0, 0.0};
                    for (int t = 0; t < 4; ++t) { /* nw, ne, sw, se; each tap zero outside the image */
                        const float tx = x0 + (t & 1), ty = y0 + (t >> 1);
                        const float wx = (t & 1) ? ix - x0 : x0 + 1.f - ix, wy = (t >> 1) ? iy - y0 : y0 + 1.f - iy;
                        if (!(tx >= 0.f && tx <= W - 1 && ty >= 0.f && ty <= H - 1)) continue;
                        const size_t off = (size_t)ty * W + (size_t)tx;
                        for (int c = 0; c < C; ++c)
                            g[c & 1] += (double)(wx * wy) * S[c * HW + off] * ref[c * HW + (size_t)y * W + x];
                    }
                    sim[(size_t)d * HW + (size_t)y * W + x] += (float)(g[0] / (C / 2));
                    sim[((size_t)D + d) * HW + (size_t)y * W + x] += (float)(g[1] / (C / 2));
                }
    }
}

/* Conv3d, kernel (kd,3,3), padding (kd/2,1,1), stride (sd,s,s); weight [Cout][Cin][kd][3][3]. */
void ref_conv3d(const float* in, const float* w, float* out, int Cin, int Cout, int D, int H, int W, int kd, int sd,
                int s) {
    const int pd = kd / 2, Do = (D + 2 * pd - kd) / sd + 1, Ho = (H + 2 - 3) / s + 1, Wo = (W + 2 - 3) / s + 1;
    for (int co = 0; co < Cout; ++co)
        for (int z = 0; z < Do; ++z)
            for (int y = 0; y < Ho; ++y)
                for (int x = 0; x < Wo; ++x) {
                    double acc = 0.0;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int a = 0; a < kd; ++a)
                            for (int b = 0; b < 3; ++b)
                                for (int c = 0; c < 3; ++c) {
                                    const int iz = z * sd - pd + a, iy = y * s - 1 + b, ix = x * s - 1 + c;
                                    if (iz < 0 || iz >= D || iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                                    acc += (double)in[(((size_t)ci * D + iz) * H + iy) * W + ix] *
                                           w[((((size_t)co * Cin + ci) * kd + a) * 3 + b) * 3 + c];
                                }
                    out[(((size_t)co * Do + z) * Ho + y) * Wo + x] = (float)acc;
                }
}

/* ConvTranspose3d in its DEFINING scatter form: out[i*2 - 1 + k] += in[i] * w[ci][co][k]; output size 2n
 * per strided axis (k3 s2 p1 output_padding 1).  kd = 1: no depth upsampling (2D layer on a 1-slice volume).
 * weight [Cin][Cout][kd][3][3]. */
void ref_deconv3d(const float* in, const float* w, float* out, int Cin, int Cout, int D, int H, int W, int kd) {
    const int Do = kd == 3 ? 2 * D : D, Ho = 2 * H, Wo = 2 * W;
    for (size_t i = 0; i < (size_t)Cout * Do * Ho * Wo; ++i) out[i] = 0.f;
    for (int ci = 0; ci < Cin; ++ci)
        for (int z = 0; z < D; ++z)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const float v = in[(((size_t)ci * D + z) * H + y) * W + x];
                    for (int co = 0; co < Cout; ++co)
                        for (int a = 0; a < kd; ++a)
                            for (int b = 0; b < 3; ++b)
                                for (int c = 0; c < 3; ++c) {
                                    const int oz = kd == 3 ? 2 * z - 1 + a : z, oy = 2 * y - 1 + b, ox = 2 * x - 1 + c;
                                    if (oz < 0 || oz >= Do || oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
                                    out[(((size_t)co * Do + oz) * Ho + oy) * Wo + ox] +=
                                        v * w[((((size_t)ci * Cout + co) * kd + a) * 3 + b) * 3 + c];
                                }
                }
}

/* BatchNorm(eval, eps 1e-5) + optional ReLU + optional residual, in place on [C][n]. */
void ref_bn_relu_add(float* x, const float* gamma, const float* beta, const float* mean, const float* var,
                     const float* skip, int C, size_t n, int relu) {
    for (int c = 0; c < C; ++c) {
        const float inv = 1.0f / sqrtf(var[c] + 1e-5f);
        for (size_t i = 0; i < n; ++i) {
            float v = (x[c * n + i] - mean[c]) * inv * gamma[c] + beta[c];
            if (relu && v < 0.f) v = 0.f;
            if (skip) v += skip[c * n + i];
            x[c * n + i] = v;
        }
    }
}

/* a6/a7: softmax over D of alpha*logits, expectation, min/max pairs, checkerboard selection, confidence.
 * mode 0 = DepthNet.forward (sel [4][H][W]), mode 1 = DepthNet.refine (sel [H][W]). */
void ref_depth_regress(const float* logits, const float* depth, float interval, float alpha, int mode, int D, int H,
                       int W, float* dsp, float* sel, float* conf) {
    const size_t HW = (size_t)H * W;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t p = (size_t)y * W + x;
            float e[4];
            for (int c = 0; c < 4; ++c) {
                float m = -INFINITY;
                for (int d = 0; d < D; ++d) m = fmaxf(m, alpha * logits[((size_t)c * D + d) * HW + p]);
                double s = 0.0, sd = 0.0;
                for (int d = 0; d < D; ++d) {
                    const double w = exp((double)(alpha * logits[((size_t)c * D + d) * HW + p] - m));
                    s += w; sd += w * depth[(size_t)d * HW + p];
                }
                e[c] = (float)(sd / s);
                dsp[c * HW + p] = e[c];
            }
            const float mean = (e[0] + e[1] + e[2] + e[3]) / 4.f;
            float var = 0.f;
            for (int c = 0; c < 4; ++c) var += (e[c] - mean) * (e[c] - mean);
            var /= 4.f;
            conf[p] = 2.f * (1.f / (1.f + expf(-interval / (sqrtf(var) + 1e-5f))) - 0.5f);
            const float sm = fminf(e[0], e[1]), sM = fmaxf(e[0], e[1]), hm = fminf(e[2], e[3]), hM = fmaxf(e[2], e[3]);
            if (mode == 1) {
                sel[p] = (y % 2 == 0) ? (x % 2 == 0 ? sm : sM) : (x % 2 == 0 ? hM : hm);
                continue;
            }
            float lo = (y % 4 == 0 || y % 4 == 2) ? sm : hm, hi = (y % 4 == 0 || y % 4 == 2) ? sM : hM;
            if (y % 4 >= 2) { const float l = 2 * lo - hi, h = 2 * hi - lo; lo = l; hi = h; }
            const float st[6] = {3 * lo - 2 * hi, 2 * lo - hi, lo, hi, 2 * hi - lo, 3 * hi - 2 * lo};
            /* windows (row%4,col%2): (0,0)[0:4] (0,1)[2:6] (1,0)[2:6] (1,1)[0:4] (2,0)[0:4] (2,1)[2:6] (3,0)[2:6] (3,1)[0:4] */
            const int hiwin = ((y % 4) % 2 == 0) ? (x % 2 == 1) : (x % 2 == 0);
            for (int k = 0; k < 4; ++k) sel[k * HW + p] = st[k + (hiwin ? 2 : 0)];
        }
}
