/*
 * dmvs.h -- C ABI of libdmvs_hip.so: the MI355X (gfx950) kernels behind DMVSNet's
 * cost-volume hot path.  This is the drop-in boundary below the Python host module
 * (dmvsnet_amd.MVSNet, same forward() signature as /root/reference/networks/mvsnet.py:188).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - every kernel is enqueued on `stream` (a hipStream_t passed as void*; 0 = null stream);
 *   - nothing allocates, nothing synchronises, nothing throws; the caller owns all buffers;
 *   - return value: 0 on success, a positive hipError_t from the launch, or a negative
 *     DMVS_E* argument error.  dmvs_error_string() decodes both;
 *   - batch size is 1 (the reference's eval loader, model.py:330-336, always uses B=1);
 *     volumes are planar fp32: activations [C][D][H][W], hypotheses [D][H][W];
 *     2D feature maps handed to the warp are pixel-major ("HWC"): element (y,x,c) at
 *     ((y*W + x)*pix_stride + c).
 *
 * Each entry point cites the reference code it replaces (file:line under /root/reference).
 */
#ifndef DMVS_H
#define DMVS_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dmvs_stream_t; /* hipStream_t */

#define DMVS_VERSION 140 /* 0.1.4 (r06): + dmvs_prob_regress / dmvs_depth_select (`prob` -> K4); 0.1.3 (r06): + K3z dmvs_conv3d_zmarch / _weight_floats / dmvs_pack_conv_weights_zmarch, + the bf16-split probe dmvs_conv3d_split_probe / _weight_floats / dmvs_pack_conv_weights_split; 0.1.2 (r05): + K3r dmvs_conv3d_coarse / _weight_floats / dmvs_pack_conv_weights_coarse; 0.1.1: DMVS_OUT_Q4 moved to bit 3 (value 8); bit 2 (value 4, r02's DMVS_OUT_HWC2: two
                            PIXEL-MAJOR halves) is retired and rejected with DMVS_EUNSUPPORTED -- a caller built against
                            version 100 can no longer get the quad-planar layout silently; dmvs_tune("k1_variant") is
                            gone (the launch variant is an argument of dmvs_warp_corr_q4) */

#define DMVS_EINVAL (-1)      /* bad dimension / null pointer */
#define DMVS_EUNSUPPORTED (-2) /* channel count or mode not compiled in */

#define DMVS_MAX_SRC_VIEWS 16

/* conv flags */
#define DMVS_RELU 1
#define DMVS_SKIP_UP2 2    /* skip is [Cout][D][Ho/2][Wo/2]: nearest x2 upsample fused into the residual add
                              (FeatureNet top-down path, module.py:328,333); K3 conv modes only */
#define DMVS_FLAG_RETIRED_4 4 /* was DMVS_OUT_HWC2 in version 100; every conv entry point returns DMVS_EUNSUPPORTED for it */
#define DMVS_OUT_Q4 8      /* out is TWO quad-planar tensors back to back, [Do][Cout/8][Ho][Wo][4] each (channels
                              [0, Cout/2) then [Cout/2, Cout); with kdepth = 1 the depth slices are the views, so every
                              view's half is one contiguous [C/4][H][W][4] map): FeatureNet's stageK / stageK_c halves
                              (module.py:326-336) written directly in the layout dmvs_warp_corr_q4 samples; K3 conv
                              modes, no residual, Cout % 8 == 0 */
#define DMVS_IN_VIEWS 16   /* dmvs_conv3d_mfma, Cin = 4, kdepth 1, DMVS_CONV_S1 only (FeatureNet's first layer, module.py:283): `in`
                              is the eval loader's image stack [D = V][3][H][W] (general_eval.py:89,186) read in place -- channel c
                              of view v at ((v * 3 + c) * H * W) -- instead of a planar [4][V][H][W] copy with a zero channel.  The
                              4th channel of the MFMA k-group carries ZERO weights and reads the next view's first channel (finite
                              pixel values: the product is exactly 0) or, for the last view, past the end of the buffer
                              descriptor (returns 0) */
/* conv modes */
#define DMVS_CONV_S1 0     /* Conv3d k3 s1 p1                         module.py:142 */
#define DMVS_CONV_S2 1     /* Conv3d k3 s2 p1                         module.py:142 */
#define DMVS_DECONV_S2 2   /* ConvTranspose3d k3 s2 p1 output_pad 1   module.py:187 */
#define DMVS_CONV2D_K5S2 3 /* Conv2d k5 s2 p2 per depth slice (kdepth = 1)    module.py:289,295 */
#define DMVS_CONV2D_K1 4   /* Conv2d 1x1 per depth slice (kdepth = 1)         module.py:301,305,306 */

int dmvs_version(void);
/* Tuning knobs (A/B measurements, autotuning); not needed for correct results.  Known names:
 *   "k3_single_buf_min_blocks"  3D conv layers with at least this many workgroups run with one LDS stage (default
 *                               0: all of them; smaller grids keep two stages)
 *   "k3_min_blocks"             the big K3 tiles are used when they yield at least this many workgroups (768)
 *   "k3_split_blocks"           two-block (Cout = 64) layers whose small tiles yield fewer workgroups than this split
 *                               their M blocks over the waves (1024)
 *   "k3_deconv_prefetch"        1 (default): transposed convs with a residual prefetch it under their MFMAs; 0: residual loads
 *                               in the epilogue (A/B)
 *   "wino_stages"               LDS stages of the 3D Winograd layers: 0 = per-layer default, 1 = one, 2 = two where they fit
 *   "wino_persistent"           0: one tile per workgroup instead of the persistent tile walk (A/B; default 1)
 *   "wino_conv0_grid"           persistent workgroups of the conv0 Winograd kernel (multiple of 8; default 512)
 *   "k3r_grid"                  persistent workgroups of a K3r launch (dmvs_conv3d_coarse): 32 .. 1024 in multiples of 32 (8 XCDs x up
 *                               to 4 cout groups); default 256 = one per CU
 *   "k3r_counted_wait"          1 (default): K3r waits for a ring stage with a counted vmcnt; 0: vmcnt(0) (bit-identical, A/B + gate)
 *   "c8_rows"                   rows per wave of the K3s row sweep (0 = chosen from the wave count)
 *   "k3z_grid"                  persistent workgroups of a K3z launch (dmvs_conv3d_zmarch; multiple of 8, 0 = as many as are resident)
 *   "k3z_zs"                    cap of a z segment's length in K3z (0 = none; results do not depend on it)
 *   "k3z_counted_wait"          ring of 3 builds only: 1 = counted vmcnt at the stage wait, 0 = vmcnt(0)
 * Returns 0, DMVS_EINVAL (bad value) or DMVS_EUNSUPPORTED (unknown name).  Process-wide, not thread-safe. */
int dmvs_tune(const char* name, int value);
const char* dmvs_error_string(int code);

/* [C_total][H][W] planar slice c0..c0+C  ->  [H][W][C] pixel-major.
 * Layout glue between FeatureNet's NCHW output (module.py:326-336, the stageK / stageK_c channel
 * split) and the warp kernel; no arithmetic. */
int dmvs_nchw_to_hwc(const float* src_chw, int c0, int C, int H, int W, float* dst_hwc, dmvs_stream_t stream);
/* the same with an explicit channel stride (floats): src points at one view's plane of a [C][V][H][W] stack. */
int dmvs_planar_to_hwc(const float* src, long chan_stride, int c0, int C, int H, int W, float* dst_hwc,
                       dmvs_stream_t stream);

/* Relative projections for all source views of one stage.
 * proj_pairs [V][2][4][4] (view 0 = reference): [v][0] extrinsic, [v][1][:3][:3] intrinsics.
 * out [V-1][12]: rot (row-major 3x3) then trans (3) of  (K_s E_s) (K_r E_r)^-1.
 * Replaces mvsnet.py:133-136 (composition) + module.py:223-225 (inverse, matmul, slicing). */
int dmvs_relative_proj(const float* proj_pairs, int V, float* out12, dmvs_stream_t stream);

/* First-stage hypothesis planes, module.py:560-579 (linear) / 598-634 (inverse).
 * depth_values [n] (only [0] and [n-1] are read).  out [D][H][W]; out_interval [1]. */
int dmvs_hypotheses_first(const float* depth_values, int n, int D, int H, int W, int inverse,
                          float* out_dhw, float* out_interval, dmvs_stream_t stream);

/* Later-stage hypothesis planes around last_depth [h][w] (previous stage's resolution),
 * then x2 bilinear upsample (align_corners=False) to [D][2h][2w] in the same kernel.
 * Replaces module.py:582-594 / 636-648 (+ 476-507, 525-554) and mvsnet.py:196,226-227,232-233.
 * ratio = depth_interval_ratio[stage]; depth_interval = (dv[n-1]-dv[0])/n is computed on device. */
int dmvs_hypotheses_next(const float* last_depth, int h, int w, const float* depth_values, int n,
                         float ratio, int D, int inverse, float* out_dhw, float* out_interval,
                         dmvs_stream_t stream);

/* Affine form of the LINEAR-depth hypotheses: every plane of a stage is
 *   plane d = base[y][x] + d * interval      (module.py:476-507, 560-579: lo + d * (hi - lo) / (D - 1))
 * so only `base` -- plane 0 of the functions above, [H][W] (the later stages: [2h][2w], upsampled) -- is produced and
 * dmvs_warp_corr_affine / dmvs_depth_regress_affine form the planes in registers: the [D][H][W] volume is neither
 * written nor read (3 x 30 / 61 / 61 MB per main pass at config 2).  Inverse-depth sampling is not affine in d and
 * keeps the volume.  out_interval as above; it is also the plane spacing. */
int dmvs_hypothesis_base_first(const float* depth_values, int n, int D, int H, int W, float* base_hw,
                               float* out_interval, dmvs_stream_t stream);
int dmvs_hypothesis_base_next(const float* last_depth, int h, int w, const float* depth_values, int n, float ratio,
                              int D, float* base_hw, float* out_interval, dmvs_stream_t stream);

/* The later-stage planes with an explicit resize factor: up = 2 is dmvs_hypotheses_next / dmvs_hypothesis_base_next,
 * up = 1 the SAME-RESOLUTION transition of pyramids deeper than the reference's three stages (a declared extension:
 * BASELINE configs[4]; the reference's resize to an equal size is the identity).  base_only: the affine form (plane 0). */
int dmvs_hypotheses_next_up(const float* last_depth, int h, int w, int up, const float* depth_values, int n, float ratio,
                            int D, int inverse, int base_only, float* out, float* out_interval, dmvs_stream_t stream);

/* K1: fused inverse-homography warp + bilinear gather + 2-group correlation + view sum.
 * Replaces CostAgg.forward (mvsnet.py:111-153) and homo_warping (module.py:212-251); the
 * [C][D][H][W] warped volume is never materialised.
 *   ref_hwc           reference feature, pixel-major, C channels, pix_stride floats per pixel
 *   src_hwc (host)    array of nsrc device pointers, same layout
 *   proj12            [nsrc][12] from dmvs_relative_proj (rows of the LOCAL source views)
 *   depth_dhw         [D][H][W] hypothesis planes
 *   sim_2dhw          [2][D][H][W]; group k = (2/C) sum_g warped[2g+k]*ref[2g+k], summed over views
 *   accumulate        0: overwrite, 1: add to what is there (view shards processed in pieces)
 * C in {8,16,32}; 1 <= nsrc <= DMVS_MAX_SRC_VIEWS. */
int dmvs_warp_corr(const float* ref_hwc, const float* const* src_hwc, int nsrc, int pix_stride,
                   const float* proj12, const float* depth_dhw, float* sim_2dhw,
                   int C, int D, int H, int W, int accumulate, dmvs_stream_t stream);

/* K1 on affine hypotheses: depth_dhw is replaced by base_hw [H][W] and step [1] (a device scalar: the interval). */
int dmvs_warp_corr_affine(const float* ref_hwc, const float* const* src_hwc, int nsrc, int pix_stride,
                          const float* proj12, const float* base_hw, const float* step, float* sim_2dhw,
                          int C, int D, int H, int W, int accumulate, dmvs_stream_t stream);

/* K1, product kernel: the same operator on QUAD-PLANAR features, [C/4][H][W][4] (channel quad q of pixel (y,x) at
 * ((q*H + y)*W + x)*4 floats) -- the layout FeatureNet's output layers write with DMVS_OUT_Q4.  A lane owns a pixel,
 * a workgroup a 32 x 8 tile x 8 (4) hypothesis planes; per source view the tile's window is bounded from the 8
 * corners of its (x, y, depth) box, staged in LDS with LDS-direct loads in 2^m channel slabs (m chosen so that the
 * window fits) and sampled with ds_read_b128; see csrc/warp_corr.hip.  Replaces the same reference code as
 * dmvs_warp_corr (mvsnet.py:111-153, module.py:212-251).
 *   depth_dhw [D][H][W], or NULL: plane d = base_hw[y][x] + d * step[0] (affine hypotheses, as dmvs_warp_corr_affine)
 *   variant   launch configuration (results agree to fp32 rounding): 0 default; low 3 bits 1 / 2 / 3 = 4 / 3 / 2
 *             workgroups per CU (40 / 53 / 80 KB windows); +8 = 4 planes per workgroup also when D > 4.  An explicit
 *             argument: no process-wide state.
 * Everything else as dmvs_warp_corr. */
int dmvs_warp_corr_q4(const float* ref_q4, const float* const* src_q4, int nsrc, const float* proj12,
                      const float* depth_dhw, const float* base_hw, const float* step, float* sim_2dhw,
                      int C, int D, int H, int W, int accumulate, int variant, dmvs_stream_t stream);

/* The same kernel on fp16 features (a declared extension: BASELINE configs[4] "fp16 features"; the reference raises
 * a dtype error there): quad-planar [C/4][H][W][4] of IEEE half, 8 bytes per pixel quad.  Products and
 * sums are fp32 (v_dot2_f32_f16); hypotheses, projections and the similarity volume stay fp32.  Halves the window
 * bytes in HBM and LDS.  Needs an even W.  Parity target: the fp32 path on the same (fp16-rounded) features. */
int dmvs_warp_corr_q4_f16(const void* ref_q4h, const void* const* src_q4h, int nsrc, const float* proj12,
                          const float* depth_dhw, const float* base_hw, const float* step, float* sim_2dhw,
                          int C, int D, int H, int W, int accumulate, int variant, dmvs_stream_t stream);

/* K2: direct LDS-tiled 3D convolution / transposed convolution, fp32 VALU, fused epilogue
 *        y = conv(x) * scale[co] + shift[co];  relu;  y += skip
 * which is Conv3d/Deconv3d + BatchNorm(eval) + ReLU (module.py:151-157, 196-202) followed by the
 * U-Net residual add (module.py:394-396).  scale/shift may be NULL (the `prob` conv, module.py:379).
 *   in  [Cin][D][H][W];  out [Cout][Do][Ho][Wo] with (Do,Ho,Wo) = (D,H,W) for S1,
 *   ((D+1)/2,(H+1)/2,(W+1)/2) for S2 (kdepth=3) and (2D,2H,2W) for DECONV_S2.
 *   kdepth = 3: full 3x3x3 kernel.  kdepth = 1: 1x3x3 kernel applied per depth slice, depth
 *   stride 1 -- the 2D bottleneck of CostRegNet_part_refine (module.py:411-414) run on [C][1][H][W].
 *   w_packed: dmvs_pack_conv_weights layout [kd][kh][kw][Cin][Cout] (Cout fastest), for DECONV the
 *   same with the ConvTranspose weight [Cin][Cout][k][k][k] re-indexed (no flip; gather form). */
int dmvs_conv3d_direct(const float* in, float* out, const float* w_packed, const float* scale,
                       const float* shift, const float* skip, int Cin, int Cout, int D, int H, int W,
                       int mode, int kdepth, int flags, dmvs_stream_t stream);

/* K3: the same operator as an implicit-GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32 /
 * 16x16x4_f32; exact fp32 products, k-ordered fmaf chain).  Same arguments and layouts as
 * dmvs_conv3d_direct except w_packed, which is the MFMA A-fragment order produced by
 * dmvs_pack_conv_weights_mfma (host).  Supported (Cin, Cout, mode, kdepth) combinations are the layers of
 * CostRegNet_part(_refine) (module.py:358-436) and, with kdepth = 1 on a [C][V][H][W] stack of views, the
 * layers of FeatureNet (module.py:283-311); anything else returns DMVS_EUNSUPPORTED.  With kdepth = 1 the depth
 * axis is a batch axis (no depth taps, no depth stride). */
int dmvs_conv3d_mfma(const float* in, float* out, const float* w_packed, const float* scale,
                     const float* shift, const float* skip, int Cin, int Cout, int D, int H, int W,
                     int mode, int kdepth, int flags, dmvs_stream_t stream);

/* Introspection (tests, tooling): which workgroup tile dmvs_conv3d_mfma would launch for this layer and input size.
 * Returns TZ * 256 + TY (tile rows along depth / height; every tile is 32 voxels wide), bit 16 set when the 16-byte
 * tile loader is eligible (W % 4 == 0; it additionally needs a 16-byte aligned input), bit 17 for the M-block-split
 * variant of a two-block (Cout = 64) layer on a small volume, or DMVS_EUNSUPPORTED.  The
 * big tiles (TY >= 8 for 3D stride-1, >= 4 for 3D stride-2, 16 / 8 for per-slice layers) are the ones the full-size
 * configs run; the parity tests use this to prove they exercise them. */
int dmvs_conv3d_mfma_plan(int Cin, int Cout, int D, int H, int W, int mode, int kdepth);

/* K3w: the stride-1 3x3(x3) layers of K3 in Winograd F(2x2, 3x3) form on the same fp32 matrix cores
 * (csrc/conv3d_wino.hip): 2.25x fewer multiplies than the direct form, fp32 inputs / products / sums, result equal to
 * dmvs_conv3d_mfma's at re-association level.  Same operator and layouts as dmvs_conv3d_mfma(mode DMVS_CONV_S1) without
 * residual:  out = relu(conv(in) * scale + shift)  (module.py:120-157; layers module.py:364, 367, 370, 406, 409, 412
 * and FeatureNet's module.py:291-292, 296-297, 309-310).  flags: DMVS_RELU, DMVS_OUT_Q4.
 *   w_packed: dmvs_pack_conv_weights_wino (host) -- G g G^T of every (cout, cin, kz) filter, formed in double, in the
 *   kernel's consumption order; dmvs_conv3d_wino_weight_floats gives its length, 0 for a layer shape not compiled.
 * Needs W % 4 == 0 and 16-byte aligned in / out, otherwise DMVS_EUNSUPPORTED (the caller then runs dmvs_conv3d_mfma). */
int dmvs_conv3d_wino(const float* in, float* out, const float* w_packed, const float* scale, const float* shift,
                     int Cin, int Cout, int D, int H, int W, int kdepth, int flags, dmvs_stream_t stream);
/* Introspection / dispatch policy: the number of workgroups dmvs_conv3d_wino would launch for this layer and input size
 * (the host uses it to leave volumes of a few dozen workgroups to dmvs_conv3d_mfma), or DMVS_EUNSUPPORTED. */
int dmvs_conv3d_wino_plan(int Cin, int Cout, int D, int H, int W, int kdepth);
long dmvs_conv3d_wino_weight_floats(int Cin, int Cout, int kdepth);
int dmvs_pack_conv_weights_wino(const float* w /* [Cout][Cin][kd][3][3] */, float* out, int Cin, int Cout, int kdepth);

/* K3r: the coarse-level stride-1 3x3(x3) layers -- conv4 (32 -> 32) and conv6 (64 -> 64) of CostRegNet_part / _part_refine
 * (module.py:367, 370, 409, 412; Conv3d / Conv2d + BatchNorm(eval) + ReLU, module.py:120-157) and their 2D forms (kdepth 1:
 * the refine nets' bottleneck, and the middle 3x3 slice of a 3D layer on a depth-1 volume) -- in Winograd F(2x2, 3x3) form
 * with REGISTER-STATIONARY filters (csrc/conv3d_coarse.hip): 256 persistent 512-thread workgroups, each wave keeps its
 * share of G g G^T in VGPRs for the whole launch and walks 8 x 8-output groups of the volume; input tiles through a ring of
 * LDS stages.  Same operator, layouts and fp32 arithmetic as dmvs_conv3d_wino without DMVS_OUT_Q4 / residual:
 *   out = relu(conv(in) * scale + shift), in [Cin][D][H][W], out [Cout][D][H][W].  flags: DMVS_RELU.
 *   w_packed: dmvs_pack_conv_weights_coarse (host); dmvs_conv3d_coarse_weight_floats = its length, 0 for a shape not compiled
 *   ((32,32) and (64,64), kdepth 3 or 1).  Any W (W % 4 != 0 or an unaligned base: dword tile loads).
 * DMVS_EUNSUPPORTED: shape not compiled or a tensor of >= 2^29 elements (the caller then runs dmvs_conv3d_wino / _mfma). */
int dmvs_conv3d_coarse(const float* in, float* out, const float* w_packed, const float* scale, const float* shift,
                       int Cin, int Cout, int D, int H, int W, int kdepth, int flags, dmvs_stream_t stream);
long dmvs_conv3d_coarse_weight_floats(int Cin, int Cout, int kdepth);
int dmvs_pack_conv_weights_coarse(const float* w /* [Cout][Cin][kd][3][3] */, float* out, int Cin, int Cout, int kdepth);

/* K3z: conv2 of the regularisation nets (module.py:364, 406: Conv3d 16 -> 16, 3x3x3, stride 1) in Winograd F(2x2, 3x3) form with
 * register-stationary filters, marching along z (csrc/conv3d_zmarch.hip): a 256-thread workgroup = the 4 transform rows, each
 * wave keeps its share of G g G^T for all 3 depth taps in VGPRs; one pipeline stage = one input plane of an 8 x 8-output column,
 * transformed once and used as depth tap 0 / 1 / 2 of three output planes; 3 persistent workgroups per CU.
 *   out = relu(conv(in) * scale + shift), in [16][D][H][W], out [16][D][H][W].  flags: DMVS_RELU.
 *   w_packed: dmvs_pack_conv_weights_zmarch (host); dmvs_conv3d_zmarch_weight_floats = its length, 0 for a shape not compiled
 *   (only (16, 16, kdepth 3)).  Needs W % 4 == 0 and 16-byte aligned tensors.
 * DMVS_EUNSUPPORTED: shape / alignment not covered or >= 2^29 elements (the caller then runs dmvs_conv3d_wino / _mfma). */
int dmvs_conv3d_zmarch(const float* in, float* out, const float* w_packed, const float* scale, const float* shift,
                       int Cin, int Cout, int D, int H, int W, int kdepth, int flags, dmvs_stream_t stream);
long dmvs_conv3d_zmarch_weight_floats(int Cin, int Cout, int kdepth);
int dmvs_pack_conv_weights_zmarch(const float* w /* [Cout][Cin][3][3][3] */, float* out, int Cin, int Cout, int kdepth);

/* K3b -- a PROBE outside the product path (csrc/conv3d_split.hip): conv1 of the regularisation nets
 * (module.py:363, 405: Conv3d 8 -> 16, 3x3x3, stride 2, BN + ReLU) with the fp32 operands split into bf16 terms
 * (x = h + m + l exactly) and multiplied on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: terms = 6 keeps hh + hm + mh + hl +
 * mm + lh (emulated fp32: error of the order of fp32 rounding), terms = 3 keeps hh + hm + mh (2^-16).
 *   in [8][D][H][W] -> out [16][(D+1)/2][(H+1)/2][(W+1)/2] = relu(conv(in) * scale + shift); flags: DMVS_RELU.
 *   w_split: dmvs_pack_conv_weights_split of the weight [16][8][3][3][3] (dmvs_conv3d_split_weight_floats 32-bit words of raw
 *   bf16 pairs; 0 for any other (Cin, Cout)).  Never selected by the product's dispatch. */
int dmvs_conv3d_split_probe(const float* in, float* out, const float* w_split, const float* scale, const float* shift,
                            int D, int H, int W, int terms, int flags, dmvs_stream_t stream);
long dmvs_conv3d_split_weight_floats(int Cin, int Cout);
int dmvs_pack_conv_weights_split(const float* w, float* out, int Cin, int Cout);

/* K3s: FeatureNet's two full-resolution layers (module.py:283-286: conv0 = Conv2d(3 -> 8) + Conv2d(8 -> 8), 3x3, stride 1,
 * pad 1, BN + ReLU) as a register-only row sweep on v_mfma_f32_4x4x1_16b_f32 (csrc/conv2d_c8.hip): with 8 output channels
 * the 16-row MFMA of dmvs_conv3d_mfma runs half empty; here one issue is 4 output channels x 64 pixels, a wave walks a
 * 62-pixel column strip down the image with the stencil's three rows in registers -- no LDS, no barrier.
 *   out[8][V][H][W] = relu(conv3x3(in) * scale + shift);  in: [Cin][V][H][W] planar, or with DMVS_IN_VIEWS (Cin = 3 only)
 *   the loader's image stack [V][3][H][W] read in place.  Cin in {3, 8}; flags: DMVS_RELU, DMVS_IN_VIEWS.
 *   w_packed: dmvs_pack_conv_weights_c8 of the nn.Conv2d weight [8][Cin][3][3] (dmvs_conv2d_c8_weight_floats floats; 0 for
 *   a Cin not compiled).  DMVS_EUNSUPPORTED beyond 2^28 output elements (the caller then runs dmvs_conv3d_mfma). */
int dmvs_conv2d_c8(const float* in, float* out, const float* w_packed, const float* scale, const float* shift,
                   int Cin, int V, int H, int W, int flags, dmvs_stream_t stream);
/* ... and both layers in ONE sweep (module.py:283-286 as a whole): out[8][V][H][W] = conv0.1(conv0.0(imgs)), the 8-channel
 * intermediate kept in registers (never stored: 2 x 303 MB per depth map at config 2).  imgs: the loader's [V][3][H][W] stack;
 * w0 / w1: dmvs_pack_conv_weights_c8 of the two weights (Cin = 3, 8); scale / shift: folded BN of each layer (required); both
 * layers end in ReLU.  Same result as two dmvs_conv2d_c8 calls, bit for bit. */
int dmvs_featurenet_conv0(const float* imgs, float* out, const float* w0_packed, const float* scale0, const float* shift0,
                          const float* w1_packed, const float* scale1, const float* shift1, int V, int H, int W,
                          dmvs_stream_t stream);
long dmvs_conv2d_c8_weight_floats(int Cin);
int dmvs_pack_conv_weights_c8(const float* w /* [8][Cin][3][3] */, float* out, int Cin);

/* FeatureNet's level-3 top-down merge (module.py:333-336) as ONE Winograd convolution (csrc/conv3d_wino.hip,
 * fpn_wino_kernel):  out = conv3x3(intra),  intra[k] = b_lat[k] + sum_j w_lat[k][j] * lat[j] + td[k] upsampled x2 (nearest),
 * zero padded -- inner2 (1x1 lateral conv + bias), the x2 upsample + add and out3 in one kernel; the 32-channel
 * full-resolution `intra` tensor is never formed.  The 1x1 lateral conv and its bias are folded into composite 3x3 filters
 * on the host (the bias as a filter on a constant-one image, so that it does not leak into the zero padding), and the
 * x2-upsampled top-down tensor needs only 9 of the 16 transform positions.  kdepth = 1 layout: lat [8][D][H][W],
 * td [32][D][H/2][W/2], out [16][D][H][W] (or its DMVS_OUT_Q4 halves).  Compiled for (Cl, Cin, Cout) = (8, 32, 16);
 * H even, W % 8 == 0, 16-byte aligned tensors, otherwise DMVS_EUNSUPPORTED (the caller then runs inner2 and out3 as two
 * layers; r01-r03's VALU-built fused variants dmvs_conv3d_mfma_fpn / dmvs_conv3d_wino_fpn were removed in r04).
 *   w_packed: dmvs_pack_conv_weights_wino_fpn(w3 [16][32][3][3], w_lat [32][8], b_lat [32]) -- length
 *   dmvs_conv3d_wino_fpn_weight_floats(); ones_hw: H*W floats of 1.0 on the device. */
int dmvs_conv3d_wino_fpn2(const float* lat, const float* td, const float* ones_hw, float* out, const float* w_packed,
                          const float* scale, const float* shift, int D, int H, int W, int flags, dmvs_stream_t stream);
long dmvs_conv3d_wino_fpn_weight_floats(void);
int dmvs_pack_conv_weights_wino_fpn(const float* w3, const float* w_lat, const float* b_lat, float* out);

/* number of floats dmvs_conv3d_mfma expects in w_packed for a layer (host helper). */
long dmvs_conv3d_mfma_weight_floats(int Cin, int Cout, int mode, int kdepth);
/* host-side packing: w is the PyTorch weight ([Cout][Cin][kd][3][3], or [Cin][Cout][kd][3][3]
 * for DECONV), kd = kdepth.  Both pointers are HOST pointers. */
int dmvs_pack_conv_weights_mfma(const float* w, float* w_packed, int Cin, int Cout, int mode, int kdepth);

/* K4: dual-depth regression.  Replaces DepthNet.forward (mvsnet.py:15-66) when mode = 0 and
 * DepthNet.refine (mvsnet.py:67-100) when mode = 1; softmax over D of alpha*logits, expectation,
 * min/max of the (small, huge) pairs, checkerboard selection, confidence.
 *   logits_4dhw [4][D][H][W]; depth_dhw [D][H][W]; interval [1] (device scalar)
 *   dsp_4hw     [4][H][W]   depth_sub_plus
 *   sel         mode 0: [4][H][W] refine hypotheses (depth_values_c); mode 1: [H][W] final depth
 *   conf_hw     [H][W] photometric confidence
 *   prob_4dhw   optional [4][D][H][W] softmax volume (training / parity only), may be NULL */
int dmvs_depth_regress(const float* logits_4dhw, const float* depth_dhw, const float* interval,
                       float alpha, int mode, int D, int H, int W, float* dsp_4hw, float* sel,
                       float* conf_hw, float* prob_4dhw, dmvs_stream_t stream);

/* `prob` -> K4 in one kernel for the passes whose volume one workgroup can hold (r06; module.py:379, 397 + mvsnet.py:19-20,
 * 68-69): the branch's `prob` head (Conv3d 8 -> 2, as dmvs_conv3d_direct runs it) followed by the softmax over D of alpha * logits
 * and the depth expectation of its two channels -- the [2][D][H][W] logits are never written.  The same operations in the same order
 * as dmvs_conv3d_direct + dmvs_depth_regress: bit-identical expectations.
 *   in [Cin][D][H][W]; w_packed as for dmvs_conv3d_direct (Cout = 2); hyp_dhw [D][H][W], or NULL: plane d = base_hw + d * step[0]
 *   dsp_2hw [2][H][W]: the branch's slice of depth_sub_plus (small: channels 0-1, huge: 2-3)
 * D must be 4 or 8, W % 4 == 0, `in` 16-byte aligned, Cin even and <= 16; otherwise DMVS_EUNSUPPORTED (callers fall back to the two
 * kernels). */
int dmvs_prob_regress(const float* in, const float* w_packed, int Cin, int D, int H, int W, const float* hyp_dhw,
                      const float* base_hw, const float* step, float alpha, float* dsp_2hw, dmvs_stream_t stream);
/* ... and K4's remainder on the four expectations (mvsnet.py:22-61 mode 0, 72-97 mode 1): min / max of the (small, huge) pairs,
 * checkerboard selection, confidence.  dsp_4hw [4][H][W] in; sel / conf_hw as dmvs_depth_regress. */
int dmvs_depth_select(const float* dsp_4hw, const float* interval, int mode, int H, int W, float* sel, float* conf_hw,
                      dmvs_stream_t stream);

/* K4 on affine hypotheses: depth_dhw is replaced by base_hw [H][W]; plane d = base + d * interval[0]. */
int dmvs_depth_regress_affine(const float* logits_4dhw, const float* base_hw, const float* interval,
                              float alpha, int mode, int D, int H, int W, float* dsp_4hw, float* sel,
                              float* conf_hw, float* prob_4dhw, dmvs_stream_t stream);

/* N4: geometric-consistency check of one (reference, source) depth-map pair -- the inner step of the fusion
 * filter.  Replaces reproject_with_depth_pytorch + check_geometric_consistency (filter/pcd.py:151-242).
 *   depth_ref, depth_src [H][W]; proj33: 33 floats folded on the host from the two cameras:
 *     [0..11]  A1, b1: K_src*xyz_src = A1*(x,y,1)*d_ref + b1          (A1 = K_s R_rel K_r^-1, b1 = K_s t_rel)
 *     [12..23] A2, t2: xyz_reprojected = A2*(xs,ys,1)*d_sampled + t2  (A2 = R_rel^-1.. K_s^-1, in the ref camera)
 *     [24..32] K_ref
 *   pixel kept iff |reprojection - pixel| < dist_thresh and |d_reproj - d_ref| / d_ref < rel_thresh.
 *   mask [H][W] u8 and depth_reproj [H][W] (0 where rejected) are written; vote_sum [H][W] i32 and depth_sum
 *   [H][W] are ACCUMULATED (the per-pixel sums filter_depth builds over the source views, pcd.py:283-300).
 *   Any of the four outputs may be NULL. */
int dmvs_geo_consistency(const float* depth_ref, const float* depth_src, const float* proj33, int H, int W,
                         float dist_thresh, float rel_thresh, unsigned char* mask, float* depth_reproj,
                         int* vote_sum, float* depth_sum, dmvs_stream_t stream);

/* N4, dynamic-threshold variant (filter/dypcd_tanks.py:164-184, the Tanks&Temples filter): the same reprojection,
 * judged by nine gates  dist < i * dist_base  and  |d_reproj - d_ref| / d_ref < i * rel_base,  i = 2..10.
 *   level_votes [9][H][W] i32, ACCUMULATED: gate i adds to level_votes[i-2] (geo_mask_sums, dypcd_tanks.py:236-252)
 *   mask / depth_reproj / vote_sum / depth_sum: as in dmvs_geo_consistency, for the last gate (i = 10)
 * Zero reference depths are not patched here (the reference divides by them: every gate is false).  Outputs may be NULL. */
int dmvs_geo_consistency_ladder(const float* depth_ref, const float* depth_src, const float* proj33, int H, int W,
                                float dist_base, float rel_base, int* level_votes, unsigned char* mask,
                                float* depth_reproj, int* vote_sum, float* depth_sum, dmvs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DMVS_H */
