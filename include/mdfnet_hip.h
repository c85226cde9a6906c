/*
 * mdfnet_hip.h -- C ABI of the MI355X (gfx950) kernels behind MDF-Net's multi-stage MVS hot path.
 *
 * The reference (zongh5a/MDF-Net) has no FFI on this path: every operator is a chain of stock
 * PyTorch ops.  Each entry point below replaces one such chain; the reference interface it
 * replaces is cited as file:line (paths relative to the reference root).
 *
 * Conventions (all entry points)
 *   - Plain pointers and sizes only.  Every `const float*` / `float*` is DEVICE memory unless the
 *     parameter comment says HOST.  The caller (PyTorch) owns every buffer; the library never
 *     allocates, frees or retains device memory.
 *   - Asynchronous: work is enqueued on `stream` (a hipStream_t passed as void*); no implicit
 *     synchronisation.  Re-entrant; callable from several host threads / one process per GPU.
 *   - Return 0 on success; <0 on failure: MDF_EARG (-1) bad argument/shape, MDF_EUNSUPPORTED (-2)
 *     configuration not built, MDF_EHIP (-3) HIP runtime error.  mdf_last_error() returns a
 *     thread-local message valid until the next call on that thread.  Never aborts.
 *   - fp32 everywhere.  "Bit-exact" below means bit-identical to torch-2.10 CPU on the same inputs.
 */
#ifndef MDFNET_HIP_H
#define MDFNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDF_ABI_VERSION 1

#define MDF_OK 0
#define MDF_EARG (-1)
#define MDF_EUNSUPPORTED (-2)
#define MDF_EHIP (-3)

/* feature-map layouts ([B,C,h,w] logical) */
#define MDF_FEA_NCHW 0
#define MDF_FEA_NHWC 1
/* volume layouts ([B,C,D,h,w] logical) */
#define MDF_VOL_NCDHW 0
#define MDF_VOL_NDHWC 1

#define MDF_MAX_SRC_VIEWS 16

int mdf_abi_version(void);
const char* mdf_last_error(void);
/* Name of the __global__ function the calling thread's most recent entry enqueued last ("" before the first launch): lets a
 * profiler attribute an entry's algorithmic bytes / flops to the kernel that served it (bench.py: roofline.per_kernel).      */
const char* mdf_last_launch(void);
/* The conv kernels keep one device-side work-item counter pair per HIP stream that has launched them (128 slots per
 * process; launches on one stream are ordered, launches on different streams may overlap).  A long-lived process that
 * destroys streams hands their slots back with this call (before hipStreamDestroy, with no conv launch of this library
 * still running on the stream).  Always returns MDF_OK; unknown streams are ignored.                               */
int mdf_release_stream(void* stream);

/* ---- a4  homo_warping (net/unit/base.py:85-126) ---------------------------------------------
 * Plane-sweep homography warp of ONE source feature map; bilinear, zero padding, the reference's
 * mixed align_corners convention (ix = px*w/(w-1) - 0.5).  Output values are bit-exact.
 *   src_fea  [B,h,w,C] (MDF_FEA_NHWC) ; C in {16,32,64}
 *   proj     [B,12]  rows of (src_proj @ inverse(ref_proj))[:3,:4], computed by the host with the
 *            same torch call as base.py:98 so kernel and oracle consume identical floats
 *   hypos    [B,D] (hypos_per_pixel=0; reference shape [B,D,1,1]) or [B,D,h,w] (=1)
 *   out      [B,C,D,h,w] (MDF_VOL_NCDHW) or [B,D,h,w,C] (MDF_VOL_NDHWC)                         */
int mdf_homo_warp_fwd(const float* src_fea, int fea_layout, const float* proj, const float* hypos,
                      int hypos_per_pixel, float* out, int out_layout, int B, int C, int D, int h, int w,
                      void* stream);

/* Test hook for "indexing bit-exact": integer corner (floor) of every sample position.
 *   x0y0 [B,D,h,w,2] int32 = (floor(ix), floor(iy)); INT32_MIN for non-finite positions,
 *   clamped to +-2^30.  Same device function as the kernels above/below.                        */
int mdf_warp_corner_indices(const float* proj, const float* hypos, int hypos_per_pixel, int32_t* x0y0,
                            int B, int D, int h, int w, void* stream);

/* ---- a5  VectorAggregate.forward, eval mode (net/unit/homoaggregate.py:25-46) ----------------
 * Fused warp + group softmax + similarity + learned view weight + weighted mean over views.
 * Fills CoreNet's Homoaggre[s] slot (net/core.py:58).  C/G must be 2 (config.py:196,205).
 *   ref_fea   [B,h,w,C] NHWC
 *   src_feas  HOST array of n_src DEVICE pointers, each [B,h,w,C] NHWC
 *   proj      [n_src,B,12]
 *   w_params  [G+4] = depth_weight.0.conv.weight[G], alpha, beta, w2, b2 where BatchNorm3d(1) is
 *             folded as ATen does in eval: alpha = gamma/sqrt(var+eps), beta = bias - mean*alpha
 *   cost      [B,G,D,h,w] (NCDHW) or [B,D,h,w,G] (NDHWC)                                         */
int mdf_warp_aggregate_vec_fwd(const float* ref_fea, const float* const* src_feas, int fea_layout,
                               const float* proj, const float* hypos, int hypos_per_pixel,
                               const float* w_params, float* cost, int cost_layout, int B, int C, int G,
                               int D, int h, int w, int n_src, void* stream);

/* ---- a5' homo_aggregate_by_variance (net/unit/homoaggregate.py:49-69) ------------------------
 * var = E[x^2] - E[x]^2 over {ref, softmax_C(warped src_v)}.  cost has C channels.               */
int mdf_warp_aggregate_var_fwd(const float* ref_fea, const float* const* src_feas, int fea_layout,
                               const float* proj, const float* hypos, int hypos_per_pixel, float* cost,
                               int cost_layout, int B, int C, int D, int h, int w, int n_src, void* stream);

/* ---- a6/a7/a8  3-D convolution layers of the regularisers (net/unit/regular.py:9-133,
 *      net/unit/base.py:50-68): Conv3d / ConvTranspose3d, k=3, pad=1, (output_padding=1), no bias,
 *      optional folded BatchNorm3d (alpha,beta per output channel), ReLU and residual add, in that
 *      order:  y = [res +] [relu] (conv(x) * alpha + beta).  fp32 MFMA (exact f32 fma chain).
 *   x        [B,Di,Hi,Wi,Cin]  NDHWC
 *   wpack    weights pre-packed by mdf_conv3d_pack_weights (device)
 *   alpha,beta [Cout] or NULL (no BN);  res [B,Do,Ho,Wo,Cout] or NULL
 *   y        [B,Do,Ho,Wo,Cout] NDHWC.   stride in {1,2}; transposed in {0,1} (transposed => stride 2)
 *   Cin,Cout in {8,16,32,64}                                                                      */
int mdf_conv3d_fwd(const float* x, const float* wpack, const float* alpha, const float* beta, const float* res,
                   float* y, int B, int Di, int Hi, int Wi, int Cin, int Cout, int stride, int transposed,
                   int relu, void* stream);

/* Packed-weight size in floats for a (Cin,Cout) k=3 layer. */
int64_t mdf_conv3d_packed_size(int Cin, int Cout);
/* Re-order torch weights into the MFMA B-fragment order (device -> device, on `stream`).
 *   w: Conv3d [Cout,Cin,3,3,3] (transposed=0) or ConvTranspose3d [Cin,Cout,3,3,3] (transposed=1). */
int mdf_conv3d_pack_weights(const float* w, float* wpack, int Cin, int Cout, int transposed, void* stream);

/* ---- a11/a12  2-D convolution layers of the feature pyramid and refinement net (net/unit/backbone.py:17-45,
 *      net/unit/refine.py:13-21, net/unit/base.py:7-25,71-82): Conv2d k in {1,3,5}, pad (k-1)/2, stride {1,2},
 *      y = [res + res_scale *] ( [up2(res_up) +] [relu]( conv(x) * alpha + beta ) )
 *      (alpha NULL: beta is the conv bias or NULL; res_up [B,Ho/2,Wo/2,Cout]: its bilinear x2 upsample,
 *      align_corners=False, is added -- the FPN top-down step backbone.py:60,62 fused into the lateral conv).
 *   x [B,H,W,Cin] NHWC, or planar [B,Cin,H,W] when planar_in != 0 (Cin < 4 only: the RGB images as eval.py hands
 *   them over, so no layout copy is needed); y [B,Ho,Wo,Cout] NHWC; wpack from mdf_conv_pack_weights (Cin 3 or 1 is
 *   zero-padded to 4).  pixel_shuffle2 != 0 (Cout = 32, or 32 -> 64 with k = 3; stride 1, no residual): y is nn.PixelShuffle(2)
 *   of the result, [B,2Ho,2Wo,Cout/4] NHWC (refine.py:19), and the weight rows must have been packed sub-pixel-major: row
 *   (dy*2+dx)*Cout/4 + oc = torch output channel oc*4 + dy*2 + dx.  (Training uses the same store for the input gradient of the
 *   k5-s2 pyramid layers, a 3x3 conv over dy whose output rows are the four parity classes of dx.)                       */
int mdf_conv2d_fwd(const float* x, const float* wpack, const float* alpha, const float* beta, const float* res,
                   float res_scale, const float* res_up, float* y, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int relu,
                   int planar_in, int pixel_shuffle2, void* stream);
int64_t mdf_conv_packed_size(int Cin, int Cout, int ntaps);
/* w: [Cout,Cin,k,k] (ntaps = k*k) or [Cout,Cin,3,3,3] (ntaps = 27), device -> device. */
int mdf_conv_pack_weights(const float* w, float* wpack, int Cin, int Cout, int ntaps, void* stream);

/* `prob` head: Conv3d(Cin->1,k3,p1,no bias) + softmax over D [+ soft-argmin]  (regular.py:43,69 /
 * :110,133 and net/unit/regress.py:5-7).  x NDHWC [B,D,h,w,Cin]; w [Cin*27] torch order [1,Cin,3,3,3];
 *   prob [B,D,h,w]; depth [B,h,w] or NULL; hypos as above (may be NULL when depth is NULL).       */
int mdf_prob_softmax_regress_fwd(const float* x, const float* w, const float* hypos, int hypos_per_pixel,
                                 float* prob, float* depth, int B, int D, int h, int wd, int Cin, void* stream);
/* Same head in partial-sum form (the fast path): first run mdf_conv2d_fwd over the B*D planes of x ([B*D,h,w,Cin]) with the
 * 2-D weight w2[kd][cin][kh][kw] = w[0][cin][kd][kh][kw], kd = 0..2, plus a 4th all-zero output channel (Cout = 4),
 * giving partials [B,D,h,w,4]; this call then forms logit[d] = P0[d-1] + P1[d] + P2[d+1], the softmax over D and the
 * soft-argmin.  D <= 96.  partials 16-byte aligned.                                                               */
int mdf_prob_from_partials_fwd(const float* partials, const float* hypos, int hypos_per_pixel, float* prob, float* depth,
                               int B, int D, int h, int wd, void* stream);
/* The partial-sum form in ONE launch: a block owns a 4 x 64 pixel tile, walks the depth axis with the MFMA step of that 2-D conv
 * and keeps the partials in registers (they never reach memory).  wpack = mdf_conv_pack_weights of the 4-channel 2-D weight above
 * (Cin_mem = Cin, Cout = 4, 9 taps).  Cin in {8,16}.  Bit-identical to mdf_conv2d_fwd + mdf_prob_from_partials_fwd; parallel over
 * pixels only, so for maps that fill the chip (B*ceil(h/4)*ceil(wd/64) blocks).                                            */
int mdf_prob_fused_fwd(const float* x, const float* wpack, const float* hypos, int hypos_per_pixel, float* prob, float* depth,
                       int B, int D, int h, int wd, int Cin, void* stream);

/* ---- tail of the refinement net as one launch (net/unit/refine.py:18-20,42-44):
 *      y = [lo +] conv2(PixelShuffle(2)(conv1(x))) [* span]     Conv2d(8,32,k3,p1,no bias) -> PixelShuffle(2) -> Conv2d(8,1,k3,p1,no bias)
 *      x NHWC [B,H,W,8]; w1pack = mdf_conv_pack_weights of conv1's weight with its output rows in PixelShuffle order
 *      (row sub*8 + oc <- channel oc*4 + sub); w2 = conv2's weight [1,8,3,3]; lo / span [B] or both NULL (the range mapping
 *      lo + y * span with torch's roundings); y [B,2H,2W].  The 8-channel map at twice the resolution never reaches memory.   */
int mdf_refine_tail_fwd(const float* x, const float* w1pack, const float* w2, const float* lo, const float* span, float* y,
                        int B, int H, int W, void* stream);

/* ---- the two full-resolution layers of the feature pyramid as one launch (net/unit/backbone.py:28, eval):
 *      y = relu(bn2(conv2(relu(bn1(conv1(x))))))   Conv2d(3,8,k3,p1) -> Conv2d(8,8,k3,p1), BatchNorm folded to (alpha, beta);
 *      x planar [N,3,H,W] as the loader hands it over, y NHWC [N,H,W,8]; w1pack / w2pack = mdf_conv_pack_weights of the two
 *      layers.  Bit-identical to the two mdf_conv2d_fwd launches; the 8-channel map between them never reaches memory.    */
int mdf_conv2d_pair_fwd(const float* x, const float* w1pack, const float* alpha1, const float* beta1, const float* w2pack,
                        const float* alpha2, const float* beta2, float* y, int N, int H, int W, void* stream);

/* ---- several 1x1 heads over one input as one launch (the feature pyramid's composed heads, net/unit/backbone.py:58-66, eval):
 *      y[h] = [up2(res_ups[h]) +] W_h x + bias_h  for h < n_heads (2 or 3); x NHWC [B,H,W,Cin], y[h] NHWC [B,H,W,couts[h]],
 *      res_ups[h] [B,H/2,W/2,couts[h]] or NULL (bilinear x2, align_corners=False, added BEFORE nothing else: torch's order
 *      up + conv), biases[h] [couts[h]] or NULL, wpacks[h] = mdf_conv_pack_weights of head h (one tap).  The arrays are HOST arrays
 *      of device pointers.  Built: Cin 64 -> (64, 32, 16) and Cin 32 -> (32, 16); every output bit-identical to mdf_conv2d_fwd.  */
int mdf_conv1x1_heads_fwd(const float* x, int n_heads, const float* const* wpacks, const float* const* biases,
                          const float* const* res_ups, float* const* ys, const int* couts, int B, int H, int W, int Cin,
                          void* stream);

/* ---- head of the refinement net as one launch (net/unit/refine.py:29,36, eval): y = conv0((depth - lo[b]) / span[b]),
 *      conv0 = Conv2d(1,8,k3,p1,no bias) with `weight` [8,1,3,3] as the module holds it; depth [B,H,W], y NHWC [B,H,W,8];
 *      lo / span [B] or both NULL (no mapping).  torch's roundings (sub, true divide); bit-identical to
 *      mdf_range_affine_fwd(mode 0) + mdf_conv2d_fwd.                                                                     */
int mdf_refine_head_fwd(const float* depth, const float* lo, const float* span, const float* weight, float* y, int B, int H, int W,
                        void* stream);

/* ---- a residual block of the refinement net as one launch (net/unit/base.py:39-47 `Res`, net/unit/refine.py:29,40, eval):
 *      y = x + scale * conv_b(relu(conv_a(x)))   both Conv2d(8,8,k3,p1,no bias); x, y NHWC [N,H,W,8], y != x;
 *      wa_pack / wb_pack = mdf_conv_pack_weights of the two layers.  Bit-identical to the two mdf_conv2d_fwd launches
 *      (relu; then res + scale * conv); the intermediate map never reaches memory.                                      */
int mdf_conv2d_res_pair_fwd(const float* x, const float* wa_pack, const float* wb_pack, float scale, float* y, int N, int H,
                            int W, void* stream);

/* ---- a9  depth_regression (net/unit/regress.py:5-7): depth = sum_d prob*hypos ----------------- */
int mdf_depth_regress_fwd(const float* prob, const float* hypos, int hypos_per_pixel, float* depth, int B, int D,
                          int h, int w, void* stream);

/* ---- a10 confidence_regress (net/unit/regress.py:9-25), n=4, pad=(1,2) ------------------------
 *   conf [B,h,w] = sum prob[idx-1..idx+2], idx = (int64) trunc(sum_d prob_d*d); idx_out int64 or NULL */
int mdf_confidence_fwd(const float* prob, float* conf, int64_t* idx_out, int B, int D, int h, int w, void* stream);
/* the same with the nearest x2 upsampling of net/core.py:76 folded in: conf2 [B,2h,2w]                            */
int mdf_confidence_up2_fwd(const float* prob, float* conf2, int B, int D, int h, int w, void* stream);
/* range mapping around the refinement net (net/unit/refine.py:29,44), per batch item b over n elements:
 *   mode 0: y = (x - lo[b]) / span[b]      mode 1: y = lo[b] + x * span[b]     (torch's separate roundings kept)   */
int mdf_range_affine_fwd(const float* x, const float* lo, const float* span, int mode, float* y, int B, long long n, void* stream);

/* ---- a3  HyposByFit.forward (net/unit/depthhypos.py:27-76) -------------------------------------
 * mode 1 = gauss1 (depthhypos.py:169-215) with hypotheses shared by all pixels: `fit_row` is
 *          row 0 of (X^T X)^-1 X^T, [B,D], computed by the host with the reference's own torch
 *          calls (the 3x3 normal matrix has cond ~1e14 in fp32);
 * mode 2 = laplace (depthhypos.py:78-125), per-pixel or shared hypotheses.
 * Step 1: s [B,h,w].  Step 2: bilinear x2 upsample of s and depth (align_corners=False), range
 * from the curve, clamps, D_out hypotheses per pixel -> hypos_out [B,D_out,2h,2w].
 *   range [B,2] = (depth_min, depth_max) as float32;  log_thresh = f32 ln(prob_thresh), computed by
 *   the host with torch.log (depthhypos.py:55,57) so it carries the reference's rounding.          */
int mdf_hypos_fit_fwd(int mode, const float* prob, const float* depth, const float* hypos, int hypos_per_pixel,
                      const float* fit_row, float* s_out, int B, int D, int h, int w, void* stream);
int mdf_hypos_from_fit_fwd(int mode, const float* s, const float* depth, const float* range, float log_thresh,
                           float* hypos_out, int B, int D_out, int h, int w, int upsample, void* stream);

/* ---- N1  depth-map consistency filter + fusion (tools/filter/dynamic_filter_gpu.py:63-103,166-238) ----------
 * One fused pass over a reference view and its n_src source views: reprojection, bilinear depth gather, the nine
 * dynamic thresholds (dist < i/thre1, rel < i/thre2, i = 2..10), geo = #(count_i >= i) >= nconditions,
 * photo = conf > photo_threshold, masked average depth.  Bit-exact vs the reference's CPU arithmetic.
 *   depth_ref, conf [h,w]; src_depths HOST array of n_src DEVICE pointers [h,w];
 *   mats [n_src][68] = per view: inverse(K_ref)[9], E_src@inverse(E_ref)[16], K_src[9], inverse(K_src)[9],
 *                       E_ref@inverse(E_src)[16], K_ref[9]  (row-major, computed by the host with the reference's calls)
 *   depth_avg [h,w]; masks [3,h,w] uint8 = (photo, geo, final)
 *   view_masks [n_src,h,w] uint16 (bit i <-> threshold i+2) or NULL; rep_out [n_src,h,w] (depth_reprojected) or NULL   */
int mdf_consistency_fuse_fwd(const float* depth_ref, const float* conf, const float* const* src_depths, const float* mats,
                             int n_src, int h, int w, float photo_threshold, int nconditions, float thre1, float thre2,
                             float* depth_avg, unsigned char* masks, unsigned short* view_masks, float* rep_out, void* stream);

/* =====================================================================================================
 * Training path (BASELINE config 3; train.py:36-45 -> loss.backward()).  The reference has no explicit backward:
 * autograd differentiates the op chains cited above.  Each entry below is the hand-written forward-in-train-mode or
 * backward of one such chain; gradients match torch autograd to fp32 summation-order tolerance.
 * ===================================================================================================== */

/* ---- BatchNorm3d in batch-statistics mode + ReLU (+ residual) around a conv layer (net/unit/base.py:61-68 with
 *      nn.BatchNorm3d.training, the blocks of net/unit/regular.py:17-41,82-108).  Activations channels-last [N,C],
 *      N = B*D*H*W voxels, C in {8,16,32,64}.
 *   stats:    sums[0..C) += sum_n y[n][c], sums[C..2C) += sum_n y^2      (sums: fp64, zeroed by the caller)
 *   finalize: aux[4C] = (a = gamma*invstd, b = beta - mean*a, mean, invstd) with the biased variance; when given,
 *             running_mean/var are updated as nn.BatchNorm does (momentum, unbiased variance) and *nbt += 1
 *   apply:    z = [res +] relu(y*a + b)
 *   bwd_reduce: red[0..C) += sum dr, red[C..2C) += sum dr*xhat,  dr = dz*[y*a+b > 0], xhat = (y-mean)*invstd
 *   bwd:      dy = gamma*invstd*(dr - red0/N - xhat*red1/N);  dgamma[c] = red1, dbeta[c] = red0
 *   ngroups: the tensor is [ngroups][N][C] and every group is one call of the module with its own batch statistics
 *   (the feature pyramid runs once per view, net/core.py:42): sums/red [ngroups][2C], aux [ngroups][4C]; dgamma/dbeta [C]
 *   are summed over the groups (one module, its calls' gradients add); finalize walks the groups in order for the running
 *   statistics and adds ngroups to *nbt.                                                                             */
int mdf_bn_stats_fwd(const float* y, long long N, int C, int ngroups, double* sums, void* stream);
int mdf_bn_finalize_fwd(const double* sums, const float* gamma, const float* beta, float eps, float momentum, long long N,
                        int C, int ngroups, float* aux, float* running_mean, float* running_var,
                        long long* num_batches_tracked, int nslices, void* stream);
int mdf_bn_relu_apply_fwd(const float* y, const float* aux, const float* res, float* z, long long N, int C, int ngroups,
                          void* stream);
/* finalize + apply as one launch (what the training path uses): z = [res +] relu(y*a + b) with (a, b) derived from `sums` in
 * the kernel; aux [ngroups][4C] is written for the backward pass, the running statistics are updated as by finalize.   */
int mdf_bn_finalize_apply_fwd(const float* y, const double* sums, const float* gamma, const float* beta, float eps, float momentum,
                              const float* res, float* z, float* aux, float* running_mean, float* running_var,
                              long long* num_batches_tracked, long long N, int C, int ngroups, int nslices, void* stream);
int mdf_bn_relu_bwd_reduce(const float* dz, const float* y, const float* aux, long long N, int C, int ngroups, double* red,
                           void* stream);
int mdf_bn_relu_bwd(const float* dz, const float* y, const float* aux, const double* red, const float* gamma, long long N,
                    int C, int ngroups, float* dy, float* dgamma, float* dbeta, int nslices, void* stream);
/* (`nslices`: the sums / reductions these three read may be spread over nslices copies [slice][group][2C] whose totals count --
 *  that is how the conv epilogues of mdf_conv*_train_fwd deliver them; 1 for mdf_bn_stats_fwd / mdf_bn_relu_bwd_reduce.) */

/* ---- weight gradient of nn.Conv3d(k3,p1,stride s) / nn.ConvTranspose3d(k3,s2,p1,op1) (net/unit/regular.py:17-43,
 *      80-110) as ONE correlation on the fp32 matrix cores:
 *          dw[a][b][tap] (+)= sum_o small[o][a] * big[stride*o + tap - 1][b]        (zero outside big)
 *      Conv3d: small = dy [B,Ds,Hs,Ws,A=Cout], big = x [B,s*Ds,s*Hs,s*Ws,Bc=Cin]      -> dw = torch [Cout,Cin,3,3,3]
 *      ConvTranspose3d: small = x [.., A=Cin], big = dy [.., Bc=Cout], stride 2         -> dw = torch [Cin,Cout,3,3,3]
 *      (the `prob` conv: A = 1).  NDHWC operands; workspace: mdf_conv3d_wgrad_workspace(...) floats.            */
int64_t mdf_conv3d_wgrad_workspace(int B, int Ds, int Hs, int Ws, int A, int Bc);
int mdf_conv3d_wgrad(const float* small_, const float* big, float* dw, float* workspace, int B, int Ds, int Hs, int Ws,
                     int A, int Bc, int stride, int accumulate, void* stream);
/* The same for the 2-D layers of the feature pyramid (net/unit/backbone.py:17-45; Conv2d k in {1,3,5}, pad (k-1)/2, stride s):
 *      dw[a][b][kh][kw] (+)= sum_o small[o][a] * big[s*o + (kh,kw) - pad][b];  small = dy [B,Hs,Ws,A], big = x [B,s*Hs,s*Ws,Bc]
 *      NHWC -> dw = torch [Cout,Cin,k,k].                                                                          */
int64_t mdf_conv2d_wgrad_workspace(int B, int Hs, int Ws, int A, int Bc, int ksize);
int mdf_conv2d_wgrad(const float* small_, const float* big, float* dw, float* workspace, int B, int Hs, int Ws, int A, int Bc,
                     int ksize, int stride, int accumulate, void* stream);
/* The same in two steps, for a whole backward pass: *_partial clears dw and leaves the partial tiles in `workspace` as
 * *nslab_out slabs of dw.numel() floats; mdf_wgrad_sum_batch adds the slabs of MANY layers into their gradient tensors in one
 * launch (host arrays of njobs slab / dw pointers, slab counts and element counts; jobs travel by value in the kernel
 * arguments).  The weight gradients are not needed before the optimizer, so a training step sums them once, at the end. */
int mdf_conv3d_wgrad_partial(const float* small_, const float* big, float* dw, float* workspace, int B, int Ds, int Hs, int Ws,
                             int A, int Bc, int stride, int* nslab_out, void* stream);
int mdf_conv2d_wgrad_partial(const float* small_, const float* big, float* dw, float* workspace, int B, int Hs, int Ws, int A, int Bc,
                             int ksize, int stride, int* nslab_out, void* stream);
int mdf_wgrad_sum_batch(const float* const* slabs, float* const* outs, const int* nslabs, const int* ns, int njobs, void* stream);
/* Deferred weight-gradient launches.  Between mdf_wgrad_batch_begin() and mdf_wgrad_batch_flush(stream) the calling thread's
 * mdf_conv3d_wgrad_partial / mdf_conv2d_wgrad_partial calls that take the LDS-staged kernel are RECORDED (operands, workspace and dw
 * must stay valid until the flush; *nslab_out is final at once); the flush launches them grouped by kernel instantiation, the layers
 * of a group as ONE launch (job table in the kernel arguments).  Results are those of the separate launches bit for bit.       */
int mdf_wgrad_batch_begin(void);
int mdf_wgrad_batch_flush(void* stream);
/* (input gradients of these layers are mdf_conv3d_fwd with re-packed weights: a stride-1 conv with flipped taps and
 *  swapped channels, the transposed conv for a stride-2 conv and vice versa.) */

/* ---- training-mode conv launches whose EPILOGUE carries the BatchNorm reductions (base.py:50-68: conv -> BatchNorm(batch
 *      statistics) -> ReLU; the statistics of a layer and the two sums of its BatchNorm backward each cost a full pass over
 *      the activations as kernels of their own -- here they ride in the conv that produces the tensor):
 *        y = [res +] conv(x)   raw (no scale / shift / ReLU), same operands and layouts as mdf_conv3d_fwd / mdf_conv2d_fwd
 *        stat_mode 1  stat_out[c] += sum y[.,c],  stat_out[C+c] += sum y[.,c]^2          (the layer's own batch statistics)
 *        stat_mode 2  the launch is the INPUT-GRADIENT conv of the next layer, so y (with res = the skip gradient) is dz of the
 *                     layer whose raw conv output is stat_y and whose (a, b, mean, invstd)[C] is stat_aux:
 *                     stat_out[c] += sum dr, stat_out[C+c] += sum dr*xhat, dr = dz*[stat_y*a+b > 0], xhat = (stat_y-mean)*invstd
 *      stat_out: fp64, zero-initialised by the caller, C = Cout in {8,16,32,64}.  2-D: `ngroups` consecutive sets of B/ngroups
 *      images are separate BatchNorm groups (one per view, net/core.py:42): stat_out [ngroups][2C], stat_aux [ngroups][4C].
 *      `nslices` (1..64): stat_out is [nslices][ngroups][2C] and the sums are the TOTALS over the slices -- a launch spreads its
 *      blocks over them so that no address takes more than a few dozen same-address atomics (~11 ns each, serialised).      */
int mdf_conv3d_train_fwd(const float* x, const float* wpack, const float* res, float* y, int B, int Di, int Hi, int Wi, int Cin,
                         int Cout, int stride, int transposed, int stat_mode, const float* stat_y, const float* stat_aux,
                         double* stat_out, int nslices, void* stream);
int mdf_conv2d_train_fwd(const float* x, const float* wpack, float* y, int B, int H, int W, int Cin_mem, int Cout, int ksize,
                         int stride, int planar_in, int stat_mode, const float* stat_y, const float* stat_aux, double* stat_out,
                         int nslices, int ngroups, void* stream);

/* ---- batched weight packing: every packed weight set a training step reads (forward convs and their input-gradient
 *      convs, train.py:36-45 after optimizer.step()), written by ONE launch.  A job is one weight set in the layout of
 *      mdf_conv3d_pack_weights (is3d = 1) / mdf_conv_pack_weights (is3d = 0), read from the parameter `src` through `mode`:
 *        0 the parameter itself [Cout][Cin][taps]          1 input gradient of a stride-1 conv: [Cin][Cout], taps mirrored
 *        2 input gradient of Conv2d(k5,s2,p2) as a 3x3 conv over dy with the 4 output-parity classes as channels
 *          (aux0 = the layer's in_channels, aux1 = first row of this part; Cin = the layer's out_channels, taps = 9)
 *        3 the `prob` conv [1][Cin][3][3][3] as 2-D conv rows per depth tap (Cout = 4, taps = 9)
 *        4 rows reordered for the PixelShuffle(2) epilogue    5 the parameter read as [Cin][Cout][taps]
 *      mdf_pack_job_fill writes job `index` into a HOST table of mdf_pack_job_bytes() bytes per job and returns the number
 *      of blocks the job takes (< 0: error code); first_block = sum of the earlier jobs' blocks.  The caller uploads the
 *      table and an int32 per-block job index once; mdf_pack_batch then re-packs everything per call.               */
int64_t mdf_pack_job_bytes(void);
int64_t mdf_pack_job_fill(void* jobs_host, int index, const float* src, float* dst, int is3d, int transposed, int mode,
                          int Cin, int Cout, int ntaps, int aux0, int aux1, int first_block);
int mdf_pack_batch(const void* jobs_dev, const int* block_job_dev, int nblocks, void* stream);

/* ---- adjoint of the FPN's bilinear x2 upsampling (backbone.py:60,62; F.interpolate(scale_factor=2, "bilinear",
 *      align_corners=False)), NHWC: dcoarse [B,h,w,C] (+)= up^T(dfine [B,2h,2w,C]); C % 4 == 0.                  */
int mdf_upsample2_bilinear_bwd(const float* dfine, float* dcoarse, int B, int h, int w, int C, int accumulate, void* stream);

/* ---- backward of the `prob` head: softmax over D + soft-argmin (regular.py:69/133, regress.py:5-7), and the input
 *      gradient of its Conv3d(C->1,k3,p1).   dlogit [B,D,h,w]; ddepth [B,h,w] and/or dprob [B,D,h,w] (either may be
 *      NULL); w torch [1,C,3,3,3]; dx NDHWC [B,D,h,w,C], C in {8,16}.                                           */
int mdf_prob_softmax_regress_bwd(const float* prob, const float* hypos, int hypos_per_pixel, const float* ddepth,
                                 const float* dprob, float* dlogit, int B, int D, int h, int w, void* stream);
int mdf_prob_conv_dgrad(const float* dlogit, const float* w, float* dx, int B, int D, int h, int wd, int C, void* stream);
/*      The same with the BatchNorm-backward sums of the regulariser's last layer taken on the way (dx is that layer's complete dz):
 *      stat_y = the layer's raw conv output [B*D*h*wd][C], stat_aux = its (a, b, mean, invstd) [4C] as mdf_bn_finalize_fwd wrote them,
 *      red = [nslices][2C] doubles zeroed by the caller (block b adds into copy b % nslices): sum dr | sum dr * xhat, the input of
 *      mdf_bn_relu_bwd (instead of a pass of mdf_bn_relu_bwd_reduce over dx and stat_y).                                          */
int mdf_prob_conv_dgrad_stat(const float* dlogit, const float* w, float* dx, int B, int D, int h, int wd, int C, const float* stat_y,
                             const float* stat_aux, double* red, int nslices, void* stream);

/* ---- training-mode VectorAggregate fused with the warp (homoaggregate.py:16-20,25-46 with the 1-channel
 *      BatchNorm3d in batch-statistics mode -- a global reduction per source view between similarity and view weight;
 *      base.py:97: no gradient through the sampling grid).  pass:
 *        0 stats       red_out[2v], [2v+1] += sum t_v, sum t_v^2  (t_v = Conv3d(G->1)(sim_v); fp64, zeroed by caller)
 *        1 forward     cost [B,D,h,w,G] and wsum [B,D,h,w] = sum_v w_v, with per-view (alpha_v, beta_v) from `par`
 *        2 bwd-reduce  red_out[2v], [2v+1] += sum dz_v, sum dz_v*xhat_v; red_out[2n], [2n+1] += d w2, d b2; aux [n_src][B*D*h*w][4]
 *                      = (w_v, dz_v, t_v, -) per sample, which pass 3 reads instead of re-deriving them from all channels
 *        3 backward    dref [B,h,w,C] (+=; zeroed by caller), dsrc[v] [B,h,w,G] (+= by fp32 atomics; zeroed by caller; the gradient of
 *                      channel 2g of every softmax pair -- channel 2g+1 gets its negative), dcw[G] (+=)
 *      par (float): [0,G) conv weight | G: w2 | G+1: b2 | G+2: gamma | G+3: 1/N | G+4+4v: alpha_v, beta_v, mean_v,
 *      invstd_v.  Features NHWC, C in {16,32,64}, G = C/2.                                                        */
int mdf_warp_aggregate_vec_train(int pass, const float* ref_fea, const float* const* src_feas, const float* proj,
                                 const float* hypos, int hypos_per_pixel, const float* par, const double* red_in,
                                 const float* dcost, float* cost, float* wsum, double* red_out, float* dref,
                                 float* const* dsrc, float* dcw, float* aux, int B, int C, int G, int D, int h, int w,
                                 int n_src, void* stream);
/* Control plane of the passes above, on the device (homoaggregate.py:16-20: the BatchNorm3d(1) of depth_weight, called
 * once per source view, :35-40):
 *   prepare       par[0,G+4) = (conv weight, w2, b2, gamma, 1/n), par[G+4, G+4+4 n_src) = 0, red[0,nred) = 0
 *   finalize      after pass 0: par[G+4+4v..] = (alpha_v, beta_v, mean_v, invstd_v) from red; running_mean / running_var /
 *                 num_batches_tracked (any may be NULL) advanced as by n_src successive calls with `momentum`
 *   bwd_finalize  after pass 3: dsrc [n_src][B,h,w,C] from the even-channel gradients dhalf [n_src][B,h,w,G] (n_half = their
 *                 total element count), and dpar = (d gamma, d beta, d w2, d b2) from pass 2's red; when dref_acc is given,
 *                 its n_ref floats (the zero-initialised buffer pass 3 accumulated d ref into) are copied to dref           */
int mdf_aggregate_train_prepare(const float* cw, const float* w2, const float* b2, const float* gamma, long long n, int G,
                                int n_src, float* par, double* red, int nred, void* stream);
int mdf_aggregate_train_finalize(const double* red, const float* gamma, const float* beta, float eps, float momentum, long long n,
                                 int G, int n_src, float* par, float* running_mean, float* running_var,
                                 long long* num_batches_tracked, void* stream);
int mdf_aggregate_train_bwd_finalize(const float* dhalf, const double* red, int n_src, long long n_half, float* dsrc,
                                     float* dpar, const float* dref_acc, float* dref, long long n_ref, void* stream);

/* ---- training loss (net/loss.py:10-27): sum over the output scales of smooth-L1 (beta 1, mean) over the pixels with
 *      gt > depth_min[b].  est, gt [B][per_batch] float; depth_min element b at floor_[b*floor_stride], float64 when
 *      floor_f64 (as the loader hands depth_range over) else float32.
 *   reduce    acc[0] += sum of the per-pixel losses, acc[1] += number of valid pixels      (fp64, zeroed by the caller)
 *   finalize  loss[0] = sum_s acc[2s]/acc[2s+1] over nscales scales; inv_count[s] = 1/acc[2s+1]
 *   bwd       dest = dloss[0] * inv_count[0] * clamp(est - gt, -1, 1) on valid pixels, 0 elsewhere                    */
int mdf_masked_smooth_l1_reduce(const float* est, const float* gt, const void* floor_, int floor_f64, int floor_stride, int B,
                                long long per_batch, double* acc, void* stream);
int mdf_masked_smooth_l1_finalize(const double* acc, int nscales, float* loss, float* inv_count, void* stream);
int mdf_masked_smooth_l1_bwd(const float* est, const float* gt, const void* floor_, int floor_f64, int floor_stride, int B,
                             long long per_batch, const float* dloss, const float* inv_count, float* dest, void* stream);

/*      The same over all output scales in ONE launch each (Loss.forward sums four scales, net/loss.py:19-25): est / gt / dest are HOST
 *      arrays of nscales (<= 8) device pointers, per_batch a host array of element counts; acc is [2*nscales] as above; a NULL dest[i]
 *      skips scale i; inv_count is the [nscales] array mdf_masked_smooth_l1_finalize wrote.                                              */
int mdf_masked_smooth_l1_reduce_multi(const float* const* est, const float* const* gt, const long long* per_batch, int nscales,
                                      const void* floor_, int floor_f64, int floor_stride, int B, double* acc, void* stream);
int mdf_masked_smooth_l1_bwd_multi(const float* const* est, const float* const* gt, const long long* per_batch, int nscales,
                                   const void* floor_, int floor_f64, int floor_stride, int B, const float* dloss,
                                   const float* inv_count, float* const* dest, void* stream);

/* ---- feature-pyramid heads in training mode, the small algebra (net/unit/backbone.py:59-63: lat2, lat3, out2, out3, out4 -- 1x1 convs
 *      around two bilinear up-samplings, evaluated through the products O2 L2, O2 L3, O3 L3; train_ops.py:FPNHeadsComposedFn).
 *      Matrices are the Conv2d weights, row-major [out][in]: O2 [c2][cm], O3 [c3][cm], L2 [cm][c2], L3 [cm][c3], biases b2, b3 [cm].
 *      _fwd writes comp = A2 = O2 L2 [c2][c2] | B3 = O2 L3 [c2][c3] | A3 = O3 L3 [c3][c3] | O2 b2 [c2] | O2 b3 [c2] | O3 b3 [c3].
 *      _bwd maps the gradients of the composed matrices (dA2, dB3, dA3), the pixel sums W2 = sum g (x) t4 [c2][cm], W3 [c3][cm] and
 *      the per-channel gradient sums s2, sc3 [c2], s3 [c3] (double, as mdf_bn_stats_fwd leaves them) onto the parameters' gradients.  */
int mdf_fpn_compose_fwd(const float* O2, const float* O3, const float* L2, const float* b2, const float* L3, const float* b3,
                        int c2, int c3, int cm, float* comp, void* stream);
int mdf_fpn_compose_bwd(const float* O2, const float* O3, const float* L2, const float* b2, const float* L3, const float* b3,
                        const float* dA2, const float* dB3, const float* dA3, const float* W2, const float* W3,
                        const double* s2, const double* sc3, const double* s3, int c2, int c3, int cm, float* dO2, float* dO3,
                        float* dL2, float* dL3, float* db2, float* db3, void* stream);

/* ---- optimizer step (train.py:14,43: torch.optim.Adam(lr) -> optimizer.step()) over all parameters in one launch.  The
 *      parameters stay the module's tensors; grads, exp_avg, exp_avg_sq are flat float buffers in parameter order.  A job
 *      is one parameter tensor (param, its offset in the flat buffers, n elements); mdf_adam_job_fill writes job `index`
 *      into a HOST table of mdf_adam_job_bytes() bytes per job and returns its block count (< 0: error code); the caller
 *      uploads the table and an int32 per-block job index once.  `step` = 1-based step count (bias corrections
 *      1 - beta^step); torch's defaults otherwise (no amsgrad; weight_decay is added to the gradient as torch does).  */
int64_t mdf_adam_job_bytes(void);
int64_t mdf_adam_job_fill(void* jobs_host, int index, float* param, long long offset, long long n, int first_block);
int mdf_adam_step(const void* jobs_dev, const int* block_job_dev, int nblocks, const float* grads, float* exp_avg,
                  float* exp_avg_sq, float lr, float beta1, float beta2, float eps, float weight_decay, long long step,
                  void* stream);
/*      The same step with its per-step scalars in DEVICE memory: hyper_dev[3] = (lr, 1 - beta1^step, sqrt(1 - beta2^step)), computed
 *      by the caller as mdf_adam_step computes them.  For a training step recorded once and replayed as a hipGraph
 *      (mdfnet_hip/graphstep.py): kernel arguments are frozen in the recording, memory is not.  */
int mdf_adam_step_hyper(const void* jobs_dev, const int* block_job_dev, int nblocks, const float* grads, float* exp_avg,
                        float* exp_avg_sq, const float* hyper_dev, float beta1, float beta2, float eps, float weight_decay,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDFNET_HIP_H */
