/*
 * bihome.h - C ABI of libbihome_hip.so: the MI355X (gfx950) kernels behind the biHomE training hot path.
 *
 * Boundary contract: plain device pointers + sizes + a hipStream_t passed as void*,
 * int status return (0 = ok, >0 = hipError_t, <0 = BH_E_*), no allocation, no global mutable state (kernel routing is
 * a pure function of the arguments; the only process state is the per-device "dynamic LDS attribute set" flags),
 * re-entrant per stream.  Builds with -DBH_TUNING (tools/ only, `make tuning`) add the bh_debug_force_tile ablation
 * hook, which IS global state; the default library does not contain it.  The reference has no FFI of its own (it is pure Python over ATen); each entry point
 * below names the reference call site (path:line under the upstream repository) whose arithmetic it
 * replaces.  The host-side mirror of the reference's plugin API (src/backbones/<Name>.Model,
 * src/heads/<Name>.Model) lives in bihome_amd/ and calls these through ctypes.
 *
 * Layouts: images/patches/perspective fields are NCHW float32 exactly as the reference hands them
 * over; *internal* feature maps of the conv stacks are NHWC float32 ("pixels x channels"), conv
 * weights are [Cout][kh][kw][Cin] (= a torch channels_last OIHW tensor), transposed-conv weights
 * [Cin][kh][kw][Cout] (= channels_last IOHW).  Homographies are row-major 3x3.
 */
#ifndef BIHOME_H
#define BIHOME_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BH_OK 0
#define BH_E_BADARG (-1)
#define BH_E_UNSUPPORTED (-2)

/* library / device probe: returns ABI version (>0); writes gcnArchName of device 0 if buf != NULL */
int bh_version(void);
int bh_device_arch(char* buf, int buflen);

/* Deterministic calls (round 3: a process-wide switch; round 4: a PER-CALL bit - the library holds no mode and no other mutable
 * state).  What torch.use_deterministic_algorithms is to the reference's ATen path (train.py:379-387).
 * By default cross-workgroup sums use hardware floating-point atomics, whose rounding depends on arrival order: two runs of the
 * same step differ in the last bits.  With the bit set a launch is order-independent:
 *   - conv family (bh_conv_desc.route & BH_ROUTE_DETERMINISTIC): BatchNorm statistics / backward sums / bias column sums accumulated
 *     in conv epilogues go through exact integer limbs inside the padded sums entries (csrc/common.h bh_det_add); weight and bias
 *     gradients: bh_conv_wgrad_det with a workspace of bh_conv_wgrad_det_bytes(d) - partial tiles added in split order (split-operand /
 *     stride-1 kernels) or integer-limb shadow entries (every other shape).  bh_conv_wgrad / bh_conv_bias_grad have no workspace
 *     and stay atomic;
 *   - BatchNorm (flags & BH_BN_DETERMINISTIC in bh_bn_fwd* / bh_bn_bwd*): the statistics pass writes, and every pass reads, the limb
 *     encoding.  A sums table must be written and read in ONE mode (the forward that leaves sums for its backward, a conv epilogue and
 *     the BatchNorm that consumes its statistics); bh_bn_fwd_coeffs looks at both encodings;
 *   - bh_warp_bwd_f, bh_triplet_l1_fwd_f, bh_oneline_loss_fwd_f, bh_scale_samples_bwd_f, bh_dsac_scores_bwd_f, bh_tail_bwd_f
 *     (flags & BH_F_DETERMINISTIC): one workgroup per sample / channel is the only writer of its sums;
 *   - bh_dlt_bwd_f: duplicates of a sample's indices are added in point order inside the wave, hypotheses in launch order;
 *   - bh_warp_fwd_f with pool = 32: the pooled coverage by a one-writer kernel (the default adds four quarter-window sums with atomics).
 * The entry points without the _f suffix are the same calls with flags = 0.  Same arithmetic otherwise: results differ from the
 * default only by the order of additions.  A HIP graph replays whatever bits its captured launches carried. */
#define BH_F_DETERMINISTIC 1          /* flags argument of the bh_*_f entry points */
#define BH_BN_DETERMINISTIC 32        /* flags argument of bh_bn_fwd / bh_bn_fwd_amax / bh_bn_bwd / bh_bn_bwd_amax */
/* Measured matrix-pipe peak for the roofline: sustained TFLOP/s of a bare v_mfma_f32_32x32x16_bf16 stream on the whole
 * chip (two workgroups of four waves per CU, ~5 ms), with constant operands (random_operands = 0) or random-bit operands (1: the data
 * toggling of real tensors - on MI355X the power-limited rate, 25-35 % below the first).  Synchronises the stream.  sink_dev: 4 bytes of
 * device memory (never written). */
int bh_probe_mfma_bf16(int random_operands, float* sink_dev, double* tflops, void* stream);
/* the same stream of v_mfma_f32_32x32x2_f32 (the fp32-input instruction of the generic, stem and small-channel kernels) */
int bh_probe_mfma_f32(int random_operands, float* sink_dev, double* tflops, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Geometry (per-sample small dense algebra, double precision inside)
 * ------------------------------------------------------------------------------------------- */

/* four_point_to_homography, src/data/utils.py:7-33 (torch branch -> kornia.get_perspective_transform):
 * corners [[0,0],[W,0],[W,H],[0,H]] (image_shape_to_corners, src/data/utils.py:36-51) + delta[B,4,2]
 * -> H[B,9] with H22 = 1, 8x8 solve with partial pivoting.  H64 feeds the warp kernels, H32 is the
 * float copy handed back to the host module. */
int bh_h4pt_fwd(const float* delta, int B, float W, float H, double* H64, float* H32, void* stream);
/* adjoint of the above: gH[B,9] (double; entry 8 ignored) -> gdelta[B,8] (overwritten) */
int bh_h4pt_bwd(const float* delta, const double* H64, const double* gH, int B, float W, float H,
                float* gdelta, void* stream);

/* DSACSoftmax.__sample_hypotheses, src/heads/ransac_utils.py:47-74, fused with forward_map_field
 * (src/heads/PerceptualHead.py:125-146) and the corner transform (PerceptualHead.py:175-178):
 * pf[B,2,h,w] perspective field, choice[B, n*P] int64 sampled indices (row-major idx = y*w + x)
 * -> Hdlt[B*n,9] (kornia.find_homography_dlt: Hartley normalisation, A^T A, smallest eigenvector,
 * denormalise, /(H22+1e-8)) and delta_hat[B*n,4,2] = H.corners - corners.
 * eig[B*n,96] doubles: eigenvectors (81, column i = vector i), eigenvalues (9), index of the
 * smallest (1), spare - saved for the adjoint. */
int bh_dlt_fwd(const float* pf, const int64_t* choice, int B, int n, int P, int h, int w,
               float* Hdlt, float* delta_hat, double* eig, void* stream);
/* adjoint: g_delta[B*n,4,2] -> g_pf[B,2,h,w] += (scatter-add at the sampled indices; caller zeroes) */
/* g_Hdlt[B*n,9] (double, NULL ok): gradient that reaches the normalised homography itself (from the hypothesis scores,
 * bh_dsac_scores_bwd), added to the one that arrives through delta_hat */
int bh_dlt_bwd(const float* pf, const int64_t* choice, const double* eig, const float* g_delta, const double* g_Hdlt,
               int B, int n, int P, int h, int w, float* g_pf, void* stream);
int bh_dlt_bwd_f(const float* pf, const int64_t* choice, const double* eig, const float* g_delta, const double* g_Hdlt,
                 int B, int n, int P, int h, int w, float* g_pf, int flags, void* stream);      /* flags: BH_F_DETERMINISTIC */

/* DSACSoftmax.__score_hypotheses ('repr_error'), src/heads/ransac_utils.py:76-128, and the
 * arg-max of softmax(-err) == arg-min of err at src/heads/PerceptualHead.py:755-757:
 * err[B,n] = sum over all h*w points of |H.coord - (coord + pf)|_1 ; best[B] int64 (first minimum). */
int bh_dsac_score(const float* pf, const float* Hdlt, int B, int n, int h, int w,
                  float* err, int64_t* best, void* stream);

/* scores[B,n] = softmax(-err) over the hypotheses (ransac_utils.py:126): the weights of the score-weighted multi-hypothesis
 * losses (PerceptualHead.py:276-280,505-511,708-710).  Adjoint of scoring + softmax: g_scores[B,n] -> g_err[B,n] (scratch,
 * overwritten), g_Hdlt[B*n,9] (double, overwritten; feed to bh_dlt_bwd) and g_pf[B,2,h,w] += (atomics; the reprojection
 * error sees every point of the field: sign(H.coord - map) per point). */
int bh_dsac_scores_fwd(const float* err, int B, int n, float* scores, void* stream);
int bh_dsac_scores_bwd(const float* pf, const float* Hdlt, const float* scores, const float* g_scores, int B, int n, int h,
                       int w, float* g_err, double* g_Hdlt, float* g_pf, void* stream);
int bh_dsac_scores_bwd_f(const float* pf, const float* Hdlt, const float* scores, const float* g_scores, int B, int n, int h,
                         int w, float* g_err, double* g_Hdlt, float* g_pf, int flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Homography warp (warp_image, src/data/utils.py:54-59 -> kornia.warp_perspective(bilinear, zeros,
 * align_corners=True); net map out(x,y) = bilinear img(H.(x,y,1)))  and the warped all-ones mask
 * + AvgPool2d(k) (src/heads/PerceptualHead.py:380-382,401,447-459) fused into the same pass.
 * ------------------------------------------------------------------------------------------- */
/* img[B,C,h,w] (NULL => only the coverage is produced), H64[B,9]; out[B,C,h,w] (may be NULL when
 * img is NULL); cov[B,h/pool,w/pool] (NULL => skipped). h,w multiples of pool; pool in {1,2,4,8,16,32}. */
int bh_warp_fwd(const float* img, const double* H64, int B, int C, int h, int w, int pool,
                float* out, float* cov, void* stream);
int bh_warp_fwd_f(const float* img, const double* H64, int B, int C, int h, int w, int pool,
                  float* out, float* cov, int flags, void* stream);         /* flags: BH_F_DETERMINISTIC (matters for pool = 32) */
/* adjoint w.r.t. H only (the image is data): g_out[B,C,h,w] (NULL ok), g_cov[B,h/pool,w/pool]
 * (NULL ok) -> gH[B,9] += (double, atomics; caller zeroes) */
int bh_warp_bwd(const float* img, const double* H64, const float* g_out, const float* g_cov,
                int B, int C, int h, int w, int pool, double* gH, void* stream);
int bh_warp_bwd_f(const float* img, const double* H64, const float* g_out, const float* g_cov,
                  int B, int C, int h, int w, int pool, double* gH, int flags, void* stream);
/* adjoint w.r.t. the IMAGE (the trained masks of the Zhang baseline are warped, src/heads/TripletHead.py:60,69, and their gradient must
 * reach the mask predictor): g_out[B,C,h,w] -> g_img[B,C,h,w] (overwritten) = transpose of the bilinear gather with the same taps and
 * zero padding.  flags & BH_F_DETERMINISTIC: scratch = bh_warp_bwd_img_scratch_doubles(...) doubles (integer-limb entries); else NULL. */
size_t bh_warp_bwd_img_scratch_doubles(int B, int C, int h, int w, int flags);
int bh_warp_bwd_img_f(const double* H64, const float* g_out, int B, int C, int h, int w, float* g_img, double* scratch, int flags,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * biHomE triplet L1 reduction (triplet_resnet_loss, double-line / l1 / channel-agnostic / str margin:
 * src/heads/PerceptualHead.py:559-561, 609-665).  Features are NHWC [B,hw,C].
 * ------------------------------------------------------------------------------------------- */
/* M1[B,hw] = sum_c |f1w-f2| - sum_c |f1-f2| ; M2[B,hw] = sum_c |f2w-f1| - sum_c |f1-f2| ;
 * numden[B,4] (double) = { sum m1w*m2*M1, sum m1w*m2, sum m2w*m1*M2, sum m2w*m1 } ; m1,m2 NULL => ones. */
int bh_triplet_l1_fwd(const float* f1, const float* f2, const float* f1w, const float* f2w,
                      const float* m1w, const float* m2w, const float* m1, const float* m2,
                      int B, int hw, int C, float* M1, float* M2, double* numden, void* stream);
int bh_triplet_l1_fwd_f(const float* f1, const float* f2, const float* f1w, const float* f2w,
                        const float* m1w, const float* m2w, const float* m1, const float* m2,
                        int B, int hw, int C, float* M1, float* M2, double* numden, int flags, void* stream);
/* loss4[4] = { loss, ln1, ln2, ln3 }: ln_k = sum_b num/max(den,1); ln3 = sum_b ||H1 H2 - I||_F^2;
 * loss = ln1 + ln2 + mu*ln3 (PerceptualHead.py:656-665) */
int bh_bihome_loss_fwd(const double* numden, const double* H1, const double* H2, int B, float mu,
                       float* loss4, void* stream);
/* adjoint of both: g_loss[1] (device scalar) -> g_f1w,g_f2w [B,hw,C] (overwritten), g_m1w,g_m2w
 * [B,hw] (overwritten), gH1,gH2 [B,9] double (overwritten) */
int bh_bihome_loss_bwd(const float* g_loss, const float* f1, const float* f2, const float* f1w,
                       const float* f2w, const float* m1w, const float* m2w, const float* m1,
                       const float* m2, const float* M1, const float* M2, const double* numden,
                       const double* H1, const double* H2, int B, int hw, int C, float mu,
                       float* g_f1w, float* g_f2w, float* g_m1w, float* g_m2w,
                       double* gH1, double* gH2, void* stream);

/* One-line variant (iHomE; triplet_resnet_loss 'one-line' / l1 / numeric margin: PerceptualHead.py:465-538):
 *   loss = sum_b [ sum_p w max(|f1w-f2|_1 - |f1-f2|_1 + margin, 0) / max(sum_p w, 1) ],  w = m1w * m2 (m2 NULL => ones)
 * T[B,hw] = pre-hinge value (kept for the adjoint), numden[B,2] double, loss[1]. */
/* Multi-hypothesis form (RANSAC_HYPOTHESIS_NO = rep > 1, PerceptualHead.py:352-361,505-511): B counts hypotheses
 * (samples * rep); f1, f2, m2 hold one entry per SAMPLE (row b / rep), f1w / m1w / T one per hypothesis; sample_w[B]
 * (NULL = 1) = DSAC score of the hypothesis, loss = sum_b sample_w[b] * loss_b; per_sample[B] (NULL ok) = loss_b. */
int bh_oneline_loss_fwd(const float* f1, const float* f2, const float* f1w, const float* m1w, const float* m2, int B, int hw,
                        int C, float margin, int rep, const float* sample_w, float* T, double* numden, float* per_sample,
                        float* loss, void* stream);
int bh_oneline_loss_fwd_f(const float* f1, const float* f2, const float* f1w, const float* m1w, const float* m2, int B, int hw,
                          int C, float margin, int rep, const float* sample_w, float* T, double* numden, float* per_sample,
                          float* loss, int flags, void* stream);
/* adjoint: g_loss[1] -> g_f1w[B,hw,C], g_m1w[B,hw] (overwritten); d loss / d sample_w[b] = g_loss * per_sample[b] */
int bh_oneline_loss_bwd(const float* g_loss, const float* f2, const float* f1w, const float* m1w, const float* m2,
                        const float* T, const double* numden, int B, int hw, int C, int rep, const float* sample_w,
                        float* g_f1w, float* g_m1w, void* stream);
/* Zhang et al. content-aware triplet loss (TripletHead.forward, src/heads/TripletHead.py:75-152) on ONE-channel full-resolution feature
 * maps f*[B,hw] (the ContentAware feature extractor, src/backbones/ContentAware.py:57-81) and masks m*[B,hw] (m1 / m2 NULL = ones: FIX_MASK):
 *   num1 = sum_p m1w m2 h(|f1w - f2| - |f1 - f2| + margin), den1 = sum_p m1w m2; num2 / den2 with (f2w, f1, m2w m1); f2w NULL: one line.
 * hinge != 0: h = max(., 0) (numeric margin, :98-107); hinge == 0: h = identity and margin is ignored (string margin, :92-97).
 * T1 / T2[B,hw] = pre-hinge values (for the adjoint), numden[B,4] double (overwritten) -> bh_bihome_loss_fwd gives ln1 + ln2 + mu ln3
 * (:150-152; one line: pass identity homographies and mu = 0).  The adjoint writes the gradients of ALL FOUR feature maps (the extractor
 * is trainable here) and of the two warped masks (overwritten); the unwarped masks are constants. */
int bh_zhang_triplet_fwd(const float* f1, const float* f2, const float* f1w, const float* f2w, const float* m1w, const float* m2w,
                         const float* m1, const float* m2, int B, int hw, float margin, int hinge, float* T1, float* T2, double* numden,
                         void* stream);
int bh_zhang_triplet_bwd(const float* g_loss, const float* f1, const float* f2, const float* f1w, const float* f2w, const float* m1w,
                         const float* m2w, const float* m1, const float* m2, const float* T1, const float* T2, const double* numden, int B,
                         int hw, int hinge, float* g_f1, float* g_f2, float* g_f1w, float* g_f2w, float* g_m1w, float* g_m2w, void* stream);
/* the same adjoint with the gradients of the UNWARPED masks (trained masks, FIX_MASK False: m2 weights line 1, m1 line 2):
 * g_m1 / g_m2[B,hw] overwritten, NULL = not wanted (bh_zhang_triplet_bwd is this call with both NULL). */
int bh_zhang_triplet_bwd_m(const float* g_loss, const float* f1, const float* f2, const float* f1w, const float* f2w, const float* m1w,
                           const float* m2w, const float* m1, const float* m2, const float* T1, const float* T2, const double* numden, int B,
                           int hw, int hinge, float* g_f1, float* g_f2, float* g_f1w, float* g_f2w, float* g_m1w, float* g_m2w,
                           float* g_m1, float* g_m2, void* stream);
/* Trained content masks (src/backbones/ContentAware.py:24-26,28-35,47-50,128-134) after the mask predictor's last BatchNorm y[N,P]
 * (P = h w pixels of a one-channel map): s = sigmoid(y); strength > 0: m = clamp(s / (max_p s * strength), 0, 1) (per-sample maximum),
 * else m = s; g = m * f (f, g NULL: the mask only).  smax[N] / imax[N]: the maximum of s and its first pixel (kept for the adjoint).
 * Adjoint: g_m (gradient of the mask from the head, NULL = none) and g_g (gradient of g from the resnet, NULL = none) -> g_y and g_f
 * (NULL = not wanted); the maximum's gradient goes to pixel imax (torch: max(1)[0] backward).  One workgroup per sample, no atomics. */
int bh_mask_fwd(const float* y, const float* f, int N, int P, float strength, float* m, float* g, float* smax, int* imax, void* stream);
int bh_mask_bwd(const float* y, const float* f, const float* m, const float* smax, const int* imax, const float* g_m, const float* g_g,
                int N, int P, float strength, float* g_y, float* g_f, void* stream);
/* y[b,:] = x[b / rep,:] * s[b] over Bn hypotheses of L floats (the score weighting of multihead_resnet_loss,
 * PerceptualHead.py:276-280) and its adjoint (g_x only for rep = 1, may be NULL; g_s[Bn] overwritten). */
int bh_scale_samples_fwd(const float* x, const float* s, int Bn, long long L, int rep, float* y, void* stream);
int bh_scale_samples_bwd(const float* g_y, const float* x, const float* s, int Bn, long long L, int rep, float* g_x, float* g_s,
                         void* stream);
int bh_scale_samples_bwd_f(const float* g_y, const float* x, const float* s, int Bn, long long L, int rep, float* g_x, float* g_s,
                           int flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Conv stacks (Rethinking._forward src/backbones/Rethinking.py:284-294 with blocks
 * src/backbones/utils.py:60-131; ResNet34 src/backbones/ResNet34.py:15-28; AuxiliaryResnet.forward
 * src/heads/PerceptualHead.py:50-76).  Replace ATen conv2d / conv_transpose2d / batch_norm / relu /
 * max_pool2d / adaptive_avg_pool2d / linear and their autograd adjoints.  Implicit GEMM on the
 * f32-input MFMA (v_mfma_f32_32x32x2_f32): exact fp32 products, fp32 accumulate.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int N, Hi, Wi, Ci;      /* input  NHWC */
    int Ho, Wo, Co;         /* output NHWC */
    int kh, kw, stride, pad;
    int transposed;         /* 0: Conv2d ; 1: ConvTranspose2d with kernel == stride, pad 0 */
    int in_nchw;            /* 1: x is NCHW (only for the network inputs, Ci <= 4) */
    int out_nchw;           /* 1: y is NCHW (only for the network output, Co <= 4) */
    int precision;          /* 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32); 1: operands rounded to bf16 while staging,
                             * fp32 accumulate (v_mfma_f32_32x32x16_bf16), tensors stay fp32 in HBM; 2 ("f32x3"): fp32
                             * accuracy on the bf16 matrix pipe - every fp32 operand is cut EXACTLY into three bf16 pieces
                             * and the six partial products of order <= 2 are accumulated in fp32 (what is dropped is
                             * below 2^-23 of a product, one fp32 rounding); taken by the halo-tiled 3x3 kernel with
                             * w_layout 2, every other kernel computes precision 2 as precision 0;
                             * 3 ("f32x2", REDUCED precision, reported separately): two bf16 pieces per operand, both rounded to
                             * nearest (x = hi + mid + e, |e| <= 2^-18 |x|), products hi*hi + hi*mid + mid*hi (~4e-6 per
                             * product: 13x the fp32 rounding, 500x below precision 1); same kernels as 2 with w_layout 3;
                             * 4 ("f16x2", round 4: fp32 accuracy at three products): two FP16 pieces per operand of the operand times
                             * a power-of-two scale per tensor (x 2^k = hi + lo + e, 11 + 11 significand bits and a sign: |e| <= 2^-23
                             * of the element down to 2^-18 of the tensor's largest magnitude, an absolute 2^-40 of it below), products
                             * hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 (~2^-22 per product: 16x below precision 3, one or two
                             * fp32 roundings), results rescaled exactly.  The scales come from MAGNITUDE RECORDS (a_bound / b_bound
                             * below); same kernels as 3 with w_layout 4; the weight gradient without both records runs as precision 2 */
    int w_layout;           /* 0: w is [Co][kh][kw][Ci].  2: as 1, made with bh_pack3x3_job.split = 1 (three bf16 pieces, 6 bytes
                             * per weight; requires precision 2).  3: as 1, made with split = 2 (two rounded bf16 pieces, 4 bytes
                             * per weight; requires precision 3).  4: as 1, made with split = 3 by bh_conv3x3_pack_f16 (two fp16 pieces
                             * and the weights' maxima; requires precision 4 and a_bound).  1 (3x3 / stride 1 / pad 1 only): w is the fragment-ordered copy made
                             * by bh_conv3x3_pack - its `pf` buffer for bh_conv_fwd*, its `pd` buffer for bh_conv_dgrad* -
                             * which the halo-tiled 3x3 kernel streams straight into registers; BH_E_UNSUPPORTED when
                             * that kernel does not take the launch (ask bh_conv_variant first) */
    int route;              /* 0: automatic kernel choice (production).  BH_ROUTE_* bits: explicit per-call routing for
                             * tests and benchmarks (e.g. drive the halo-tiled 3x3 kernel on a grid it would decline) */
    /* precision 4 only (NULL otherwise; the only per-call members of the descriptor): magnitude records (BH_AMAX_FLOATS floats each,
     * device memory) of the operand tensors - a_bound: the tensor the launch convolves (x for forward / weight gradient - of the
     * BatchNorm OUTPUT for the *_bnin forms -, gy for dgrad); b_bound: gy of the weight gradient.  A record holds an upper bound
     * of max |element| as the maximum over its 16 slots (one per 128-byte line; non-negative floats): written by bh_absmax, by the
     * *_amax forms of the BatchNorm entry points (measured) or by bh_bn_fwd_coeffs_amax (|gamma| sqrt(rows) + |beta|).  A bound
     * that is too small overflows fp16 (inf / NaN in the result, never a silently wrong value); one that is too large by 2^j
     * costs j of the 18 binades of full precision. */
    const float* a_bound;
    const float* b_bound;
} bh_conv_desc;
#define BH_AMAX_FLOATS 512        /* 16 slots x 32 floats (csrc/common.h BH_AMAX_*) */
/* record[slot] = max(record[slot], max |x[i]|) over n floats: the record must be zeroed (or hold an earlier bound to be combined with).
 * One streaming pass: the fallback where no producer kernel left a record. */
int bh_absmax(const float* x, long long n, float* record, void* stream);
#define BH_ROUTE_GENERIC_CONV 1   /* fwd / dgrad: never the halo-tiled 3x3 kernel (generic implicit GEMM) */
#define BH_ROUTE_HALO_SMALL 2     /* fwd / dgrad: let the halo-tiled 3x3 kernel take grids below its workgroup minimum */
#define BH_ROUTE_NO_STEM7 4       /* fwd: never the dedicated 7x7/2 stem kernel */
#define BH_ROUTE_WGRAD_GENERIC 8  /* wgrad: generic split-K kernel instead of the stride-1 fast path */
#define BH_ROUTE_WGRAD_3TAP 16    /* wgrad: three taps per workgroup in the stride-1 fast path */
#define BH_ROUTE_C3_ONE_SUBTILE 32  /* fwd / dgrad: halo-tiled 3x3 kernel with one 8x8 sub-tile per workgroup (64-channel tile) */
#define BH_ROUTE_C3_ONE_POSITION 64 /* fwd / dgrad: halo-tiled 3x3 kernel never walks two tile positions per workgroup */
#define BH_ROUTE_DETERMINISTIC 128  /* every launch of this call is order-independent (see "Deterministic calls" above) */
#define BH_ROUTE_C3_PC 512          /* fwd / dgrad, precision 4, 64-channel tile: the persistent producer / consumer kernel for every launch it supports
                                     * (default: only where it is the faster one) */
#define BH_ROUTE_C3_TILE_WG 256     /* fwd / dgrad, precision 4: the one-workgroup-per-tile halo kernel instead of the persistent producer /
                                     * consumer kernel (round 5) - same convolution results bit for bit */
#define BH_ROUTE_WX3_PC 1024        /* wgrad, precision 4, 64-channel blocks: the EIGHT-wave producer / consumer form of wgrad_x3_kernel (bitwise the
                                     * four-wave result).  Faster alone (-2 us per launch); the default four-wave form leaves 200 registers per
                                     * SIMD lane to co-resident kernels of a second stream, which is worth more in a two-stream step (round 5) */
#define BH_ROUTE_WX3_SHARED 2048    /* wgrad, split-operand 3x3 kernels: the launch shares the GPU with another stream - 160 workgroups instead of one
                                     * per CU (fewer, longer workgroups: the other stream's launches start at once on the free CUs, and the split-K
                                     * partial blocks shrink with the workgroup count).  Same-box in-step A/B, round 5: -0.19 ms per two-stream step */
#define BH_ROUTE_GEMM_X3 4096       /* fwd / dgrad, precision 2 / 4, round 6: layers that run on the generic implicit-GEMM kernel (strided and 1x1 convs,
                                     * 128- / 256-channel transposed convs, their dgrads) cut their operands exactly into three bf16 pieces and make six
                                     * products per product (the arithmetic of the f32x3 3x3 kernels) instead of running the fp32-input MFMA: error against
                                     * float64 <= the fp32-input MFMA's (tests/test_conv_kernels_gpu.py), those launches ~10 % shorter, the step 0.05 ms.
                                     * Opt-in: without it these layers are bit-identical to precision 0. */

/* One 3x3 layer's weights for bh_conv3x3_pack: w[Co][3][3][Ci] (Co, Ci multiples of 32) -> pf (forward operand order) and
 * pd (dgrad operand order: transposed, taps flipped), Co*9*Ci floats each (split: 1.5x that); either may be NULL. */
typedef struct {
    const float* w;
    float* pf;
    float* pd;
    int Co, Ci;
    int split;              /* 0: fp32 fragments (w_layout 1).  1: three bf16 pieces per weight (w_layout 2): pf / pd then hold
                             * Co*9*Ci*6 bytes each.  2: two rounded bf16 pieces (w_layout 3): Co*9*Ci*4 bytes each.  3: two fp16
                             * pieces of w 2^k (w_layout 4; bh_conv3x3_pack_f16): Co*9*Ci*4 + 64 bytes each - the last 64 bytes hold
                             * sixteen partial maxima of |w|, from which the pack and every consumer derive k */
    int reserved;
} bh_pack3x3_job;
/* Packs the weights of njobs layers in one launch (jobs_dev: device array).  Call after every optimizer step (the
 * parameters changed) before the next forward; frozen layers need it once. */
int bh_conv3x3_pack(const bh_pack3x3_job* jobs_dev, int njobs, void* stream);
/* the same for tables with split = 3 jobs: one more launch in front that takes the layers' max |w| */
int bh_conv3x3_pack_f16(const bh_pack3x3_job* jobs_dev, int njobs, void* stream);

/* Which kernel a launch described by d would run: which = 0 forward, 1 dgrad, 2 wgrad, 3 wgrad through a workspace (bh_conv_wgrad_det),
 * 4 / 5 bh_conv_dgrad_bnreduce with the ReLU mask from z / from y (bn_groups = its groups), 6 bh_conv_dgrad_colsum.  Writes the kernel template
 * instantiation (the symbol rocprofv3 lists, e.g. "conv3x3_halo_kernel<false,64,false,2>"; several launches joined by
 * '+') into buf[n].  Runs the real dispatch code with the launches replaced by a name record, so it cannot drift from it.
 * accumulate / bn_groups (0: plain bh_conv_fwd, > 0: bh_conv_fwd_bnstats with that many groups) mirror the arguments of
 * the call being described (they influence routing). */
int bh_conv_variant(const bh_conv_desc* d, int which, int accumulate, int bn_groups, char* buf, int n);

#ifdef BH_TUNING
/* ablation / tuning hook of the -DBH_TUNING build (tools/ only; process-global state) */
int bh_debug_force_tile(int bm, int bn);
#endif
/* y = conv(x, w) (+ bias[Co] if bias != NULL) */
int bh_conv_fwd(const float* x, const float* w, const float* bias, float* y, const bh_conv_desc* d, void* stream);
/* bh_conv_fwd on the generic implicit-GEMM kernel (Conv2d of any geometry, ConvTranspose2d with k == s) that also leaves the magnitude
 * record of its output: amax_y (BH_AMAX_FLOATS floats, zeroed by the caller) receives max |y| - what the fp16-piece 3x3 kernels (precision 4)
 * need of their source tensor; round 4: the transposed convs in front of the decoder units' 3x3 convs (their outputs were measured by a
 * separate bh_absmax pass, 4 launches of 25-55 us per step). */
int bh_conv_fwd_amax(const float* x, const float* w, const float* bias, float* y, const bh_conv_desc* d, float* amax_y, void* stream);

/* y = act(conv(x, w) + bias + res): res (NULL ok) has the layout of y, relu != 0 applies max(.,0).  The inference path:
 * an eval-mode BatchNorm is folded into (w, bias) by the host and its ReLU / residual add ride in the conv epilogue. */
int bh_conv_fwd_act(const float* x, const float* w, const float* bias, const float* res, float* y, const bh_conv_desc* d,
                    int relu, void* stream);
/* BatchNorm-on-load (f32x3 3x3 layers, round 2): x is the INPUT of a training-mode BatchNorm(+ReLU) whose output this convolution
 * is the only consumer of; the kernels apply y = max(x * scale + shift, relu ? 0 : -inf) per channel while staging their operand
 * (zero padding stays zero), so the BatchNorm's output tensor is never written or read.  table[groups][C] x (scale, shift) comes from
 * bh_bn_fwd_coeffs (which also makes the running-statistics update); the images of x are `groups` equal stacks along N.
 * bh_conv_fwd_bnin: packed f32x3 forward (w_layout 2), optional BatchNorm sums of ITS output as in bh_conv_fwd_bnstats (sums NULL: none);
 * bh_conv_wgrad_bnin: the f32x3 weight gradient (workspace form) with the same transform on x.  BH_E_UNSUPPORTED when the launch is
 * not taken by those kernels (ask bh_conv_variant; table > 4 KB). */
typedef struct {
    const float* table;
    int groups;
    int relu;
} bh_bn_in;
int bh_bn_fwd_coeffs(const double* stats, const float* gamma, const float* beta, float* running_mean, float* running_var, int groups,
                     int rows, int C, float eps, float momentum, float* table, void* stream);
/* bh_bn_fwd_coeffs + the magnitude record of the BatchNorm's OUTPUT (what the *_bnin consumers convolve): the a-priori bound
 * max_c 2 (|gamma_c| sqrt(rows) + |beta_c|) - a normalised sample is at most sqrt(rows - 1) standard deviations from its mean. */
int bh_bn_fwd_coeffs_amax(const double* stats, const float* gamma, const float* beta, float* running_mean, float* running_var, int groups,
                          int rows, int C, float eps, float momentum, float* table, float* amax_y, void* stream);
int bh_conv_fwd_bnin(const float* x, const float* w, const float* bias, float* y, const bh_conv_desc* d, double* sums, int groups,
                     const bh_bn_in* bni, void* stream);
int bh_conv_wgrad_bnin(const float* x, const float* gy, float* gw, float* gbias, const bh_conv_desc* d, float* ws, long long ws_bytes,
                       const bh_bn_in* bni, void* stream);
/* BatchNorm ADJOINT on load (round 6; round-5 VERDICT item 2): the weight gradient of a 3x3 / stride-1 convolution whose output z feeds a
 * training-mode BatchNorm (+ReLU, + optional residual in front of the ReLU: src/backbones/utils.py:85-131 conv -> bn -> relu chains),
 * computed from the gradient d of that BatchNorm's OUTPUT instead of the gradient of z:
 *     gw[co][tap][ci] += sum_pixels g[pixel][co] x[pixel + tap][ci],   g = sc (mask(d) - k1 - xhat(z) k2)
 * with the BatchNorm's adjoint g applied while the kernel stages its operand - sc = gamma / std, xhat = (z - mean) / std from the forward
 * sums `stats`; k1 = sum mask(d) / rows, k2 = sum mask(d) xhat / rows from `sums`, which bh_conv_dgrad_bnreduce accumulated when it
 * completed d; mask = (y > 0), y read from `y` (a residual was added in front of the ReLU) or recomputed as z sc + shift (y NULL); relu 0:
 * no mask.  The launch therefore needs neither the BatchNorm's adjoint pass (bh_bn_bwd) nor its output tensor: it can start as soon as
 * the dgrad that made d has finished, next to the HBM-bound adjoint pass instead of behind it.  x, bni as bh_conv_wgrad_bnin (bni NULL:
 * x as it is).  desc->precision = 4 with desc->a_bound = the magnitude record of x and desc->b_bound = the record of d (max |d|:
 * bh_conv_dgrad_bnreduce_amax leaves it); the fp16 scale of g comes from the a-priori bound max_c |sc| (max |d| + |k1| + sqrt(rows) |k2|).
 * ws / ws_bytes: the partial-block workspace (bh_conv_wgrad_det_bytes).  BH_E_UNSUPPORTED where wgrad_x3_kernel's 64-channel fp16-piece
 * form does not apply (the caller runs bh_bn_bwd and bh_conv_wgrad_* as before). */
/* Round 6: the weight gradients of n = 1 .. 4 layers of ONE geometry (precision 4, 64-channel blocks: what wgrad_x3_kernel's fp16-piece form
 * takes) in ONE launch + one reduce launch: gw[i] += x[i]^T gy[i].  descs[i]: identical geometry / route, each with its own magnitude records
 * (a_bound of x[i], b_bound of gy[i]); bni: NULL, or n BatchNorm-on-load tables of equal groups / relu.  ws >= n-independent
 * bh_conv_wgrad_det_bytes(descs[0]) bytes.  Fewer workgroups share a layer's pixels, so a LAUNCH writes the <= 256 partial blocks that each
 * single launch writes (37.7 MB less traffic per extra layer) and n - 1 kernel / reduce launch pairs disappear.  Deterministic; equal to the
 * single launches up to the order of the split-K sums.  BH_E_UNSUPPORTED where that kernel form does not apply. */
int bh_conv_wgrad_batch(int n, const float* const* x, const float* const* gy, float* const* gw, const bh_conv_desc* const* descs, float* ws,
                        long long ws_bytes, const bh_bn_in* const* bni, void* stream);
typedef struct {
    const float* z;        /* the BatchNorm's input = the convolution's output [N,H,W,Co] */
    const float* y;        /* the saved output behind the ReLU, or NULL (mask recomputed from z) */
    const double* stats;   /* forward sums of z: bh_bn_stats_doubles(groups, Co) */
    const double* sums;    /* backward sums (sum mask(d), sum mask(d) xhat), same layout */
    const float* gamma;    /* NULL: 1 */
    const float* beta;     /* NULL: 0 */
    float eps;
    int relu;
    int groups;            /* statistics groups: equal stacks of images along N */
} bh_bn_adj;
int bh_conv_wgrad_bnadj(const float* x, const float* d_out, float* gw, const bh_conv_desc* desc, float* ws, long long ws_bytes,
                        const bh_bn_in* bni, const bh_bn_adj* bna, void* stream);
/* y = conv(x, w) + bias, and sums (bh_bn_stats_doubles(groups, Co) doubles, caller-zeroed) += per-channel (sum y, sum y^2) of each of the
 * `groups` sub-batches stacked along N: the batch statistics of the BatchNorm that follows, accumulated in the conv
 * epilogue (halo-tiled 3x3 kernel; generic implicit GEMM incl. ConvTranspose2d when the pixels of a group are a
 * multiple of 128; one extra statistics launch otherwise).  Pass the buffer to bh_bn_fwd with flags bit3.  NHWC output. */
int bh_conv_fwd_bnstats(const float* x, const float* w, const float* bias, float* y, const bh_conv_desc* d, double* sums,
                        int groups, void* stream);
/* gx = conv^T(gy, w)  (overwritten; accumulate != 0: gx += ..., used where gradient branches join).
 * d->in_nchw: gx is written NCHW (gradient w.r.t. an NCHW network input; no accumulate, not transposed). */
int bh_conv_dgrad(const float* gy, const float* w, float* gx, const bh_conv_desc* d, int accumulate, void* stream);
/* dgrad of a stride-2 conv (3x3 pad 1 or 1x1 pad 0, even Hi/Wi, NHWC) by output parity class: four dense stride-1
 * launches on class-packed weights instead of one adjoint gather that multiplies 3/4 zeros.  wpack: scratch of
 * Co*kh*kw*Ci floats (rewritten by every call).  BH_E_UNSUPPORTED for any other geometry (use bh_conv_dgrad). */
int bh_conv_dgrad_s2(const float* gy, const float* w, float* gx, const bh_conv_desc* d, int accumulate, float* wpack,
                     void* stream);
/* The BatchNorm (+ReLU) whose OUTPUT gradient a dgrad produces: z = its input, y = its output (needed for the ReLU mask
 * only when a residual was added; NULL: the mask is recomputed from z), stats = its forward sums (bh_bn_fwd). */
typedef struct bh_bn_reduce {
    const float* z;
    const float* y;
    const double* stats;
    const float* gamma;
    const float* beta;
    float eps;
    int relu;
    float* amax_d;         /* round 6, optional: caller-zeroed magnitude record (BH_AMAX_FLOATS floats) that receives max |mask(d)| of the gradient
                              written - what bh_conv_wgrad_bnadj's scale bound needs; NULL: not measured */
} bh_bn_reduce;
/* bh_conv_dgrad that also accumulates, in its epilogue, the backward sums of that BatchNorm (sum g*mask, sum g*mask*xhat
 * per group and channel) into `sums` (bh_bn_stats_doubles(groups, Ci) doubles, caller-zeroed): pass them to bh_bn_bwd
 * with flags bit4 and its reduce pass is skipped.  gx must be the FINAL gradient (accumulate = 1 adds to the partial
 * sum already in gx).  Only for shapes the halo-tiled 3x3 kernel takes (3x3, stride 1, pad 1, H, W multiples of 8,
 * Co % 32 == 0, Ci % 32 == 0, fp32 or bf16 operands); BH_E_UNSUPPORTED otherwise and nothing is written. */
int bh_conv_dgrad_bnreduce(const float* gy, const float* w, float* gx, const bh_conv_desc* d, int accumulate,
                           const bh_bn_reduce* bnr, double* sums, int groups, void* stream);
/* bh_conv_dgrad (no accumulate) that also adds, in its epilogue, the per-channel sums of the gradient it writes into
 * `sums` (bh_bn_stats_doubles(1, Ci) doubles, caller-zeroed; entry (0, c, 0) = column sum).  The column sums of a conv's
 * input gradient ARE the bias gradient of the layer that produced that input (ConvTranspose2d + bias in the decoder
 * units, src/backbones/utils.py:60-82): bh_bias_grad_from_sums adds them to gbias[C] and the separate streaming pass of
 * bh_conv_bias_grad over that gradient disappears.  Only where the halo-tiled 3x3 kernel applies (else BH_E_UNSUPPORTED). */
int bh_conv_dgrad_colsum(const float* gy, const float* w, float* gx, const bh_conv_desc* d, double* sums, void* stream);
int bh_bias_grad_from_sums(const double* sums, float* gbias, int groups, int C, void* stream);
/* Second half of the two-step dgrad of the extractor's 7x7/2 stem on a grayscale (Ci = 1) or RGB (Ci = 3, in_nchw) patch:
 * Tm[N*Ho*Wo][ldT] = gy x w^T (one 1x1 bh_conv_fwd launch; column = tap*Ci + c, ldT >= 49*Ci padded)
 * -> gx[N,Ci,Hi,Wi] = col2im(Tm). */
int bh_col2im_c1(const float* Tm, float* gx, const bh_conv_desc* d, int ldT, void* stream);
/* The same dgrad for ONE input channel in one kernel (round 4: no tap table in HBM): gy[N,Ho,Wo,64] NHWC, w[64][7][7][1] -> gx[N,1,2Ho,2Wo]
 * (overwritten).  The frozen extractor's conv1 on a warped patch (src/heads/PerceptualHead.py:52-55,377,398).  BH_E_UNSUPPORTED for other
 * geometries: the caller keeps the two-pass form for those.  No atomics: every image tile has one writer. */
int bh_stem7_dgrad_c1(const float* gy, const float* w, float* gx, const bh_conv_desc* d, void* stream);
/* Round 6: that dgrad WITH the adjoint of the homography warp that produced the extractor's input (src/data/utils.py:54-59 `warp_image`
 * as called by src/heads/PerceptualHead.py:371,392 `_warp`, followed by `auxiliary_resnet` :377,398): the gradient of the warped patch
 * has one consumer - dL/dH - so the thread that sums a pixel's gradient applies the warp's adjoint on the spot (the pixel's bilinear tap,
 * four gathered pixels of the SOURCE patch src[N,1,Hi,Wi], the coverage term of the pool-averaged all-ones mask when g_cov[N,Hi/pool,
 * Wi/pool] is given: PerceptualHead.py:380-382,401,447-459) and gH[N,9] += the nine sums.  Equal to bh_stem7_dgrad_c1 followed by
 * bh_warp_bwd(src, H64, gx, g_cov, N, 1, Hi, Wi, pool, gH) up to the order of the double sums; the 4 Hi Wi-byte gradient image is not
 * written (gx = NULL) or written as before (gx != NULL).  gH accumulates through f64 atomics: deterministic callers
 * (BH_F_DETERMINISTIC) use the two separate calls.  BH_E_UNSUPPORTED as bh_stem7_dgrad_c1, or when pool is not a power of two <= 32. */
int bh_stem7_dgrad_c1_warp(const float* gy, const float* w, float* gx, const bh_conv_desc* d, const float* src, const double* H64,
                           const float* g_cov, int pool, double* gH, void* stream);
/* Round 6: the one-plane 7x7 / 2 stem ON the homography warp of a source patch - `auxiliary_resnet(_warp(patch, H))` of
 * src/heads/PerceptualHead.py:371-377,392-398 (warp_image: src/data/utils.py:54-59) in one launch: y[N,Ho,Wo,64] = conv(warp(src[N,1,Hi,Wi],
 * H64[N,9])) with the fp16-piece stem kernel (d->precision = 4) making the warped pixels while it fetches its patches; warped[N,1,Hi,Wi]
 * (the image, as bh_warp_fwd writes it - bitwise) and cov[N,Hi/4,Wi/4] (the pool-averaged warped all-ones mask, :380-382,447-459 - bitwise
 * bh_warp_fwd's) are written on the way, either may be NULL; bn_sums / groups as bh_conv_fwd_bnstats (NULL: no statistics).  Equal to
 * bh_warp_fwd(src, H64, N, 1, Hi, Wi, 4, warped, cov) followed by bh_conv_fwd_bnstats(warped, ...) - bitwise in y as well (the same
 * pixels into the same arithmetic).  BH_E_UNSUPPORTED where the fp16-piece stem does not apply, or pool != 4: the caller makes the two calls. */
int bh_stem7_fwd_warp(const float* src, const double* H64, int pool, const float* w, const float* bias, float* y, const bh_conv_desc* d,
                      float* warped, float* cov, double* bn_sums, int groups, void* stream);
/* Weight gradient of the backbone's 7x7 / 2 stem on TWO stacked image planes (Rethinking.py:31, ResNet34.py:17) in a dedicated kernel
 * (round 4): x[N,2,Hi,Wi] NCHW, gy[N,Ho,Wo,64] NHWC, gw[64][7][7][2] +=.  Partial sums of the persistent workgroups go through the
 * caller's workspace (bh_stem7_wgrad_ws_bytes(d) bytes, 0 = geometry not taken) and are added in workgroup order: no atomics, bitwise
 * repeatable.  bh_conv_wgrad computes the same gradient for every geometry. */
size_t bh_stem7_wgrad_ws_bytes(const bh_conv_desc* d);
int bh_stem7_wgrad(const float* x, const float* gy, float* gw, const bh_conv_desc* d, float* ws, size_t ws_bytes, void* stream);
/* gw += x^T * gy ; gbias += sum gy  (accumulated: caller zeroes; gbias NULL ok) */
int bh_conv_wgrad(const float* x, const float* gy, float* gw, float* gbias, const bh_conv_desc* d, void* stream);
/* Deterministic form of bh_conv_wgrad for the stride-1 "same" 3x3 / 5x5 / 7x7 layers its fast path takes (Co, Ci multiples
 * of 64, power-of-two maps): the split-K workgroups store their partial tiles into ws (bh_conv_wgrad_det_bytes(d) bytes, 0 =
 * shape not supported) and a second launch adds them in a fixed order - bitwise repeatable gw, no fp32 atomics.
 * With precision 2 (f32x3: 3x3 layers, H and W multiples of 8, channels multiples of 64) this is the FAST form - the kernel
 * keeps one 64 x 9 x 64 block per workgroup and storing + reducing <= 256 of them costs less than the atomics tail.
 * BH_E_UNSUPPORTED for other shapes (use bh_conv_wgrad). */
/* In a deterministic call (BH_ROUTE_DETERMINISTIC) every shape has a workspace form (bh_conv_wgrad_det_bytes > 0) and gbias, when given,
 * is accumulated deterministically as well (its entries are the last Co * 32 bytes of the workspace). */
long long bh_conv_wgrad_det_bytes(const bh_conv_desc* d);
int bh_conv_wgrad_det(const float* x, const float* gy, float* gw, float* gbias, const bh_conv_desc* d, float* ws, long long ws_bytes,
                      void* stream);
/* the bias part alone: gbias[Co] += sum of gy over all output pixels (what bh_conv_wgrad does when gbias != NULL) */
int bh_conv_bias_grad(const float* gy, float* gbias, const bh_conv_desc* d, void* stream);

/* Training-mode BatchNorm2d over `groups` independent sub-batches stacked along N (the reference
 * runs the backbone once per direction and the extractor once per patch, each call with its own
 * batch statistics - Rethinking.py:296-313, PerceptualHead.py:358-398 - this build stacks those
 * calls into one launch and keeps the statistics separate per group).  x,y: [groups*rows, C].
 *   stats[groups, C, 2] double = {mean, biased var}; running stats are updated group after group
 *   with `momentum`, unbiased variance, exactly like consecutive nn.BatchNorm2d calls.
 *   flags: bit0 relu, bit1 residual add (y = act(bn(x) + res)), BH_BN_DETERMINISTIC. eval mode: use_running = 1. */
/* stats (forward): bh_bn_stats_doubles() doubles = [groups,C,2] per-channel sums (sum x, sum x^2), each entry on its
 * own 128-byte line (same-line f64 atomics serialise; layout in csrc/common.h).  Accumulated with f64 atomics: MUST BE
 * ZERO on entry unless eval mode; flags bit3 = the sums were already accumulated by bh_conv_fwd_bnstats.  Kept for the
 * adjoint.  scratch (backward): bh_bn_scratch_doubles() doubles, need not be initialised (coefficient table +
 * per-chunk partial sums, reduced deterministically). */
int bh_bn_stats_doubles(int groups, int C);
int bh_bn_scratch_doubles(int groups, int C);
/* C: a multiple of 4 (<= 1024, C/4 dividing 256), or 1 (round 3: the one-channel BatchNorms of the Zhang feature extractor /
 * mask predictor, ContentAware.py:24-26,68-70 - csrc/bn1.hip, same contract). */
int bh_bn_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
              const float* res, float* y, double* stats, int groups, int rows, int C, float eps, float momentum,
              int flags, int use_running, void* stream);
/* the same, and amax_y (BH_AMAX_FLOATS floats, zeroed by the caller) receives the measured max |y| (precision 4 consumers) */
int bh_bn_fwd_amax(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                   const float* res, float* y, double* stats, int groups, int rows, int C, float eps, float momentum,
                   int flags, int use_running, float* amax_y, void* stream);
/* adjoint. gy: grad w.r.t. y; y: the forward output (for the relu mask); x: forward input.
 * -> gx (overwritten), gres (written when non-NULL: = masked gy), ggamma/gbeta += (NULL ok => frozen).
 * flags bit2 (only without residual): recompute the ReLU mask from x (y is not read, may be NULL).
 * flags bit4: `scratch` already holds the gradient sums (bh_bn_stats_doubles() layout) accumulated by
 * bh_conv_dgrad_bnreduce - one launch instead of three. */
int bh_bn_bwd(const float* gy, const float* y, const float* x, const float* gamma, const float* beta, const double* stats,
              float* gx, float* gres, float* ggamma, float* gbeta, double* scratch,
              int groups, int rows, int C, float eps, int flags, int use_running,
              const float* running_mean, const float* running_var, void* stream);

/* the same, and amax_gx (BH_AMAX_FLOATS floats, zeroed by the caller) receives the measured max |gx| */
int bh_bn_bwd_amax(const float* gy, const float* y, const float* x, const float* gamma, const float* beta, const double* stats,
                   float* gx, float* gres, float* ggamma, float* gbeta, double* scratch,
                   int groups, int rows, int C, float eps, int flags, int use_running,
                   const float* running_mean, const float* running_var, float* amax_gx, void* stream);

/* BatchNorm (+ReLU) + MaxPool2d(3, 2, 1) in one pass (round 4: conv1-bn1-relu-maxpool of the backbone stem, Rethinking.py:31-36, and of the
 * perceptual extractor): x[N,Hi,Wi,C] -> y[N,Ho,Wo,C], idx (one byte per element: window position 0..8 of the first maximum, as
 * bh_maxpool3s2_fwd; NULL ok).  stats / flags / use_running / momentum / amax_y as in bh_bn_fwd_amax (flags: bit0 relu, bit3 sums ready,
 * BH_BN_DETERMINISTIC).  The activation between BatchNorm and pooling is never stored: the adjoint is bh_maxpool3s2_bwd followed by bh_bn_bwd
 * with the mask recomputed from x (flags bit2). */
int bh_bn_maxpool_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* y,
                      unsigned char* idx, double* stats, int groups, int N, int Hi, int Wi, int C, float eps, float momentum, int flags,
                      int use_running, float* amax_y, void* stream);
/* Adjoint of bh_bn_maxpool_fwd in one call (round 5): gy[N,Ho,Wo,C] and idx as written by the forward, x the BatchNorm input -> gx[N,Hi,Wi,C]
 * (and ggamma / gbeta +=, NULL ok).  The full-resolution gradient between pooling and BatchNorm is never stored (bh_maxpool3s2_bwd + bh_bn_bwd
 * wrote and re-read it: 850 -> 490 MB on the extractor stem, /root/reference/src/heads/PerceptualHead.py:50-60).  scratch: bh_bn_scratch_doubles
 * doubles; flags: bit0 relu (mask recomputed from x), BH_BN_DETERMINISTIC; use_running / running_* / amax_gx as in bh_bn_bwd_amax.  The sums are
 * chunk partials added in fixed order: bitwise reproducible in either mode. */
int bh_bn_maxpool_bwd(const float* gy, const unsigned char* idx, const float* x, const float* gamma, const float* beta, const double* stats,
                      float* gx, float* ggamma, float* gbeta, double* scratch, int groups, int N, int Hi, int Wi, int C, float eps, int flags,
                      int use_running, const float* running_mean, const float* running_var, float* amax_gx, void* stream);
/* BatchNorm (+ReLU, no residual) adjoint whose output gradient is the dgrad of a 1x1 / stride-1 convolution with KC = 16 or 32 output channels
 * (round 5): gs[groups*rows][KC] is the gradient of THAT CONVOLUTION's output, w[KC][C] its kernel-layout weight ([Co][1][1][Ci]); the
 * BatchNorm's output gradient g = gs w is rebuilt per element in both passes and never stored (the decoder units' BatchNorm + ReLU + 1x1
 * conv, /root/reference/src/backbones/utils.py:60-82).  Equivalent to bh_conv_dgrad (1x1) followed by bh_bn_bwd_amax with flags bit0 | bit2
 * (training mode), up to the summation order of the KC products.  scratch: bh_bn_scratch_doubles; flags: bit0 relu, BH_BN_DETERMINISTIC. */
int bh_bn_bwd_from_1x1(const float* gs, const float* w, int KC, const float* x, const float* gamma, const float* beta, const double* stats,
                       float* gx, float* ggamma, float* gbeta, double* scratch, int groups, int rows, int C, float eps, int flags,
                       float* amax_gx, void* stream);
/* Two-branch join (round 4): y = act(bn_a(xa) + bn_b(xb)), both BatchNorms in training mode with their own statistics tables (already
 * accumulated: by the producers' epilogues or bh_bn_stats-style passes) - the end of ResNet50DeconvBlock / the strided ResNet34ConvBlock
 * (src/backbones/utils.py:60-82, 85-112) without writing the normalised lower branch.  flags: bit0 relu, BH_BN_DETERMINISTIC.  Running
 * statistics of both are updated (NULL: skipped).  amax_y as in bh_bn_fwd_amax.
 * Adjoint: gy, y (ReLU mask), xa, xb -> gxa, gxb (overwritten), ggamma / gbeta of both += (NULL ok); scratch: bh_bn_join_scratch_doubles()
 * doubles; amax_gxa / amax_gxb (NULL ok) receive max |gxa| / max |gxb|.  One reduce pass for the three sums (sum d, sum d xhat_a,
 * sum d xhat_b with d = gy [y > 0]) and one apply pass. */
/* the statistics pass alone: stats (zeroed, bh_bn_stats_doubles() doubles) += per-channel (sum x, sum x^2); flags: BH_BN_DETERMINISTIC */
int bh_bn_stats(const float* x, double* stats, int groups, int rows, int C, int flags, void* stream);
int bh_bn_join_scratch_doubles(int groups, int C);
int bh_bn_join_fwd(const float* xa, const float* xb, const float* gamma_a, const float* beta_a, float* rmean_a, float* rvar_a,
                   const float* gamma_b, const float* beta_b, float* rmean_b, float* rvar_b, const double* stats_a, const double* stats_b,
                   float* y, int groups, int rows, int C, float eps_a, float eps_b, float momentum_a, float momentum_b, int flags,
                   float* amax_y, void* stream);
int bh_bn_join_bwd(const float* gy, const float* y, const float* xa, const float* xb, const float* gamma_a, const float* gamma_b,
                   const double* stats_a, const double* stats_b, float* gxa, float* gxb, float* ggamma_a, float* gbeta_a, float* ggamma_b,
                   float* gbeta_b, double* scratch, int groups, int rows, int C, float eps_a, float eps_b, int flags, float* amax_gxa,
                   float* amax_gxb, void* stream);
/* The same adjoint with the ReLU mask RECOMPUTED from xa, xb (round 5): y is not read - the kernels evaluate the forward kernel's own
 * expression y = fma(xa, sc_a, fma(xb, sc_b, sh_a + sh_b)) with the same coefficient arithmetic, so the mask is bitwise the forward's
 * decision and the result bitwise bh_bn_join_bwd's; one of four input streams less in both passes (1.34 -> 1.07 GB at the full-resolution
 * join of ResNet50DeconvBlock, /root/reference/src/backbones/utils.py:60-82).  beta_a / beta_b: the BatchNorms' shifts (NULL: 0). */
int bh_bn_join_bwd_remask(const float* gy, const float* xa, const float* xb, const float* gamma_a, const float* beta_a, const float* gamma_b,
                          const float* beta_b, const double* stats_a, const double* stats_b, float* gxa, float* gxb, float* ggamma_a,
                          float* gbeta_a, float* ggamma_b, float* gbeta_b, double* scratch, int groups, int rows, int C, float eps_a, float eps_b,
                          int flags, float* amax_gxa, float* amax_gxb, void* stream);

/* Fused tail of the Zeng backbone, `layer8` (src/backbones/Rethinking.py:145-147):
 *   Conv2d(Ci,Cm,1,bias) -> BatchNorm2d(Cm) -> ReLU -> Conv2d(Cm,Co,1,bias), NHWC x[groups*rows,Ci] -> NCHW out[N,Co,h,w]
 * (hw = h*w pixels per image, rows = pixels per group).  The Cm-channel intermediate is never materialised: its batch
 * statistics are derived from the first/second moments of x (a 1x1 conv is linear).  Ci in {8,16}, Cm multiple of
 * 64 (<= 256), Co <= 4.  ws: bh_tail_ws_doubles() doubles (kept for the adjoint); scratch: bh_tail_scratch_floats(). */
int bh_tail_ws_doubles(int groups, int Ci, int Cm);
int bh_tail_scratch_floats(int groups, int Ci, int Cm);
int bh_tail_fwd(const float* x, const float* w1, const float* b1, const float* gamma, const float* beta,
                float* running_mean, float* running_var, const float* w2, const float* b2, float* out, double* ws,
                int groups, int rows, int hw, int Ci, int Cm, int Co, float eps, float momentum, int use_running,
                void* stream);
/* Same call with a per-call route (no process state): 0 = the library's choice - round 4: for Ci = 16, Co <= 2, rows % 32 == 0 and
 * hw % 32 == 0 the Ci -> Cm product runs on the matrix pipe in the exact three-bf16-piece arithmetic (tail_fwd_mfma_kernel);
 * bit 0 (BH_TAIL_ROUTE_VALU_FWD) keeps the per-pixel VALU kernel (tests compare the two, tools time them). */
#define BH_TAIL_ROUTE_VALU_FWD 1
#define BH_TAIL_ROUTE_LDS_MOMENTS 2      /* bit 1: the input moments by the LDS-slab kernel instead of the matrix-pipe one (Ci = 16) */
int bh_tail_fwd_route(const float* x, const float* w1, const float* b1, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, const float* w2, const float* b2, float* out, double* ws,
                      int groups, int rows, int hw, int Ci, int Cm, int Co, float eps, float momentum, int use_running,
                      int route, void* stream);
/* adjoint: gout[N,Co,h,w] -> gx[groups*rows,Ci] (overwritten, NULL ok); gw1[Cm][Ci], ggamma, gbeta, gw2[Co][Cm], gb2 +=
 * Round 4: pixels whose output gradient is exactly zero are skipped by the reduction (they add exactly zero: on the biHomE path only
 * the DSAC-sampled points of the field carry a gradient), and gx = c0 - M x (the BatchNorm mean terms, an affine map of x made once
 * per group) + the Cm-channel term where the gradient is non-zero.  Dense gradients take the per-lane channel loop as before. */
int bh_tail_bwd(const float* gout, const float* x, const float* w1, const float* b1, const float* gamma,
                const float* beta, const float* w2, const double* ws, const float* running_mean,
                const float* running_var, float* gx, float* gw1, float* ggamma, float* gbeta, float* gw2, float* gb2,
                float* scratch, int groups, int rows, int hw, int Ci, int Cm, int Co, float eps, int use_running,
                void* stream);
int bh_tail_bwd_f(const float* gout, const float* x, const float* w1, const float* b1, const float* gamma,
                  const float* beta, const float* w2, const double* ws, const float* running_mean,
                  const float* running_var, float* gx, float* gw1, float* ggamma, float* gbeta, float* gw2, float* gb2,
                  float* scratch, int groups, int rows, int hw, int Ci, int Cm, int Co, float eps, int use_running,
                  int flags, void* stream);                                   /* flags: BH_F_DETERMINISTIC (the bias gradient gb2) */

/* Synthetic pair generator (next-row f1): HomographyNetPrep + PhotometricDistortSimple + DictToGrayscale +
 * DictStandardize of src/data/transforms.py:296-330,344-378,441-725 for B samples in one launch.
 * images[n_images,3,Hs,Ws] float RGB 0..255 (resident); img_idx[B]; origin[B,2] = top-left corner (x0,y0) of the
 * patch; Hpatch[B,9] double = bh_h4pt_fwd(delta) in patch coordinates; photo[B,2,6] = per image of the pair
 * {brightness delta, contrast factor applied before the HSV part, saturation factor, hue delta (degrees), contrast factor
 * applied after the HSV part, channel-permutation index 0..5 into ((0,1,2),(0,2,1),(1,0,2),(1,2,0),(2,0,1),(2,1,0))}
 * (transforms.py:141-245; OpenCV float HSV), or NULL for no distortion.  The distortion is applied to the image before
 * it is warped, as upstream.  patch1 = crop, patch2(x) = image(origin + Hpatch.x) bilinear; both standardised
 * ((gray/255 - mean)/std), [B,1,P,P]. */
int bh_synth_pairs(const float* images, const int* img_idx, const float* origin, const double* Hpatch, const float* photo,
                   int B, int n_images, int Hs, int Ws, int P, float mean, float std, float* patch1, float* patch2,
                   void* stream);

/* MaxPool2d(3, 2, 1) NHWC. argmax[N,Ho,Wo,C] (uint8, NULL ok in inference): window position 0..8 of the first
 * maximum (ATen's tie rule); the adjoint gathers through it (no atomics, no recomputation). */
int bh_maxpool3s2_fwd(const float* x, float* y, unsigned char* argmax, int N, int Hi, int Wi, int C, void* stream);
int bh_maxpool3s2_bwd(const unsigned char* argmax, const float* gy, float* gx, int N, int Hi, int Wi, int C, void* stream);
/* AdaptiveAvgPool2d(1) NHWC -> [N,C], and adjoint */
int bh_gap_fwd(const float* x, float* y, int N, int HW, int C, void* stream);
int bh_gap_bwd(const float* gy, float* gx, int N, int HW, int C, void* stream);
/* out = a + b (elementwise, used to join gradient branches) */
int bh_add(const float* a, const float* b, float* out, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
