/*
 * og_decoder.h -- C ABI of libog_decoder.so: the OffsetGuided decoder hot path as
 * hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (hellojialee/OffsetGuided) is pure Python: it has no native boundary of
 * its own.  Each entry point below replaces the torch-op sequence of one reference
 * function (file:line given per function, relative to the reference repository root); the
 * Python package offsetguided_amd.decoder binds them with ctypes behind the reference's own
 * names (hmp_NMS, topK_channel, joint_dets, LimbsCollect, GreedyGroup, PostProcess).
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - all tensor pointers are DEVICE pointers to dense, contiguous fp32 / int64 / int32 data
 *     in the reference's NCHW layout; nothing is copied or retained;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call is
 *     stream-ordered, asynchronous, and performs no allocation and no host synchronisation
 *     (hipGraph-capturable); scratch comes from the caller through `workspace`;
 *   - return value: 0 on success, negative OG_E* code on failure; og_last_error() returns a
 *     thread-local message for the last failure on the calling thread; nothing throws;
 *   - inputs must be finite (NaN ordering of torch.topk is not reproduced);
 *   - no CPU fallback exists: without a HIP device every compute entry point fails.
 */
#ifndef OG_DECODER_H
#define OG_DECODER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OG_ABI_VERSION 3

#define OG_OK 0
#define OG_EINVAL (-1)    /* bad argument (shape, k, alignment, null pointer)        */
#define OG_ENOSPC (-2)    /* workspace too small                                     */
#define OG_EHIP (-3)      /* HIP runtime error (launch failure, no device)           */
#define OG_EUNSUPPORTED (-4)

int og_abi_version(void);
const char *og_last_error(void);

/* Number of HIP devices visible, or a negative OG_E* code. */
int og_device_count(void);

/* ---- a5: F.interpolate(hmps, scale_factor=4, mode='bicubic')  decoder/factory.py:74-75 ----
 * src (planes,h,w) -> dst (planes,4h,4w); torch-CPU fp32 arithmetic, bit-exact (A=-0.75,
 * align_corners=False, index-clamped taps). */
int og_upsample_bicubic4_f32(const float *src, long planes, int h, int w, float *dst, void *stream);

/* ---- a5: F.interpolate(offs, scale_factor=4, mode='bilinear')  decoder/factory.py:77-78 ----
 * Full materialisation; the decode path does not need it (og_collect_limbs_f32 samples the
 * low-res map at the peaks with the same arithmetic). */
int og_upsample_bilinear4_f32(const float *src, long planes, int h, int w, float *dst, void *stream);

/* ---- a6: hmp_NMS  decoder/heatmap.py:15-35 ----
 * out = heat * (maxpool3x3(zero-padded heat) == heat); heat/out (planes,H,W). */
int og_hmp_nms_f32(const float *heat, long planes, int H, int W, float *out, void *stream);

/* ---- a7: topK_channel  decoder/heatmap.py:38-49 ----
 * Per-plane top-k of `scores` (planes, hw), sorted by value descending; ties: lower flat
 * index first (-0.0 == +0.0).  Outputs scores (planes,k) f32 and inds (planes,k) i64; the
 * caller derives ys = inds / w (floor) and xs = inds % w.
 * workspace: og_topk_workspace_bytes(planes, H, W, k). */
int og_topk_channel_f32(const float *scores, long planes, int H, int W, int k, float *out_scores,
                        int64_t *out_inds, void *workspace, size_t workspace_bytes, void *stream);

/* ---- a8: joint_dets = topK_channel(hmp_NMS(hmps), k)  decoder/heatmap.py:52-59 ----
 * Fused single pass over the hi-res heatmaps: the NMS map is never written.  Same outputs
 * and tie rule as og_topk_channel_f32 applied to og_hmp_nms_f32's result (zero-valued filler
 * entries, if a plane has fewer than k positive peaks, are the lowest flat indices whose NMS
 * value is zero).  Requires 2*(H+W)-4 >= k.
 * workspace: og_topk_workspace_bytes(planes, H, W, k). */
int og_nms_topk_f32(const float *hmps, long planes, int H, int W, int k, float *out_scores,
                    int64_t *out_inds, void *workspace, size_t workspace_bytes, void *stream);

/* ---- a5+a8 fused: joint_dets(F.interpolate(hmps, x4, 'bicubic'), k)  decoder/factory.py:74-75 + heatmap.py:52-59 ----
 * hmps_lr (planes,h,w) = stride-4 head output.  The (planes,4h,4w) hi-res heatmap is never
 * materialised: every band computes its hi-res rows on the fly (same rounding as
 * og_upsample_bicubic4_f32) and feeds them to the same NMS / top-k machinery.  Outputs, tie rule
 * and workspace exactly as og_nms_topk_f32 on the upsampled tensor (indices are hi-res flat
 * indices): workspace og_topk_workspace_bytes(planes, 4h, 4w, k). */
int og_upsample_nms_topk_f32(const float *hmps_lr, long planes, int h, int w, int k, float *out_scores,
                             int64_t *out_inds, void *workspace, size_t workspace_bytes, void *stream);

size_t og_topk_workspace_bytes(long planes, int H, int W, int k);

/* ---- a9+a10: LimbsCollect.generate_limbs  decoder/collect.py:62-236 (+ _channel_dets :246-254) ----
 * scores/inds : (N,C,k) from og_nms_topk_f32 on (N,C,H,W) hi-res heatmaps
 * offs        : off_is_lowres ? (N,2L,H/4,W/4) stride-4 head output, bilinearly sampled at the
 *               from-peaks exactly as factory.py:77-78 + collect.py:143-147 would
 *             : (N,2L,H,W) hi-res offsets, gathered
 * jf/jt       : device int32[L] from/to joint channel per limb (LimbsCollect.pack_jtypes)
 * limbs       : (N,L,k,13) [x1,y1,v1,x2,y2,v2,ind1,ind2,min_dist,len,score,scale1,scale2]
 * No scale / jitter heads (scales are the constant 4, collect.py:117-122). */
int og_collect_limbs_f32(const float *scores, const int64_t *inds, const float *offs, int off_is_lowres,
                         int N, int C, int H, int W, const int32_t *jf, const int32_t *jt, int L, int k,
                         float thre_hmp, float min_len, float resize_factor, float *limbs, void *stream);

/* Same with `vector_nd` offset components per limb: 2 = og_collect_limbs_f32; 4 = the `cat_flip_offs` form
 * (decoder/factory.py:115-127 -> collect.py:62 vector_nd=4): offs (N,4L,..) from og_flip_cat_f32, match distance =
 * the 4-D norm of (guide - to, guide' - to). */
int og_collect_limbs_nd_f32(const float *scores, const int64_t *inds, const float *offs, int off_is_lowres,
                            int vector_nd, int N, int C, int H, int W, const int32_t *jf, const int32_t *jt, int L,
                            int k, float thre_hmp, float min_len, float resize_factor, float *limbs, void *stream);

/* Same with the keypoint-scale head (decoder/collect.py:111-122, :257-262): limbs columns 11 / 12 = the scale map of
 * the from / to joint channel at the from-peak / the matched to-peak instead of the constant 4.
 * scales_mode 0: no head (scales NULL); 1: scales (N,C,H,W) at input resolution, gathered; 2 / 3: scales (N,C,H/4,W/4)
 * = the head output, sampled as F.interpolate(x4, 'bicubic' / 'bilinear') would (decoder/factory.py:80-82). */
int og_collect_limbs_ex_f32(const float *scores, const int64_t *inds, const float *offs, int off_is_lowres,
                            int vector_nd, const float *scales, int scales_mode, int N, int C, int H, int W,
                            const int32_t *jf, const int32_t *jt, int L, int k, float thre_hmp, float min_len,
                            float resize_factor, float *limbs, void *stream);

/* ... and with the jitter-offset head (decoder/collect.py:127-138, :154-165, :210-214; include_jitter_offset with
 * use_jitter_offset): jitter (N,2,..) = the two shared refinement channels; jitter_mode 0: none; 1: maps at input
 * resolution; 3: stride-4 head output, sampled as F.interpolate(x4, 'bilinear') would (decoder/factory.py:84-88).
 * The guide point is refined by the vector read at its truncated coordinates ([x][y] indexing of the reference:
 * square inputs only) and the limb's end points move by the vectors at their own peaks. */
int og_collect_limbs_full_f32(const float *scores, const int64_t *inds, const float *offs, int off_is_lowres,
                              int vector_nd, const float *scales, int scales_mode, const float *jitter, int jitter_mode,
                              int N, int C, int H, int W, const int32_t *jf, const int32_t *jt, int L, int k,
                              float thre_hmp, float min_len, float resize_factor, float *limbs, void *stream);

/* ---- a8+a9+a10 in ONE call: LimbsCollect.generate_limbs  decoder/collect.py:62-236 on (N,C,H,W) hi-res heatmaps ----
 * = og_nms_topk_f32 (joint_dets, decoder/heatmap.py:52-59) followed by og_collect_limbs_full_f32, same arguments and
 * bit-identical limbs.  topk_scores / topk_inds: optional (N,C,k) outputs of the joint_dets stage (both or neither).
 * Two launches queued back to back -- band top-k, then ONE kernel that merges the band lists and pairs the limbs (shapes
 * whose merge stage does not fit the LDS finish with og_collect_limbs_full_f32's kernel).  `flags` is reserved: pass 0 (rounds 2
 * and 3 selected two one-launch forms there -- a persistent kernel and the merge + pairing by last arrivers of the band launch --
 * both bit-identical and slower; they live on as tools/experiments/k1_single.inc / EXPERIMENTS.md).
 * workspace: og_generate_limbs_workspace_bytes(N, C, H, W, k) bytes, 16-byte aligned, ZERO-FILLED by the caller
 * (hipMemset) before its first use; every call leaves it ready for the next one (any shape).
 */
int og_generate_limbs_f32(const float *hmps_hr, const float *offs, int off_is_lowres, int vector_nd,
                          const float *scales, int scales_mode, const float *jitter, int jitter_mode,
                          int N, int C, int H, int W, const int32_t *jf, const int32_t *jt, int L, int k,
                          float thre_hmp, float min_len, float resize_factor, float *topk_scores,
                          int64_t *topk_inds, float *limbs, int flags, void *workspace, size_t workspace_bytes,
                          void *stream);

size_t og_generate_limbs_workspace_bytes(int N, int C, int H, int W, int k);

/* ---- flip-test without a merge pass: PostProcess.flip_augment (decoder/factory.py:98-146, the averaged form) folded into the
 * loads of its two consumers.  Both take the head outputs of [images | mirrored images] (2N leading) and compute every value
 * they read exactly as og_flip_merge_f32 would have written it -- (a + flipW(b)[perm]) / 2, x offsets negated, the limbs of
 * `reserve_mask` un-averaged -- so the results are bit-identical to og_flip_merge_f32 followed by og_upsample_bicubic4_f32 /
 * og_generate_limbs_f32 (2-component offsets sampled from the stride-4 map, no scale / jitter head), one pass over the
 * 3 x 5.6 MB/img stride-4 maps and one launch fewer.
 *   og_upsample_bicubic4_flip_f32: hm_pair (2N,C,h,w) -> dst (N,C,4h,4w); kp_perm int32[C] (config.heatmap_hflip).
 *   og_generate_limbs_flip_f32: offs_pair (2N,2L,H/4,W/4); limb_perm / reserve_mask int32[L] (config.offset_hflip); the other
 *     arguments, the workspace and the outputs as og_generate_limbs_f32. */
int og_upsample_bicubic4_flip_f32(const float *hm_pair, const int32_t *kp_perm, int N, int C, int h, int w, float *dst, void *stream);
int og_generate_limbs_flip_f32(const float *hmps_hr, const float *offs_pair, const int32_t *limb_perm, const int32_t *reserve_mask,
                               int N, int C, int H, int W, const int32_t *jf, const int32_t *jt, int L, int k, float thre_hmp,
                               float min_len, float resize_factor, float *topk_scores, int64_t *topk_inds, float *limbs,
                               void *workspace, size_t workspace_bytes, void *stream);

/* ---- K1-fused (the production path): generate_limbs straight from the STRIDE-4 head outputs.  The x4 bicubic of
 * decoder/factory.py:74-75 runs inside the NMS kernel (bit-identical to og_upsample_bicubic4_f32), offsets / scale / jitter maps are
 * sampled at the peaks: neither hi-res tensor of factory.py:74-88 is built.  Same limbs, same optional (N,C,k) lists, same two
 * launches and the same workspace as og_generate_limbs_f32 on (N,C,4h,4w): og_generate_limbs_workspace_bytes(N, C, 4h, 4w, k).
 *   hmps_lr (N,C,h,w), offs_lr (N,vector_nd*L,h,w); scales_lr (N,C,h,w) with scales_mode 2 / 3 (bicubic / bilinear) or NULL / 0;
 *   jitter_lr (N,2,h,w) with jitter_mode 3 or NULL / 0.
 * og_generate_limbs_fused_flip_f32: flip-test (decoder/factory.py:98-146, averaged form, 2-component offsets, no scale / jitter
 *   head) folded into both consumers: hm_pair_lr (2N,C,h,w), offs_pair_lr (2N,2L,h,w) = head outputs of [images | mirrored images];
 *   kp_perm int32[C], limb_perm / reserve_mask int32[L] device arrays (config.heatmap_hflip / offset_hflip).  Bit-identical to
 *   og_flip_merge_f32 + og_generate_limbs_fused_f32. */
int og_generate_limbs_fused_f32(const float *hmps_lr, const float *offs_lr, int vector_nd, const float *scales_lr, int scales_mode,
                                const float *jitter_lr, int jitter_mode, int N, int C, int h, int w, const int32_t *jf,
                                const int32_t *jt, int L, int k, float thre_hmp, float min_len, float resize_factor,
                                float *topk_scores, int64_t *topk_inds, float *limbs, void *workspace, size_t workspace_bytes,
                                void *stream);
int og_generate_limbs_fused_flip_f32(const float *hm_pair_lr, const int32_t *kp_perm, const float *offs_pair_lr,
                                     const int32_t *limb_perm, const int32_t *reserve_mask, int N, int C, int h, int w,
                                     const int32_t *jf, const int32_t *jt, int L, int k, float thre_hmp, float min_len,
                                     float resize_factor, float *topk_scores, int64_t *topk_inds, float *limbs, void *workspace,
                                     size_t workspace_bytes, void *stream);

/* ---- a12: GreedyGroup.group_skeletons  decoder/group.py:39-185 (+ :187-240) ----
 * One workgroup per image, device resident (replaces .cpu().numpy() + Pool.starmap,
 * decoder/factory.py:91-94).
 * limbs (N,L,k,13) -> poses (N,mmax,n_kp,6) [x,y,v,scale,limb_score,global_idx],
 * counts int32[N] = poses per image, status int32[N] = 0 ok / 1 subset table overflowed
 * (more than `mmax` partial skeletons alive; results for that image are then invalid).
 * Any skeleton / top-k: the partial-skeleton table and, for large L*k, the staged candidate rows move from LDS to the
 * workspace (L*k <= ~6 000 at mmax 128; OG_EUNSUPPORTED beyond).
 * workspace: og_group_workspace_bytes(N, L, k, n_kp, mmax). */
int og_greedy_group_f32(const float *limbs, int N, int L, int k, const int32_t *jf, const int32_t *jt,
                        int n_kp, double person_thre, float dist_max, int use_scale, int sort_dim, int mmax,
                        float *poses, int32_t *counts, int32_t *status, void *workspace, size_t workspace_bytes,
                        void *stream);

size_t og_group_workspace_bytes(int N, int L, int k, int n_kp, int mmax);

/* ---- a4: PostProcess.flip_augment (vector-addition form)  decoder/factory.py:98-146 ----
 * hm (2N,C,h,w), off (2N,2L,h,w) -> hm_out (N,C,h,w), off_out (N,2L,h,w).
 * kp_perm int32[C], limb_perm int32[L], reserve_mask int32[L] (1 = keep the un-averaged
 * original, config.offset_hflip()[1]) are device arrays. */
int og_flip_merge_f32(const float *hm, const float *off, int N, int C, int L, int h, int w,
                      const int32_t *kp_perm, const int32_t *limb_perm, const int32_t *reserve_mask,
                      float *hm_out, float *off_out, void *stream);

/* ---- a4, cat_flip_offs=True form  decoder/factory.py:115-127 ----
 * Same inputs; off_out (N,4L,h,w): per limb [x, y, mirrored x, mirrored y] (reserve limbs repeat x, y). */
int og_flip_cat_f32(const float *hm, const float *off, int N, int C, int L, int h, int w,
                    const int32_t *kp_perm, const int32_t *limb_perm, const int32_t *reserve_mask,
                    float *hm_out, float *off_out, void *stream);

/* ---- backbone epilogues (bf16, channels-last / NHWC activations of the inference engine) ----
 * Stand-alone epilogue / layout passes.  In the engine every convolution carries its epilogue itself (og_conv*_ below);
 * og_bias_act_* is what remains for a convolution that torch ran (InferenceEngine(strict=False) only), og_upsample2_add_* the
 * hourglass merge at the levels whose last convolution is not the tiled kernel.
 *
 * og_bias_act_bf16: x (pixels, channels) bf16, in place:  x = act(x + bias[c] (+ skip))
 *   = convolution.forward models/hourglass_104.py:26-30 (BN folded into bias, ReLU) and
 *     residual.forward :70-79 (bn2 + skip, ReLU).  bias fp32[channels]; skip bf16 like x or NULL;
 *     channels % 8 == 0; fp32 arithmetic, one rounding to bf16.
 * og_upsample2_add_bf16: up (n,H,W,channels) += nearest_x2(low (n,H/2,W/2,channels))
 *   = kp_module.forward :183-190 (up1 + up2 merge). */
int og_bias_act_bf16(void *x, const float *bias, const void *skip, long pixels, int channels, int relu, void *stream);
int og_upsample2_add_bf16(void *up, const void *low, long n, int H, int W, int channels, void *stream);

/* Engine boundary conversions (models/networks.py:189-194 hands fp32 NCHW in and out; the engine computes bf16 NHWC):
 * og_nchw_f32_to_nhwc_bf16: images (N,3,H,W) fp32 -> (N,H,W,3) bf16, one pass.
 * og_nhwc_bf16_to_nchw_f32: channels [first, first+channels) of src (N,H,W,src_channels) bf16, plus bias[first+c]
 *   (fp32[src_channels] or NULL), -> dst (N,channels,H,W) fp32: the head maps of models/heads.py:48-70,116-142 out of
 *   ONE 1x1 convolution that evaluates all heads. */
int og_nchw_f32_to_nhwc_bf16(const float *src, void *dst, long N, int C, int H, int W, void *stream);
int og_nhwc_bf16_to_nchw_f32(const void *src, int src_channels, int first_channel, int channels, const float *bias,
                             float *dst, long N, int H, int W, void *stream);

/* ---- evaluate.py input side (the rescale by cv2.resize is NOT included) ----
 * CenterPad + ToTensor + Normalize, transforms/pad.py:40-66 + evaluate.py:163-168: img (h,w,3) uint8 RGB on the device ->
 * out (3,target_h,target_w) fp32 = (px/255 - mean)/std with px the image pixel or `fill3` (124,116,104 in the
 * reference); mean3/std3/fill3 are HOST arrays of 3 floats; ltrb (host int[4], may be NULL) receives the paddings that
 * the annotations / annotations_inverse need. */
int og_center_pad_normalize_u8(const unsigned char *img, int h, int w, int target_h, int target_w, const float *mean3,
                               const float *std3, const float *fill3, float *out, int *ltrb, void *stream);

/* RescaleLongAbsolute's cv2.resize(INTER_CUBIC) (transforms/scale.py:27) for (h,w,3) uint8 images in HBM -> (new_h,new_w,3).
 * OpenCV's published 8-bit algorithm (fixed-point taps, replicated border); pinned bit-exactly to the CPU restatement
 * oracle/og_oracle.c:ogo_resize_cubic_u8 -- parity with cv2 itself is unpinned (third-party, absent from the build). */
int og_resize_cubic_u8(const unsigned char *src, int h, int w, unsigned char *dst, int new_h, int new_w, void *stream);

/* GT encoder input: the full-resolution uint8 mask_miss (N,h,w), 0 / 255, to the boolean mask at output
 * resolution -- cv2.resize(fx = fy = 1 / stride, INTER_CUBIC) / 255 > 0.7, encoder/heatmap.py:56-60, encoder/offset.py:46-50.
 * out: (N, round(h / stride), round(w / stride)) bytes 0 / 1.  Same published 8-bit algorithm as og_resize_cubic_u8
 * (pinned to oracle/og_oracle.c:ogo_shrink_mask_miss_u8; parity with cv2 itself unpinned). */
int og_shrink_mask_miss_u8(const unsigned char *mask, int N, int h, int w, int stride, unsigned char *out, void *stream);

/* The whole input chain of evaluate.py:150-168 in one pass: rescale to (new_h,new_w) as above, pad to (target_h,target_w)
 * with the fill colour -- corner_pad 0: CenterPad (transforms/pad.py:35-62, the --long-edge chain), 1: RightDownPad
 * (transforms/pad.py:70-118, the --fixed-height chain: left = top = 0) --, ToTensor, Normalize -> out fp32
 * (3,target_h,target_w); ltrb as og_center_pad_normalize_u8.  The resized uint8 image is never stored. */
int og_rescale_pad_normalize_u8(const unsigned char *img, int h, int w, int new_h, int new_w, int target_h, int target_w,
                                int corner_pad, const float *mean3, const float *std3, const float *fill3, float *out,
                                int *ltrb, void *stream);

/* The same chain for a whole batch in ONE launch (evaluate.py:157-182 collates the images of a batch before the network sees
 * them): `raw` = the batch's uint8 images packed back to back in HBM; offsets (host long[n]) = byte offset of image i;
 * hw4 (host int[4 n]) = (h, w, new_h, new_w) of image i; out fp32 (n,3,target_h,target_w); ltrb (host int[4 n] or NULL).
 * Bit-identical to n calls of og_rescale_pad_normalize_u8. */
int og_rescale_pad_normalize_batch_u8(const unsigned char *raw, const long *offsets, const int *hw4, int n, int target_h,
                                      int target_w, int corner_pad, const float *mean3, const float *std3, const float *fill3,
                                      float *out, int *ltrb, void *stream);

/* ---- network stem: convolution(7, 3, 128, stride=2) + BN + ReLU  models/hourglass_104.py:283, :16-30 ----
 * images (N,3,H,W) fp32 (H, W multiples of 32) -> out (N,H/2,W/2,128) bf16 NHWC, input conversion and epilogue fused.
 * w_packed bf16 [128][7 kernel rows][8 taps][4 channels] (tap 7 and channel 3 zero) = the BN-folded weight
 * (128,3,7,7) permuted to (cout, ky, kx, ch) and zero-padded; bias fp32[128]. */
int og_stem7x7_bf16(const float *images, const void *w_packed, const float *bias, void *out, int N, int H, int W, int relu,
                    void *stream);

/* ---- 3x3 stride-1 pad-1 convolution with the epilogue fused, for the small inner hourglass levels ----
 * out = act(conv3x3(x, w) + bias (+ skip)):  convolution.forward models/hourglass_104.py:26-30 / residual.forward
 * :70-79 with BN folded.  x (N,H,W,Cin), w (Cout,3,3,Cin) [= channels_last (Cout,Cin,3,3)], skip/out (N,H,W,Cout),
 * all bf16; bias fp32[Cout]; Cin, Cout multiples of 64; fp32 accumulation, one rounding to bf16.
 * Split-K implicit GEMM on MFMA: meant for N*H*W of a few hundred to a few thousand pixels, where library
 * kernels leave most CUs idle.  workspace: og_conv3x3_workspace_bytes(N*H*W, Cin, Cout), 256-byte aligned,
 * ZERO-INITIALISED once by the caller (its first 256 bytes are read as the zero padding and never written). */
int og_conv3x3_bf16(const void *x, const void *w, const float *bias, const void *skip, void *out, int N, int H, int W,
                    int Cin, int Cout, int relu, void *workspace, size_t workspace_bytes, void *stream);
size_t og_conv3x3_workspace_bytes(long pixels, int Cin, int Cout);        /* upper bound for any H x W with N*H*W = pixels */
size_t og_conv3x3_workspace_bytes_nhw(int N, int H, int W, int Cin, int Cout);  /* exact for this shape */
/* General form for the remaining convolutions of the hourglass: ksize 1 (pad 0) or 3 (pad 1), stride 1 or 2 --
 * residual.conv1 with stride 2 and the 1x1 projection `skip` (models/hourglass_104.py:54-57, :63-68), the 1x1
 * inters_/cnvs_ junction (:239-250) and the head convolutions (models/heads.py).  x (N,Hin,Win,Cin),
 * w (Cout,ksize,ksize,Cin), skip/out (N,Hout,Wout,Cout) with Hout = (Hin + 2*(ksize/2) - ksize)/stride + 1;
 * same dtype / channel / workspace rules as og_conv3x3_bf16 (which is og_conv2d_bf16 with ksize 3, stride 1);
 * workspace: og_conv2d_workspace_bytes (0 = unsupported shape). */
int og_conv2d_bf16(const void *x, const void *w, const float *bias, const void *skip, void *out, int N, int Hin, int Win,
                   int Cin, int Cout, int ksize, int stride, int relu, void *workspace, size_t workspace_bytes,
                   void *stream);
size_t og_conv2d_workspace_bytes(int N, int Hin, int Win, int Cin, int Cout, int ksize, int stride);
/* A whole projection residual tail in one launch: out = act(conv(x) + conv1x1(x2, stride2) + bias) -- residual.forward
 * models/hourglass_104.py:70-79 with the `skip` branch (:63-68) a 1x1 convolution + BN: bn2(conv2(.)) + skip(x), ReLU.
 * The projection is appended along K: w_cat (Cout, ksize*ksize*Cin + Cin2) = [conv weight (Cout,k,k,Cin) | projection
 * weight (Cout,Cin2)], x2 (N,H2,W2,Cin2) sampled at (y*stride2, x*stride2); bias = the sum of both folded biases.
 * Always the split-K kernel (meant for the small levels); workspace: og_conv2d_proj_workspace_bytes. */
int og_conv2d_proj_bf16(const void *x, const void *w_cat, const float *bias, const void *x2, void *out, int N, int Hin,
                        int Win, int Cin, int Cout, int ksize, int stride, int H2, int W2, int Cin2, int stride2, int relu,
                        void *workspace, size_t workspace_bytes, void *stream);
size_t og_conv2d_proj_workspace_bytes(int N, int Hin, int Win, int Cin, int Cout, int ksize, int stride, int Cin2);
/* ---- band-resident convolution for the SMALL levels (20x20 / 10x10 / 5x5 at batch 8): csrc/conv_band.hip.  A workgroup owns
 * (image, band of output rows, 16 output channels) and all of K; K is split over its four waves (wave w = a quarter of the input
 * channels, all nine taps), whose weight fragments go from the pre-packed image straight into registers; the band's input rows sit
 * in LDS once (zero pixels between the rows: a tap is a plain address shift); the waves' fp32 partial tiles meet in LDS: no slabs,
 * no tickets, nothing but finished activations is handed between workgroups.  Same arithmetic as og_conv2d_* / og_conv2d_proj_*
 * (fp32 accumulation over the same products, one rounding; the summation ORDER differs, so results agree to fp32 rounding, not bit
 * for bit).  Replaces convolution.forward models/hourglass_104.py:26-30 and residual.forward :70-79 (BN folded), stride 1 or 2,
 * with the residual's 1x1 projection `skip` (:63-68) as extra K steps.
 *   og_conv_band_supported: 0 = not served (needs 64 <= Cin (and Cin2) <= 512 in multiples of 32, Cout % 16 == 0, stride 1 | 2,
 *     pad 1, input width <= 112, a band's rows in 160 KiB of LDS); otherwise the number of workgroups of the launch.
 *   og_conv_band_pack_w16: w (Cout,3,3,Cin) = the memory of a channels_last (Cout,Cin,3,3) tensor [+ w2 (Cout,Cin2), or NULL with
 *     Cin2 = 0] -> packed, Cout * (9*Cin + Cin2) elements, once per layer.
 *   og_conv_band_*: x (N,Hin,Win,Cin), skip / out (N,H,W,Cout) with H = (Hin-1)/stride + 1, x2 (N,H2,W2,Cin2) sampled at
 *     (y*stride2, x*stride2) or NULL; bias fp32[Cout] (with a projection: the sum of both folded biases). */
int og_conv_band_supported(int N, int Hin, int Win, int Cin, int Cout, int stride, int H2, int W2, int Cin2, int stride2);
int og_conv_band_pack_w16(const void *w, const void *w2, int Cin, int Cout, int Cin2, void *packed, void *stream);
int og_conv_band_bf16(const void *x, const void *w_packed, const float *bias, const void *skip, const void *x2, void *out, int N,
                      int Hin, int Win, int Cin, int Cout, int stride, int relu, int H2, int W2, int Cin2, int stride2, void *stream);
int og_conv_band_f16(const void *x, const void *w_packed, const float *bias, const void *skip, const void *x2, void *out, int N,
                     int Hin, int Win, int Cin, int Cout, int stride, int relu, int H2, int W2, int Cin2, int stride2, void *stream);
/* ---- the same 3x3 stride-1 convolution for the LARGE levels (160x160 / 80x80 / 40x40 at 640x640 input), on weights tiled
 * once in advance: csrc/conv3x3_tiled.inc -- halo-tiled direct convolution, two 4-wave workgroups per CU, 32-channel K steps,
 * every weight stage one contiguous 8 KiB LDS image.  Same arithmetic and epilogue as og_conv3x3_bf16 (fp32 accumulation, the
 * residual initialises the accumulators, one rounding).
 *   og_conv3x3_tiled_supported: 0 = shape not served (needs Cout % 128 == 0, Cin % 64 == 0, and H, W multiples of 16, or
 *     W == 40 / W == 20 with H % 4 == 0), otherwise the tile kind (16 x 16, 40 x 4, 20 x 4 pixels).
 *   og_conv3x3_tiled_workspace_bytes: 0 for most shapes (workspace may then be NULL).  A level whose output tiles alone would
 *     not fill the chip (40 x 40 and 20 x 20 at batch 8) is also split along K over 2-3 workgroups per tile, which meet through
 *     fp32 slabs + arrival tickets in the workspace: 256-byte aligned, ZERO-INITIALISED once by the caller, reusable by any
 *     later call on the same stream.
 *   og_conv3x3_pack_w16: w (Cout,3,3,Cin) 16-bit (bf16 or fp16 alike) -> packed, the same number of bytes, laid out
 *     [Cout/128][Cin/32][9 taps][128 rows x 64 B] with the k order / slot swizzle the kernel's fragment reads expect;
 *     order 0 = taps in their own order (for og_conv3x3_tiled_*), order 1 = the order the stride-2 kernel consumes them
 *     (0 2 6 8 | 3 5 | 1 7 | 4, for og_conv3x3s2_tiled_*); order 2 / 3 = a 1x1 weight (Cout, Cin) in cout tiles of 128 / 64 rows
 *     with k in its natural order (for og_conv1x1_tiled_* / og_conv1x1_heads_*).
 *   og_conv3x3_tiled_bf16: x (N,H,W,Cin), skip / out (N,H,W,Cout), bias fp32[Cout]; replaces convolution.forward
 *     models/hourglass_104.py:26-30 / residual.forward :70-79 (BN folded) like og_conv3x3_bf16. */
/* Optional hint for the NEXT convolution launch issued by this host thread -- og_conv3x3_tiled_* / og_conv3x3_tiled_up2_* /
 * og_conv3x3s2_tiled_* / og_conv_band_* / og_conv2d_* / og_conv2d_proj_* / og_conv3x3_* (every launch of the five takes and clears it;
 * the 1x1 kernels og_conv1x1_* and the stem do not look at it): [w_next, w_next + bytes) = the packed weights of the layer that will run AFTER that launch.  The launch's
 * workgroups touch those lines at entry (values unused), so that the next layer of a dependent chain -- the 20x20 / 10x10 / 5x5 levels,
 * whose layers are bound by the latency of first-touch weight reads -- finds its weights in the memory-side cache instead of HBM.
 * Purely a performance hint: results do not depend on it; NULL / 0 clears it. */
void og_conv_next_weights_hint(const void *w_next, size_t bytes);
int og_conv3x3_tiled_supported(int N, int H, int W, int Cin, int Cout);
int og_conv3x3_pack_w16(const void *w, int Cin, int Cout, int order, void *packed, void *stream);
size_t og_conv3x3_tiled_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int og_conv3x3_tiled_bf16(const void *x, const void *w_packed, const float *bias, const void *skip, void *out, int N, int H,
                          int W, int Cin, int Cout, int relu, void *workspace, size_t workspace_bytes, void *stream);
/* The last convolution below an hourglass merge and the merge itself in one launch (kp_module.forward, models/hourglass_104.py:
 * 170-176: up2 = upsample(low3); return up1 + up2): up (N,2H,2W,Cout), holding up1, += nearest_x2(act(conv3x3(x) + bias + skip)),
 * the convolution's result rounded to 16 bits first -- og_conv3x3_tiled_bf16 followed by og_upsample2_add_bf16, bit for bit; the
 * (N,H,W,Cout) tensor in between is never written.  Shapes and workspace as og_conv3x3_tiled_bf16. */
int og_conv3x3_tiled_up2_bf16(const void *x, const void *w_packed, const float *bias, const void *skip, void *up, int N, int H,
                              int W, int Cin, int Cout, int relu, void *workspace, size_t workspace_bytes, void *stream);
/* Stride 2 (residual.conv1 of the down-sampling residuals, models/hourglass_104.py:54-57 with stride 2, and the second `pre`
 * layer :214-217) on the same kernel structure: x (N,Hin,Win,Cin) -> out (N,Hin/2,Win/2,Cout), pad 1; weights packed with
 * order 1; the four input-parity phases of a tile are gathered straight from the NHWC input by the LDS-DMA.
 * og_conv3x3s2_tiled_supported(N, Hin, Win, Cin, Cout): 0 = not served; needs Hin, Win even, Cout % 128 == 0, Cin % 64 == 0 and
 * an output of 8k x 16k pixels (1: 16 x 8 output tiles) or 40 wide with an even height (2: 40 x 2 tiles, the 80 -> 40 level). */
int og_conv3x3s2_tiled_supported(int N, int Hin, int Win, int Cin, int Cout);
int og_conv3x3s2_tiled_bf16(const void *x, const void *w_packed, const float *bias, const void *skip, void *out, int N, int Hin,
                            int Win, int Cin, int Cout, int relu, void *stream);
/* ---- pointwise (1x1) convolutions of the large levels (csrc/conv3x3_tiled.inc, conv1x1_tiled_kernel) ----
 * og_conv1x1_tiled_bf16: out (N,H,W,Cout) = act(W [x1 | x2] + bias (+ skip)): output pixel (y, x) reads x1 (N,H1,W1,C1) at
 *   (y * stride1, x * stride1); an optional second input x2 of the SAME shape and stride is concatenated along K -- the
 *   inters_ / cnvs_ junction relu(bn(conv(inter)) + bn(conv(feat))) of models/hourglass_104.py:239-250, :291-292 in one launch --
 *   and stride 2 with one input is the 1x1 projection `skip` of the down-sampling residuals (:63-67).  w_packed: the
 *   (Cout, C1 + C2) weight through og_conv3x3_pack_w16(order 2); C1, C2 multiples of 64, Cout of 128; bias / skip may be null.
 * og_conv1x1_heads_bf16: all heads of the decoded stack as one 1x1 convolution (models/heads.py:48-70, :116-142, no
 *   activation): x (N,H,W,C) -> up to four dense fp32 NCHW tensors outs[i] (N,head_channels[i],H,W), written straight from the
 *   fp32 accumulators (bias added in fp32, no 16-bit rounding).  w_packed: the concatenated head weights padded to Cout (a
 *   multiple of 64) through og_conv3x3_pack_w16(order 3); bias fp32[Cout]. */
int og_conv1x1_tiled_bf16(const void *x1, int C1, int H1, int W1, int stride1, const void *x2, int C2, int H2, int W2, int stride2,
                          const void *w_packed, const float *bias, const void *skip, void *out, int N, int H, int W, int Cout,
                          int relu, void *stream);
int og_conv1x1_heads_bf16(const void *x, int C, const void *w_packed, const float *bias, int N, int H, int W, int Cout, int n_heads,
                          const int *head_channels, float *const *outs, void *stream);
/* Debug aid: later og_conv3x3_bf16 launches write [workgroup][8] u64 s_memrealtime (100 MHz) marks into `buf`
 * (device memory, 64 B per workgroup); NULL switches it off. */
/* ---- the same entry points for fp16 activations / weights (the reference evaluates in fp16 through apex O2,
 * evaluate.py:92,198-201): v_mfma_f32_16x16x32_f16 instead of ..._bf16, fp32 accumulation, identical layouts, arguments,
 * workspaces (og_conv*_workspace_bytes) and error behaviour; models.InferenceEngine(dtype=torch.float16). ---- */
int og_bias_act_f16(void *x, const float *bias, const void *skip, long pixels, int channels, int relu, void *stream);
int og_upsample2_add_f16(void *up, const void *low, long n, int H, int W, int channels, void *stream);
int og_nchw_f32_to_nhwc_f16(const float *src, void *dst, long N, int C, int H, int W, void *stream);
int og_nhwc_f16_to_nchw_f32(const void *src, int src_channels, int first_channel, int channels, const float *bias,
                            float *dst, long N, int H, int W, void *stream);
int og_stem7x7_f16(const float *images, const void *w_packed, const float *bias, void *out, int N, int H, int W, int relu,
                   void *stream);
int og_conv3x3_f16(const void *x, const void *w, const float *bias, const void *skip, void *out, int N, int H, int W,
                   int Cin, int Cout, int relu, void *workspace, size_t workspace_bytes, void *stream);
int og_conv3x3_tiled_f16(const void *x, const void *w_packed, const float *bias, const void *skip, void *out, int N, int H,
                         int W, int Cin, int Cout, int relu, void *workspace, size_t workspace_bytes, void *stream);
int og_conv3x3_tiled_up2_f16(const void *x, const void *w_packed, const float *bias, const void *skip, void *up, int N, int H,
                         int W, int Cin, int Cout, int relu, void *workspace, size_t workspace_bytes, void *stream);
int og_conv3x3s2_tiled_f16(const void *x, const void *w_packed, const float *bias, const void *skip, void *out, int N, int Hin,
                           int Win, int Cin, int Cout, int relu, void *stream);
int og_conv1x1_tiled_f16(const void *x1, int C1, int H1, int W1, int stride1, const void *x2, int C2, int H2, int W2, int stride2,
                         const void *w_packed, const float *bias, const void *skip, void *out, int N, int H, int W, int Cout,
                         int relu, void *stream);
int og_conv1x1_heads_f16(const void *x, int C, const void *w_packed, const float *bias, int N, int H, int W, int Cout, int n_heads,
                         const int *head_channels, float *const *outs, void *stream);
int og_conv2d_f16(const void *x, const void *w, const float *bias, const void *skip, void *out, int N, int Hin, int Win,
                  int Cin, int Cout, int ksize, int stride, int relu, void *workspace, size_t workspace_bytes, void *stream);
int og_conv2d_proj_f16(const void *x, const void *w_cat, const float *bias, const void *x2, void *out, int N, int Hin,
                       int Win, int Cin, int Cout, int ksize, int stride, int H2, int W2, int Cin2, int stride2, int relu,
                       void *workspace, size_t workspace_bytes, void *stream);

/* ---- training losses, value + gradient in one pass ----
 * og_focal_l2_loss_f32: models/losses.py:31-58 + HeatMapsLoss :174-176.  pred/gt (N,C,hw) fp32, mask_miss
 *   (N,hw) bytes (0 = unlabelled); *sum += sum of 0.5 (s-s*)^2 |1-st|^gamma over labelled elements with finite
 *   gt (caller zeroes *sum); grad (N,C,hw) = d sum / d pred.
 * og_offset_l1_loss_f32: offset_instance_l1_loss :87-92 + OffsetMapsLoss :237-242.  e = |pred/ps - gt/ps| kept
 *   if e >= margin (sqrt(e) if sqrt_re); sum_count[0] += sum, sum_count[1] += count; grad = d sum / d pred
 *   (the caller scales by 1 / (1 + count)). */
int og_focal_l2_loss_f32(const float *pred, const float *gt, const unsigned char *mask_miss, int N, int C, long hw,
                         float tau, float gamma, float *sum, float *grad, void *stream);
int og_offset_l1_loss_f32(const float *pred, const float *gt, const float *gt_ps, const unsigned char *mask_miss, int N,
                          int C, long hw, float margin, int sqrt_re, float *sum_count, float *grad, void *stream);

/* ---- ground-truth encoder ----
 * joints (N,P,n_kp,4) fp32 rows [x, y, v, scale] in input-image pixels (transforms/annotations.py:46-50), P = padded
 * person count, n_persons int32[N] (NULL: all P rows are used; rows with v <= 0 never contribute).
 * og_encode_heatmaps_f32: HeatMapGenerator.create_heatmaps encoder/heatmap.py:125-197 -> hm (N,n_kp,h,w) and, if not
 *   NULL, bg (N,1,h,w) = 1 - max over channels (:78); h = in_h/stride, w = in_w/stride.
 * og_encode_offsets_f32: OffsetMapGenerator.create_offsetmaps encoder/offset.py:98-197 -> off (N,2L,h,w) (inf where no
 *   limb is defined), pscale (N,2L,h,w) (1 there), and, if not NULL, scale (N,n_kp,h,w) (nan there); jf/jt int32[L],
 *   sigmas fp32[n_kp] (config COCO_PERSON_SIGMAS) are device arrays.
 * Offsets/scales bit-exact vs the reference; heatmaps to ~2e-7 (device exp). */
int og_encode_heatmaps_f32(const float *joints, const int32_t *n_persons, int N, int P, int n_kp, int in_w, int in_h,
                           int stride, int sigma, float clip_thre, float *hm, float *bg, void *stream);
/* og_encode_jitter_f32: HeatMapGenerator.create_jitter_offset encoder/heatmap.py:199-255 -> jit (N,2,h,w): vector from
 * the cell centre to the nearest annotated keypoint inside a fill_size window, inf elsewhere (bit-exact). */
int og_encode_jitter_f32(const float *joints, const int32_t *n_persons, int N, int P, int n_kp, int in_w, int in_h,
                         int stride, int fill_size, float *jit, void *stream);
int og_encode_offsets_f32(const float *joints, const int32_t *n_persons, int N, int P, int n_kp, const int32_t *jf,
                          const int32_t *jt, int L, int in_w, int in_h, int stride, int fill_size, float min_jscale,
                          const float *sigmas, float *off, float *scale, float *pscale, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OG_DECODER_H */
