/*
 * sslam_oracle.h - CPU restatement of the semantic-slam per-frame extraction + matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library, and only as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py against golden vectors
 * produced by running the reference's own Python modules (tests/golden/make_golden.py, run where
 * /root/reference exists).
 *
 * Canonical fp32 evaluation order.  The reference computes in fp32 through torch / numpy BLAS whose summation
 * order is unspecified.  This restatement fixes ONE order per stage (documented per function); the HIP kernels
 * implement exactly the same order, so GPU == oracle bit-for-bit (indices *and* floats) while
 * oracle == torch within 1e-4 on values and exactly on indices (fixtures are checked to be free of near-ties).
 *   - every contraction (conv, linear, similarity) is one fused-multiply-add chain in increasing k,
 *     starting from the bias (or 0):  acc = fmaf(a[k], b[k], acc).  This is what gfx950's
 *     v_mfma_f32_32x32x2_f32 computes.
 *   - reductions (BatchNorm / LayerNorm statistics, 1x1 conv, L2 norm) use the fixed trees described below.
 *   - exp() is the polynomial ora_expf() below, built from fmaf / rintf / integer ops only.
 */
#ifndef SSLAM_ORACLE_H
#define SSLAM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_C 384 /* ViT-S/16 embed dim; dino_backbone.py:50 */

/* A2 - dino_backbone.py:91-106.  tokens (n_frames, tokens_per_frame, 384); the first n_prefix tokens of each
 * frame are dropped; BatchNorm1d statistics are taken over `group` consecutive frames (group=1 reproduces the
 * reference's B=1 calls, group=B its batched call).  train!=0: batch mean / biased variance (written to
 * out_mean/out_var, (n_frames/group, 384)); train==0: run_mean/run_var.  out_feat (n_frames, cells, 384). */
void ora_bn_tokens(const float *tokens, int n_frames, int tokens_per_frame, int n_prefix, int group,
                   const float *gamma, const float *beta, const float *run_mean, const float *run_var,
                   int train, float eps, float *out_feat, float *out_mean, float *out_var);

/* A3 - keypoint_selector.py:45-67.  feat (n_frames, G, G, 384) NHWC; w1 (hs,384,3,3), b1 (hs), w2 (hs), b2 (1).
 * sal (n_frames, G, G).  hs must be a multiple of 64. */
void ora_selector_saliency(const float *feat, int n_frames, int G, const float *w1, const float *b1,
                           const float *w2, const float *b2, int hs, float *sal);

/* A4 - keypoint_selector.py:209-226 (one frame). */
void ora_nms(const float *sal, int G, int radius, float *out);

/* torch.quantile(v, q) with linear interpolation, as called at keypoint_selector.py:106,140 */
float ora_quantile(const float *v, int n, double q);

/* A5 - keypoint_selector.py:69-207.  sal (n_frames, G, G) -> kp_xy (n_frames, K, 2) fp32 (x, y) patch units,
 * scores (n_frames, K), idx (n_frames, K) flat cell index y*G+x, status (n_frames): 0 ok, 1 = the reference's
 * torch.topk would raise (k larger than the number of cells).  top-k order: value descending, flat
 * index ascending. */
void ora_select_keypoints(const float *sal, int n_frames, int G, int K, int radius, double pct, float *kp_xy,
                          float *scores, int32_t *idx, int32_t *status);

/* A6 - dino_backbone.py:114-152 (grid_sample bilinear, align_corners=True, zero padding).
 * feat (n_frames, G, G, 384), kp_xy (n_frames, K, 2) -> out (n_frames, K, 384). */
void ora_gather(const float *feat, int n_frames, int G, const float *kp_xy, int K, float *out);

/* A7 - descriptor_refiner.py:58-126.  x (rows, 384) -> desc (rows, d_out).  w: 4 + 8*n_blocks pointers in
 * state_dict order: input_proj.{weight,bias}, per block {norm1.w, norm1.b, fc1.w, fc1.b, norm2.w, norm2.b,
 * fc2.w, fc2.b}, output_proj.{weight,bias}.  hidden = 384. */
void ora_refine(const float *x, int rows, const float *const *w, int n_blocks, int d_out, float *desc);

/* A8 - dino_backbone.py:154-178 */
void ora_patch_to_pixel(const float *kp, int n, float *out);
void ora_pixel_to_patch(const float *px, int n, float *out);

/* similarity matrix row/column arg-max (first maximal index), the core of M1..M5.
 * d1 (n, d), d2 (m, d); nn12 (n), s12 (n): row arg-max / max; nn21 (m), s21 (m): column arg-max / max. */
void ora_sim_argmax(const float *d1, int n, const float *d2, int m, int d, int32_t *nn12, float *s12,
                    int32_t *nn21, float *s21);

/* full similarity matrix S (n, m) in the canonical chain order (for the M2 / M4 ratio-test checks) */
void ora_sim_matrix(const float *d1, int n, const float *d2, int m, int d, float *S);

/* M1 - visualize_matches_sequence.py:106-197.  intensity1/2 may be NULL.  Returns the match count; matches
 * (cap, 2) int64 ascending in idx1, quality (cap). */
int ora_match_with_quality(const float *d1, int n, const float *d2, int m, int d, const float *sc1,
                           const float *sc2, double saliency_weight, double min_saliency, double min_sim,
                           const float *int1, const float *int2, double min_intensity, int64_t *matches,
                           float *quality);

/* A0 - visualize_matches_sequence.py:59-67,72: Pillow antialiased resize (filter: 0 bilinear, 1 bicubic) of a
 * uint8 HWC RGB image to (size,size), returned as uint8 HWC in resized (may be NULL) and, if chw != NULL, as
 * normalised fp32 CHW ((x/255 - mean)/std, ImageNet constants). */
void ora_resize_rgb(const uint8_t *img, int h, int w, int size, int filter, uint8_t *resized, float *chw);

/* A9 - visualize_matches_sequence.py:88-95: bicubic resize -> "L" -> /255 -> lookup at round(pixel coords). */
void ora_intensity(const uint8_t *img, int h, int w, int size, const float *kp_pixel, int K, float *out);
void ora_gray_resized(const uint8_t *img, int h, int w, int size, uint8_t *gray);

/* the canonical exp / sigmoid, exported so tests can compare them with the device versions */
float ora_expf(float x);
float ora_sigmoid(float x);

int ora_num_threads(void);
void ora_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
