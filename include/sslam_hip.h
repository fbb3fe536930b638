/*
 * sslam_hip.h - C ABI of libsslam_hip.so: the MI355X (gfx950) implementation of the semantic-slam per-frame
 * feature-extraction + descriptor-matching hot path.
 *
 * The reference (Siverteh/semantic-slam-master) is pure Python: there is no FFI / plugin layer to mirror, so this
 * header defines the boundary a maintainer binds with ctypes (INTEGRATION.md shows the stub).  One entry per fused
 * stage; each cites the reference code it replaces (paths relative to the reference repo).
 *
 * Conventions (all entries):
 *   - every pointer is a CALLER-OWNED DEVICE pointer unless the name ends in _host; the library never copies to
 *     the host and NEVER allocates or frees device memory.  The two entries whose fastest form needs scratch take it
 *     from the caller (sslam_selector_saliency_ws, sslam_sim_argmax_ws; sizes from sslam_workspace_bytes or the
 *     per-entry *_workspace_bytes); their forms without a workspace argument run a scratch-free launch shape with
 *     the same bits;
 *   - `stream` is a hipStream_t passed as void* (PyTorch: torch.cuda.current_stream().cuda_stream); calls only
 *     enqueue work - no synchronisation, no host read-back;
 *   - return value: SSLAM_OK or a negative SSLAM_E_* code; launch failures are reported via hipGetLastError;
 *   - stateless and thread-safe given distinct streams / workspaces.  The SSLAM_* environment variables named below
 *     are TEST-ONLY knobs (A/B timing, forcing a launch form in the parity tests): the environment is read once,
 *     when the library is loaded, never per call; every form they select produces the same bits;
 *   - all floating point is IEEE fp32 evaluated in the canonical order documented in oracle/sslam_oracle.h
 *     (contractions = one fused-multiply-add chain in increasing k on v_mfma_f32_32x32x2_f32), so results are
 *     bit-identical to the CPU oracle.
 */
#ifndef SSLAM_HIP_H
#define SSLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSLAM_OK 0
#define SSLAM_E_INVALID (-1)     /* null pointer / non-positive size / misaligned pointer */
#define SSLAM_E_UNSUPPORTED (-2) /* shape outside what the kernels are built for */
#define SSLAM_E_LAUNCH (-3)      /* hipGetLastError() != hipSuccess after the launch */

#define SSLAM_C 384   /* backbone embed dim (ViT-S/16), dino_backbone.py:50 */
#define SSLAM_HID 384 /* refiner hidden dim, configs/train_config.yaml:13 */
#define SSLAM_D 128   /* descriptor dim, configs/train_config.yaml:12 */

/* library version (major*10000 + minor*100 + patch) and the gfx target it was compiled for */
int sslam_version(void);
const char *sslam_arch(void);
/* number of kernel launches enqueued by this process so far (lets tests prove the HIP path ran) */
long long sslam_launch_count(void);

/* One caller-owned device scratch buffer (bytes, multiple of 256; 0 if none is needed) that serves every *_ws entry of
 * a pipeline step enqueued on one stream - the stages run in stream order and share it: n_frames frames of a G x G grid
 * through sslam_selector_saliency_ws, n_pairs pairs of K keypoints through sslam_sim_argmax_ws (n_pairs may be 0). */
long long sslam_workspace_bytes(int n_frames, int G, int K, int n_pairs);
/* per-entry needs (what sslam_workspace_bytes takes the maximum of) */
long long sslam_selector_saliency_workspace_bytes(int n_frames, int G);
long long sslam_sim_argmax_workspace_bytes(int n2, int n_pairs);

/* TEST-ONLY: override one of the load-time knobs (name = its environment variable, e.g. "SSLAM_CONV_TAIL"); unset != 0
 * restores the value read from the environment when the library was loaded (else the built-in default).  Not thread-safe against running calls; product code never calls it.  Knobs:
 * SSLAM_M1_VARIANT, SSLAM_CONV_VARIANT, SSLAM_CONV_LATENCY_ROWS, SSLAM_CONV_LAT2_ROWS, SSLAM_CONV_NO_HALO, SSLAM_CONV_TAIL,
 * SSLAM_CONVBF_NO_HALO, SSLAM_CONVBF_TAIL, SSLAM_CONVBF_VARIANT, SSLAM_VIT_NO_FUSED_MLP, SSLAM_BN_FORM,
 * SSLAM_RT_STOP (csrc/common.h says what each selects). */
int sslam_test_set_knob(const char *name, long long value, int unset);

/* ---- weight packing (host side, plain C++; run once per checkpoint) ------------------------------------------
 * The kernels read weights in an LDS-image order: K split into chunks, each chunk stored [n][k] with k permuted
 * inside groups of 8 as (0,2,4,6,1,3,5,7) so that one ds_read_b128 feeds four consecutive 32x32x2 MFMA steps.
 *
 * sslam_pack_conv3x3_host: w (hs,384,3,3) [keypoint_selector.py:31, state_dict key conv.0.weight]
 *                          -> out (9*384*hs floats).
 * sslam_pack_linear_host:  w (n_out, k_in) [nn.Linear weight, descriptor_refiner.py:35,43,103,105]
 *                          -> out (n_out*k_in floats) as [k/8][n][8]; k_in % 8 == 0. */
int sslam_pack_conv3x3_host(const float *w_host, int hs, float *out_host);
int sslam_pack_linear_host(const float *w_host, int n_out, int k_in, float *out_host);

/* ---- A0: preprocessing.  Replaces transforms.Compose([Resize, ToTensor, Normalize]) applied at
 * visualize_matches_sequence.py:59-67,72 (Pillow antialiased bilinear resize, /255, ImageNet mean/std).
 * sslam_resample_table_host builds Pillow's fixed-point coefficient table for one axis (filter 0 = bilinear,
 * 1 = bicubic): bounds (out_size*2 int32: first input index, tap count), coefs (out_size*ksize int32);
 * returns ksize (<= SSLAM_MAX_TAPS) or a negative error.
 * img (n, h, w, 3) uint8 -> out (n, 3, size, size) fp32. */
#define SSLAM_MAX_TAPS 32
int sslam_resample_table_host(int in_size, int out_size, int filter, int32_t *bounds_host, int32_t *coefs_host,
                              int coefs_capacity);
int sslam_preprocess_u8(const uint8_t *img, int n, int h, int w, int size, const int32_t *bounds_h,
                        const int32_t *coefs_h, int ksize_h, const int32_t *bounds_v, const int32_t *coefs_v,
                        int ksize_v, float *out_chw, void *stream);
/* The same arithmetic, written as the patch-embedding operand of sslam_vit_forward_patches instead of the planar image:
 * out (n, (size/16)^2, 768) bf16, row = patch (py, px), k = c*256 + ky*16 + kx - the normalised fp32 value of
 * sslam_preprocess_u8 rounded to bf16 (round-to-nearest-even) once; size % 16 == 0.  SSLAM_E_UNSUPPORTED for resampling
 * ratios the tiled kernel does not cover (more than 7 horizontal taps, or an image base that is not dword-aligned): take
 * sslam_preprocess_u8 + sslam_vit_forward then. */
int sslam_preprocess_u8_patches(const uint8_t *img, int n, int h, int w, int size, const int32_t *bounds_h,
                                const int32_t *coefs_h, int ksize_h, const int32_t *bounds_v, const int32_t *coefs_v,
                                int ksize_v, void *out_patches_bf16, void *stream);

/* ---- A2: token drop + BatchNorm1d over tokens.  Replaces DinoBackbone.forward after the ViT call,
 * dino_backbone.py:91-106.  tokens (n_frames, tokens_per_frame, 384); statistics over `group` consecutive frames
 * (group = 1 reproduces per-frame B=1 calls).  train != 0: batch statistics, also written to
 * out_mean / out_var (n_frames/group, 384; biased variance); train == 0: run_mean / run_var are used.
 * out_feat (n_frames, cells, 384) with cells = tokens_per_frame - n_prefix. */
int sslam_bn_tokens(const float *tokens, int n_frames, int tokens_per_frame, int n_prefix, int group,
                    const float *gamma, const float *beta, const float *run_mean, const float *run_var, int train,
                    float eps, float *out_feat, float *out_mean, float *out_var, void *stream);

/* ---- A3: saliency CNN.  Replaces KeypointSelector.forward, keypoint_selector.py:45-67
 * (conv3x3 384->hs + ReLU + conv1x1 hs->1 + sigmoid).  feat (n_frames, G, G, 384) NHWC; w1_packed from
 * sslam_pack_conv3x3_host; b1 (hs), w2 (hs), b2 (1); hs in {128, 256}.  sal (n_frames, G, G).
 * Four launch shapes, chosen by size, all bit-identical (same fma chain per output; csrc/selector.hip): the halo
 * and the stage form of the 128-row throughput kernel, and two latency forms for few frames.  The second latency form
 * (two workgroups per 32-cell tile; up to six 28x28 frames) needs 16 bytes of scratch per cell: the _ws entry takes it from
 * the caller (workspace_bytes >= sslam_selector_saliency_workspace_bytes; workspace may be NULL: then, and in the entry
 * without a workspace, the 8-wave latency form runs instead). */
int sslam_selector_saliency(const float *feat, int n_frames, int G, const float *w1_packed, const float *b1,
                            const float *w2, const float *b2, int hs, float *sal, void *stream);
int sslam_selector_saliency_ws(const float *feat, int n_frames, int G, const float *w1_packed, const float *b1,
                               const float *w2, const float *b2, int hs, float *sal, void *workspace,
                               long long workspace_bytes, void *stream);

/* ---- A3, bf16 THROUGHPUT mode (BASELINE.json configs[1] "bf16 conv stack").  Same layer as
 * sslam_selector_saliency with bf16 operands (round-to-nearest-even), fp32 accumulation on v_mfma_f32_32x32x16_bf16 and
 * the fp32 epilogue.  NOT index-exact against the fp32 reference: callers report the agreement rate next to it.
 * sslam_f32_to_bf16: the bf16 copy of the feature map (n % 8 == 0, 16-byte aligned pointers).
 * sslam_pack_conv3x3_bf16_host: w (hs,384,3,3) fp32 -> 9*384*hs bf16 in MFMA-fragment order. */
int sslam_f32_to_bf16(const float *in, void *out_bf16, long long n, void *stream);
/* sslam_bn_tokens that also writes the bf16 copy of out_feat (same shape) in the same pass */
int sslam_bn_tokens_bf16copy(const float *tokens, int n_frames, int tokens_per_frame, int n_prefix, int group,
                             const float *gamma, const float *beta, const float *run_mean, const float *run_var, int train,
                             float eps, float *out_feat, void *out_feat_bf16, float *out_mean, float *out_var, void *stream);
int sslam_pack_conv3x3_bf16_host(const float *w_host, int hs, void *out_bf16_host);
int sslam_selector_saliency_bf16(const void *feat_bf16, int n_frames, int G, const void *w1_packed_bf16, const float *b1,
                                 const float *w2, const float *b2, int hs, float *sal, void *stream);

/* ---- A4 + A5 (+ A8): NMS + percentile threshold + branchy top-k.  Replaces KeypointSelector.select_keypoints /
 * _apply_nms, keypoint_selector.py:69-226, and DinoBackbone.patch_to_pixel, dino_backbone.py:154-165.
 * sal (n_frames, G, G) -> kp_xy (n_frames, K, 2) fp32 (x, y) patch units; scores (n_frames, K);
 * idx (n_frames, K) int32 flat cell index (may be NULL); kp_pixel (n_frames, K, 2) = kp*16+8 (may be NULL);
 * status (n_frames) int32: 0 ok, 1 = the reference's torch.topk would raise (K exceeds the cells).
 * No host synchronisation: the whole data-dependent control flow runs on the device, one workgroup per frame.
 * G*G <= 4096, K <= 4096, 0 <= nms_radius <= 8. */
int sslam_select_keypoints(const float *sal, int n_frames, int G, int K, int nms_radius, double min_score_percentile,
                           float *kp_xy, float *scores, int32_t *idx, float *kp_pixel, int32_t *status,
                           void *stream);

/* ---- A6: bilinear feature gather.  Replaces DinoBackbone.extract_at_keypoints, dino_backbone.py:114-152
 * (grid_sample bilinear, align_corners=True, zero padding).  feat (n_frames, G, G, 384), kp_xy (n_frames, K, 2)
 * -> out (n_frames, K, 384). */
int sslam_gather(const float *feat, int n_frames, int G, const float *kp_xy, int K, float *out, void *stream);

/* ---- A7: descriptor MLP.  Replaces DescriptorRefiner.forward / ResidualBlock.forward,
 * descriptor_refiner.py:58-126.  Refiner weights are passed as ONE packed device buffer laid out by
 * sslam_refiner_pack_host (offsets in floats are returned by sslam_refiner_layout).
 * x (rows, 384) -> desc (rows, 128). */
typedef struct {
    int n_blocks;          /* residual blocks (num_layers - 2; 2 in the shipped config) */
    long long total;       /* floats in the packed buffer */
    long long in_w, in_b;  /* packed input_proj.weight, input_proj.bias */
    long long blk[8][8];   /* per block: norm1.w, norm1.b, fc1.w(packed), fc1.b, norm2.w, norm2.b, fc2.w(packed), fc2.b */
    long long out_w, out_b;
} sslam_refiner_layout_t;
int sslam_refiner_layout(int n_blocks, sslam_refiner_layout_t *layout_host);
/* w_host: 4 + 8*n_blocks host pointers in state_dict order (see oracle/sslam_oracle.h ora_refine) */
int sslam_refiner_pack_host(const float *const *w_host, int n_blocks, float *out_host);
int sslam_refine(const float *x, long long rows, const float *packed, int n_blocks, float *desc, void *stream);

/* ---- A6 + A7 fused: gather straight into the MLP's LDS tile (the pipeline's fast path). */
int sslam_gather_refine(const float *feat, int n_frames, int G, const float *kp_xy, int K, const float *packed,
                        int n_blocks, float *desc, void *stream);

/* ---- A6 + A7, bf16 THROUGHPUT mode (BASELINE.json configs[1]): the same gather + MLP with bf16
 * GEMM operands (v_mfma_f32_32x32x16_bf16), fp32 accumulation / residual / LayerNorm statistics / L2 normalisation;
 * LayerNorm is folded into the following GEMM (W*gamma, column sums, b + W beta are precomputed by the packer).
 * NOT bit-exact against the fp32 reference.  packed_bf16: sslam_refiner_bf16_bytes(n_blocks) bytes written by
 * sslam_refiner_pack_bf16_host from the same pointer list as sslam_refiner_pack_host. */
long long sslam_refiner_bf16_bytes(int n_blocks);
int sslam_refiner_pack_bf16_host(const float *const *w_host, int n_blocks, void *out_host);
int sslam_refine_bf16(const float *x, long long rows, const void *packed_bf16, int n_blocks, float *desc, void *stream);
int sslam_gather_refine_bf16(const float *feat, int n_frames, int G, const float *kp_xy, int K, const void *packed_bf16,
                             int n_blocks, float *desc, void *stream);

/* ---- A9: per-keypoint intensity.  Replaces visualize_matches_sequence.py:87-95 (Pillow BICUBIC resize to
 * (size,size) -> "L" -> /255 -> gray[round(y), round(x)]); only the pixels that are looked up are resampled.
 * img (n, h, w, 3) uint8; kp_pixel (n, K, 2); tables from sslam_resample_table_host(filter = 1). */
int sslam_keypoint_intensity(const uint8_t *img, int n, int h, int w, int size, const int32_t *bounds_h,
                             const int32_t *coefs_h, int ksize_h, const int32_t *bounds_v, const int32_t *coefs_v,
                             int ksize_v, const float *kp_pixel, int K, float *out, void *stream);

/* ---- M1..M5 core: cosine-similarity GEMM + row / column arg-max.  Replaces torch.mm + argmax(dim=1) +
 * argmax(dim=0) at visualize_matches_sequence.py:144-148 (and visualize_matches.py:105-109, train.py:423-425,
 * test/test_descriptor_quality.py:116-123, test/test_tracking.py:159-160).
 * For each of n_pairs pairs p: d1 = desc1 + p*stride1 (n1 x 128), d2 = desc2 + p*stride2 (n2 x 128)
 * (strides in floats; stride 0 broadcasts).  nn12/s12 (n_pairs, n1): first arg-max / max over d2 for each row of
 * d1; nn21/s21 (n_pairs, n2): the column direction.  second12 (n_pairs, n1): the largest similarity of the row with
 * the winner removed (-inf if n2 == 1) - what the ratio tests of visualize_matches.py:116-121 and
 * test/test_descriptor_quality.py:129-131 need, without sorting rows.  s12 / s21 / second12 may be NULL.
 * sslam_sim_argmax_ws, batched calls (n_pairs >= 16) with workspace_bytes >= sslam_sim_argmax_workspace_bytes(n2, n_pairs)
 * = n_pairs*n2*8 bytes of caller-owned scratch: the similarity matrix is evaluated once and the column direction reduced
 * with 64-bit (value, ~index) keys and atomic max - deterministic.  Smaller calls, a NULL / short workspace, and the entry
 * without a workspace evaluate it once per direction and need no scratch.  Same bits either way.
 * Precondition: finite descriptors (the refiner's L2-normalised rows are).  With NaN / Inf in the inputs the VALUES and the
 * choice among candidates are unspecified (torch.argmax would return the first NaN), but every index written stays inside
 * [0, n2) resp. [0, n1), so sslam_match_finalize and the sibling matchers never index out of range. */
int sslam_sim_argmax(const float *desc1, long long stride1, int n1, const float *desc2, long long stride2, int n2,
                     int n_pairs, int32_t *nn12, float *s12, int32_t *nn21, float *s21, float *second12, void *stream);
int sslam_sim_argmax_ws(const float *desc1, long long stride1, int n1, const float *desc2, long long stride2, int n2,
                        int n_pairs, int32_t *nn12, float *s12, int32_t *nn21, float *s21, float *second12,
                        void *workspace, long long workspace_bytes, void *stream);

/* ---- M1: mutual check + thresholds + quality + ordered compaction.  Replaces
 * SequenceMatcher.match_with_quality, visualize_matches_sequence.py:149-197, given the arg-max arrays above.
 * scores / intensities are (n_pairs-strided) per-frame arrays like the descriptors; intensity1/2 may be NULL.
 * Outputs per pair: matches (cap = n1 rows of 2 int64, ascending idx1), quality (n1 fp32), count (1 int32). */
int sslam_match_finalize(const int32_t *nn12, const float *s12, const int32_t *nn21, int n1, int n2, int n_pairs,
                         const float *scores1, long long sstride1, const float *scores2, long long sstride2,
                         const float *intensity1, const float *intensity2, float w_desc, float w_sal,
                         float min_saliency, float min_sim, float min_intensity, int64_t *matches, float *quality,
                         int32_t *count, void *stream);

/* ---- A1: DINOv3 ViT-S/16 forward.  Replaces the third-party call
 * `self.dino.forward_features(images)` at dino_backbone.py:85 (timm model "vit_small_patch16_dinov3"): 16x16 patch
 * embedding, [CLS] + 4 register tokens, 12 pre-LN blocks (6 heads x 64, q/v/proj bias, axial RoPE theta 100 on the
 * patch tokens, LayerScale, MLP 1536 GELU), final LayerNorm.  bf16 MFMA operands, fp32 accumulation / LayerNorm /
 * softmax / residual stream: tolerance-level parity with an fp32 evaluation (not bit-exact).
 * All pointers are DEVICE pointers; vectors fp32; matrices bf16, nn.Linear (n_out, k_in) re-ordered by
 * sslam_vit_pack_linear_host into the order the GEMM kernel streams them: one 1 KB MFMA fragment per (192-column tile,
 * k-step of 16, 32-column slice): element (n, k) -> [n/192][k/16][(n%192)/32][(k%16)/8][n%32][k%8].
 * wqkv = rows [q_proj; k_proj; v_proj] (1152, 384), bqkv likewise with ZEROS for the k rows (no key bias), the q rows of
 * both multiplied by log2(e)/sqrt(64) (softmax runs in the exp2 domain); wo / bo and wdown / bdown multiplied row-wise by
 * the block's LayerScale (ls1, ls2) - the caller folds these constants before packing (sslam_amd/vit_hip.py does);
 * patch_w = Conv2d weight reshaped (384, 768); prefix = [cls; reg0..3] (5, 384); rope_cos / rope_sin (G*G, 64) fp32.
 * images_chw (n, 3, size, size) fp32 (the output of sslam_preprocess_u8) -> tokens_out (n, 5 + (size/16)^2, 384). */
typedef struct {
    const float *ln1_g, *ln1_b;
    const void *wqkv;
    const float *bqkv;
    const void *wo;
    const float *bo, *ln2_g, *ln2_b;
    const void *wup;
    const float *bup;
    const void *wdown;
    const float *bdown;
    const void *wmlp;      /* optional (may be NULL): up_proj + down_proj as ONE stream for the fused MLP kernel
                              (sslam_vit_pack_mlp_host); used for batches of more than ~8 frames, wup / wdown otherwise */
} sslam_vit_layer_t;
typedef struct {
    const void *patch_w;
    const float *patch_b, *prefix;
    sslam_vit_layer_t layer[12];
    const float *norm_g, *norm_b, *rope_cos, *rope_sin;
} sslam_vit_weights_t;
/* host helper: fp32 nn.Linear weight (n_out, k_in), n_out % 192 == 0, k_in % 384 == 0 -> the packed bf16 image above */
int sslam_vit_pack_linear_host(const float *w, int n_out, int k_in, uint16_t *out);
/* host helper: up_proj (1536, 384) + down_proj (384, 1536) [rows times row_scale = LayerScale 2, or NULL] -> the 2 x 1536 x 384
 * bf16 stream of the fused MLP kernel (per 64-wide hidden chunk: 48 KB of up rows, then 48 KB of down columns) */
int sslam_vit_pack_mlp_host(const float *w_up, const float *w_down, const float *row_scale, uint16_t *out);
long long sslam_vit_workspace_bytes(int n_frames, int size);
int sslam_vit_forward(const float *images_chw, int n_frames, int size, const sslam_vit_weights_t *weights_host_struct,
                      void *workspace, long long workspace_bytes, float *tokens_out, void *stream);
/* the same forward from the bf16 patch rows of sslam_preprocess_u8_patches (n, (size/16)^2, 768): no fp32 image, no
 * im2patch pass; bit-identical tokens (the patch embedding rounds the image to bf16 either way) */
int sslam_vit_forward_patches(const void *patches_bf16, int n_frames, int size, const sslam_vit_weights_t *weights_host_struct,
                              void *workspace, long long workspace_bytes, float *tokens_out, void *stream);

/* ---- A1 with the reference's numerics: the same forward in fp32 (the reference's timm model runs in fp32,
 * dino_backbone.py:85) on v_mfma_f32_32x32x2_f32 - fp32 operands, fma-chain contractions, fp32 LayerNorm / softmax / erf-GELU /
 * residual stream.  Agrees with an fp32 torch evaluation of the same weights to ~1e-5 relative (summation order).
 * All pointers DEVICE pointers to fp32; nothing is folded into the weights (ls1 / ls2 are the LayerScale vectors).  The four
 * per-layer matrices - wqkv = rows [q_proj; k_proj; v_proj] (1152, 384), wo (384, 384), wup (1536, 384), wdown (384, 1536) - are
 * re-ordered by sslam_vit_f32_pack_linear_host into the MFMA fragment order their kernel streams from L2 (same values, same
 * size); bqkv has zeros for the k rows; patch_w is the Conv2d weight viewed as (384, 768), as it is; prefix, rope_cos / rope_sin
 * as in sslam_vit_weights_t - DINOv3's construction, the 32 angles of a cell tiled twice: columns d and d + 32 of a row are equal
 * and this entry reads columns 0..31 only (a caller with other tables must not use it; sslam_amd/vit_hip.py checks).
 * Workspace: sslam_vit_f32_workspace_bytes(n_frames, size) bytes (x, LayerNorm output, q / k / v, MLP hidden: 13.7 KB per token;
 * for n_frames <= SSLAM_ATTN_KEY_SPLIT_MAX_FRAMES also the key-split attention's partials, 7.9 KB per token). */
typedef struct {
    const float *ln1_g, *ln1_b, *wqkv, *bqkv, *wo, *bo, *ls1, *ln2_g, *ln2_b, *wup, *bup, *wdown, *bdown, *ls2;
} sslam_vit_layer_f32_t;
typedef struct {
    const float *patch_w, *patch_b, *prefix;
    sslam_vit_layer_f32_t layer[12];
    const float *norm_g, *norm_b, *rope_cos, *rope_sin;
} sslam_vit_weights_f32_t;
/* host helper: fp32 nn.Linear weight (n_out, k_in), n_out % 128 == 0, k_in % 32 == 0 -> element (n, k) at
 * [n/32][k/8][(k%8)/4][n%32][k%4] */
int sslam_vit_f32_pack_linear_host(const float *w, int n_out, int k_in, float *out);
long long sslam_vit_f32_workspace_bytes(int n_frames, int size);
int sslam_vit_forward_f32(const float *images_chw, int n_frames, int size, const sslam_vit_weights_f32_t *weights_host_struct,
                          void *workspace, long long workspace_bytes, float *tokens_out, void *stream);
/* The same forward with the attention's launch form named by the caller.  The reference's callers run the backbone at B = 1
 * (visualize_matches_sequence.py:72-74) and B = 4 (train.py:300-302): a launch of a few frames leaves most of the chip idle
 * while 42 workgroups per frame walk all key tiles one after the other.  attention_form:
 *   SSLAM_ATTN_ONE_PASS  (0) one workgroup per (frame, head, 128 queries) over all keys - the throughput form;
 *   SSLAM_ATTN_KEY_SPLIT (1) five workgroups per (frame, head, 128 queries), one contiguous key range each, un-normalised
 *                            partials (O, running maximum, row sum) in the workspace, merged in the fixed order 0..4 by a second
 *                            launch: deterministic, independent of the batch, ~1e-6 relative from the one-pass form (another
 *                            summation order of the same softmax); n_frames <= SSLAM_ATTN_KEY_SPLIT_MAX_FRAMES, else
 *                            SSLAM_E_INVALID (the workspace holds the partials only up to that size).  The same form runs the
 *                            down projection (K = 1536) with the four waves of a workgroup summing one K quarter each, the
 *                            quarters added in a fixed order: the few-frame form of the forward as a whole.
 * sslam_vit_forward_f32 is this entry with form = KEY_SPLIT when n_frames <= SSLAM_ATTN_KEY_SPLIT_MAX_FRAMES, else ONE_PASS; a
 * caller that cuts one batch into several launches passes the form of the WHOLE batch to each, so that a frame's tokens do not
 * depend on where the cuts fall (sslam_amd/vit_hip.py does). */
#define SSLAM_ATTN_ONE_PASS 0
#define SSLAM_ATTN_KEY_SPLIT 1
#define SSLAM_ATTN_KEY_SPLIT_MAX_FRAMES 8
int sslam_vit_forward_f32_form(const float *images_chw, int n_frames, int size, const sslam_vit_weights_f32_t *weights_host_struct,
                               void *workspace, long long workspace_bytes, float *tokens_out, int attention_form, void *stream);

#ifdef __cplusplus
}
#endif
#endif
