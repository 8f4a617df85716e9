/* C ABI of the MI355X-native SegGPT hot path of kyle-dorman/beach_seg.
 *
 * This is the drop-in boundary for the keyword call the reference makes on the object returned by
 * `load_model` (src/util/ml_util.py:7-13):
 *     self.model(pixel_values=, prompt_pixel_values=, prompt_masks=, labels=, embedding_type="instance")
 * at src/model.py:139-144 (infer), :245-251 (train), :282-288 (val); the callee's signature of record is
 * transformers/models/seggpt/modeling_seggpt.py:831-844 ("HF:" below).  The reference is pure Python, so the
 * "FFI for this path" is a ctypes binding (beach_seg_amd/_native.py; INTEGRATION.md shows the stub a
 * maintainer would add to src/util/ml_util.py).
 *
 * Conventions: every entry point returns 0 on success, non-zero on error with a thread-local message from
 * bsg_last_error().  All device pointers are caller-owned (torch tensors); the library never allocates or
 * frees device memory.  Everything is enqueued on the caller's HIP stream (`stream` = hipStream_t), there are
 * no hidden synchronisations, so a call sequence is hipGraph-capturable.  One handle per device; not
 * thread-safe.  Tensors are contiguous; image tensors are NCHW fp32 exactly as the reference passes them.
 */
#ifndef BEACH_SEG_AMD_H
#define BEACH_SEG_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSG_DTYPE_F32 0  /* parity mode: fp32 storage, exact-f32 MFMA */
#define BSG_DTYPE_BF16 1 /* throughput mode: bf16 storage / bf16 MFMA, fp32 accumulate + fp32 residual stream */
#define BSG_DTYPE_F16 2  /* IEEE-half storage / f16 MFMA (same rate as bf16, 8x less operand round-off), fp32 accumulate +
                            fp32 residual stream; the dgrad chain runs on a power-of-two multiple of the gradient chosen on
                            device from max |grad_pred| (halves hold 6e-5 .. 65504) and is un-scaled in the last GEMM */
#define BSG_MAX_TAPS 8
#define BSG_GLOBAL_WEIGHTS 16 /* weight-table slots before the per-layer blocks */
#define BSG_LAYER_WEIGHTS 20  /* slots per encoder layer */

/* Mirror of the SegGptConfig fields the path reads (HF:configuration_seggpt.py:57-75). */
typedef struct bsg_config {
  int hidden_size, num_layers, num_heads;
  int canvas_h, canvas_w; /* image_size: prompt stacked over query on H */
  int patch_size, mlp_dim, decoder_hidden, merge_index;
  int num_taps;
  int taps[BSG_MAX_TAPS]; /* intermediate_hidden_state_indices */
  float layer_norm_eps;
  int dtype;
  int embed_split; /* bf16 only: 1 = the patch embedding (HF:108) and its dgrad run as split-precision GEMMs, K tripled:
                      activations [hi | hi | lo] against weights [W_hi | W_lo | W_hi] (x = hi + lo in bf16), so the input
                      pixels and the prompt-pixel gradient are not quantised to 8 bits; weight slots 0 / 1 then hold
                      T[D][3*768] / T[768][3*D].  0.6 % of the FLOPs. */
  int gemm_x3;     /* BSG_DTYPE_F32 only: 1 = "float32 at three f16 MFMAs".  Storage, softmax, LayerNorm, residual stream and the
                      3x3 conv stay exact f32; the Linear GEMMs (the 256 x 256 kernel) and the three attention kernels use every
                      f32 operand fragment as hi = f16(x), lo = f16(x - hi) and accumulate hi*hi + hi*lo + lo*hi in fp32 (22-bit
                      operands) on the f16 matrix cores -- up to ~3x the rate of v_mfma_f32_*_f32; measured: the errors of the
                      exact mode on every reference vector at twice its step rate.  The caller stores every Linear weight (slots 0, 1, 6, 7 and
                      2, 3, 5, 6, 10, 11, 13, 14 of a layer) PRE-SPLIT: multiplied by 2^5 (exact; keeps hi and lo in f16's normal
                      range), and every 16-byte chunk of a row holding [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] (f16) of its four
                      values in place of the four floats (beach_seg_amd.seggpt.build_weight_table); the library multiplies the
                      accumulators by 2^-5; the dgrad chain runs on the device-chosen power-of-two multiple
                      of the gradient, as in BSG_DTYPE_F16.  Masks are no longer guaranteed bit-exact (error ~4x exact f32's). */
} bsg_config;

typedef struct bsg_model bsg_model;

/* Weight table (device pointers, caller-owned, must outlive the handle).  "T" = activation dtype of the
 * config; "wT" = the same Linear weight transposed ([in][out]) for the dgrad GEMMs.
 * global: 0 patch_w T[D][768]   1 patch_wT T[768][D] (embed_split: see bsg_config)   2 tok_table_instance f32[2][N][D]
 *         3 tok_table_semantic f32[2][N][D]   4 final_ln_g   5 final_ln_b   6 dec_w T[256*C][taps*D]
 *         7 dec_wT T[taps*D][256*C]   8 dec_b f32   9 conv_w T[C co][9][C ci]   10 conv_wT T[C ci][9][C co]
 *         (taps flipped)   11 conv_b   12 dec_ln_g   13 dec_ln_b   14 head_w f32[3][C]   15 head_b f32[3]
 *         (C = decoder_hidden: 64, the reference checkpoint, or 128, BASELINE config 5)
 * layer l at 16 + 20*l: 0 ln1_g 1 ln1_b 2 qkv_w 3 qkv_wT 4 qkv_b 5 proj_w 6 proj_wT 7 proj_b 8 ln2_g 9 ln2_b
 *         10 fc1_w 11 fc1_wT 12 fc1_b 13 fc2_w 14 fc2_wT 15 fc2_b 16 rel_pos_h f32[2Hp-1][64] 17 rel_pos_w f32
 *         18 rel_cat T[LH + LW][64], LH = roundup16(2Hp), LW = roundup16(2Wp): rel_pos_h rows at 0, rel_pos_w rows
 *         at LH, zero rows elsewhere   19 rel_catT T[64][LH + LW] (its transpose)
 * tok_table[kind][t] folds conv bias (or mask_token for masked tokens of the mask stream), segment token,
 * bicubic-resized position embedding and type token (HF:163-206): pure constants of the checkpoint. */
int bsg_create(const bsg_config* cfg, const void* const* weights, int n_weights, bsg_model** out);
void bsg_destroy(bsg_model* m);

/* Bytes of caller-provided, ZERO-INITIALISED workspace for a batch of `batch` samples.  `train` != 0 adds the
 * activations saved for bsg_backward. */
size_t bsg_workspace_bytes(const bsg_model* m, int batch, int train);
/* Offset/size of a named workspace region (test & debugging aid); returns non-zero if unknown. */
int bsg_workspace_region(const bsg_model* m, int batch, int train, const char* name, int layer, size_t* offset,
                         size_t* bytes);

/* HF:831-951 SegGptForImageSegmentation.forward with the default bool_masked_pos and feature_ensemble=False.
 * pixel_values / prompt_pixel_values / prompt_masks: f32 (B,3,H/2,W); pred_masks: f32 (B,3,H,W).
 * `labels` is not a parameter: under the default mask HF never feeds it to the network (HF:706-715) and the
 * reference ignores `out.loss` (src/model.py:292).  embedding_type: 0 = "instance", 1 = "semantic".
 * save_for_backward != 0 keeps activations in `workspace` for a following bsg_backward on the same workspace. */
int bsg_forward(bsg_model* m, void* stream, int batch, const float* pixel_values, const float* prompt_pixel_values,
                const float* prompt_masks, int embedding_type, float* pred_masks, void* workspace,
                size_t workspace_bytes, int save_for_backward);

/* The same forward for a caller that reads pred_masks on canvas rows >= first_row only -- the fused train step: the
 * reference's loss and its decode of the prediction cover the query half (src/model.py:53-57, process_pred_masks), so
 * first_row = H/2 there.  The encoder runs in full; the decoder (tap LayerNorms, decoder_embed, 3x3 conv, head) runs over the
 * token rows that reach those pixels and, with save_for_backward, the ones bsg_backward_rows(first_row) reads back.  Rows of
 * pred_masks above the first computed 16-row tile are NOT written.  first_row = 0 is bsg_forward; the rows that are written
 * hold the same bits as bsg_forward's.  A following backward must be bsg_backward_rows with a first_row >= this one (a smaller
 * one is refused: it would read decoder activations this forward did not write). */
int bsg_forward_rows(bsg_model* m, void* stream, int batch, const float* pixel_values, const float* prompt_pixel_values,
                     const float* prompt_masks, int embedding_type, int first_row, float* pred_masks, void* workspace,
                     size_t workspace_bytes, int save_for_backward);

/* The same forward with feature_ensemble=True (HF:414-423; the few-shot caller src/predict_no_prompt.py:283-304): the
 * `batch` rows are K prompts for ONE query; in every block the query-half of the attention-block output is replaced
 * by its mean over the prompts (per stream kind before the merge block, over all rows from the merge block on).
 * Inference only: no activations are saved. */
int bsg_forward_ensemble(bsg_model* m, void* stream, int batch, const float* pixel_values,
                         const float* prompt_pixel_values, const float* prompt_masks, int embedding_type,
                         float* pred_masks, void* workspace, size_t workspace_bytes);

/* dgrad-only backward of the frozen network (what Lightning's loss.backward() executes, src/model.py:233-269):
 * grad_pred f32 (B,3,H,W) -> grad_prompt_pixel_values f32 (B,3,H/2,W).  Weights receive no gradient
 * (src/util/ml_util.py:9-10). */
int bsg_backward(bsg_model* m, void* stream, int batch, const float* grad_pred, float* grad_prompt_pixel_values,
                 void* workspace, size_t workspace_bytes);
/* BSG_DTYPE_F16 overflow guard: the backward checks the prompt gradient it produced; workspace region "gscale"
 * (bsg_workspace_region(m, batch, 1, "gscale", -1, ...)) holds, as int32 at byte offset 64: [0] 1 if the LAST backward on
 * this workspace produced a non-finite gradient (the caller must then skip its optimiser step, as torch's GradScaler
 * does), [1] the back-off exponent in force (raised by 2 per overflow OF THE DGRAD CHAIN, i.e. only when grad_pred itself
 * was finite: 4x more headroom for the next backward; lowered by 1 after 1000 clean backwards), [2] clean backwards since
 * the last change, [3] dgrad overflows so far, [4] 1 if the LAST backward's input was itself non-finite -- grad_pred, or the
 * pred_masks of the train-mode forward on this workspace ([6], raised by bsg_forward, consumed by the backward) -- i.e. a bad
 * batch or a forward overflow: the step is dropped, the scale is left alone, [5] backwards dropped for that reason so far.  Device
 * memory: read it on the stream (no host round trip is needed to act on it).  Always 0 for the other dtypes.  The state
 * lives in the caller's workspace because this library owns no device memory: one guard per workspace. */

/* Same, for a grad_pred the caller guarantees to be zero on canvas rows < first_row (what bsg_loss_fwd_bwd produces
 * with first_row = H/2: the reference loss only covers the query half, src/model.py:53-57).  The decoder dgrad then
 * runs only over the token rows that can receive a gradient.  first_row = 0 is bsg_backward. */
int bsg_backward_rows(bsg_model* m, void* stream, int batch, const float* grad_pred, int first_row,
                      float* grad_prompt_pixel_values, void* workspace, size_t workspace_bytes);

/* SegGptLoss of the reference (src/model.py:40-64).  variant 0 reproduces the unsqueeze(1) batch broadcast
 * of :61; variant 1 is the per-sample masked mean (identical at B = 1).  pred f32 (B,3,2h,w), labels f32
 * (B,3,h,w), yesdata u8 (B,h,w).  loss_out: 1 float (device).  grad_pred may be NULL.
 * scratch: >= bsg_loss_scratch_bytes(h, w) bytes, contents irrelevant. */
size_t bsg_loss_scratch_bytes(int h, int w);
int bsg_loss_fwd_bwd(void* stream, int batch, int h, int w, const float* pred, const float* labels,
                     const uint8_t* yesdata, float beta, int variant, float* loss_out, float* grad_pred,
                     void* scratch, size_t scratch_bytes);

/* The same loss with the label image left un-materialised: class_ids u8 (B,h,w) are the label classes
 * (batch["mask"], src/model.py:236), palette_norm f32 (B,K,3) the normalised palette of create_palette
 * (src/model.py:215-231); the label pixel is palette_norm[b][id][c] -- the bits Normalize(torch_apply_mask_rgb(palette,
 * mask)) (src/model.py:238-239) would hold -- and yesdata = (id != 0) (src/model.py:255).  Same scratch. */
int bsg_loss_fwd_bwd_ids(void* stream, int batch, int h, int w, int K, const float* pred, const uint8_t* class_ids,
                         const float* palette_norm, float beta, int variant, float* loss_out, float* grad_pred,
                         void* scratch, size_t scratch_bytes);

/* torch_apply_mask_rgb + Normalize (src/util/ml_util.py:114-132, src/data.py:345; call sites src/model.py:211-212,
 * 238-239): class_ids u8 (B,h,w), palette u8 (B,K,3) -> out f32 (B,3,h,w) = (palette[b][id][c] / 255 - mean[c]) / std[c],
 * each step one correctly rounded float32 operation (bit-exact against the reference's CPU result).  mean = 0, std = 1
 * gives torch_apply_mask_rgb alone.  Ids >= K read entry K-1 (torch would raise). */
int bsg_mask_rgb_norm(void* stream, int batch, int h, int w, int K, const uint8_t* class_ids, const uint8_t* palette,
                      const float mean[3], const float std[3], float* out);

/* process_pred_masks (src/model.py:155-175): arg-min over K normalised palette colours on the bottom half.
 * pred f32 (B,3,2h,w), palette_norm f32 (B,K,3); either output may be NULL. */
int bsg_decode_argmin(void* stream, int batch, int h, int w, int K, const float* pred, const float* palette_norm,
                      int64_t* out_i64, uint8_t* out_u8);

/* HF decode used by the reference's few-shot caller (src/predict_no_prompt.py:297-303 ->
 * SegGptImageProcessor.post_process_semantic_segmentation, HF:image_processing_seggpt.py:300-332): bottom half of
 * pred f32 (B,3,2h,w), x * std + mean, clip(x * 255, 0, 255), arg-min of the squared distance to palette f32 (K,3)
 * (integer colours, shared by the batch) -> out u8 (B,h,w). */
int bsg_decode_hf(void* stream, int batch, int h, int w, int K, const float* pred, const float* palette,
                  const float* mean3, const float* std3, uint8_t* out_u8);

/* prepare_prompt's gather (src/model.py:197 stack + data.py:224 Normalize) and its backward (scatter-add of
 * grad / std into the rows of the flat prompt-gradient buffer). params/grads: f32 (P,3,h,w); idx: i32 (B). */
int bsg_prompt_gather(void* stream, int batch, int h, int w, const float* params, const int32_t* idx,
                      const float mean[3], const float std[3], float* out);
int bsg_prompt_grad_scatter(void* stream, int batch, int h, int w, const float* grad_pixels, const int32_t* idx,
                            const float std[3], float* grad_params);

/* torch.optim.AdamW step (src/model.py:398) on the `n_active` prompt rows listed in `active` (device i32).
 * `touched` (device u8 indexed by ROW, may be NULL): rows with touched[row] == 0 are skipped entirely, the way
 * torch skips a Parameter whose .grad is None.  step_sizes / bc2_sqrts: device f32 [n_active],
 * lr/(1-beta1^t) and sqrt(1-beta2^t) of each listed row. */
int bsg_adamw_step(void* stream, int n_active, long row_elems, float* params, const float* grads, float* exp_avg,
                   float* exp_avg_sq, const int32_t* active, const uint8_t* touched, const float* step_sizes,
                   const float* bc2_sqrts, float lr, float beta1, float beta2, float eps, float weight_decay,
                   float grad_scale);

/* Predict-loop glue (src/predict.py:259-260, 120-159, 100): nearest-resize each (hin,win) u8 class mask to
 * (crop,crop), one-hot vote into the u8 (mh,mw,K) mosaic counters with clipping; crops i32 (n,4) =
 * (xmin,ymin,xmax,ymax), must not overlap within one call.  Then arg-max over K. */
int bsg_vote_paste(void* stream, int n_crops, const uint8_t* masks, int hin, int win, int crop,
                   const int32_t* crops, uint8_t* counter, int mh, int mw, int K);
int bsg_vote_argmax(void* stream, const uint8_t* counter, long n_pixels, int K, uint8_t* out);

/* Tile front-end (src/data.py:88-96, src/util/geo_util.py:316-341): cut n windows (xmin,ymin, side = crop; zero
 * padding outside the mosaic) out of a u8 HWC (mh,mw,3) mosaic, resize each to (S,S) bit-for-bit like Pillow's
 * BICUBIC on 8-bit images (two integer passes, u8 intermediate), then /255 and (x - mean) / std into NCHW f32.
 * coef i32 [S][kmax], bounds i32 [S][2] = (first source index, tap count) per output coordinate: Pillow's
 * precompute_coeffs + normalize_coeffs_8bpc for crop -> S, built by the host (beach_seg_amd.data.pil_bicubic_tables).
 * out f32 (n,3,S,S) and/or out_u8 (n,S,S,3) (the resized bytes themselves); either may be NULL. */
int bsg_tile_frontend(void* stream, const uint8_t* mosaic, int mh, int mw, int n_crops, const int32_t* crops, int crop,
                      int S, const int32_t* coef, const int32_t* bounds, int kmax, const float* mean3,
                      const float* std3, float* out, uint8_t* out_u8);

/* `tif_image` (src/util/geo_util.py:449-470; 8 bands: src/util/multichannel_img.py:7-29) on device: bands (C,H,W) f32
 * (in_dtype 0, what the reference reads with out_dtype=float32, geo_util.py:385) or u16 (in_dtype 1, promoted exactly),
 * nodata u8 (H,W) or NULL -> out_rgb u8 (H,W,3), the mosaic format bsg_tile_frontend consumes.  C = 4: R = band 4, G =
 * band 3, B = mean(band 1, band 2), clip to [min, min + 3000] over the valid pixels, per-channel divide by the max,
 * nodata -> 0, truncating x255 -- bit-exact against the reference's own output; C = 8: log10(1 + band-group mean), per-
 * channel min / max stretch.  scratch: >= 32 bytes of device memory, contents irrelevant. */
int bsg_tif_image(void* stream, int C, int H, int W, int in_dtype, const void* bands, const uint8_t* nodata,
                  uint8_t* out_rgb, void* scratch);

/* Train-time augmentation of src/data.py:195-224 with EXPLICIT random parameters (the reference draws them inside
 * kornia): per sample params[b] = {flags, ex0, ey0, ew, eh}, flags bit 0 = vertical flip, bit 1 = horizontal flip, bit 2 =
 * add `noise` (f32 (B,3,h,w), already scaled by gauss_std and shifted by gauss_mean), bit 3 = RandomSharpness applied,
 * bit 4 = ColorJiggle applied, bit 5 = the erased box is also set to class 0 in mask_out; the e* box is erased to 0 (ew = 0: none).  color (NULL: neither colour operation): f32
 * (B,6) = {brightness, contrast, saturation, hue factor (in turns, kornia's hue_factor), sharpness factor, order code}
 * with order code = i0 | i1<<2 | i2<<4 | i3<<6, operation i0 first (0 brightness, 1 contrast, 2 saturation, 3 hue).
 * Order: flips -> ColorJiggle -> RandomSharpness -> erase -> noise -> Normalize.  The two colour operations follow
 * kornia's published definitions (kornia.enhance.adjust_brightness / adjust_contrast / adjust_saturation / adjust_hue /
 * sharpness); kornia is not installable here, so they are "parity unpinned".  img f32 (B,3,h,w) in [0,1] -> out; mask u8
 * (B,h,w) -> mask_out follows the flips only (both NULL to skip).  scratch: batch*3*h*w floats (forward) / twice that
 * (backward), only with color.  bsg_train_aug_bwd is the gradient wrt img (the learnable prompt pixels,
 * src/model.py:197-205); it needs the forward's img when color is given. */
int bsg_train_aug(void* stream, int batch, int h, int w, const float* img, const uint8_t* mask, const int32_t* params,
                  const float* color, const float* noise, const float mean[3], const float std[3], float* out,
                  uint8_t* mask_out, float* scratch);
int bsg_train_aug_bwd(void* stream, int batch, int h, int w, const float* grad_out, const float* img, const int32_t* params,
                      const float* color, const float std[3], float* grad_img, float* scratch);

/* State update of MulticlassF1Score(num_classes=K, ignore_index) (src/model.py:85-93, 256, 295): confmat[t][p] += 1 over
 * the n pixels whose target t != ignore_index (pass -1 for none); confmat u64 [K][K], K <= 16; pred as i64 or u8 (the
 * other NULL).  tp / fp / fn are the diagonal / column sums / row sums; the matrix is what ranks all-reduce (SUM). */
int bsg_confusion_update(void* stream, long n, int K, int ignore_index, const int64_t* pred_i64, const uint8_t* pred_u8,
                         const uint8_t* target, uint64_t* confmat);

/* The NT GEMM kernel on its own (unit tests and micro-benchmarks): out[M][N] = A[M][K] W[N][K]^T (+ bias[N]),
 * A / W / out in the dtype given (BSG_DTYPE_*), bias f32 or NULL.  dtype 3: f32 A / out with W in the pre-split format of
 * bsg_config.gemm_x3 (three f16 MFMAs on 22-bit operand splits; N > 192). */
int bsg_op_gemm(void* stream, int dtype, int M, int N, int K, const void* A, const void* W, const float* bias,
                void* out);

/* The same kernel with one of the fused epilogues of the encoder blocks (unit tests), T = bf16 or f16 (dtype 1 / 2), acc = A W^T:
 *   epilogue 1: out T = gelu(acc + bias), out2 T = gelu'(acc + bias) (out2 may be NULL)   -- fc1, erf GELU (HF ACT2FN["gelu"])
 *   epilogue 2: out f32 = acc + bias + aux f32 (out may alias aux)                         -- proj / fc2 into the fp32 residual stream
 *   epilogue 3: out T = acc * aux T                                                        -- dfc2 times the saved gelu'
 * aux / out / out2 row-major [M][N].  Shapes with M, N multiples of 256 and K of 128 take gemm_nt_kernel_v5, others v3. */
int bsg_op_gemm_epilogue(void* stream, int dtype, int epilogue, int M, int N, int K, const void* A, const void* W, const float* bias,
                         const void* aux, void* out, void* out2);

/* The fused attention kernels on their own (unit tests and micro-benchmarks), T = bf16 or f16 (dtype 1 / 2): qkv T[S*N][3*nh*64] (q | k | v column
 * blocks, head h at columns h*64), N = hp*wp tokens per stream; rel_cat / rel_catT as in the weight table (slots 18 / 19);
 * `which` bit 0: forward -> out T[S*N][nh*64], lse2 f32[S][nh][hp*32]; bit 1: dQ (needs out, lse2 of a forward and dout
 * T[S*N][nh*64]) -> dqkv q columns; bit 2: dK, dV (needs the tables a dQ launch left in `scratch`) -> dqkv k, v columns, by the
 * kernel the backward uses (one wave per SIMD where the token grid allows it).  A/B bits, same inputs and outputs as bit 2:
 * bit 3 the one-wave-per-SIMD kernel, bit 4 the eight-wave kernel, bit 5 two four-wave workgroups per CU.
 * scratch: >= bsg_op_attention_scratch_bytes bytes. */
size_t bsg_op_attention_scratch_bytes(int S, int nh, int hp);
int bsg_op_attention(void* stream, int dtype, int which, int S, int nh, int hp, int wp, const void* qkv, const void* rel_cat,
                     const void* rel_catT, const void* dout, void* out, float* lse2, void* dqkv, void* scratch,
                     size_t scratch_bytes);

/* Row windows for the dQ (bit 1) and dK/dV (bits 2, 4, 5) kernels of the NEXT bsg_op_attention call of this thread, as
 * bsg_backward_rows applies them (unit tests): dq is wanted for the queries [dq_begin, dq_end) only (dq_end = 0: to N; rows of
 * 128-query workgroups wholly outside come back zero); dK/dV streams the queries >= q_begin (a multiple of 64: the caller
 * guarantees dout == 0 below) and writes only the first key_rows key rows (0 = all), rounded up to the kernel's row group
 * (exact for the one-wave-per-SIMD kernel, 8 / 4 rows for bits 4 / 5).  The windows are consumed by that call. */
int bsg_op_attention_windows(int dq_begin, int dq_end, int q_begin, int key_rows);

/* Optional per-launch timing (HIP events recorded on the caller's stream around the kernels of one category):
 * 0 GEMM, 1 attention fwd, 2 attention bwd dQ, 3 attention bwd dK/dV, 4 3x3 conv.  bsg_profile_read waits for the
 * recorded events and returns the summed kernel time, the summed ALGORITHMIC flops and the launch count since
 * the last bsg_profile_reset.  Disabled by default (no events, nothing recorded). */
int bsg_profile_enable(bsg_model* m, int enable);
int bsg_profile_read(bsg_model* m, int category, double* total_ms, double* total_flops, long* launches);
int bsg_profile_reset(bsg_model* m);

const char* bsg_last_error(void);
const char* bsg_build_info(void);

#ifdef __cplusplus
}
#endif
#endif
