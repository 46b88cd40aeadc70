/* ullava_hip.h -- C ABI of libullava_hip.so, the MI355X (gfx950) kernels of the u-LLaVA forward path.
 *
 * Element types: the functions named *_bf16 below take bfloat16 tensors (the reference's configured dtype, configs/train/ullava.yaml
 * bf16: true); each has a twin *_f16 for IEEE binary16 (declared at the end of this file).  Byte / fp32 work with mixed dtypes takes
 * a dtype code: ULL_DT_F32 / ULL_DT_BF16 / ULL_DT_F16.
 *
 * The reference (OPPOMKLab/u-LLaVA) has no FFI: its hot path is PyTorch nn.Modules calling ATen.  Each entry
 * point below replaces the ATen calls of the reference lines it cites (paths relative to the reference repo;
 * "hf:" = transformers, the reference's pinned third-party dependency, cited from v5.15.0).
 *
 * Conventions: borrowed device pointers (no ownership transfer), caller-provided outputs and
 * workspace, all tensors bf16 (raw uint16 bits) unless noted, element strides, `stream` = hipStream_t (NULL =
 * default stream), no hidden synchronisation, no internal threads, re-entrant per stream.
 * Return 0 on success, negative ULL_ERR_* otherwise (the Python host raises RuntimeError).
 */
#ifndef ULLAVA_HIP_H
#define ULLAVA_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ULL_DT_F32 0
#define ULL_DT_BF16 1
#define ULL_DT_F16 2

#define ULL_OK 0
#define ULL_ERR_ARG (-1)    /* null pointer / non-positive size */
#define ULL_ERR_SHAPE (-2)  /* alignment or shape constraint violated */
#define ULL_ERR_LAUNCH (-3) /* HIP launch error */
#define ULL_ERR_LDS (-4)    /* problem does not fit the 160 KiB LDS budget of the kernel */

/* ull_gemm_bf16 epilogue flags */
#define ULL_EPI_BIAS 1
#define ULL_EPI_ACT_QUICK_GELU (1 << 1) /* hf: activations.py QuickGELUActivation (CLIP MLP) */
#define ULL_EPI_ACT_GELU (2 << 1)       /* torch.nn.GELU (erf): SAM MLPBlock, mlp2x projector, mask-decoder upscaling */
#define ULL_EPI_ACT_RELU (3 << 1)       /* seg/det projectors, SAM decoder MLPs */
#define ULL_EPI_RESID 8                 /* out = bf16(R + bf16(linear)) : residual adds of LlamaDecoderLayer / CLIPEncoderLayer */
#define ULL_EPI_SWIGLU 16               /* W = gate/up rows interleaved in groups of 16; out[N/2] = silu(gate)*up (hf: LlamaMLP.forward) */
#define ULL_EPI_OUT_F32 32              /* C is float32 */
#define ULL_EPI_W_TILED 64                /* W stored tile-major (see below) */
#define ULL_EPI_X_TILED 128               /* X stored tile-major */
#define ULL_EPI_BIAS_ROUNDED 256        /* with ULL_EPI_BIAS: out = round(round(X W^T) + bias), at::linear's unfused matmul + add_ path
                                         * (non-contiguous 3-D input: SAM TwoWayTransformer layer 0 k/v/q projections of the image keys) */

/* C[M,N] = epilogue(X[M,K] * W[N,K]^T).  K % 64 == 0, ldx % 8 == 0, ldw % 8 == 0.
 * Replaces every nn.Linear on the path: hf llama/modeling_llama.py LlamaAttention q/k/v/o_proj, LlamaMLP;
 * hf clip/modeling_clip.py CLIPAttention, CLIPMLP; models/ullava_core.py:117-129 (vision_projector), :325 (lm_head);
 * models/ullava.py:86-118 (seg/det projector, det_decoder); segment_anything/modeling/image_encoder.py:235-260,
 * common.py:13-26, transformer.py:220-242, mask_decoder.py:169-191. */
/* flags also accepts ULL_EPI_W_TILED (64) / ULL_EPI_X_TILED (128): that operand is stored tile-major
 * [rows/256][K/64][256][64] (rows zero-padded to a multiple of 256; every 256 x 64 K-tile = 32 contiguous KiB, which makes an L2
 * miss cheaper: +4 % on the LLaMA layer GEMMs).  Only with M >= 1024, N >= 512, K >= 128 (the 256 x 256 kernel); ldw / ldx ignored.
 * ws / ws_bytes: caller-owned scratch for the stream-K tail (a partial last round of 256x256 tiles is split along K into fp32
 * slabs of 256 KiB there and summed by a finalize launch on the same stream).  NULL / 0 = never split.  The library keeps no
 * buffer and no per-stream state of its own: concurrent calls on different streams are independent as long as each passes its own
 * workspace (ull_gemm_streamk_ws_bytes() is enough for any shape).  Whether to split at all is the caller's policy.
 * Bits 16..22 of flags are per-call tuning overrides for tools/ (ULL_GEMM_TUNE_*); 0 = shipped heuristics. */
#define ULL_GEMM_TUNE_GROUP_M(g) (((g) & 15) << 16) /* tile raster: M-tiles walked before the next N-tile */
#define ULL_GEMM_TUNE_SMALL_KERNEL (1 << 20)        /* force the 128x128 kernel */
#define ULL_GEMM_TUNE_WAVES8 (1 << 21)              /* 256x256 tile on 8 waves of 128x64, whatever the shape */
#define ULL_GEMM_TUNE_WAVES4 (1 << 22)              /* 256x256 tile on 4 waves of 128x128, whatever the shape */
#define ULL_GEMM_TUNE_STAGED_EPILOGUE 512           /* 4-wave kernel: LDS-staged epilogue where the register-direct one would run (tests / tools) */
int ull_gemm_bf16(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* R,
                  int64_t ldr, int64_t M, int64_t N, int64_t K, int flags, void* ws, int64_t ws_bytes, void* stream);

/* hf: LlamaAttention q/k/v_proj (one [3D, D] matrix) + apply_rotary_pos_emb in the GEMM epilogue: C[M, N] = X W^T with the rotary
 * embedding applied to the output columns [0, rope_cols) (q heads then k heads, head_dim = 128; the v columns pass through).  Same
 * three roundings as ull_rope_inplace_bf16 on the Linear's bf16 output.  rope_cos / rope_sin: [M, 64] from ull_rope_table_bf16.
 * flags: ULL_EPI_W_TILED / ULL_EPI_X_TILED / tuning bits only; ws / ws_bytes as for ull_gemm_bf16. */
int ull_gemm_qkv_rope_bf16(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                           const void* rope_cos, const void* rope_sin, int64_t rope_cols, int64_t head_dim, int flags, void* ws,
                           int64_t ws_bytes, void* stream);

/* hf: LlamaRotaryEmbedding.forward for a flat list of positions: cos_out / sin_out bf16 [tokens, half] = bf16(cos / sin(pos * inv_freq))
 * (fp32 trigonometry, one rounding).  positions int64 [tokens], inv_freq float32 [half]. */
int ull_rope_table_bf16(const void* positions, const void* inv_freq, int64_t tokens, int64_t half, void* cos_out, void* sin_out,
                        void* stream);

/* Bytes of stream-K workspace that cover every shape ull_gemm_bf16 may split (256 slabs of 256 x 256 fp32 = 64 MiB). */
int64_t ull_gemm_streamk_ws_bytes(void);

/* The same Linear for decode steps (M <= 4 rows, K % 8 == 0): a pure weight stream, one wave per output feature, no LDS/MFMA.
 * Same flags, layouts and rounding points as ull_gemm_bf16.  Reached from generate() after the prefill
 * (models/ullava_core.py:357-395 keeps only the last token once a KV cache exists). */
int ull_gemv_bf16(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* R,
                  int64_t ldr, int64_t M, int64_t N, int64_t K, int flags, void* stream);

/* The same contract for 2 <= M <= 16 (batched decode steps) on the matrix cores: 16 output features x 16 padded rows per 16x16x32
 * MFMA, weights streamed once, K % 32 == 0. */
int ull_gemm_skinny_bf16(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* R,
                         int64_t ldr, int64_t M, int64_t N, int64_t K, int flags, void* stream);

/* ull_gemv_bf16 with the preceding LlamaRMSNorm fused in: C = epilogue(rmsnorm(X; norm_w, eps) * W^T), M * K <= 16384.
 * Decode-step form of input_layernorm -> q/k/v_proj and post_attention_layernorm -> gate/up_proj (hf LlamaDecoderLayer.forward). */
int ull_gemv_rmsnorm_bf16(const void* X, int64_t ldx, const void* norm_w, float eps, const void* W, int64_t ldw, void* C, int64_t ldc,
                          const void* bias, const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, int flags, void* stream);

/* Decode-step q|k|v projection with RoPE and the KV-cache append in its epilogue (M = B * S <= 4 tokens): one launch for
 * hf LlamaAttention.forward's q/k/v_proj + apply_rotary_pos_emb + cache update (models/ullava_core.py:357-395 feeds one token per step).
 * W = [3 * H * hd, K] (q | k | v rows); norm_w (may be null) = the preceding LlamaRMSNorm; cos_tab / sin_tab = ull_rope_table_bf16 of the
 * step's positions [B * S, hd / 2].  Q_out [B * S, H * hd] (row pitch ldq) receives the rotated queries, k_cache[b, h, past + s, :] the
 * rotated keys, vt_cache[b, h, :, slot(past + s)] the values.  Same bits as ull_gemv_rmsnorm_bf16 followed by ull_rope_append_bf16. */
int ull_gemv_qkv_rope_append_bf16(const void* X, int64_t ldx, const void* norm_w, float eps, const void* W, int64_t ldw, void* Q_out, int64_t ldq,
                                  const void* cos_tab, const void* sin_tab, void* k_cache, void* vt_cache, int64_t B, int64_t S, int64_t H,
                                  int64_t hd, int64_t K, int64_t smax, int64_t past, void* stream);

/* y = w * bf16(x * rsqrt(mean(x^2) + eps)).  hf: LlamaRMSNorm.forward. */
int ull_rmsnorm_bf16(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, int64_t rows, int64_t D, float eps, void* stream);

/* models/ullava_core.py:327-338: shifted CrossEntropyLoss(logits[:, :-1], labels[:, 1:]), ignore_index -100.
 * out float[2] = {sum of token losses, counted tokens} (caller zeroes it; loss = out[0] / out[1]). */
int ull_shifted_cross_entropy_bf16(const void* logits, int64_t ld, const void* labels, int64_t B, int64_t S, int64_t V, void* out,
                                   void* stream);

/* torch.nn.LayerNorm over the last dim (hf CLIPEncoderLayer.layer_norm1/2; SAM Block.norm1/2, TwoWayAttentionBlock.norm1-4). */
int ull_layernorm_bf16(const void* x, int64_t ldx, const void* w, const void* b, void* y, int64_t ldy, int64_t rows, int64_t D, float eps,
                       void* stream);

/* hf: CLIPVisionEmbeddings.forward (cat(class, patches) + position_embedding) fused with CLIPVisionModel.pre_layrnorm. */
int ull_clip_embed_ln_bf16(const void* patch, int64_t ldp, const void* cls, const void* pos, const void* w, const void* b, void* y,
                           int64_t ldy, int64_t n_img, int64_t tokens, int64_t D, float eps, void* stream);

/* Attention.  Sk <= 1024: score rows stay in registers; larger Sk: two-pass streaming kernel with the same rounding points.
 * hf: llama eager_attention_forward (causal + key padding mask), clip eager_attention_forward (no mask): scale_mode 1
 * multiplies after the matmul like both.  SAM encoder (image_encoder.py:235-260,354-392): q_scale = hd^-0.5 applied to Q
 * as a bf16 tensor op, scale_mode 0, and the decomposed rel-pos bias added to the bf16 scores (rel_h then rel_w): rel_mode 1 =
 * rel_h [B*H,Sq,rel_kh] / rel_w [B*H,Sq,rel_kw] precomputed (ull_sam_relpos_bf16); rel_mode 2 = rel_h/rel_w are the module's raw
 * rel_pos_h [2*rel_kh-1,hd] / rel_pos_w [2*rel_kw-1,hd] parameters and the kernel builds the tables itself on the MFMA (the 14 x 14
 * windows have their own entry, ull_sam_window_attention_bf16).  SAM decoder (transformer.py:220-242): scale_mode 2 divides by sqrt(hd).
 * Q/K: [B,H,S,hd] by strides, hd contiguous.  Vt: [B,H,hd,vt_len] as written by ull_transpose_v_bf16 (vt_len % 64 == 0); or, with
 * vt_len = 0, V itself [B,H,S,hd] by (vt_bs, vt_hs, vt_ds = token stride): the LLaMA (hd 128, Sk <= 1024) and CLIP (hd 64, Sk <= 704)
 * prefill kernels and the SAM global-attention kernel (64 x 64 grid, rel_mode 2) transpose it on the fly through the LDS
 * (ULL_ERR_SHAPE where no such kernel exists: Sq <= 16, other flavors).
 * key_mask: int32 [B,Sk] (nonzero = attend) or NULL.  zeros: >= 16 readable zero bytes (head-dim padding source). */
int ull_attention_bf16(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_ss, const void* K, int64_t k_bs, int64_t k_hs, int64_t k_ss,
                       const void* Vt, int64_t vt_bs, int64_t vt_hs, int64_t vt_ds, int64_t vt_len, void* O, int64_t o_bs, int64_t o_hs,
                       int64_t o_ss, const void* key_mask, int64_t B, int64_t H, int64_t Sq, int64_t Sk, int64_t hd, int causal,
                       int scale_mode, float scale, float q_scale, const void* rel_h, const void* rel_w, int64_t rel_kh,
                       int64_t rel_kw, int rel_mode, const void* zeros, void* stream);

/* hf: apply_rotary_pos_emb on n_heads consecutive heads (q heads then k heads of a fused QKV row), in place.
 * positions int64 [tokens]; inv_freq float32 [hd/2] computed by the host exactly as LlamaRotaryEmbedding does. */
int ull_rope_inplace_bf16(void* x, int64_t row_stride, const void* positions, const void* inv_freq, int64_t tokens, int64_t n_heads,
                          int64_t hd, void* stream);

/* Decode step (models/ullava_core.py:357-395 with a KV cache): the same RoPE on the H q heads (in place) and the H k heads of
 * fused [q|k|v] rows, plus the append of the new tokens' k (roped) / v to the cache: K cache [B,H,smax,hd] row past+s, V^T cache
 * [B,H,hd,smax] column = the key-permuted slot of position past+s (layout of ull_transpose_v_bf16).  qkv rows = B*S tokens. */
int ull_rope_append_bf16(void* qkv, int64_t row_stride, const void* positions, const void* inv_freq, int64_t B, int64_t S, int64_t H,
                         int64_t hd, void* k_cache, void* vt_cache, int64_t smax, int64_t past, void* stream);

/* V [B,S,H,hd] -> Vt [B,H,hd,pitch] (pitch % 64 == 0), zeros for keys >= S, keys permuted inside each 32-key block
 * (slot 8g+4a+r <- key 16a+4g+r): the K-contiguous A-operand layout of P*V matching the P register layout. */
int ull_transpose_v_bf16(const void* v, int64_t v_bs, int64_t v_ss, void* vt, int64_t B, int64_t S, int64_t H, int64_t hd, int64_t pitch,
                         void* stream);

/* Patch extraction for conv(kernel = stride = ps): out[(img,py,px)][(c*ps+ky)*ps+kx], zero padded to Kp columns.
 * hf: CLIPVisionEmbeddings.patch_embedding; segment_anything/modeling/image_encoder.py:395-426 (PatchEmbed). */
int ull_im2col_bf16(const void* img, void* out, int64_t n_img, int64_t C, int64_t H, int64_t W, int64_t ps, int64_t Kp, void* stream);

/* models/ullava_core.py:205-226,248-251: per-sample start/end token counts, first start position, running feature index.
 * spans int32 [B,4] = {kind 0 text / 1 image / 2 video, first start pos, feature index, error bits}: bit 0 = start/end counts
 * differ (the reference's assert, :209-211), bit 1 = an id outside [0, vocab) (nn.Embedding's IndexError, :191; vocab <= 0: unchecked). */
int ull_mm_spans(const void* ids, int64_t B, int64_t S, int64_t img_start, int64_t img_end, int64_t vid_start, int64_t vid_end,
                 int64_t vocab, void* spans, void* stream);

/* One greedy decoding step of HF `generate` (the method the reference's models inherit: models/ullava.py:350-361) for the whole batch:
 * next = argmax(logits[b, 0..V)) with the first index on ties (torch.argmax); rows whose unfinished[b] == 0 get `pad` instead when
 * has_pad; seq[b*seq_ld + pos] = token (int64); a live row that emitted one of the n_eos ids in `eos` (int64, device) becomes finished;
 * alive[0] (int32, caller-zeroed) += number of rows still unfinished.  logits: 16-bit, rows row_stride elements apart. */
int ull_greedy_step_bf16(const void* logits, int64_t row_stride, int64_t B, int64_t V, void* unfinished, const void* eos, int64_t n_eos,
                         int64_t pad, int has_pad, void* seq, int64_t seq_ld, int64_t pos, void* alive, void* stream);

/* models/ullava_core.py:191,243-245,266-268: token-embedding lookup with the projected visual tokens spliced in after
 * the first start token (the torch.cat of the reference, done as one gather).  Image i's n_img_tok feature rows start
 * at row i*img_pitch + img_off of img_feat (pitch = patches + 1, off = 1 skips the CLS row without a copy).
 * ids outside [0, vocab) are clamped (no out-of-bounds read; ull_mm_spans reports them). */
int ull_embed_splice_bf16(const void* ids, const void* table, const void* img_feat, int64_t n_img_tok, int64_t img_pitch,
                          int64_t img_off, const void* vid_feat, int64_t n_vid_tok, const void* spans, void* out, int64_t B, int64_t S,
                          int64_t D, int64_t vocab, void* stream);

/* models/ullava_core.py:173-178: f[b,t,tok_pitch,d] (patches = tokens tok_off..tok_off+N) -> concat([mean over n (temporal), mean over t (spatial)], dim=1). */
int ull_video_pool_bf16(const void* f, void* out, int64_t B, int64_t T, int64_t N, int64_t D, int64_t tok_pitch, int64_t tok_off,
                        void* stream);

/* dst[i,:] = src[idx[i],:]  (models/ullava.py:190-199: boolean-mask gather of [SEG]/[LOC] rows, done BEFORE the projector). */
int ull_gather_rows_bf16(const void* src, int64_t lds, const void* idx, void* dst, int64_t ldd, int64_t n, int64_t D, void* stream);

/* nn.Dropout in training mode (PEFT lora_dropout, train_ullava.py:222): y = keep[i] ? round(x[i] / (1 - p)) : 0, scale = 1 / (1 - p), keep = one
 * byte per element from the caller's RNG.  Its own backward (apply to dy with the same mask). */
int ull_dropout_apply_bf16(const void* x, const void* keep, void* y, int64_t n, float scale, void* stream);

/* out = bf16(a + b[row % b_rows])  (bf16 tensor adds: queries + query_pe, keys + key_pe, x + pos_embed). */
int ull_add_rows_bf16(const void* a, const void* b, void* out, int64_t rows, int64_t D, int64_t b_rows, void* stream);

/* ---- SAM (models/segment_anything/modeling) -- token-major / channels-last layouts ---------------------------------- */

/* image_encoder.py:263-289 window_partition: x [B,H,W,C] -> [B*nW, ws*ws, C], zero rows for the F.pad region. */
int ull_window_partition_bf16(const void* x, void* out, int64_t B, int64_t H, int64_t W, int64_t C, int64_t ws, void* stream);

/* image_encoder.py:292-318 window_unpartition fused with the residual add of Block.forward (:190): out = shortcut + unpart(win). */
int ull_window_unpartition_add_bf16(const void* win, const void* shortcut, void* out, int64_t B, int64_t H, int64_t W, int64_t C, int64_t ws,
                                    void* stream);

/* image_encoder.py:321-392 get_rel_pos + einsum("bhwc,hkc->bhwk") / ("bhwc,wkc->bhwk"): q [NB,nH,KH*KW,hd] by strides ->
 * out_h [NB*nH, KH*KW, KH], out_w [NB*nH, KH*KW, KW].  rel_pos_h [2*KH-1, hd], rel_pos_w [2*KW-1, hd] (tables of another length go
 * through ull_interp_rows_linear_bf16 first). */
int ull_sam_relpos_bf16(const void* q, int64_t q_bs, int64_t q_hs, int64_t q_ss, const void* rel_pos_h, const void* rel_pos_w, void* out_h,
                        void* out_w, int64_t NB, int64_t nH, int64_t KH, int64_t KW, int64_t hd, void* stream);

/* image_encoder.py:176-190 (Block.forward between norm1 and proj) for the 14 x 14 windows, on tokens that stay in IMAGE order:
 * window_partition with its zero padding + Attention with the decomposed rel-pos bias + window_unpartition, one launch.  qkv
 * [B*H*W, ld] rows q|k|v of 3*nH*hd, out [B*H*W, ldo]; pad_row = the q|k|v row of a padded token = the qkv bias (the reference pads
 * the normalised activations with zeros); rel_pos_h / rel_pos_w [27, hd].  ws = 14, hd = 80; no V^T image is needed. */
int ull_sam_window_attention_bf16(const void* qkv, int64_t ld, const void* pad_row, const void* rel_pos_h, const void* rel_pos_w, void* out,
                                  int64_t ldo, int64_t B, int64_t H, int64_t W, int64_t nH, int64_t hd, int64_t ws, float q_scale,
                                  const void* zeros, void* stream);

/* image_encoder.py:336-343: the linear resize get_rel_pos applies to a rel-pos table whose length differs from 2*size-1:
 * x [L, C] -> y [M, C] = F.interpolate(x as [1, C, L], size = M, mode = "linear") with ATen's CPU rounding (weights rounded to the element type). */
int ull_interp_rows_linear_bf16(const void* x, void* y, int64_t L, int64_t M, int64_t C, void* stream);

/* common.py:31-43 LayerNorm2d on channels-last rows [rows, C] with the reference's bf16 op chain; gelu != 0 fuses the
 * nn.GELU that follows it in mask_decoder.py:53-64 output_upscaling. */
int ull_layernorm2d_cl_bf16(const void* x, const void* w, const void* b, void* y, int64_t rows, int64_t C, float eps, int gelu, void* stream);

/* image_encoder.py:117-124 ("prevent overflow"): an fp16 model runs the neck under torch.autocast(float32), so the neck's two LayerNorm2d
 * (common.py:31-43) see fp32 convolution outputs and compute in fp32 with their fp16 weight / bias promoted.  fp16 build only (a bf16 model
 * goes straight through ull_layernorm2d_cl_bf16).  x[row] = xa[row] + xb[row] * xb_scale (xb may be null): fp32 channels-last rows [rows, C];
 * y = w * ((x - mean) / sqrt(var + eps)) + b in fp32.  y_lo == null: y_hi = fp16(y), the `.to(float16)` that ends the neck.  y_lo != null:
 * y_hi = fp16(y), y_lo = fp16((y - y_hi) * 2^11): the two-term split that lets the following 3x3 convolution run as fp16 GEMMs with fp32
 * accumulation on an fp32-accurate input, conv(y) = conv(y_hi) + 2^-11 conv(y_lo). */
int ull_neck_layernorm2d_f32in_f16(const void* xa, const void* xb, float xb_scale, const void* w, const void* b, void* y_hi, void* y_lo,
                                   int64_t rows, int64_t C, float eps, void* stream);

/* image_encoder.py:100-106 neck Conv2d(k=3, padding=1): x [B,H,W,C] -> cols [B*H*W, 9*C] in (ky,kx,ci) order. */
int ull_im2col3x3_bf16(const void* x, void* out, int64_t B, int64_t H, int64_t W, int64_t C, void* stream);

/* mask_decoder.py:150-158: masks[n,t,Y,X] = hyper_in[n,t,:] . upscaled[n,:,Y,X]; `up` is the blocked output of the two
 * ConvTranspose2d(k=2,s=2) GEMMs: [n][G*G cells][d1][d2][C].  masks bf16 [n,T,4G,4G]. */
int ull_mask_matmul_bf16(const void* hyper, const void* up, void* masks, int64_t n, int64_t T, int64_t C, int64_t G, void* stream);

/* sam.py:137-172 postprocess_masks: F.interpolate(bilinear, align_corners=False) in fp32 of n images of dtype in_dtype (ULL_DT_*;
 * strided crop). */
int ull_bilinear_f32(const void* in, int in_dtype, int64_t in_img_stride, int64_t in_row_stride, int64_t in_h, int64_t in_w, void* out,
                     int64_t n, int64_t out_h, int64_t out_w, void* stream);

/* Fused ViT patch embedding: Conv2d(C, N, kernel = stride = ps) as one GEMM whose A tiles are LDS-DMA'd straight from the pixels
 * (no im2col buffer).  hf CLIPVisionEmbeddings.patch_embedding (models/ullava_core.py:131-159) and SAM PatchEmbed
 * (segment_anything/modeling/image_encoder.py:395-427).  img contiguous [n_img, C, H, W] bf16; Wp [N, Kp] packed as
 * Wp[n][(c*ps + ky)*16 + kx] = w[n][c][ky][kx] (kx < ps), zero elsewhere, Kp = ceil(C*ps*16 / 64)*64; out [n_img*(H/ps)*(W/ps), N];
 * zeros = 16 zero bytes; bias [N] or null.  Nothing behind the image buffer is read. */
int ull_patchify_bf16(const void* img, int64_t n_img, int64_t C, int64_t H, int64_t W, int64_t ps, const void* Wp, int64_t Kp,
                      const void* bias, void* out, int64_t ldc, int64_t N, const void* zeros, void* stream);

/* ---- image pre/post-processing either side of the path (byte / integer work, bit-exact against the host libraries) ------------- */

/* One separable pass of Pillow's 8-bit resampler (libImaging/Resample.c ImagingResampleHorizontal_8bpc / Vertical_8bpc) over a
 * uint8 [H, W, C] image: axis 1 -> dst [H, out_size, C], axis 0 -> dst [out_size, W, C].  bounds int32 [out_size, 2] = (first tap,
 * tap count); coeffs int32 [out_size, ksize] = the 22-bit fixed-point taps (precompute_coeffs + normalize_coeffs_8bpc, computed
 * by the host).  Replaces PIL.Image.resize inside transformers CLIPImageProcessor (dataset/processors/clip_processor.py:93) and
 * inside ResizeLongestSide.apply_image (models/segment_anything/utils/transforms.py:27-35). */
int ull_resample_u8(const void* src, int64_t H, int64_t W, int64_t C, int axis, int64_t out_size, const void* bounds, const void* coeffs,
                    int64_t ksize, void* dst, void* stream);

/* uint8 [H, W, 3] -> [3, OH, OW] of dtype out_dtype (ULL_DT_*): dst[c][y][x] = lut[c][src[top+y][left+x][c]] for y < copy_h, x < copy_w, else 0.
 * lut fp32 [3, 256] holds the reference's normalisation of every byte value.  CLIP: center crop + /255 + (x-mean)/std
 * (transformers rescale + normalize); SAM: (x-mean)/std + zero pad to 1024 (dataset/tools/mask_toolbox.py:15-25). */
int ull_u8_lut_chw(const void* src, int64_t H, int64_t W, int64_t C, int64_t top, int64_t left, const void* lut, void* dst, int64_t OH,
                   int64_t OW, int64_t copy_h, int64_t copy_w, int out_dtype, void* stream);

/* evaluation/tools.py:29-41 intersectionAndUnionGPU(K = 2) on (logits > 0) (trainers/ullava_trainer.py:44): logits fp32
 * [n, hw], target uint8 [n, hw]; counts int32 [n, 6] += {inter0, inter1, out0, out1, tgt0, tgt1}, ignore_index pixels dropped. */
int ull_mask_iou_counts(const void* logits, const void* target, int64_t n_masks, int64_t hw, int ignore_index, void* counts, void* stream);

/* ---- forward values of the training losses (models/loss.py; combined by models/ullava.py:283-312) ------------------------------ */

/* Per mask and per 1/64th of its pixels: {sum BCE-with-logits, sum (sigmoid/scale)*t, sum sigmoid/scale, sum t/scale} ->
 * part float [n_masks, 64, 4].  The host adds the 64 partials and forms sigmoid_ce_loss (loss.py:72-89) and dice_loss (:45-69). */
int ull_mask_loss_sums_f32(const void* logits, const void* target, int64_t n_masks, int64_t hw, float scale, void* part, void* stream);

/* bbox_l1_loss / bbox_giou_loss numerators (loss.py:92-110): out float[2] = {sum |pred - gt|, sum (1 - GIoU(pred_i, gt_i)) over
 * predictions with x1 >= x0 and y1 >= y0}.  pred [n,4] of dtype pred_dtype (ULL_DT_*), gt [n,4] fp32, xyxy. */
int ull_box_losses_f32(const void* pred, int pred_dtype, const void* gt, int64_t n, void* out, void* stream);

/* ---- fused two-way mask decoder (north_star: "SAM's MaskDecoder cross-attention fused into one LDS-resident kernel") ----------------
 * models/segment_anything/modeling/transformer.py:62-106,151-182,220-242; mask_decoder.py:137-164.  Fixed dims: width 256, cross-attention
 * internal width 128, 8 heads, T <= 8 tokens per prompt.  All weights in nn.Linear layout, every 16-bit rounding point of the
 * reference graph kept.  queries / qpe / out [n, T, 256]; keys [n, P, 256]; pos [P, 256]. */

/* Token self attention (:151-160, :220-242), one (prompt, head) per block: q/k/v projections of the head + attention over the T tokens;
 * att [n, T, 256] = concatenated heads.  first != 0: layer 0 (no positional add). */
int ull_sam_self_attn_heads_bf16(const void* queries, const void* qpe, int64_t n, int64_t T, int first, const void* wq, const void* bq,
                                 const void* wk, const void* bk, const void* wv, const void* bv, void* att, void* stream);

/* out [n, T, 256] = LayerNorm(res + att Wo^T + bo) (res NULL: no residual), att [n, T, din], din = 256 or 128; then up to three token-side
 * projections for the attention that follows: projs = HOST array of 3 x {w [128, 256], b [128], out [n, T, 128]} device pointers (w NULL =
 * unused; projs NULL = none), add_pe[i] != 0: the projection's input is out + qpe. */
int ull_sam_out_ln_bf16(const void* att, int64_t din, const void* res, const void* qpe, int64_t n, int64_t T, const void* wo, const void* bo,
                        const void* ln_w, const void* ln_b, float eps, void* out, const void* const* projs, const int* add_pe, void* stream);

/* MLP block + norm3 (:168-171) in two launches (hidden split over hidden / 256 blocks per prompt, then sum + residual + LayerNorm and the
 * projections as in ull_sam_out_ln_bf16).  part_ws: float32 [n * (hidden / 256) * 8 * 256] caller-owned scratch. */
int ull_sam_token_mlp_ln_bf16(const void* queries, const void* qpe, int64_t n, int64_t T, int64_t hidden, const void* w1, const void* b1,
                              const void* w2, const void* b2, const void* ln_w, const void* ln_b, float eps, void* part_ws, void* out,
                              const void* const* projs, const int* add_pe, void* stream);

/* The four hyper-network MLPs + the IoU head (mask_decoder.py:137-164) in one launch.  ptrs: HOST array of 30 device pointers =
 * 5 MLPs x {w1, b1, w2, b2, w3, b3} (hyper 0..3 on token rows 1..4, then the IoU head on row 0); hyper [n, 4, hyper_out], iou [n, n_iou]. */
int ull_sam_small_mlps_bf16(const void* hs, int64_t n, int64_t T, const void* const* ptrs, int64_t n_mask_tokens, int64_t hyper_out,
                            int64_t n_iou, void* hyper, void* iou, void* stream);

/* token -> image cross attention core (:162-166 and the final attention :100-105), two launches: (1) per 128-key tile the v and k
 * projections on the MFMA from ONE LDS-resident tile (weights streamed through LDS in K-slices) and the scaled scores against
 * qproj [n, T, 128]; (2) one (prompt, head) per block: exact fp32 softmax over all P keys and P V -> att [n, T, 128].  late_bias_kv != 0:
 * bias added after the rounding of the k / v products (layer 0, see ULL_EPI_BIAS_ROUNDED).  Caller-owned scratch: scores_ws [n*8*8*P]
 * and vproj_ws [n*P*128] elements.  P = 4096 or 1024. */
int ull_sam_t2i_attention_bf16(const void* qproj, const void* keys, const void* pos, int64_t n, int64_t T, int64_t P, const void* wk,
                               const void* bk, const void* wv, const void* bv, int late_bias_kv, void* scores_ws, void* vproj_ws, void* att,
                               void* stream);

/* image -> token cross attention + norm4 (:173-180) in ONE launch per 128 image rows: q projection on the MFMA (tile LDS-resident), softmax
 * over the T tokens against kproj / vproj [n, T, 128], P V, out-projection on the MFMA, residual, LayerNorm; out [n, P, 256] != keys. */
int ull_sam_i2t_attention_ln_bf16(const void* keys, const void* pos, const void* kproj, const void* vproj, int64_t n, int64_t T, int64_t P,
                                  const void* wq, const void* bq, const void* wo, const void* bo, int late_bias_q, const void* ln_w,
                                  const void* ln_b, float eps, void* out, void* stream);

/* ---- backward kernels (train_ullava.py / train_ullava_core.py; torch autograd of the cited forward ops) --------
 * Gradients are evaluated in fp32 from the stored 16-bit tensors and rounded once on output.  Linear layers need no entry point of
 * their own: dX = dY W and dW = dY^T X are ull_gemm_bf16 on transposed operands. */

/* hf LlamaRMSNorm backward: dx [rows, D]; dw float32 [D] += sum over rows (caller zeroes it; may be NULL). */
int ull_rmsnorm_bwd_bf16(const void* x, int64_t ldx, const void* w, const void* dy, int64_t lddy, void* dx, int64_t lddx, void* dw,
                         int64_t rows, int64_t D, float eps, void* stream);

/* hf LlamaMLP activation on the UNFUSED gate/up projection kept in ULL_EPI_SWIGLU's interleaved column order (groups of 16 gate |
 * 16 up): a [M, I] = silu(gate) * up with the reference's two roundings; backward: dgu [M, 2I] from da [M, I]. */
int ull_swiglu_fwd_bf16(const void* gu, void* a, int64_t M, int64_t I, int halves, void* stream);
int ull_swiglu_bwd_bf16(const void* gu, const void* da, void* dgu, int64_t M, int64_t I, int halves, void* stream);

/* ReLU backward as a selection (seg / det projector MLPs, the mask decoder's MLPs): dx[i] = y[i] > 0 ? dy[i] : 0 over n elements. */
int ull_relu_mask_bf16(const void* y, const void* dy, void* dx, int64_t n, void* stream);

/* Backward of ull_rope_inplace_bf16 (the rotation is orthogonal: the transposed rotation of the gradient), in place. */
int ull_rope_bwd_inplace_bf16(void* dx, int64_t row_stride, const void* positions, const void* inv_freq, int64_t tokens, int64_t n_heads,
                              int64_t hd, void* stream);

/* Backward of O = softmax(mult * Q K^T + mask) V (hf eager_attention_forward; SAM transformer.py:220-242 with mult = 1/sqrt(hd)).
 * All tensors [B, H, S, hd] by element strides (batch, head, seq), hd contiguous: strides = int64[24] for Q, K, V, O, dO, dQ, dK, dV.
 * causal / key_mask as in ull_attention_bf16.  scratch: float32 [2 * B * H * Sq].  Sk <= ~4900 (LDS). */
int ull_attention_bwd_bf16(const void* Q, const void* K, const void* V, const void* O, const void* dO, void* dQ, void* dK, void* dV,
                           const int64_t* strides, const void* key_mask, int64_t B, int64_t H, int64_t Sq, int64_t Sk, int64_t hd, int causal,
                           float mult, void* scratch, void* stream);

/* The same backward on the matrix cores, for hd = 64 / 128: same arguments plus Qt, Kt, dOt = the ull_transpose_v_bf16 images
 * [B, H, hd, pitch] of Q, K and dO (pitch >= ceil64(max(Sq, Sk))); every stride a multiple of 4 elements, token strides of 8.
 * hd = 128 (the LLaMA block) stages the operand tiles through the LDS and takes the transposed fragments from the transposing LDS
 * read: Qt / Kt / dOt are not read there and may be NULL.  scratch: float32 [2 * B * H * ceil64(Sq)].  No atomics, same fp32
 * softmax recomputation. */
int ull_attention_bwd_mfma_bf16(const void* Q, const void* K, const void* V, const void* O, const void* dO, const void* Qt, const void* Kt,
                                const void* dOt, int64_t pitch, void* dQ, void* dK, void* dV, const int64_t* strides, const void* key_mask,
                                int64_t B, int64_t H, int64_t Sq, int64_t Sk, int64_t hd, int causal, float mult, void* scratch, void* stream);

/* Backward of ull_shifted_cross_entropy_bf16: dlogits [B, S, V] (same ld); stats = the forward's float[2]; gout = upstream gradient
 * of the mean loss (one float32 on the device). */
int ull_shifted_cross_entropy_bwd_bf16(const void* logits, int64_t ld, const void* labels, int64_t B, int64_t S, int64_t V, const void* stats,
                                       const void* gout, void* dlogits, void* stream);

/* Backward of ull_embed_splice_bf16: demb [B, S, D] -> d_table float32 [vocab, D] += (caller zeroes; NULL = not needed), d_img /
 * d_vid = gradients of the projected visual features (same layouts as the forward's inputs; NULL = not needed).  Rows inside a
 * spliced span never reach d_table (reference models/ullava_core.py:243-245: torch.cat drops the placeholder rows), with or without
 * d_img / d_vid.  detach_text != 0 = projector_from_scratch (:230-240, :255-264): in samples that carry an image / video only the
 * start- and end-token rows accumulate into d_table; text-only samples keep every row. */
int ull_embed_splice_bwd_bf16(const void* ids, const void* demb, void* d_table, void* d_img, int64_t n_img_tok, int64_t img_pitch,
                              int64_t img_off, void* d_vid, int64_t n_vid_tok, const void* spans, int64_t B, int64_t S, int64_t D, int64_t vocab,
                              int detach_text, void* stream);

/* torch.nn.LayerNorm backward (SAM transformer.py norm1-4 / norm_final_attn): dx; dw, db float32 [D] += (caller zeroes; may be NULL). */
int ull_layernorm_bwd_bf16(const void* x, int64_t ldx, const void* w, const void* dy, int64_t lddy, void* dx, int64_t lddx, void* dw, void* db,
                           int64_t rows, int64_t D, float eps, void* stream);

/* common.py:31-43 LayerNorm2d backward on channels-last rows; gelu != 0: dy is the gradient AFTER the fused nn.GELU. */
int ull_layernorm2d_cl_bwd_bf16(const void* x, const void* w, const void* b, const void* dy, void* dx, void* dw, void* db, int64_t rows,
                                int64_t C, float eps, int gelu, void* stream);

/* torch.nn.GELU (erf) as a stand-alone op of the training path, and its backward. */
int ull_gelu_fwd_bf16(const void* x, void* y, int64_t n, void* stream);
int ull_gelu_bwd_bf16(const void* x, const void* dy, void* dx, int64_t n, void* stream);

/* Backward of ull_mask_matmul_bf16: dhyper float32 [n, T, C] += (caller zeroes), dup in `up`'s blocked layout.  T <= 8, C <= 32. */
int ull_mask_matmul_bwd_bf16(const void* hyper, const void* up, const void* dmasks, void* dhyper, void* dup, int64_t n, int64_t T, int64_t C,
                             int64_t G, void* stream);

/* out[i] = scale * sum_r x[r, i] over R contiguous slabs of n elements (fp32 accumulation, one rounding): the local reduction of the
 * direct-exchange gradient reduce-scatter over xGMI (u-llava_amd/dist.py; the reference delegates this to DeepSpeed ZeRO-2,
 * configs/deepspeed/bf16_zero2.json:5-11). */
int ull_sum_slabs_bf16(const void* x, void* out, int64_t R, int64_t n, float scale, void* stream);

/* out float32 [N] = column sums of x [rows, N] (bias gradients). */
int ull_colsum_bf16(const void* x, int64_t ld, int64_t rows, int64_t N, void* out, void* stream);

/* y[c][r] = x[r][c] (x [R, C] with row stride ldx, y [C, R] with row stride ldy): the transposed operands of the Linear backward
 * (torch.autograd of nn.Linear: dX = dY W, dW = dY^T X), which run through the NT GEMM of ull_gemm_bf16. */
int ull_transpose2d_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t R, int64_t C, void* stream);

/* Backward of ull_mask_loss_sums_f32 (g float32 [n_masks, 4] -> dlogits float32 [n_masks, hw]), of ull_box_losses_f32 (gw float32[2] ->
 * dpred float32 [n, 4]) and the adjoint of ull_bilinear_f32 (din float32, zeroed by the caller; atomics). */
int ull_mask_loss_sums_bwd_f32(const void* logits, const void* target, const void* g, int64_t n_masks, int64_t hw, float scale, void* dlogits,
                               void* stream);
int ull_box_losses_bwd_f32(const void* pred, int pred_dtype, const void* gt, int64_t n, const void* gw, void* dpred, void* stream);
int ull_bilinear_bwd_f32(const void* dout, void* din, int64_t in_img_stride, int64_t in_row_stride, int64_t in_h, int64_t in_w, int64_t n,
                         int64_t out_h, int64_t out_w, void* stream);

/* AdamW on a shard of the flattened parameters (the optimizer transformers.Trainer builds for train_ullava.py:273-293, under ZeRO stage 2
 * with bf16: configs/deepspeed/bf16_zero2.json): fp32 master / first / second moments updated in place, torch's single-tensor AdamW
 * arithmetic in fp32 (decoupled weight decay, bias correction with `step` = the count after this update).  grad (dtype code ULL_DT_*) is
 * multiplied by grad_scale first; param_out (dtype code) receives the updated parameters, or NULL. */
int ull_adamw_step_f32(void* master, void* m, void* v, const void* grad, int grad_dtype, void* param_out, int param_dtype, int64_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream);

/* out[0] (fp32 on the device, caller-zeroed) += sum of squares of n gradient elements: the local part of clip_grad_norm_'s total norm. */
int ull_sumsq_f32(const void* g, int grad_dtype, int64_t n, void* out, void* stream);

/* ---- coarse entries (round 6): ONE call enqueues a whole stack of layers on the stream --------------------------------------------------
 * Host-side composition of the per-op entries above (u-llava_amd/csrc/layers.hip holds no kernel): the same launches, the same dispatch rules
 * as the per-op path, bit-identical results; what they remove is a host round trip per launch (8 Python ranks share one host).
 * A Linear is described once: */
typedef struct ull_linear {
    const void* w;       /* [n, k] row-major, row pitch ldw (what the GEMV / skinny kernels and the 128 x 128 GEMM read) */
    const void* w_tiled; /* its ULL_EPI_W_TILED copy for the 256 x 256 kernel, or NULL */
    const void* bias;    /* [n] or NULL */
    int64_t n, k, ldw;
} ull_linear;
typedef struct ull_llama_layer { /* hf LlamaDecoderLayer: input_layernorm, q|k|v (one [3D, D] matrix), o_proj, post_attention_layernorm, gate|up */
    const void* ln1;             /* (interleaved in 16-row groups, ULL_EPI_SWIGLU), down_proj */
    const void* ln2;
    ull_linear qkv, o, gu, down;
} ull_llama_layer;
typedef struct ull_clip_layer { /* hf CLIPEncoderLayer: layer_norm1, q|k|v (one [3D, D] matrix + bias), out_proj, layer_norm2, fc1 (quick_gelu), fc2 */
    const void *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    ull_linear qkv, out, fc1, fc2;
} ull_clip_layer;
typedef struct ull_sam_block { /* image_encoder.py:128-193 Block: norm1, attn.qkv, attn.proj, norm2, mlp.lin1 (GELU), mlp.lin2 */
    const void *n1_w, *n1_b, *n2_w, *n2_b;
    ull_linear qkv, proj, lin1, lin2;
    const void* rel_pos_h; /* [2 * side - 1, hd] with side = window (14) or the grid (64): already resized to that length */
    const void* rel_pos_w;
    int64_t window;        /* 14, or 0 = global attention */
} ull_sam_block;

/* hf LlamaModel.forward's layer loop at prefill shapes (modeling_llama.py:347-419 via models/ullava_core.py:312-322): per layer RMSNorm ->
 * q|k|v GEMM with RoPE epilogue -> causal attention on the fused q|k|v buffer -> o_proj + residual -> RMSNorm -> gate|up GEMM + SwiGLU ->
 * down_proj + residual.  x_in [T = B * S, D] is read only; layer l's output goes to x_out[l] ([T, D]; pointers may repeat, an entry must not
 * alias x_in while x_in is still the layer's input, i.e. for l = 0); x_mid [T, D], xn [T, D], qkv [T, 3D], att [T, D], act [T, I] are scratch.
 * rope_cos / rope_sin [T, 64] from ull_rope_table_bf16; key_mask int32 [B, S] or NULL.  hd = 128, 16 < S <= 1024, D % 64 == 0 (anything else:
 * ULL_ERR_SHAPE -- use the per-op entries).  ws / ws_bytes / sk_min_k: the stream-K policy of ull_gemm_bf16 (split where K >= sk_min_k;
 * sk_min_k < 0 = never). */
int ull_llama_prefill_layers_bf16(const ull_llama_layer* layers, int64_t n_layers, const void* x_in, void* const* x_out, void* x_mid, void* xn,
                                  void* qkv, void* att, void* act, const void* rope_cos, const void* rope_sin, const void* key_mask, int64_t B,
                                  int64_t S, int64_t H, int64_t hd, int64_t I, float eps, void* ws, int64_t ws_bytes, int64_t sk_min_k,
                                  const void* zeros, void* stream);

/* The same loop for generation steps (T = B * S <= 4 new tokens on a filled KV cache; models/ullava_core.py:357-395): per layer
 * ull_gemv_qkv_rope_append_bf16 -> split-key attention over the cache -> o_proj -> RMSNorm + gate|up + SwiGLU -> down_proj, the GEMV / skinny-GEMM
 * choice per Linear as in the per-op path.  k_cache[l] [B, H, smax, hd], vt_cache[l] [B, H, hd, smax] (permuted, see ull_transpose_v_bf16);
 * q [T, D], att [T, D], xn [T, max(D, I)], act [T, I], x_mid [T, D] scratch; rope tables [T, hd / 2] of the step's positions. */
int ull_llama_decode_layers_bf16(const ull_llama_layer* layers, int64_t n_layers, const void* x_in, void* const* x_out, void* x_mid, void* xn,
                                 void* q, void* att, void* act, const void* rope_cos, const void* rope_sin, const void* key_mask,
                                 void* const* k_cache, void* const* vt_cache, int64_t B, int64_t S, int64_t H, int64_t hd, int64_t I, int64_t smax,
                                 int64_t past, float eps, const void* zeros, void* stream);

/* hf CLIPEncoder.forward's layer loop (modeling_clip.py:353-384; models/ullava_core.py:146-158 reads hidden_states[-2], so the caller passes
 * the first 23 layers): h [T = n_img * S, D] is updated in place; h_mid [T, D], y [T, D], qkv [T, 3D], att [T, D], f [T, I] scratch.
 * hd = 64, 16 < S <= 704. */
int ull_clip_layers_bf16(const ull_clip_layer* layers, int64_t n_layers, void* h, void* h_mid, void* y, void* qkv, void* att, void* f, int64_t n_img,
                         int64_t S, int64_t H, int64_t hd, int64_t I, float eps, void* ws, int64_t ws_bytes, int64_t sk_min_k, const void* zeros,
                         void* stream);

/* ImageEncoderViT.forward's block loop (image_encoder.py:110-116; Block.forward :165-193) on image-order tokens x [B * g * g, C], in place:
 * 14 x 14 window blocks through ull_sam_window_attention_bf16, global blocks through ull_attention_bf16 (rel_mode 2).  g = 64, hd = 80. */
int ull_sam_blocks_bf16(const ull_sam_block* blocks, int64_t n_blocks, void* x, void* x_mid, void* y, void* qkv, void* att, void* f, int64_t B,
                        int64_t g, int64_t nH, int64_t hd, int64_t I, float eps, void* ws, int64_t ws_bytes, int64_t sk_min_k, const void* zeros,
                        void* stream);

/* ==== fp32 build of the inference path (u-llava_amd/csrc/f32.hip): `--dtype fp32` of inference_ullava.py:25,164-168 ====
 * float32 tensors everywhere, no intermediate roundings (the reference's fp32 graph has none); same arguments, layouts and flags as the *_bf16
 * function of the same name, with these differences: ull_gemm_f32 accepts any M / N / K / strides, ignores ws and refuses the tile-major
 * flags; ull_attention_f32 takes V either as rows (vt_len = 0) or as the ull_transpose_v image for ANY shape, head_dim <= 128, rel_mode 0 / 1
 * only.  Contractions run on v_mfma_f32_16x16x4_f32 (IEEE fp32 products and sums).  Correctness path: no fused / tiled fast path has an
 * fp32 twin (qkv+RoPE epilogue, patchify, window attention, fused mask decoder, coarse layer stacks), and pure data movement (embed_splice,
 * gather_rows, window_partition) is served by the 16-bit entries on the same bytes viewed as rows of twice as many 16-bit elements. */
int ull_gemm_f32(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, int flags, void* ws, int64_t ws_bytes, void* stream);
int ull_attention_f32(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_ss, const void* K, int64_t k_bs, int64_t k_hs, int64_t k_ss, const void* Vt, int64_t vt_bs, int64_t vt_hs, int64_t vt_ds, int64_t vt_len, void* O, int64_t o_bs, int64_t o_hs, int64_t o_ss, const void* key_mask, int64_t B, int64_t H, int64_t Sq, int64_t Sk, int64_t hd, int causal, int scale_mode, float scale, float q_scale, const void* rel_h, const void* rel_w, int64_t rel_kh, int64_t rel_kw, int rel_mode, const void* zeros, void* stream);
int ull_transpose_v_f32(const void* v, int64_t v_bs, int64_t v_ss, void* vt, int64_t B, int64_t S, int64_t H, int64_t hd, int64_t pitch, void* stream);
int ull_rmsnorm_f32(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, int64_t rows, int64_t D, float eps, void* stream);
int ull_layernorm_f32(const void* x, int64_t ldx, const void* w, const void* b, void* y, int64_t ldy, int64_t rows, int64_t D, float eps, void* stream);
int ull_clip_embed_ln_f32(const void* patch, int64_t ldp, const void* cls, const void* pos, const void* w, const void* b, void* y, int64_t ldy, int64_t n_img, int64_t tokens, int64_t D, float eps, void* stream);
int ull_layernorm2d_cl_f32(const void* x, const void* w, const void* b, void* y, int64_t rows, int64_t C, float eps, int gelu, void* stream);
int ull_rope_inplace_f32(void* x, int64_t row_stride, const void* positions, const void* inv_freq, int64_t tokens, int64_t n_heads, int64_t hd, void* stream);
int ull_rope_append_f32(void* qkv, int64_t row_stride, const void* positions, const void* inv_freq, int64_t B, int64_t S, int64_t H, int64_t hd, void* k_cache, void* vt_cache, int64_t smax, int64_t past, void* stream);
int ull_im2col_f32(const void* img, void* out, int64_t n_img, int64_t C, int64_t H, int64_t W, int64_t ps, int64_t Kp, void* stream);
int ull_im2col3x3_f32(const void* x, void* out, int64_t B, int64_t H, int64_t W, int64_t C, void* stream);
int ull_video_pool_f32(const void* f, void* out, int64_t B, int64_t T, int64_t N, int64_t D, int64_t tok_pitch, int64_t tok_off, void* stream);
int ull_add_rows_f32(const void* a, const void* b, void* out, int64_t rows, int64_t D, int64_t b_rows, void* stream);
int ull_window_unpartition_add_f32(const void* win, const void* shortcut, void* out, int64_t B, int64_t H, int64_t W, int64_t C, int64_t ws, void* stream);
int ull_sam_relpos_f32(const void* q, int64_t q_bs, int64_t q_hs, int64_t q_ss, const void* rel_pos_h, const void* rel_pos_w, void* out_h, void* out_w, int64_t NB, int64_t nH, int64_t KH, int64_t KW, int64_t hd, void* stream);
int ull_interp_rows_linear_f32(const void* x, void* y, int64_t L, int64_t M, int64_t C, void* stream);
int ull_mask_matmul_f32(const void* hyper, const void* up, void* masks, int64_t n, int64_t T, int64_t C, int64_t G, void* stream);
int ull_greedy_step_f32(const void* logits, int64_t row_stride, int64_t B, int64_t V, void* unfinished, const void* eos, int64_t n_eos, int64_t pad, int has_pad, void* seq, int64_t seq_ld, int64_t pos, void* alive, void* stream);
int ull_shifted_cross_entropy_f32(const void* logits, int64_t ld, const void* labels, int64_t B, int64_t S, int64_t V, void* out, void* stream);
/* ==== END fp32 build ==== */

/* ==== BEGIN fp16 twins (generated by tools/gen_header_f16.py) ==== */
/* IEEE binary16 build of every dtype-dependent entry point: same arguments, layouts, flags and rounding points as the *_bf16
 * function of the same name; every 16-bit element is an fp16 instead of a bf16 (the reference's `--dtype fp16`,
 * inference_ullava.py:26,164-168).  Both builds live in the same library; the host picks by tensor dtype. */
int ull_gemm_f16(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, int flags, void* ws, int64_t ws_bytes, void* stream);
int ull_gemm_qkv_rope_f16(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, const void* rope_cos, const void* rope_sin, int64_t rope_cols, int64_t head_dim, int flags, void* ws, int64_t ws_bytes, void* stream);
int ull_rope_table_f16(const void* positions, const void* inv_freq, int64_t tokens, int64_t half, void* cos_out, void* sin_out, void* stream);
int ull_gemv_f16(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, int flags, void* stream);
int ull_gemm_skinny_f16(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, int flags, void* stream);
int ull_gemv_rmsnorm_f16(const void* X, int64_t ldx, const void* norm_w, float eps, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, int flags, void* stream);
int ull_gemv_qkv_rope_append_f16(const void* X, int64_t ldx, const void* norm_w, float eps, const void* W, int64_t ldw, void* Q_out, int64_t ldq, const void* cos_tab, const void* sin_tab, void* k_cache, void* vt_cache, int64_t B, int64_t S, int64_t H, int64_t hd, int64_t K, int64_t smax, int64_t past, void* stream);
int ull_rmsnorm_f16(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, int64_t rows, int64_t D, float eps, void* stream);
int ull_shifted_cross_entropy_f16(const void* logits, int64_t ld, const void* labels, int64_t B, int64_t S, int64_t V, void* out, void* stream);
int ull_layernorm_f16(const void* x, int64_t ldx, const void* w, const void* b, void* y, int64_t ldy, int64_t rows, int64_t D, float eps, void* stream);
int ull_clip_embed_ln_f16(const void* patch, int64_t ldp, const void* cls, const void* pos, const void* w, const void* b, void* y, int64_t ldy, int64_t n_img, int64_t tokens, int64_t D, float eps, void* stream);
int ull_attention_f16(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_ss, const void* K, int64_t k_bs, int64_t k_hs, int64_t k_ss, const void* Vt, int64_t vt_bs, int64_t vt_hs, int64_t vt_ds, int64_t vt_len, void* O, int64_t o_bs, int64_t o_hs, int64_t o_ss, const void* key_mask, int64_t B, int64_t H, int64_t Sq, int64_t Sk, int64_t hd, int causal, int scale_mode, float scale, float q_scale, const void* rel_h, const void* rel_w, int64_t rel_kh, int64_t rel_kw, int rel_mode, const void* zeros, void* stream);
int ull_rope_inplace_f16(void* x, int64_t row_stride, const void* positions, const void* inv_freq, int64_t tokens, int64_t n_heads, int64_t hd, void* stream);
int ull_rope_append_f16(void* qkv, int64_t row_stride, const void* positions, const void* inv_freq, int64_t B, int64_t S, int64_t H, int64_t hd, void* k_cache, void* vt_cache, int64_t smax, int64_t past, void* stream);
int ull_transpose_v_f16(const void* v, int64_t v_bs, int64_t v_ss, void* vt, int64_t B, int64_t S, int64_t H, int64_t hd, int64_t pitch, void* stream);
int ull_im2col_f16(const void* img, void* out, int64_t n_img, int64_t C, int64_t H, int64_t W, int64_t ps, int64_t Kp, void* stream);
int ull_greedy_step_f16(const void* logits, int64_t row_stride, int64_t B, int64_t V, void* unfinished, const void* eos, int64_t n_eos, int64_t pad, int has_pad, void* seq, int64_t seq_ld, int64_t pos, void* alive, void* stream);
int ull_embed_splice_f16(const void* ids, const void* table, const void* img_feat, int64_t n_img_tok, int64_t img_pitch, int64_t img_off, const void* vid_feat, int64_t n_vid_tok, const void* spans, void* out, int64_t B, int64_t S, int64_t D, int64_t vocab, void* stream);
int ull_video_pool_f16(const void* f, void* out, int64_t B, int64_t T, int64_t N, int64_t D, int64_t tok_pitch, int64_t tok_off, void* stream);
int ull_gather_rows_f16(const void* src, int64_t lds, const void* idx, void* dst, int64_t ldd, int64_t n, int64_t D, void* stream);
int ull_dropout_apply_f16(const void* x, const void* keep, void* y, int64_t n, float scale, void* stream);
int ull_add_rows_f16(const void* a, const void* b, void* out, int64_t rows, int64_t D, int64_t b_rows, void* stream);
int ull_window_partition_f16(const void* x, void* out, int64_t B, int64_t H, int64_t W, int64_t C, int64_t ws, void* stream);
int ull_window_unpartition_add_f16(const void* win, const void* shortcut, void* out, int64_t B, int64_t H, int64_t W, int64_t C, int64_t ws, void* stream);
int ull_sam_relpos_f16(const void* q, int64_t q_bs, int64_t q_hs, int64_t q_ss, const void* rel_pos_h, const void* rel_pos_w, void* out_h, void* out_w, int64_t NB, int64_t nH, int64_t KH, int64_t KW, int64_t hd, void* stream);
int ull_sam_window_attention_f16(const void* qkv, int64_t ld, const void* pad_row, const void* rel_pos_h, const void* rel_pos_w, void* out, int64_t ldo, int64_t B, int64_t H, int64_t W, int64_t nH, int64_t hd, int64_t ws, float q_scale, const void* zeros, void* stream);
int ull_interp_rows_linear_f16(const void* x, void* y, int64_t L, int64_t M, int64_t C, void* stream);
int ull_layernorm2d_cl_f16(const void* x, const void* w, const void* b, void* y, int64_t rows, int64_t C, float eps, int gelu, void* stream);
int ull_im2col3x3_f16(const void* x, void* out, int64_t B, int64_t H, int64_t W, int64_t C, void* stream);
int ull_mask_matmul_f16(const void* hyper, const void* up, void* masks, int64_t n, int64_t T, int64_t C, int64_t G, void* stream);
int ull_patchify_f16(const void* img, int64_t n_img, int64_t C, int64_t H, int64_t W, int64_t ps, const void* Wp, int64_t Kp, const void* bias, void* out, int64_t ldc, int64_t N, const void* zeros, void* stream);
int ull_sam_self_attn_heads_f16(const void* queries, const void* qpe, int64_t n, int64_t T, int first, const void* wq, const void* bq, const void* wk, const void* bk, const void* wv, const void* bv, void* att, void* stream);
int ull_sam_out_ln_f16(const void* att, int64_t din, const void* res, const void* qpe, int64_t n, int64_t T, const void* wo, const void* bo, const void* ln_w, const void* ln_b, float eps, void* out, const void* const* projs, const int* add_pe, void* stream);
int ull_sam_token_mlp_ln_f16(const void* queries, const void* qpe, int64_t n, int64_t T, int64_t hidden, const void* w1, const void* b1, const void* w2, const void* b2, const void* ln_w, const void* ln_b, float eps, void* part_ws, void* out, const void* const* projs, const int* add_pe, void* stream);
int ull_sam_small_mlps_f16(const void* hs, int64_t n, int64_t T, const void* const* ptrs, int64_t n_mask_tokens, int64_t hyper_out, int64_t n_iou, void* hyper, void* iou, void* stream);
int ull_sam_t2i_attention_f16(const void* qproj, const void* keys, const void* pos, int64_t n, int64_t T, int64_t P, const void* wk, const void* bk, const void* wv, const void* bv, int late_bias_kv, void* scores_ws, void* vproj_ws, void* att, void* stream);
int ull_sam_i2t_attention_ln_f16(const void* keys, const void* pos, const void* kproj, const void* vproj, int64_t n, int64_t T, int64_t P, const void* wq, const void* bq, const void* wo, const void* bo, int late_bias_q, const void* ln_w, const void* ln_b, float eps, void* out, void* stream);
int ull_rmsnorm_bwd_f16(const void* x, int64_t ldx, const void* w, const void* dy, int64_t lddy, void* dx, int64_t lddx, void* dw, int64_t rows, int64_t D, float eps, void* stream);
int ull_swiglu_fwd_f16(const void* gu, void* a, int64_t M, int64_t I, int halves, void* stream);
int ull_swiglu_bwd_f16(const void* gu, const void* da, void* dgu, int64_t M, int64_t I, int halves, void* stream);
int ull_relu_mask_f16(const void* y, const void* dy, void* dx, int64_t n, void* stream);
int ull_rope_bwd_inplace_f16(void* dx, int64_t row_stride, const void* positions, const void* inv_freq, int64_t tokens, int64_t n_heads, int64_t hd, void* stream);
int ull_attention_bwd_f16(const void* Q, const void* K, const void* V, const void* O, const void* dO, void* dQ, void* dK, void* dV, const int64_t* strides, const void* key_mask, int64_t B, int64_t H, int64_t Sq, int64_t Sk, int64_t hd, int causal, float mult, void* scratch, void* stream);
int ull_attention_bwd_mfma_f16(const void* Q, const void* K, const void* V, const void* O, const void* dO, const void* Qt, const void* Kt, const void* dOt, int64_t pitch, void* dQ, void* dK, void* dV, const int64_t* strides, const void* key_mask, int64_t B, int64_t H, int64_t Sq, int64_t Sk, int64_t hd, int causal, float mult, void* scratch, void* stream);
int ull_shifted_cross_entropy_bwd_f16(const void* logits, int64_t ld, const void* labels, int64_t B, int64_t S, int64_t V, const void* stats, const void* gout, void* dlogits, void* stream);
int ull_embed_splice_bwd_f16(const void* ids, const void* demb, void* d_table, void* d_img, int64_t n_img_tok, int64_t img_pitch, int64_t img_off, void* d_vid, int64_t n_vid_tok, const void* spans, int64_t B, int64_t S, int64_t D, int64_t vocab, int detach_text, void* stream);
int ull_layernorm_bwd_f16(const void* x, int64_t ldx, const void* w, const void* dy, int64_t lddy, void* dx, int64_t lddx, void* dw, void* db, int64_t rows, int64_t D, float eps, void* stream);
int ull_layernorm2d_cl_bwd_f16(const void* x, const void* w, const void* b, const void* dy, void* dx, void* dw, void* db, int64_t rows, int64_t C, float eps, int gelu, void* stream);
int ull_gelu_fwd_f16(const void* x, void* y, int64_t n, void* stream);
int ull_gelu_bwd_f16(const void* x, const void* dy, void* dx, int64_t n, void* stream);
int ull_mask_matmul_bwd_f16(const void* hyper, const void* up, const void* dmasks, void* dhyper, void* dup, int64_t n, int64_t T, int64_t C, int64_t G, void* stream);
int ull_sum_slabs_f16(const void* x, void* out, int64_t R, int64_t n, float scale, void* stream);
int ull_colsum_f16(const void* x, int64_t ld, int64_t rows, int64_t N, void* out, void* stream);
int ull_transpose2d_f16(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t R, int64_t C, void* stream);
int ull_llama_prefill_layers_f16(const ull_llama_layer* layers, int64_t n_layers, const void* x_in, void* const* x_out, void* x_mid, void* xn, void* qkv, void* att, void* act, const void* rope_cos, const void* rope_sin, const void* key_mask, int64_t B, int64_t S, int64_t H, int64_t hd, int64_t I, float eps, void* ws, int64_t ws_bytes, int64_t sk_min_k, const void* zeros, void* stream);
int ull_llama_decode_layers_f16(const ull_llama_layer* layers, int64_t n_layers, const void* x_in, void* const* x_out, void* x_mid, void* xn, void* q, void* att, void* act, const void* rope_cos, const void* rope_sin, const void* key_mask, void* const* k_cache, void* const* vt_cache, int64_t B, int64_t S, int64_t H, int64_t hd, int64_t I, int64_t smax, int64_t past, float eps, const void* zeros, void* stream);
int ull_clip_layers_f16(const ull_clip_layer* layers, int64_t n_layers, void* h, void* h_mid, void* y, void* qkv, void* att, void* f, int64_t n_img, int64_t S, int64_t H, int64_t hd, int64_t I, float eps, void* ws, int64_t ws_bytes, int64_t sk_min_k, const void* zeros, void* stream);
int ull_sam_blocks_f16(const ull_sam_block* blocks, int64_t n_blocks, void* x, void* x_mid, void* y, void* qkv, void* att, void* f, int64_t B, int64_t g, int64_t nH, int64_t hd, int64_t I, float eps, void* ws, int64_t ws_bytes, int64_t sk_min_k, const void* zeros, void* stream);
/* ==== END fp16 twins ==== */

#ifdef __cplusplus
}
#endif
#endif /* ULLAVA_HIP_H */
