/* muse_hip.h -- C ABI of libmuse_hip.so: the MI355X (gfx950) hot path of lucidrains/muse-maskgit-pytorch.
 *
 * The reference has no FFI / plugin seam of its own (it is pure PyTorch); the drop-in boundary is its Python
 * class surface (muse_maskgit_pytorch/__init__.py:1-4) and, one level down, the Attend(q, k, v, mask) operator
 * (attend.py:109).  Each entry point below names the reference code it replaces (file:line relative to
 * /root/reference/muse_maskgit_pytorch/).  The reference-side binding is a ctypes stub: see INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++ / torch types.  Every function returns int: MM_OK (0) or a
 *     negative MM_ERR_* code; mm_last_error() returns the calling thread's last message.  Nothing throws or
 *     aborts across the ABI.
 *   - All tensor pointers are CALLER-OWNED DEVICE pointers (the host framework allocates; this library never
 *     frees them).  Scratch memory is a caller-provided workspace sized by the matching *_workspace_bytes().
 *   - Every launch goes to the explicit `stream` (a hipStream_t); no hidden synchronisation, no default-stream
 *     use.  Handles are immutable after create: share them across threads, one in-flight call per
 *     (handle, workspace).
 *   - dtypes: token ids / mask indices int64 (torch.long, mmp.py:519); scores / logits fp32; activations and
 *     Linear / conv / embedding weights bf16 (raw uint16 bits); norm gains, biases, scales, null_kv fp32;
 *     masks uint8 (1 = keep).
 */
#ifndef MUSE_HIP_H
#define MUSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_ABI_VERSION 9

#define MM_OK 0
#define MM_ERR_SHAPE (-1)
#define MM_ERR_DTYPE (-2)
#define MM_ERR_ALIGN (-3)
#define MM_ERR_ARCH (-4)
#define MM_ERR_HIP (-5)
#define MM_ERR_WORKSPACE (-6)
#define MM_ERR_UNSUPPORTED (-7)

#define MM_NOISE_NONE 0     /* plain argmax of logits / temperature                                        */
#define MM_NOISE_GUMBEL 1   /* caller supplies -log(-log(u)) per (token, vocab) -- bit-exact parity mode     */
#define MM_NOISE_UNIFORM 2  /* caller supplies u; the kernel applies mmp.py:403-408                          */
#define MM_NOISE_PHILOX 3   /* on-device Philox2x32-10, counter = (global token row, step, vocab/2), key = seed */

typedef void* mm_stream_t;  /* hipStream_t */

int mm_abi_version(void);
const char* mm_last_error(void);
/* MM_OK iff the current HIP device is gfx950; MM_ERR_ARCH otherwise (product path refuses to run elsewhere). */
int mm_device_check(void);

/* ------------------------------------------------------------------------------------------------ operators */

/* nn.Linear(bias=False): out[m][n] = sum_k x[m][k] * w[n][k]            (mmp.py:85,88,118-124,225,233)
 * x bf16 [M][ldx], w bf16 [N][ldw], K % 64 == 0 (zero-pad), ldx/ldw % 8 == 0.
 * out_f32 != 0: out is fp32 [M][ldc] else bf16.  resid_f32 (optional, fp32 [M][ldc], may alias out when
 * out_f32) is added in the epilogue: the residual adds of TransformerBlocks.forward (mmp.py:189-193). */
int mm_gemm_bf16(mm_stream_t stream, const void* x, int64_t ldx, const void* w, int64_t ldw, int M, int N, int K,
                 void* out, int64_t ldc, int out_f32, const float* resid_f32);

/* to_logits + classifier-free guidance in one pass:
 *   out[m][n] = null + (cond - null) * cond_scale,  cond = x_cond[m].w[n], null = x_null[m].w[n]
 * (mmp.py:250-254 + 332).  out fp32 [M][ldc]. */
int mm_gemm_cfg_logits(mm_stream_t stream, const void* x_cond, const void* x_null, int64_t ldx, const void* w,
                       int64_t ldw, int M, int N, int K, float* out, int64_t ldc, float cond_scale);

/* x[row] = token_emb[ids[row]] + pos_emb[row % n]  (mmp.py:322-323); tables bf16, x fp32 [rows][dim]. */
int mm_embed(mm_stream_t stream, const int64_t* ids, int rows, int n, const void* token_emb, int vocab_rows,
             const void* pos_emb, int dim, float* x);

/* LayerNorm (mmp.py:63-70): fp32 in, bf16 out, eps 1e-5; beta may be NULL (it is a zero buffer in the reference).
 * row_index (optional int32 [rows]) gathers source rows. */
int mm_layernorm(mm_stream_t stream, const float* x, int64_t ldx, int rows, int dim, const float* gamma,
                 const float* beta, const int32_t* row_index, void* out, int64_t ldo);

/* GEGLU + LayerNorm(inner) (mmp.py:72-77, 86-87): h bf16 [rows][2*Fp] = [x half | gate half]; out bf16 [rows][Fp]
 * with columns >= F zero.  gamma / beta: fp32 [Fp] (zero-padded past F; read with 16-byte loads). */
int mm_geglu_ln(mm_stream_t stream, const void* h, int64_t ldh, int rows, int F, int Fp, const float* gamma,
                const float* beta, void* out, int64_t ldo);

/* Linear(D, 2F) + GEGLU in one pass (mmp.py:85 + :72-77): w1 GEGLU-interleaved as in mm_ff_weights; out bf16 [M][Fp] =
 * gate * gelu_erf(x), columns >= F come out as zero.  Then LayerNorm(inner) (mmp.py:87) on that: mm_layernorm_inner. */
int mm_gemm_geglu(mm_stream_t stream, const void* x, int64_t ldx, const void* w1, int64_t ldw, int M, int Fp, int K,
                  void* out, int64_t ldc);
int mm_layernorm_inner(mm_stream_t stream, const void* a, int64_t lda, int rows, int F, int Fp, const float* gamma,
                       const float* beta, void* out, int64_t ldo);

/* The Attend seam (attend.py:109-140): softmax(scale * q k^T, key mask) v.  dim_head = 32, 64 or 128 (last argument).
 * Element strides (batch, head, token) per operand, d contiguous.  nk = number of keys.
 * normalize != 0 additionally fuses mmp.py:145-153: q,k are L2-normalised and scaled by q_scale/k_scale [dim_head]
 * in-kernel and a learned null key/value (null_k/null_v fp32 [heads][dim_head], raw parameters) is prepended.
 * key_mask: optional uint8 [B][nk] (1 = keep), row stride km_sb. */
int mm_attend(mm_stream_t stream, const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const void* k,
              int64_t k_sb, int64_t k_sh, int64_t k_sn, const void* v, int64_t v_sb, int64_t v_sh, int64_t v_sn,
              void* out, int64_t o_sb, int64_t o_sh, int64_t o_sn, int B, int H, int nq, int nk,
              const uint8_t* key_mask, int64_t km_sb, int normalize, const float* q_scale, const float* k_scale,
              const float* null_k, const float* null_v, float scale, int dim_head);

/* Re-mask step (mmp.py:558-563): per sample the k highest scores (ties: lower index) get ids = mask_id; every
 * other slot's score becomes -1e5 (what mmp.py:609 leaves there).  rows_out (optional int32 [B*k]) receives the
 * flat positions b*n + pos of the masked tokens, ascending per sample. */
int mm_mask_step(mm_stream_t stream, float* scores, int64_t* ids, int B, int n, int k, int64_t mask_id,
                 int32_t* rows_out);

/* top-k filter + Gumbel argmax + confidence (mmp.py:576-580, 603-606, 403-418) on R rows of CFG-combined fp32
 * logits [R][ld].  rows (optional) gives each row's flat token position (default: r).  Results are scattered:
 * ids[pos] = pred, scores[pos] = 1 - softmax(logits)[pred]; pred_out / score_out (optional) are compact [R].
 * temperature must already be max(T, 1e-10) (mmp.py:411).  Noise (GUMBEL/UNIFORM) is indexed by flat token
 * position: noise[pos * noise_ld + v]. */
int mm_sample_rows(mm_stream_t stream, const float* logits, int64_t ld, int R, int V, int k_keep,
                   const int32_t* rows, float temperature, int noise_kind, const float* noise, int64_t noise_ld,
                   uint64_t seed, uint64_t row_offset, uint32_t step, int64_t* ids, float* scores,
                   int64_t* pred_out, float* score_out);

/* ---- sampling without the logits round trip (csrc/sampling_fused.hip; mmp.py:576-609, SURVEY.md 8d "fused floor").
 * The guidance-logits GEMM emits, per token row and 256-column piece, a record {max, sum exp, 128-bit mask of the kept granules} and the kept GRANULES
 * (two adjacent columns whose larger value reaches thr_lo[row]) compacted in column order, instead of the logits; mm_fused_sample finishes the row (exact
 * k-th largest, Gumbel argmax, confidence) from those.  Buffers, caller-owned: stats 32 B x [R][V/256]; cand 16 B x [R][V/256][MM_FUSED_SLOT] (= 128 float2
 * entries per piece); fail_flag int32 [1] (zero it; set to 1 if some row's candidates could not be proven to contain its kept set -- then repeat on the
 * logits path, mm_gemm_cfg_logits + mm_sample_rows).  V % 256 == 0.
 *   mm_fused_threshold : thr_lo[r] = mean_r + z sigma_r of row r's logits over the vocabulary, from the row's embeddings (bf16 cond / null,
 *                        combined with cond_scale) and the vocabulary statistics of to_logits (wmean fp32 [D], wcov bf16 [D][D], D % 64 == 0);
 *                        z = mm_fused_z(k_keep, V, margin) (normal quantile of the kept fraction minus a safety margin in sigmas);
 *                        ws: mm_fused_threshold_workspace_bytes(R, D) bytes of scratch
 *   mm_gemm_cfg_logits_fused : mm_gemm_cfg_logits whose epilogue emits stats / cand instead of writing the logits; x_null == NULL: x_cond holds
 *                        the already mixed embeddings (mm_cfg_mix) and the product is the single pass mm_generate runs (cond_scale ignored)
 *   mm_fused_emit      : the same emission from materialised logits (tests; shapes the 256-column GEMM does not take)
 *   mm_fused_sample    : the finishing kernel; remaining arguments as in mm_sample_rows.  A row whose candidates cannot be proven complete is
 *                        appended to fail_rows[atomicAdd(fail_count, 1)] (device int32 [fail_cap] / [1], optional) for the caller to finish on
 *                        the logits path (mm_generate does, on the device); without a list, or when it is full, *fail_flag is set to 1 */
#define MM_FUSED_SLOT 64
float mm_fused_z(int k_keep, int V, float margin);
/* Distribution-free form of the bound (round 5): thr[r] = the mm_fused_quantile_rank(k_keep, V, S)-th largest of the S sampled logits sub[r][0 .. S) of row r
 * (S <= 4096 vocabulary columns drawn once per model, mm_transformer_desc.logits_wsub): rank = S k / V + 4.5 standard deviations of the hypergeometric count. */
int mm_fused_quantile_rank(int k_keep, int V, int S);
int mm_fused_quantile(mm_stream_t stream, const float* sub, int64_t ld, int R, int S, int rank, float* thr);
size_t mm_fused_threshold_workspace_bytes(int R, int D);
int mm_fused_threshold(mm_stream_t stream, const void* emb_cond, const void* emb_null, int64_t ld, int R, int D, float cond_scale, const float* wmean,
                       const void* wcov, float z, void* ws, float* thr);
int mm_gemm_cfg_logits_fused(mm_stream_t stream, const void* x_cond, const void* x_null, int64_t ldx, const void* w, int64_t ldw, int M, int N, int K,
                             float cond_scale, const float* thr, void* stats, void* cand);
int mm_fused_emit(mm_stream_t stream, const float* logits, int64_t ld, int R, int V, const float* thr, void* stats, void* cand);
int mm_fused_sample(mm_stream_t stream, const float* thr, const void* stats, const void* cand, int R, int V, int k_keep, const int32_t* rows,
                    float temperature, int noise_kind, const float* noise, int64_t noise_ld, uint64_t seed, uint64_t row_offset, uint32_t step,
                    int64_t* ids, float* scores, int64_t* pred_out, float* score_out, int32_t* fail_flag, int32_t* fail_rows, int32_t* fail_count,
                    int fail_cap);

/* Training-forward losses of Transformer.forward (mmp.py:337-348), forward only:
 *   mm_ce_loss : F.cross_entropy over the vocabulary with ignore_index, mean over the non-ignored rows; logits fp32 [R][ld],
 *                labels int64 [R], row_loss_ws fp32 [R] scratch, out fp32 [1] (NaN when every row is ignored, like torch).
 *   mm_bce_loss: F.binary_cross_entropy_with_logits(x, y), mean; x, y fp32 [n] (TokenCritic, dim_out == 1). */
int mm_ce_loss(mm_stream_t stream, const float* logits, int64_t ld, int R, int V, const int64_t* labels, int64_t ignore_index,
               float* row_loss_ws, float* out);
int mm_bce_loss(mm_stream_t stream, const float* x, const float* y, int n, float* out);

/* ---- fp8 engine (BASELINE configs[4]: "fp8 MFMA weights").
 * mm_quantize_e4m3_rows: w fp32 [rows][ldw] -> wq OCP-e4m3 bytes [rows][Kp] (columns K..Kp-1 zero; Kp % 128 == 0 for mm_gemm_fp8) with one
 * scale per row, scale = max|w_row| / 448 (1 for an all-zero row), wq = rne(w / scale). */
int mm_quantize_e4m3_rows(mm_stream_t stream, const float* w, int64_t ldw, int rows, int K, int Kp, void* wq, float* scale);
/* fp8 engine (BASELINE configs[4] "fp8 MFMA weights"; the Linear layers mmp.py:85,88,118-124,233): e4m3 x e4m3 products on the K = 128 fp8 MFMA
 * (v_mfma_f32_16x16x128_f8f6f4), fp32 accumulation, per-row scales on both operands applied to the accumulators:
 *     out[m][n] = x_scale[m] * w_scale[n] * sum_k xq[m][k] * wq[n][k]
 * mm_quantize_act_e4m3: activation rows (bf16, or fp32 when x_is_f32) -> e4m3 [rows][Kp] + scale[rows] (max |x| / 448, the weights' rule of
 *     mm_quantize_e4m3_rows); Kp % 128 == 0 for the GEMM (columns K..Kp-1 are written as zero).
 * mm_gemm_fp8: K % 128 == 0 (padded), N % 128 == 0, ldx / ldw in bytes (= elements).  epilogue 0: out bf16 [M][ldc]; 1: GEGLU on w1 packed
 *     value/gate-interleaved in 64-row blocks (mm_ff_weights.w1), out bf16 [M][N / 2]; 2: out fp32 [M][ldc] = resid_f32 (same layout, may alias
 *     out, may be NULL) + product.  Self-defined numerics: the oracle is the fp32 restatement on the de-quantised operands. */
int mm_quantize_act_e4m3(mm_stream_t stream, const void* x, int x_is_f32, int64_t ldx, int rows, int K, int Kp, void* xq, float* scale);
int mm_gemm_fp8(mm_stream_t stream, const void* xq, int64_t ldx, const float* x_scale, const void* wq, int64_t ldw, const float* w_scale, int M, int N, int K,
                void* out, int64_t ldc, int epilogue, const float* resid_f32);

/* Nearest-codebook vector quantisation (the north star's "L2 nearest-codebook VQ lookup"; EXTENSION with a self-defined oracle:
 * the reference's VectorQuantize branch, vqgan_vae.py:297-303, 336-342, 433-435, cannot run).  x fp32 [N][ldx] (C used), codebook fp32
 * [K][C], C % 4 == 0, C <= 256.  ids[r] = argmin_k |x_r - e_k|^2, or argmax_k of the cosine similarity when cosine != 0
 * (vq use_cosine_sim = True, the reference's default kwargs); ties -> lower index.  aux_ws: K floats of scratch.
 * mm_vq_gather: out[r][:] = codebook[ids[r]][:]. */
int mm_vq_nearest(mm_stream_t stream, const float* x, int64_t ldx, int N, int C, const float* codebook, int K, int cosine,
                  float* aux_ws, int64_t* ids);
int mm_vq_gather(mm_stream_t stream, const int64_t* ids, int64_t N, int C, const float* codebook, float* out);

/* ---- backward operators of the transformer training step (MaskGit.forward, mmp.py:623-741, which the reference
 *      differentiates with torch autograd).  Activation gradients are bf16, the residual-stream gradient and every parameter
 *      gradient fp32.  Linear layers: dX = dY * W and dW = dY^T * X are mm_gemm_bf16 calls on transposed copies
 *      (mm_transpose_bf16).  INTEGRATION.md lists the per-layer call sequence. */

/* Weight-gradient GEMM out[M][N] = x[M][K] * w[N][K]^T (fp32) with a LONG contraction K (= the token count) and a small
 * output: the K loop is split over `splits` workgroups per tile (partial slabs in ws, splits*M*N floats), then reduced in a fixed
 * order.  mm_gemm_wgrad_splits(M, N, K) proposes a split count (1 = use mm_gemm_bf16). */
int mm_gemm_wgrad_splits(int M, int N, int K);
int mm_gemm_wgrad(mm_stream_t stream, const void* x, int64_t ldx, const void* w, int64_t ldw, int M, int N, int K, int splits,
                  float* ws, float* out);
/* The same weight gradient WITHOUT transposed copies (round 6, csrc/gemm_tn.hip): dW fp32 [N][K] = dY^T X for row-major bf16 dY [rows][ldy] (first N columns) and
 * X [rows][ldx] (first K columns) -- the operands are read as they are (LDS-DMA + transposing LDS reads), rows need not be a multiple of anything.  N and K
 * multiples of 128 (MM_ERR_UNSUPPORTED otherwise: transpose and use mm_gemm_wgrad).  ws: mm_gemm_wgrad_tn_splits(rows, N, K) x N x K floats when that is > 1
 * (split over the rows, slabs summed in a fixed order).  Deterministic; a different summation order from mm_gemm_wgrad. */
int mm_gemm_wgrad_tn_splits(int rows, int N, int K);
/* 1 where mm_train_step takes this form instead of transposed copies + mm_gemm_wgrad (the small projections and the head: measured, csrc/gemm_tn.hip); a driver that
 * wants the step's bits follows it (training.py does). */
int mm_gemm_wgrad_tn_prefer(int rows, int N, int K, int64_t ldy, int64_t ldx);
int mm_gemm_wgrad_tn(mm_stream_t stream, const void* dy, int64_t ldy, const void* x, int64_t ldx, int rows, int N, int K, float* ws, float* out);
/* out[c][r] = in[r][c]; bf16, strides in elements (multiples of 8). */
int mm_transpose_bf16(mm_stream_t stream, const void* in, int64_t rows, int64_t cols, int64_t ld_in, void* out, int64_t ld_out);
/* out[i] = bf16(x[i]). */
int mm_f32_to_bf16(mm_stream_t stream, const float* x, void* out, int64_t count);
/* out[c] = sum_p part[p][c] in index order (deterministic reduction of per-workgroup partials). */
int mm_colsum_f32(mm_stream_t stream, const float* part, int nparts, int D, float* out);

/* LayerNorm (mmp.py:63-70) backward.  x fp32 [.][ldx] (the forward input), dy bf16 [rows][lddy], gamma fp32 [D]; row_index
 * (optional int32 [rows]): dy row r belongs to x row row_index[r] (the forward normalised a gathered subset) and dx row
 * row_index[r] receives its gradient.  dx fp32: accumulate != 0 adds into it (residual stream), else overwrites those rows.
 * dgamma fp32 [D] is overwritten; ws: mm_ln_bwd_workspace_floats(rows, D) floats of scratch. */
int64_t mm_ln_bwd_workspace_floats(int rows, int D);
int mm_layernorm_bwd(mm_stream_t stream, const float* x, int64_t ldx, const void* dy, int64_t lddy, const float* gamma,
                     const int32_t* row_index, int rows, int D, float* dx, int64_t lddx, int accumulate, float* dgamma, float* ws);

/* Backward of mm_geglu_ln (GEGLU mmp.py:72-77 + the FeedForward's inner LayerNorm mmp.py:86): h bf16 [rows][ldh] = [x | gate]
 * (each Fp wide, F valid), dz bf16 [rows][lddz] gradient of the normalised output, gamma fp32 [Fp] (padded); dh bf16
 * [rows][lddh] = gradient w.r.t. h (padding columns zero), dgamma fp32 [Fp]; ws: mm_ln_bwd_workspace_floats(rows, Fp). */
int mm_geglu_ln_bwd(mm_stream_t stream, const void* h, int64_t ldh, const void* dz, int64_t lddz, const float* gamma, int rows,
                    int F, int Fp, void* dh, int64_t lddh, float* dgamma, float* ws);

/* d(mean cross-entropy over R rows)/d(logits) (mmp.py:343): dl bf16 [R][ldd] = (softmax(logits[r]) - onehot(labels[r])) * scale,
 * scale = 1 / (number of rows in the mean); every row must carry a valid label (gather the non-ignored rows first). */
int mm_ce_bwd(mm_stream_t stream, const float* logits, int64_t ld, int R, int V, const int64_t* labels, float scale, void* dl,
              int64_t ldd);

/* Backward of the TokenCritic head + loss (mmp.py:345-346, 383-386): logits x [rows] = e [rows][D] . w [D], loss = mean BCE-with-logits
 * against y [rows].  de bf16 [rows][ldde] = g * w, dw fp32 [D] = sum g * e with g = (sigmoid(x) - y) / rows;
 * ws: mm_ln_bwd_workspace_floats(rows, D). */
int mm_bce_head_bwd(mm_stream_t stream, const void* e, int64_t lde, const float* x, const float* y, const float* w, int rows, int D,
                    void* de, int64_t ldde, float* dw, float* ws);

/* Embedding backward (mmp.py:322-323): dx fp32 [B*n][D]; dpos fp32 [n][D] is overwritten; dtoken fp32 [rows of the table][D] must be zeroed by the
 * caller and receives, per id, the sum of the gradient rows carrying that id -- deterministic (one writer per id, a fixed order of additions).
 * ws == NULL: one serial chain per id and column (slow when one id owns most rows: the mask id of a training batch).  ws of
 * mm_embed_bwd_workspace_bytes(B, n, D) bytes: two levels -- per 256-row block, then over the blocks (ABI 9; a different association of the same sum). */
size_t mm_embed_bwd_workspace_bytes(int B, int n, int D);
int mm_embed_bwd(mm_stream_t stream, const int64_t* ids, int B, int n, int D, const float* dx, float* dtoken, float* dpos, void* ws, size_t ws_bytes);
/* dst[row_index[r]][:] = src[r][:] for r < R; bf16, D % 8 == 0 (gradient of a row gather; dst zeroed by the caller). */
int mm_scatter_rows_bf16(mm_stream_t stream, const void* src, const int32_t* row_index, int R, int D, void* dst);

/* out[i] = bf16(sum over p of parts[p][i]), fp32 accumulation in index order; parts bf16 [P][n] contiguous, n % 8 == 0 (the partial
 * key / value gradients of the 256-query chunks of a long sequence). */
int mm_sum_parts_bf16(mm_stream_t stream, const void* parts, int P, int64_t n, void* out);

/* Backward of mm_attend in its Muse form (normalize = 1, null key/value, scale 8; mmp.py:137-162, attend.py:109-140).
 * q/k/v: the forward inputs (raw projections), o: the forward output, dout: its gradient (all bf16, strided like mm_attend).
 * Outputs: dqn / dkn = gradients w.r.t. the NORMALISED, scaled q / k (feed mm_qk_norm_bwd), dv; dnk / dnv fp32 [B*H][64] =
 * gradients of the normalised null key / the null value per (batch, head).  nq in {64, 128, 256}; longer sequences: one call per
 * 256-query chunk (the log-sum-exp is per query, so chunks are independent), partial dkn / dv / dnk / dnv summed afterwards. */
int mm_attention_bwd(mm_stream_t stream, const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const void* k, int64_t k_sb,
                     int64_t k_sh, int64_t k_sn, const void* v, int64_t v_sb, int64_t v_sh, int64_t v_sn, const void* o,
                     int64_t o_sb, int64_t o_sh, int64_t o_sn, const void* dout, int64_t do_sb, int64_t do_sh, int64_t do_sn,
                     void* dqn, int64_t dq_sb, int64_t dq_sh, int64_t dq_sn, void* dkn, int64_t dk_sb, int64_t dk_sh, int64_t dk_sn,
                     void* dv, int64_t dv_sb, int64_t dv_sh, int64_t dv_sn, float* dnk, float* dnv, int B, int H, int nq, int nk,
                     const uint8_t* key_mask, int64_t km_sb, const float* q_scale, const float* k_scale, const float* null_k,
                     const float* null_v, float scale);

/* y = l2norm(x) * scale per head of 64 (mmp.py:151-153) backward: dx = (g - xh (xh . g)) / |x|, g = dy * scale; dscale partials.
 * rows x heads_per_row vectors of 64.  x: bf16 [rows][ldx] (head hh at column hh*64), or x_f32 fp32 [H][64] broadcast (vector i
 * uses row i % H: the null key).  dy: bf16 [rows][lddy] or dy_f32 fp32 [rows*heads_per_row][64].  dx likewise (bf16 / fp32).
 * dscale_part fp32 [mm_qk_norm_bwd_blocks(rows*heads_per_row)][64]: reduce with mm_colsum_f32. */
int64_t mm_qk_norm_bwd_blocks(int64_t nvec);
int mm_qk_norm_bwd(mm_stream_t stream, const void* x, int64_t ldx, const float* x_f32, int H, const void* dy, int64_t lddy,
                   const float* dy_f32, const float* scale, int64_t rows, int heads_per_row, void* dx, int64_t lddx,
                   float* dx_f32, float* dscale_part);

/* The uniforms MM_NOISE_PHILOX draws for rows [row_offset, row_offset + rows) at `step`: out fp32 [rows][V]. */
int mm_philox_uniform(mm_stream_t stream, uint64_t seed, uint64_t row_offset, uint32_t step, int rows, int V,
                      float* out);

/* ---- VQGanVAE operators, activations NHWC bf16 (vqgan_vae.py:223-281, 422-441) */

/* Implicit-GEMM convolution.  in bf16 [B][Hin][Win][Cin] (Cin % 8 == 0); w bf16 [Cout][Kp], k = (ty*TW+tx)*Cin+ci,
 * Kp = TH*TW*Cin rounded up to 64 (zero padded).  Virtual output grid Hv x Wv per image; input pixel of tap
 * (ty,tx) = (y*stride + ty + off_y, x*stride + tx + off_x), zero outside.  Output pixel (y*os+py, x*os+px) of an
 * Hout x Wout image -> ConvTranspose2d(4,2,1) is four calls with os=2.  Epilogue: + bias[Cout],
 * LeakyReLU(0.1) if act, + resid (bf16, same shape as out).  out_nchw_f32 == 1 writes fp32 [B][Cout][Hout][Wout]; == 2 writes fp32 NHWC (Cout % 4 == 0)
 * and takes resid as fp32 NHWC -- the precision tier's form: `in` holds P bf16 term segments per pixel (Cin = P x channels, mm_split_rows) and w the matching
 * per-tap segment pack, so the product is exact to fp32. */
int mm_conv2d_nhwc(mm_stream_t stream, const void* in, int B, int Hin, int Win, int Cin, const void* w, int Cout,
                   int TH, int TW, int stride, int off_y, int off_x, int Hv, int Wv, int os, int py, int px,
                   int Hout, int Wout, const float* bias, int act, const void* resid, void* out, int out_nchw_f32);

int mm_glu_nhwc(mm_stream_t stream, const void* x, int64_t rows, int C, void* out);            /* nn.GLU(dim=1)    */
int mm_groupnorm_nhwc(mm_stream_t stream, const void* x, int B, int HW, int C, int groups, const float* gamma,
                      const float* beta, int act, float* stats_ws /* [B*groups*2] */, void* out); /* nn.GroupNorm */
/* LFQ.indices_to_codes + project_out (vqgan_vae.py:430-432): ids int64 [count] -> bf16 [count][C];
 * w fp32 [C][bits], b fp32 [C]; w == NULL means no projection (C == bits). */
int mm_lfq_decode(mm_stream_t stream, const int64_t* ids, int64_t count, int bits, int C, const float* w,
                  const float* b, void* out);
/* LFQ.forward in eval mode (vqgan_vae.py:424): x bf16 [count][C] -> ids int64 [count], out bf16 [count][C]. */
int mm_lfq_encode(mm_stream_t stream, const void* x, int64_t count, int C, int bits, const float* w_in,
                  const float* b_in, const float* w_out, const float* b_out, int64_t* ids, void* out);
int mm_nchw_f32_to_nhwc8_bf16(mm_stream_t stream, const float* img, int B, int C, int H, int W, void* out);
int mm_nhwc_bf16_to_nchw_f32(mm_stream_t stream, const void* x, int B, int C, int H, int W, float* out);

/* ---- composite VQGanVAE entry points (vqgan_vae.py:422-441): the ResnetEncDec layer list + LFQ on one stream, one call per encode / decode.
 * A layer is one entry of ResnetEncDec.encoders / .decoders (vqgan_vae.py:223-232); weights packed as for mm_conv2d_nhwc (bf16 [Cout][Kp]),
 * biases / GroupNorm parameters fp32. */
#define MM_VAE_STEM 0   /* Conv2d(channels, dim, k, padding k/2): w[0] packed with Cin padded to 8, b[0]                       */
#define MM_VAE_DOWN 1   /* Conv2d(4, stride 2, pad 1) + LeakyReLU(0.1): w[0], b[0]                                             */
#define MM_VAE_RES 2    /* ResBlock: conv3 w[0] b[0], GroupNorm gn_g/gn_b[0], conv3 w[1] b[1], GroupNorm [1], conv1 w[2] b[2]   */
#define MM_VAE_GLU 3    /* GLUResBlock: conv3 (C -> 2C) w[0] b[0], GLU, GroupNorm [0], conv3 w[1] b[1], GLU, GroupNorm [1], conv1 w[2] b[2] */
#define MM_VAE_UP 4     /* ConvTranspose2d(4,2,1) + LeakyReLU(0.1): w[0..3] = parity matrices (py*2+px), b[0]                   */
#define MM_VAE_HEAD 5   /* Conv2d(dim, channels, 1): w[0], b[0]; writes the NCHW fp32 image                                     */
typedef struct mm_vae_layer {
    int32_t kind, cout, k, groups;
    const void* w[4];
    const float* b[3];
    const float* gn_g[2];
    const float* gn_b[2];
} mm_vae_layer;
typedef struct mm_vae_desc {
    int32_t channels, encoded_dim, bits, n_enc, n_dec;
    int32_t half;                   /* round 6 (ABI 8): 1 = DECODE-ONLY handle on fp16 storage -- the decoder's conv weights packed as fp16 x 1 / alpha, activations
                                       NHWC fp16, single fp16 terms on the fp16 MFMA (mm_conv2d_nhwc_half); n_enc must be 0                                        */
    const mm_vae_layer* enc;        /* host array [n_enc]: stem, then down-sampling convolutions / residual blocks in list order */
    const mm_vae_layer* dec;        /* host array [n_dec]: GLU blocks / up-sampling convolutions in list order, head last       */
    const float* lfq_wi; const float* lfq_bi;      /* project_in  [bits][encoded_dim], [bits]  (NULL when encoded_dim == bits) */
    const float* lfq_wo; const float* lfq_bo;      /* project_out [encoded_dim][bits], [encoded_dim]                            */
    float alpha;                    /* half != 0: the inverse of the power-of-two scale the fp16 conv weights were packed with (0 = 1) */
    int32_t reserved;
} mm_vae_desc;
typedef struct mm_vae mm_vae_t;
int mm_vae_create(const mm_vae_desc* desc, mm_vae_t** out);
void mm_vae_destroy(mm_vae_t* vae);
size_t mm_vae_decode_workspace_bytes(const mm_vae_t* vae, int B, int h, int w);
size_t mm_vae_encode_workspace_bytes(const mm_vae_t* vae, int B, int H, int W);
/* VQGanVAE.decode_from_ids (vqgan_vae.py:427-438): ids int64 [B][h][w] -> image fp32 [B][channels][h*2^ups][w*2^ups] */
int mm_vae_decode_from_ids(const mm_vae_t* vae, mm_stream_t stream, const int64_t* ids, int B, int h, int w, float* image, void* workspace,
                           size_t workspace_bytes);
/* VQGanVAE.encode (vqgan_vae.py:422-425): image fp32 [B][channels][H][W] -> ids int64 [B][h][w] (+ the quantized feature map fp32
 * [B][encoded_dim][h][w] when fmap_out != NULL) */
int mm_vae_encode(const mm_vae_t* vae, mm_stream_t stream, const float* image, int B, int H, int W, float* fmap_out, int64_t* ids_out, void* workspace,
                  size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------------ parity engine (fp32)
 * Precision level L0: fp32 storage + fp32 MFMA (v_mfma_f32_32x32x2_f32), the reference's operator sequence one to one, no
 * fusion that changes a rounding point.  This is what `set_precision('parity')` of the Python classes runs: logits / pixels within 1e-3
 * of the reference's fp32 CPU run and bit-equal token ids at full size (tests/test_gpu_base_size.py).  All pointers fp32 unless noted. */

/* nn.Linear: out[m][n] = sum_k x[m][k] * w[n][k] (+ bias[n]) (LeakyReLU(0.1) if act) (+ resid[m][n], same stride as out)  (mmp.py:85,88,118-124,233,332) */
int mm_f32_gemm(mm_stream_t stream, const float* x, int64_t ldx, const float* w, int64_t ldw, int M, int N, int K, float* out, int64_t ldc,
                const float* bias, int act, const float* resid);
/* Conv2d / one parity class of ConvTranspose2d(4,2,1) as in mm_conv2d_nhwc, on NHWC fp32; w [Cout][TH*TW*Cin], k = (ty*TW+tx)*Cin+ci
 * (vqgan_vae.py:224-232, 255-261, 271-277) */
int mm_f32_conv2d_nhwc(mm_stream_t stream, const float* in, int B, int Hin, int Win, int Cin, const float* w, int Cout, int TH, int TW, int stride,
                       int off_y, int off_x, int Hv, int Wv, int os, int py, int px, int Hout, int Wout, const float* bias, int act,
                       const float* resid, float* out, int out_nchw);
int mm_f32_layernorm(mm_stream_t stream, const float* x, int64_t ldx, int rows, int D, const float* gamma, const float* beta, float* out,
                     int64_t ldo);                                                                              /* mmp.py:63-70 */
/* GEGLU (mmp.py:72-77): h [rows][ldh] = [x half (F) | gate half (F)] -> out[r][c] = gate * gelu_erf(x) */
int mm_f32_geglu(mm_stream_t stream, const float* h, int64_t ldh, int64_t rows, int F, float* out, int64_t ldo);
/* out = null + (cond - null) * cond_scale  (mmp.py:254) */
int mm_f32_cfg_combine(mm_stream_t stream, const float* cond, const float* null_, float cond_scale, int64_t n, float* out);
/* x[row] = token_emb[ids[row]] + pos_emb[row % n] (mmp.py:322-323); pos_emb NULL = plain gather (condition ids, mmp.py:316) */
int mm_f32_embed(mm_stream_t stream, const int64_t* ids, int64_t rows, int n, const float* token_emb, int vocab_rows, const float* pos_emb, int D,
                 float* x, int64_t ldx);
/* mask[row] = any(text_embeds[row] != 0)  (mmp.py:304) */
int mm_f32_text_mask(mm_stream_t stream, const float* text_embeds, int64_t rows, int D, uint8_t* mask);
/* The same attention as fp16 TERM PRODUCTS on the fp16 matrix pipe (csrc/attention_x2.hip, the 'f16x2' tier's self-attention kernel: q, k, v split into two fp16
 * terms in registers, three products per block, fp32 softmax), fp32 result.  Shape class: dim_head 64, nk in {128, 192, 256}, nq >= 128, no key mask
 * (MM_ERR_UNSUPPORTED otherwise).  Element strides (batch, head, token), d contiguous, 16-byte aligned rows. */
int mm_attend_terms(mm_stream_t stream, const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                    const float* v, int64_t v_sb, int64_t v_sh, int64_t v_sn, float* out, int64_t o_sb, int64_t o_sh, int64_t o_sn, int B, int H, int nq, int nk,
                    int normalize, const float* q_scale, const float* k_scale, const float* null_k, const float* null_v, float scale);
/* mm_attend semantics (attend.py:109-140 + mmp.py:145-157) on fp32 operands, dim_head 32 / 64 / 128 */
int mm_f32_attend(mm_stream_t stream, const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                  const float* v, int64_t v_sb, int64_t v_sh, int64_t v_sn, float* out, int64_t o_sb, int64_t o_sh, int64_t o_sn, int B, int H, int nq, int nk,
                  const uint8_t* key_mask, int64_t km_sb, int normalize, const float* q_scale, const float* k_scale, const float* null_k,
                  const float* null_v, float scale, int dim_head);
int mm_f32_glu_nhwc(mm_stream_t stream, const float* x, int64_t rows, int C, float* out);
int mm_f32_groupnorm_nhwc(mm_stream_t stream, const float* x, int B, int HW, int C, int groups, const float* gamma, const float* beta, int act,
                          float* out);
int mm_f32_lfq_decode(mm_stream_t stream, const int64_t* ids, int64_t count, int bits, int C, const float* w, const float* b, float* out);
/* ids from the signs of the projected features t_in [count][bits] (LFQ.forward in eval mode, vqgan_vae.py:424) */
int mm_f32_lfq_bits(mm_stream_t stream, const float* t_in, int64_t count, int bits, int64_t* ids);
int mm_f32_nchw_to_nhwc(mm_stream_t stream, const float* in, int B, int C, int HW, float* out);
int mm_f32_nhwc_to_nchw(mm_stream_t stream, const float* in, int B, int C, int HW, float* out);

/* ---- precision tier 'bf16x3' (mm_transformer_desc.split_products): operand form of an fp32 matrix.  out bf16 [rows][products * K] = the
 * first `products` (3, 5 or 6) of the segments [h | m | l | h | m | h], x = h + m + l exactly (h = bf16(x), m = bf16(x - h), l = x - h - m).
 * Feed the result to mm_gemm_bf16 / mm_gemm_cfg_logits with K' = products * K against a weight packed the same way. */
int mm_split_rows(mm_stream_t stream, const float* x, int64_t ldx, int64_t rows, int K, int products, void* out);

/* ---- precision tier 'f16x2' (round 4): the same engine on fp16 TERMS and v_mfma_f32_16x16x32_f16.  x ~ h + l, h = fp16(x), l = fp16(x - h): 22
 * significand bits (relative error <= 2^-22; subnormal terms are taken un-flushed, absolute floor 2^-25), so ONE product of two such sums needs only
 * the pairs h.h + l.h + h.l -- X' = [xh | xl | xh] against W' = [wh | wh | wl]: THREE products for GENERAL fp32 weights (the bf16 split needs six), TWO
 * ([xh | xl] . [wh | wh]) when every weight is a single fp16 term (any bf16-representable checkpoint in fp16 range).  The weight terms are packed
 * times a power of two `1 / alpha` (host side: the largest |w| of the model lands in [2^13, 2^14), so that low terms are normal fp16 numbers) and every
 * accumulator is multiplied by `alpha` -- exact.  Operand code of every `products` argument of this header: MM_SPLIT_F16 | 2 or MM_SPLIT_F16 | 3
 * (mm_split_rows, mm_cfg_mix, mm_gemm_split, mm_transformer_desc.split_products).  Measured on the reference's own fp32 checkpoint at
 * BASELINE configs[1] size: logits within 4e-6 of the reference, ids 100 % at every decode step. */
#define MM_SPLIT_F16 0x100
/* mm_gemm_split only, with MM_SPLIT_F16 (round 5): the caller states that X' / W' are GENUINE term-segment packs of mm_split_rows / the weight packing -- segment 2
 * of X' repeats segment 0, segment 1 of W' repeats segment 0 -- so the kernels may stage every term plane ONCE and run the products of a k-block from that one
 * staging (csrc/gemm_terms.hip; same terms, summation order per 32-deep k-block hh, lh, hl instead of all hh, all lh, all hl).  Without the bit the operator is
 * the plain matrix product of the two packs.  mm_transformer_* / mm_generate on an 'f16x2' model always state it (they own their packs). */
#define MM_SPLIT_SHARED 0x200

/* out fp32 [M][ldc] (+ resid_f32) = X' . W'^T over term-segment packs: products = 3 / 5 / 6 -> bf16 terms (== mm_gemm_bf16 with out_f32), alpha unused;
 * MM_SPLIT_F16 | 2 / 3 -> fp16 terms on the fp16 MFMA, accumulators x alpha.  K = segments x inner width, a multiple of 64. */
int mm_gemm_split(mm_stream_t stream, const void* x, int64_t ldx, const void* w, int64_t ldw, int M, int N, int K, int products, float alpha,
                  float* out, int64_t ldc, const float* resid_f32);

/* FF w1 of the 'f16x2' tier (round 5, csrc/gemm_terms.hip): out_split = the term-segment pack [hh | hl | hh][:P] (16-bit container, ldo elements per row,
 * segment length Fp = N / 2) of gate * gelu(x) (exact erf, fp32; mmp.py:72-77) with [x | gate] = alpha X' . W'^T, W' the term segments of the GEGLU-INTERLEAVED
 * w1 rows (mm_ff_weights.w1); ln_part (optional) fp32 [M][N / 64][2] = (sum, sum of squares) of the row's fp32 products per 32 output columns -- what
 * mm_ff_weights.w2_folded's LayerNorm(inner) fold consumes.  products = MM_SPLIT_F16 | 2 / 3; the kernel's shape class only (segment length % 64 (3) / 128 (2),
 * N % 256 == 0, >= 256 tiles of 256 rows x 256 weight rows unless mm_debug_set2(1)): MM_ERR_UNSUPPORTED otherwise (the caller then runs mm_gemm_split + the
 * GEGLU / LayerNorm / split pass). */
int mm_gemm_split_geglu(mm_stream_t stream, const void* x, int64_t ldx, const void* w, int64_t ldw, int M, int N, int K, int products, float alpha,
                        void* out_split, int64_t ldo, float* ln_part);

/* mm_conv2d_nhwc on fp16 term operands: `in` holds MM_SPLIT_F16 | P segments per pixel (Cin = P x channels), w the matching per-tap pack (x 1 / alpha);
 * fp32 output only: out_nchw_f32 = 1 (NCHW) or 2 (NHWC, optional fp32 NHWC residual). */
int mm_conv2d_nhwc_f16(mm_stream_t stream, const void* in, int B, int Hin, int Win, int Cin, const void* w, int Cout,
                       int TH, int TW, int stride, int off_y, int off_x, int Hv, int Wv, int os, int py, int px,
                       int Hout, int Wout, const float* bias, int act, const float* resid_f32, float* out, int out_nchw_f32, float alpha);
/* Round 6 -- the half-precision VAE decode: the same convolution on SINGLE fp16 terms with fp16 activation storage.  `in` NHWC fp16, w fp16 [Cout][Kp] packed
 * x 1 / alpha (a power of two: ops.f16_weight_scale), fp32 accumulation; out NHWC fp16 (out_nchw_f32 = 0; optional resid NHWC fp16) or NCHW fp32 (= 1).  Same MFMA
 * rate as bf16, 11 significand bits instead of 8: decoded pixels 2e-4 of the image scale instead of 1.6e-3 (values saturate at +-65504). */
int mm_conv2d_nhwc_half(mm_stream_t stream, const void* in, int B, int Hin, int Win, int Cin, const void* w, int Cout,
                        int TH, int TW, int stride, int off_y, int off_x, int Hv, int Wv, int os, int py, int px,
                        int Hout, int Wout, const float* bias, int act, const void* resid, void* out, int out_nchw_f32, float alpha);
/* ... with the operand code stated (round 5): products = MM_SPLIT_F16 | 2 / 3 segments per pixel (Cin = P x channels) and per tap of the weight rows; with
 * MM_SPLIT_SHARED (genuine packs: [xh | xl | xh] per pixel, [wh | wh | wl] per tap) the 256 x 128 kernel stages every term plane once (three products, channels % 32 == 0). */
int mm_conv2d_nhwc_terms(mm_stream_t stream, const void* in, int B, int Hin, int Win, int Cin, const void* w, int Cout,
                         int TH, int TW, int stride, int off_y, int off_x, int Hv, int Wv, int os, int py, int px,
                         int Hout, int Wout, const float* bias, int act, const float* resid_f32, float* out, int out_nchw_f32, float alpha, int products);

/* Classifier-free guidance applied to the EMBEDDINGS (round 3): to_logits is linear (mmp.py:332), so null + (cond - null) * s of the two passes'
 * logits (mmp.py:254) equals to_logits(e) with e = e_null + (e_cond - e_null) * s -- ONE [R x V x D] product instead of two.  mm_generate and
 * Transformer.forward_with_cond_scale mix first and multiply once (mm_gemm_bf16 / the fused-sampling GEMM on the mixed rows).
 * products == 0: bf16 rows [R][ld] in, bf16 [R][D] out (rounded once); products = 3 / 5 / 6: term-segment packs in and out (exact fp32 mix). */
int mm_cfg_mix(mm_stream_t stream, const void* emb_cond, const void* emb_null, int64_t ld, int64_t rows, int D, int products, float cond_scale,
               void* out);

/* ------------------------------------------------------------------------------------------------ transformer */

typedef struct mm_attn_weights {
    const float* ln_gamma;   /* norm.gamma [D]                                                            */
    const float* ln_beta;    /* norm.beta  [D] (zero buffer; may be NULL)                                  */
    const void* w_q;         /* bf16 [I][D]   to_q.weight                                                  */
    const void* w_kv;        /* bf16 [2I][D]  to_kv.weight; self-attn: pack q|kv contiguously (w_kv == w_q + I*D) */
    const void* w_out;       /* bf16 [D][I]   to_out.weight                                                */
    const float* null_k;     /* fp32 [H][64]  null_kv[0]                                                   */
    const float* null_v;     /* fp32 [H][64]  null_kv[1]                                                   */
    const float* q_scale;    /* fp32 [64]                                                                  */
    const float* k_scale;    /* fp32 [64]                                                                  */
    /* fp8 engine (mm_transformer_desc.fp8): w_q / w_kv / w_out are OCP e4m3 [rows][in] and these are their per-row scales (fp32 [I] / [2I] / [D]);
     * the cross-attention's w_kv stays bf16 (w_kv_scale unused there: the context projection runs once per generate).  NULL otherwise. */
    const float* w_q_scale;
    const float* w_kv_scale;
    const float* w_out_scale;
    /* optional, bf16 engine (all three or none; round 4): the block's LayerNorm folded into its first projection -- LayerNorm(x) . W^T =
     * rstd * (x . Wg^T) - rstd * mean * c1 + c2, so the projection reads the RAW bf16 residual rows (written, with their statistics, by the
     * residual-adding epilogue in front of it) and no LayerNorm pass runs:                                              */
    const void* w_q_ln;      /* bf16, w_q's shape (self-attention: the concatenated q|k|v matrix [3I][D]) = bf16(W[n][k] * ln_gamma[k])       */
    const float* ln_c1;      /* [rows of w_q_ln]: sum_k float(w_q_ln[n][k])                                            */
    const float* ln_c2;      /* [rows of w_q_ln]: sum_k ln_beta[k] * W[n][k]  (NULL when ln_beta is NULL / zero)         */
} mm_attn_weights;

typedef struct mm_ff_weights {
    const float* ln1_gamma;  /* [D]   */
    const float* ln1_beta;
    const void* w1;          /* bf16 [2*Fp][D], GEGLU-interleaved: per 128-row tile t and half-tile w (0,1), rows
                              * [128t+64w, +32) = gelu-half features 64t+32w.., rows [128t+64w+32, +32) = the gate-half
                              * features of the same 32 output columns (features >= F are zero rows)            */
    const float* ln2_gamma;  /* [Fp]: F gains, zero padded */
    const float* ln2_beta;
    const void* w2;          /* bf16 [D][Fp], columns >= F zero                                            */
    /* optional (all three or none): LayerNorm(inner) folded into the w2 GEMM -- z . W2^T with z = (a - mean) * rstd * gamma + beta
     * is rstd * (a . W2g^T) - rstd * mean * c1 + c2, which saves the LayerNorm's pass over the [rows][Fp] activation:          */
    const void* w2_folded;   /* bf16 [D][Fp] = bf16(w2[o][f] * ln2_gamma[f])                                   */
    const float* ln2_c1;     /* [D]: sum_f float(w2_folded[o][f])                                          */
    const float* ln2_c2;     /* [D]: sum_f ln2_beta[f] * float(w2[o][f])                                   */
    /* fp8 engine: w1 (same GEGLU interleave) / w2 are e4m3 [2*Fp][D] / [D][Fp] with these per-row scales (fp32 [2*Fp] / [D]); w2_folded unused */
    const float* w1_scale;
    const float* w2_scale;
    /* optional, bf16 engine (round 4): LayerNorm(dim) in front of w1 folded the same way (see mm_attn_weights.w_q_ln)           */
    const void* w1_ln;       /* bf16 [2*Fp][D], w1's GEGLU-interleaved row order = bf16(w1[n][k] * ln1_gamma[k])                    */
    const float* ln1_c1;     /* [2*Fp] in the same row order: sum_k float(w1_ln[n][k])                                   */
    const float* ln1_c2;     /* [2*Fp]: sum_k ln1_beta[k] * w1[n][k]  (NULL when ln1_beta is NULL / zero)                */
    /* optional, 'f16x2' tier (round 5, ABI 7): w1 as term segments [wh | wh | wl][:P] of its GEGLU-INTERLEAVED rows (the row order of `w1` above; in the
     * tier `w1` itself stays plain [x half | gate half]).  With it AND w2_folded (term segments of w2[o][f] * ln2_gamma[f]) / ln2_c1 / ln2_c2 the feed-forward
     * runs as two kernels: w1 + GEGLU + term split + LayerNorm(inner) partial sums (csrc/gemm_terms.hip), w2 with the LayerNorm folded in               */
    const void* w1_terms_geglu;
} mm_ff_weights;

typedef struct mm_layer_weights {
    mm_attn_weights self_attn;
    mm_attn_weights cross_attn;
    mm_ff_weights ff;
} mm_layer_weights;

typedef struct mm_transformer_desc {
    int32_t dim, depth, heads, dim_head, ff_inner, ff_inner_padded;
    int32_t seq_len, num_tokens, vocab_rows, dim_out, text_dim, self_cond;
    const void* token_emb;          /* bf16 [vocab_rows][D]  (vocab_rows = num_tokens + 1 with the mask id)  */
    const void* pos_emb;            /* bf16 [seq_len][D]                                                     */
    const void* text_proj;          /* bf16 [D][text_dim] or NULL = nn.Identity (mmp.py:233)                 */
    const mm_layer_weights* layers; /* host array [depth]                                                    */
    const float* final_gamma;       /* transformer_blocks.norm                                               */
    const float* final_beta;
    const void* to_logits;          /* bf16 [dim_out][D]                                                     */
    mm_ff_weights self_cond_ff;     /* self_cond_to_init_embed (used only when self_cond != 0)               */
    /* optional (both or none): statistics of to_logits over the vocabulary -- mean of its rows (fp32 [D]) and their covariance (bf16 [D][D]).  With them
     * mm_generate samples without materialising the logits (mm_fused_*): a row's k-th largest logit is bounded from the row's embedding. */
    const float* logits_wmean;
    const void* logits_wcov;
    /* Precision tier 'bf16x3' (0 = the bf16 engine above).  P = 3, 5 or 6: fp32-grade results on the bf16 matrix pipe -- every activation that
     * feeds a Linear is kept as the exact three-term bf16 split of its fp32 value, x = h + m + l, and multiplied as a K-concatenation of P
     * term pairs X' = [xh|xm|xl|xh|xm|xh][:P] . W' = [wh|wh|wh|wm|wm|wl][:P] (fp32 accumulation in the MFMA): P = 3 when every weight is
     * bf16-representable (m = l = 0), 5 for two-term weights, 6 for general fp32 weights.  With P != 0 the pointers above change meaning:
     * every Linear weight (w_q, w_kv, w_out, w1, w2, text_proj, to_logits) is the segment pack bf16 [out][P * in]; w1 is the PLAIN
     * [2*Fp][P*D] matrix (rows [0, F) = gelu half, rows [Fp, Fp + F) = gate half, the rest zero; w2_folded / ln2_c1 / ln2_c2 unused);
     * token_emb / pos_emb are fp32 tables; ctx and the `embed` output are bf16 [.][P * D]; q|k|v, GEMM outputs, the residual stream and
     * the logits are fp32; attention runs on the fp32 MFMA.  Reference arithmetic matched: fp32 end to end (mmp.py:240-259, 279-335). */
    int32_t split_products;         /* 0 | 3 | 5 | 6 (bf16 terms) | MM_SPLIT_F16 | 2 | MM_SPLIT_F16 | 3 (fp16 terms: 'f16x2', see above -- the same pointer meanings with
                                     * [out][P * in] packs of fp16 terms x 1 / split_alpha) */
    /* fp8 engine (BASELINE configs[4] "fp8 MFMA weights"), fp8 != 0: the Linear weights of the layers (and of self_cond_ff) are OCP e4m3 rows with the
     * per-row scales of mm_attn_weights / mm_ff_weights (mm_quantize_e4m3_rows); their input activations are quantised per token row inside the producing
     * kernels and the products run on v_mfma_f32_16x16x128_f8f6f4 (mm_gemm_fp8).  Embeddings, text_proj, the cross-attention's w_kv, attention, to_logits
     * and the sampling are the bf16 engine's.  dim, heads * dim_head and ff_inner_padded must be multiples of 128; split_products must be 0.
     * Self-defined numerics: the oracle is the fp32 restatement with the same per-row fake quantisation at every Linear of the layers. */
    int32_t fp8;
    float split_alpha;              /* fp16 terms only: the power of two every GEMM accumulator is multiplied by (0 = 1); the weight terms were packed x 1 / split_alpha */
    /* bf16 engine, LayerNorm(dim) folded into the GEMMs around it (default).  The fold multiplies the bf16 image of the RAW residual row, so its rounding error in
     * normalised units grows with |row mean| / (row standard deviation): negligible for random-init-like statistics, visible for a checkpoint whose residual
     * stream carries a large DC offset.  ln_fold_off != 0 runs every LayerNorm as its own kernel (the round-3 engine; same arithmetic otherwise).  ln_probe
     * (optional, device float[1], zeroed by the caller): every fold consumer first records max(|mean| * rstd) of the rows it reads (atomicMax) -- the host side
     * probes once per packed model and falls back to ln_fold_off above MM_LN_FOLD_MAX_RATIO (muse_maskgit.py, Transformer.set_layernorm_fold). */
    int32_t ln_fold_off;
    float* ln_probe;
    /* optional: logits_wsub_rows (<= 4096) rows of to_logits at vocabulary indices drawn once per model, bf16 [rows][dim] (plain bf16 values on every engine).
     * With them the fused sampler's per-row lower bound of the k-th largest logit is DISTRIBUTION-FREE: the row's logits at the sampled columns are computed by
     * a small GEMM and the bound is their (rows k / V + 4.5 sigma)-th largest -- it holds for peaky / heavy-tailed / multi-modal logits alike, where the
     * Gaussian estimate from logits_wmean / logits_wcov (used when this is NULL) fails every row and the call falls back to materialised logits. */
    const void* logits_wsub;
    int32_t logits_wsub_rows;
} mm_transformer_desc;
#define MM_LN_FOLD_MAX_RATIO 4.0f

typedef struct mm_transformer mm_transformer_t;

int mm_transformer_create(const mm_transformer_desc* desc, mm_transformer_t** out);
void mm_transformer_destroy(mm_transformer_t* model);

/* Context of Transformer.forward (mmp.py:302-318): ctx bf16 [B][m][D] (m = L + nc), key_mask uint8 [B][m].
 * text_embeds fp32 [B][L][text_dim] zero-padded (t5.py:93); cond_ids (optional) int64 [B][nc].
 * drop_text != 0 builds the classifier-free "null" mask (text keys off, cond-id keys on). */
size_t mm_context_workspace_bytes(const mm_transformer_t* model, int B, int L);
int mm_transformer_context(const mm_transformer_t* model, mm_stream_t stream, const float* text_embeds, int B,
                           int L, const int64_t* cond_ids, int nc, int drop_text, void* ctx, uint8_t* key_mask,
                           void* workspace, size_t workspace_bytes);

/* Transformer.forward without the loss branch (mmp.py:322-335): ids int64 [B][n]; ctx / key_mask as above;
 * self_cond_embed optional fp32 [B*n][D].  Outputs (each optional): embed bf16 [B*n][D] (the final LayerNorm
 * output), logits fp32 [B*n][dim_out]. */
size_t mm_transformer_workspace_bytes(const mm_transformer_t* model, int B, int n, int m);
int mm_transformer_forward(const mm_transformer_t* model, mm_stream_t stream, const int64_t* ids, int B, int n,
                           const void* ctx, const uint8_t* key_mask, int m, const float* self_cond_embed,
                           void* embed_out, float* logits_out, void* workspace, size_t workspace_bytes);

/* The cross-attention block of ONE layer as an operator (round 5): x (fp32 [seqs * n][dim], updated in place) += CrossAttention(LayerNorm(x), ctx) -- mmp.py:139-162,
 * 191 -- exactly as mm_transformer_forward runs it for this model: on the bf16 engine's headline shape class (dim = inner = 512, 8 heads x 64, <= 35 context tokens,
 * LayerNorm(dim) fold on) the one-kernel form of csrc/cross_fold.hip, otherwise (or with mm_debug_set bit 1 << 31) q projection + attention + output projection.
 * ctx bf16 [seqs][m][dim], key_mask uint8 [seqs][m] or NULL.  bf16 engine only.  Test surface of the block (tests/test_gpu_ops.py). */
size_t mm_cross_attention_block_workspace_bytes(const mm_transformer_t* model, int seqs, int n, int m);
int mm_cross_attention_block(const mm_transformer_t* model, mm_stream_t stream, int layer, float* x, int seqs, int n, const void* ctx, const uint8_t* key_mask, int m,
                             void* workspace, size_t workspace_bytes);

/* MaskGit.generate's decode loop (mmp.py:519-615), whole loop on `stream` with no host synchronisation:
 * CFG double pass batched as 2B sequences, cross-attention K/V of the context computed once, logits and
 * sampling only at the currently masked rows.  Host supplies the per-step schedule computed with the
 * reference's own arithmetic (mask_counts[t] = max(int(cos(t*pi/2)*n), 1), temperatures[t] = max(T0*(steps
 * left)/T, 1e-10)).  noise (GUMBEL/UNIFORM modes): fp32 [timesteps][B][n][V].  Outputs: ids int64 [B][n],
 * scores fp32 [B][n].  Optional traces [timesteps][B][n]: trace_masked_ids (ids after the re-mask scatter),
 * trace_ids / trace_scores (state after each step). */
#define MM_GEN_NO_FUSED_SAMPLING 1   /* flags: always materialise the logits (mm_gemm_cfg_logits + mm_sample_rows) */
#define MM_GEN_CAN_REMASK 2          /* flags: can_remask_prev_masked (mmp.py:608-609): every position is sampled and scored at every step */
typedef struct mm_generate_params {
    int32_t batch, n, timesteps, k_keep, noise_kind, nc, L, flags;
    float cond_scale;
    float pad0;
    uint64_t seed, row_offset;
    const int32_t* mask_counts;     /* host [timesteps] */
    const float* temperatures;      /* host [timesteps] */
    const float* text_embeds;       /* device fp32 [B][L][text_dim] */
    const int64_t* cond_ids;        /* device int64 [B][nc] or NULL */
    const float* noise;             /* device or NULL */
    int64_t* ids;                   /* device out */
    float* scores;                  /* device out */
    int64_t* trace_masked_ids;
    int64_t* trace_ids;
    float* trace_scores;
    /* device int32 [2], zeroed by the caller, or NULL (= no fused sampling).  Fused sampling bounds every row's k-th largest logit BEFORE the
     * logits exist; the finishing kernel verifies the bound per row.  A row it cannot be proven for (heavy-tailed logits) is finished on the
     * logits path INSIDE the same step, on the device (its logits recomputed, sample_rows' kernel; the ids never depend on which path sampled a
     * row); status[1] counts those rows.  Only when more than 128 rows fail in one step is status[0] set to 1: the ids of that call are then
     * invalid, repeat it with MM_GEN_NO_FUSED_SAMPLING. */
    int32_t* status;
    /* ---- decode variants, all inside the same loop (zero / NULL = off).  cond_scale == 1 runs the single conditional pass (mmp.py:247-248); a
     * transformer created with self_cond feeds every step's cond-pass embed into the next step (mmp.py:325-328, 574).
     * Critic scores (mmp.py:590-601): scores = critic(ids) + (u - 0.5) * critic_noise_scale * (timesteps - 1 - step) / timesteps replace the
     * sampler's 1 - p.  `critic` = a TokenCritic (transformer with dim_out == 1, its own embeddings / context); critic_head_w / _b = a SelfCritic
     * (mmp.py:352-374): Linear(dim, 1) (bf16 [dim] weight, fp32 [1] bias) on this model's cond-pass embed of the new ids.  Exclusive. */
    const mm_transformer_t* critic;
    const void* critic_head_w;
    const float* critic_head_b;
    const float* critic_noise;      /* device fp32 [timesteps][B][n]: the U(0,1) draws of mmp.py:601 (required with a critic) */
    float critic_noise_scale;
    float pad1;
    void* critic_workspace;         /* device, >= mm_generate_critic_workspace_bytes(critic or model, ...) */
    size_t critic_workspace_bytes;
    /* optional device uint64[2] = {seed, row_offset}: with MM_NOISE_PHILOX the sampling kernels read their keys from here AT EXECUTION TIME instead of the
     * `seed` / `row_offset` fields above -- a hipGraph captured once around mm_generate then replays with whatever the host wrote into the buffer before
     * the replay (MaskGit.generate(graph=True)): fresh noise per replay, bit-identical to an eager call with the same keys. */
    const uint64_t* seed_dev;
} mm_generate_params;

size_t mm_generate_workspace_bytes(const mm_transformer_t* model, int B, int n, int L, int nc);
/* workspace of the critic passes: pass the TokenCritic handle, or the generator's own handle for a SelfCritic */
size_t mm_generate_critic_workspace_bytes(const mm_transformer_t* critic, int B, int n, int L, int nc);
int mm_generate(const mm_transformer_t* model, mm_stream_t stream, const mm_generate_params* params,
                void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------------ training step (round 4)
 * MaskGit.forward's differentiable core (muse_maskgit_pytorch.py:623-741 through Transformer.forward :279-348) as ONE call: forward with saved
 * activations, mean cross-entropy over the R labelled rows, and the hand-written backward, ~1150 launches on one stream with no allocation and no
 * host synchronisation in between (capturable).  Parameters are the caller's fp32 tensors (nn.Parameter layouts), bf16 operand copies are made per
 * step inside the workspace; every gradient is WRITTEN (not accumulated) as fp32 in the parameter's own shape.  The operators and their order are
 * those of the operator-by-operator driver (training.py): the loss and every gradient are bit-identical to it.  Scope: dim_head 64, n in {64, 128,
 * 256}, batch * n / dim / dim_out / text_dim multiples of 64, no self-conditioning, no conditioning ids, cross-entropy head (the other variants run on
 * the operator-by-operator driver).  null_kv: [2][heads][1][64] (k then v), q_scale / k_scale [64]; beta pointers: the LayerNorms' zero buffers (may be
 * NULL only for ff.b2); text_proj NULL = nn.Identity (text_dim == dim; d_text_proj unused). */
typedef struct mm_train_attn {
    const float *gamma, *beta, *to_q, *to_kv, *q_scale, *k_scale, *null_kv, *to_out;
    float *d_gamma, *d_to_q, *d_to_kv, *d_q_scale, *d_k_scale, *d_null_kv, *d_to_out;
} mm_train_attn;
typedef struct mm_train_ff {
    const float *g1, *b1, *w1, *g2, *b2, *w2;      /* w1 [2F][D] (rows [0, F) gelu half, [F, 2F) gate half), w2 [D][F] */
    float *d_g1, *d_w1, *d_g2, *d_w2;
} mm_train_ff;
typedef struct mm_train_layer {
    mm_train_attn sa, ca;
    mm_train_ff ff;
} mm_train_layer;
typedef struct mm_train_desc {
    int32_t dim, depth, heads, ff_inner, seq_len, vocab_rows, dim_out, text_dim;
    const float *token_emb, *pos_emb, *text_proj, *final_gamma, *final_beta, *to_logits;
    float *d_token_emb, *d_pos_emb, *d_text_proj, *d_final_gamma, *d_to_logits;
    const mm_train_layer* layers;      /* host array [depth] */
} mm_train_desc;
size_t mm_train_step_workspace_bytes(const mm_train_desc* desc, int B, int n, int L, int R);
/* ids int64 [B][n]; text_embeds fp32 [B][L][text_dim]; ctx_mask uint8 [B][L] (1 = attend); row_index int32 [R] flat positions b * n + pos that carry a
 * label, labels_rows int64 [R]; loss_out fp32 [1]; logits_rows_out (optional) fp32 [R][dim_out]. */
int mm_train_step(const mm_train_desc* desc, mm_stream_t stream, const int64_t* ids, int B, int n, const float* text_embeds, int L, const uint8_t* ctx_mask,
                  const int32_t* row_index, const int64_t* labels_rows, int R, float* loss_out, float* logits_rows_out, void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------------ multi-GPU: the one collective of the path
 * Inference shards by sample (one process per GPU); mm_generate needs no communication (every reduction of MaskGit.generate is per sample,
 * mmp.py:561,576,580,603).  What remains is ONE all-gather of the generated token grids: ids int64 [count] per rank (all < codebook size) travel as
 * int32 over RCCL (xGMI) on the caller's stream and arrive as int64 [world * count] in rank order.  RCCL is resolved at run time (the RCCL already
 * loaded in the process, else librccl.so): MM_ERR_UNSUPPORTED when there is none.  Bootstrap like ncclCommInitRank: rank 0 calls
 * mm_comm_unique_id (128 bytes), shares them by any host channel, every rank calls mm_comm_create on its own device. */
typedef struct mm_comm mm_comm_t;
int mm_comm_unique_id(void* id_out_128_bytes);
int mm_comm_create(const void* unique_id_128_bytes, int rank, int world, mm_comm_t** out);
void mm_comm_destroy(mm_comm_t* comm);
int mm_comm_world(const mm_comm_t* comm);
int mm_comm_rank(const mm_comm_t* comm);
size_t mm_allgather_ids_workspace_bytes(const mm_comm_t* comm, int64_t count);
int mm_allgather_ids(mm_comm_t* comm, mm_stream_t stream, const int64_t* ids, int64_t count, int64_t* out, void* workspace, size_t workspace_bytes);

/* ---- in-library kernel timing (bench.py's roofline leg).  When enabled, mm_generate brackets its two dominant
 * kernels with HIP events on the launch stream: slot 0 = the CFG to_logits GEMM (MFMA-bound), slot 1 = sample_rows
 * (HBM-bound).  mm_profile_read synchronises those events and returns, per slot, the launch count, the summed
 * duration in milliseconds and the summed algorithmic work (flops for slot 0, bytes for slot 1), then resets. */
#define MM_PROF_SLOTS 2
/* kernel ablation switches for tools/ (0 = product behaviour) */
int mm_debug_set(int flags);
/* second word (round 5): 1 = the term-sharing GEMM of the 'f16x2' tier (csrc/gemm_terms.hip) whatever the tile count (tests run small batches through the
 * production kernels with it), 2 = term sharing off (A/B), 4 (round 6; was the environment variable MM_TRAIN_SIDE=0) = mm_train_step without its side stream (A/B),
 * 8 / 16 / 32 = the VAE decode's 256 x 256 convolution tile / fused head / parity-batched ConvTranspose off (A/B), 64 = the 'f16x2' tier's cross-attention as
 * separate attention + output projection launches instead of csrc/cross_vw_x2.hip, 128 = that kernel without its in-kernel LayerNorm + q projection, 256 = the tier's null-half constant row in the feed-forward's LayerNorm pass, 1024 = csrc/gemm_tn.hip off in mm_train_step, 4096 = the two-sweep cross-entropy backward on long rows, 8192 = the head's dW beside (not behind) its dX in mm_train_step (A/B, tests).
 * The library reads NO environment variable: every switch is an explicit call. */
int mm_debug_set2(int flags);
/* Race / determinism screen (tools/determinism_stress.py, tests): with a device buffer registered (NULL = off) every
 * mm_transformer_forward writes one 64-bit position-sensitive checksum per operator output, in launch order, to device_buf[0..];
 * mm_debug_trace_count() = entries of the last pass.  Two passes over the same inputs must produce identical traces. */
int mm_debug_trace(uint64_t* device_buf, int capacity);
int mm_debug_trace_count(void);
/* additionally keep raw copies of the operator outputs #first, #first + step, ... of a traced pass (stride_bytes apart; NULL = off) */
int mm_debug_capture(void* device_buf, size_t stride_bytes, int first, int step);
int mm_profile_enable(int enable);
int mm_profile_read(int slot, int64_t* launches, double* total_ms, double* total_work);

#ifdef __cplusplus
}
#endif
#endif /* MUSE_HIP_H */
