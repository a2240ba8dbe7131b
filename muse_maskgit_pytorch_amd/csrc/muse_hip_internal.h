// Internal (C++) declarations shared between the kernel translation units and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/muse_hip.h"

typedef uint16_t bf16_t;

enum { MODE_DENSE = 0, MODE_CFG = 1, MODE_CONV = 2 };
enum { OUT_BF16 = 0, OUT_F32 = 1, OUT_NCHW_F32 = 2 };
enum { ACT_NONE = 0, ACT_LEAKY = 1 };
enum { EPI_NONE = 0, EPI_GEGLU = 1 };

struct GemmArgs {
    int mode;
    // weight operand W[N][ldw], K contiguous (zero padded to a multiple of 64)
    const bf16_t* W; int N; int ldw; int K;
    // activation operand: M rows
    int M;
    const bf16_t* X; int ldx;     // dense / cfg-cond / conv input (NHWC)
    const bf16_t* X2;             // cfg-null rows
    // conv geometry (MODE_CONV): virtual output grid Hv x Wv per image, taps TH x TW
    int Hin, Win, Cin, TW, stride, off_y, off_x, Hv, Wv, Ktrue;
    int os, py, px, Hout, Wout;   // output pixel = (y*os+py, x*os+px) in an Hout x Wout image
    // epilogue
    void* out; long ldc; int out_kind;
    const float* bias; const float* resid_f32; const bf16_t* resid_bf16; long ldr;
    int act; float cfg_scale;
    int epi;                      // EPI_GEGLU: W rows are GEGLU-interleaved, the tile emits N/2 columns of gate*gelu(x)
    int tiles_m, tiles_n;         // filled by mm_gemm_launch
    int group_m;                  // gemm_wide.hip: row-tiles per group of the XCD-grouped tile order (filled by its launcher)
    int splits; long split_stride; // split-K (weight gradients): gridDim.y = splits, split s sums k-tiles [s*K/splits, (s+1)*K/splits) into out + s*split_stride floats
    // LayerNorm(inner) folded into the FF GEMM pair (model.hip ff_block):
    //   w1 + GEGLU (any kernel of the family): ln_part != NULL -> per output row and 64 output columns the sum and sum of squares of
    //     the bf16 outputs go to ln_part[(row * ln_np + part) * 2 ..], part = output column / 64, ln_np = N / 128;
    //   w2 (128x128 / 256x128 kernel, fp32 + residual): ln_c1 != NULL -> out = rstd*acc - rstd*mean*ln_c1[n] + ln_c2[n] + resid with
    //     mean / rstd of each row from the ln_np partials over ln_F valid features.
    float* ln_part; int ln_np; int ln_F; const float* ln_c1; const float* ln_c2;
    // fused sampling (MODE_CFG on gemm_cfg.hip only): when fs_stats != NULL the logits are NOT written; every 256-column piece of a row emits its
    // statistics and its candidates >= fs_thr[row] instead (common.h fused_emit_piece)
    const float* fs_thr; float4* fs_stats; float4* fs_cand;
    const int* m_dev;             // optional device-side row count (<= M): row tiles at or beyond it exit immediately (128x128 kernel only)
    int wide_tok;                 // MODE_DENSE, fp32 out: request the persistent 128-token x 256-column kernel of gemm_cfg.hip (WIDE_MIX: the guidance logits of
                                  // the MIXED embedding as one pass; takes fs_* like MODE_CFG); falls back to the other kernels when not eligible
    int debug;                    // ablation bits (mm_debug_set): 1 = no epilogue stores, 2 = no DMA after tile 0, 4 = no MFMA, 8 = force the 128x128 kernel, 4096 = no persistent kernel
    // 'f16x2' precision tier: both operands hold fp16 TERMS (common.h split2_f16) -> v_mfma_f32_16x16x32_f16; the fp32 accumulators are multiplied by
    // `alpha` (the inverse of the power-of-two scale the weight terms were packed with; 0 = 1) before anything else in the epilogue.  fp32 output only,
    // dense and convolution modes, on the 128x128 / 256x128 kernels and the fused-sampling logits kernel.
    int f16; float alpha;
    // round 6 (f16 only): 16-bit outputs / residuals of this launch are fp16, not bf16 (out_kind OUT_BF16 / resid_bf16 name the 16-bit container): the single-term
    // fp16 VAE decode -- NHWC fp16 activations in and out, the head convolution to NCHW fp32
    int half_io;
    // round 6, gemm_wide_conv.hip: the 1 x 1 head convolution (Conv2d(dim, channels, 1), vqgan_vae.py:232) in the epilogue of the last up-sampling convolution
    // (N == 256): head_w = the head's packed 16-bit weights [head_c][head_ldw] (same storage type and scale as W), head_b fp32 [head_c]; `out` = NCHW fp32 image
    const bf16_t* head_w; const float* head_b; int head_c; int head_ldw;
    // ... and all four parity classes of a ConvTranspose2d(4, 2, 1) in one launch: par_w[py * 2 + px] = that class's packed 2 x 2 weights (par_w[1] != NULL selects
    // the form; W / off_y / off_x / py / px are then ignored: tap offset (py - 1, px - 1), output phase (py, px), os = 2)
    const bf16_t* par_w[4];
    // ... with `terms` = 2 / 3 the caller also states that X' / W' are equal-length term segments [xh | xl | xh][:terms] / [wh | wh | wl][:terms] (K = terms x the
    // segment length): gemm_terms.hip then stages every term plane once and runs the products of a k-block from that one staging (0: unknown -- plain fp16 GEMM
    // of depth K).  With EPI_GEGLU (terms != 0 only): W rows GEGLU-interleaved, `out` = the term-segment pack [hh | hl | hh][:terms] of gate * gelu(x)
    // (16-bit container, ldc elements per row, segment length N / 2), ln_part = (sum, sum of squares) per row and 32 output columns (ln_np = N / 64).
    int terms;
    int terms_nodup;      // EPI_GEGLU on term operands: do not write the repeated h segment of the output pack (its only reader, FF w2, runs a term-sharing k-loop)
    // LayerNorm(dim) folded into the GEMMs around it (round 4, bf16 engine; model.hip):
    //   PRODUCER -- the fp32-residual epilogue of the 128x128 / 256x128 kernels (out = resid + acc): with xb_out != NULL it also writes the new residual
    //     row as bf16 (xb_out [M][ldxb]) and, per row and 64 columns, the (sum, sum of squares) of the fp32 values to st_part[(row * st_np + i) * 2 ..],
    //     st_np = 2 ceil(N / 128).  add_row (optional, fp32 [N]) is added to rows >= add_row_from after the residual: (acc + resid) + add_row.
    //   CONSUMER -- bf16-output dense GEMMs (gemm_wide plain / GEGLU, the 128x128 kernel): with in_c1 != NULL the operand rows X are the RAW bf16 residual
    //     rows and W carries the gains (bf16(W . diag gamma)); the epilogue applies out = rstd * acc - rstd * mean * in_c1[n] + in_c2[n] (in_c2 optional)
    //     with the row's mean / rstd from the in_np partials in_part[(row * in_np + i) * 2 ..] over in_F features -- before GEGLU when there is one.
    bf16_t* xb_out; long ldxb; float* st_part; int st_np; int st_gran; const float* add_row; int add_row_from;      // st_gran: columns per partial (64 for N <= 512, else 128; set by mm_gemm_launch)
    const float* in_part; int in_np; int in_F; const float* in_c1; const float* in_c2;
};
extern int g_mm_debug;
extern int g_mm_debug2;      // mm_debug_set2 (round 5): 1 = gemm_terms.hip whatever the tile count (tests: small batches through the production kernels), 2 = gemm_terms.hip and every term-sharing k-loop off (A/B), 4 = mm_train_step on the caller's stream only (A/B), 8 = gemm_wide_conv.hip off (A/B: convolutions on the 256 x 128 kernel), 16 = the VAE head not fused into the last up-sampling convolution (A/B), 32 = a ConvTranspose2d's parity classes as four launches (A/B), 64 = the 'f16x2' tier's cross-attention as attention + output projection launches instead of cross_vw_x2.hip (A/B, tests), 128 = cross_vw_x2.hip behind LayerNorm-split + q GEMM launches (A/B, tests), 256 = the tier's null-half constant cross-attention row added by the feed-forward's LayerNorm-split pass instead of the self-attention's output projection (A/B), 1024 = the training step's weight gradients all through transposed copies (gemm_tn.hip off; A/B), 4096 = ce_bwd on long rows as two sweeps (A/B), 8192 = mm_train_step's head dW launched beside the head's dX instead of behind it (A/B)

int mm_gemm_launch(GemmArgs a, hipStream_t stream);
bool mm_gemm_big_eligible(const GemmArgs& a);
bool mm_gemm_big_split_eligible(const GemmArgs& a);      // split-K (a.splits > 1) on the 256 x 128 tile: >= 4096 of K per split      // gemm_big.hip: 256x128 tile, 3-stage counted-vmcnt pipeline
int mm_gemm_big_launch(GemmArgs a, hipStream_t stream);
bool mm_gemm_pers_eligible(const GemmArgs& a);
bool mm_gemm_cfg2_eligible(const GemmArgs& a);
bool mm_gemm_wide_eligible(const GemmArgs& a);      // gemm_wide.hip: 256 x 256 x 64 tile, one barrier per 64-deep step
int mm_gemm_wide_launch(GemmArgs a, hipStream_t stream);
bool mm_gemm_wide_fused_eligible(const GemmArgs& a);      // the fused-sampling logits GEMM on the same k-loop (persistent)
int mm_gemm_wide_fused_launch(GemmArgs a, hipStream_t stream);     // gemm_cfg.hip: persistent 128 tokens x 256 columns, guidance logits
int mm_gemm_cfg2_launch(GemmArgs a, hipStream_t stream);
bool mm_gemm_wide_conv_eligible(const GemmArgs& a);      // gemm_wide_conv.hip (round 6): NHWC convolutions on the persistent 256 x 256 x 64 tile (+ the fused 1 x 1 head)
int mm_gemm_wide_conv_launch(GemmArgs a, hipStream_t stream);
bool mm_gemm_terms_eligible(const GemmArgs& a);      // gemm_terms.hip (round 5): fp16 term-segment operands, every term plane staged once
int mm_gemm_terms_launch(GemmArgs a, hipStream_t stream);
#ifdef MM_TOOLS_PP      // tools/experiments/gemm_pp.hip (round 5's measured-and-rejected forms; tools build only, see tools/build_timing.sh)
bool mm_gemm_pp_fused_selected(const GemmArgs& a);
int mm_gemm_pp_fused_launch(GemmArgs a, hipStream_t stream);
#endif
// gemm_pers.hip: persistent 256x128, stores overlapped with the next tile
int mm_gemm_pers_launch(GemmArgs a, hipStream_t stream);

// error plumbing (thread-local message, never throws across the ABI)
int mm_set_error(int code, const char* msg);
int mm_set_hip_error(hipError_t e, const char* where);
int mm_check_launch(const char* kernel);

// ---- kernel launchers implemented in the other .hip files (all asynchronous on `stream`)
int k_embed(hipStream_t s, const int64_t* ids, int rows, int n, int pos_offset, const bf16_t* tok, int vocab_rows,
            const bf16_t* pos, int D, float* x);
int k_layernorm(hipStream_t s, const float* x, long ldx, int rows, int D, const float* gamma, const float* beta,
                const int32_t* row_index, bf16_t* out, long ldo);
int k_geglu_ln(hipStream_t s, const bf16_t* h, long ldh, int rows, int F, int Fp, const float* gamma, const float* beta,
               bf16_t* out, long ldo);
int k_layernorm_addvec(hipStream_t s, float* x, long ldx, int rows, int D, const float* gamma, const float* beta, const float* addvec,
                       int add_from, bf16_t* out, long ldo);
int k_fold_image(hipStream_t s, const float* x, long ldx, int rows, int D, bf16_t* xb, long ldxb, float* stp, int np);      // fold producer outputs of a stream no GEMM just wrote
int k_ln_fold_ratio(hipStream_t s, const float* stp, int rows, int np, int F, float* out);      // max |mean| * rstd of the rows a LayerNorm(dim)-fold consumer reads -> atomicMax(*out)
int k_gather_rows16(hipStream_t s, const void* src, long src_pitch_bytes, const int32_t* rows, int R, int row_add, int row_bytes, void* dst);
int k_gather_rows16_multi(hipStream_t s, int njobs, const void* const* src, const long* src_pitch_bytes, const int* row_add, const int* row_bytes, void* const* dst,
                          const int32_t* rows, int R);
// final LayerNorm of both guidance passes + guidance mix in the embedding (+ <e, wmean> per row) over the (optionally gathered) rows, in one pass
int k_final_mix(hipStream_t s, const float* xc, const float* xn, long ldx, int rows, int D, const float* gamma, const float* beta, const int32_t* row_index,
                float cond_scale, bf16_t* out, const float* wmean, float* mu);
int k_gather_rows16_counted(hipStream_t s, const void* src, long src_pitch_bytes, const int32_t* rows, const int32_t* count, int cap, int row_bytes, void* dst,
                            int32_t* total);
int k_ln_bf16(hipStream_t s, const bf16_t* a, long lda, int rows, int F, int Fp, const float* gamma, const float* beta,
              bf16_t* out, long ldo);
int k_f32_to_bf16(hipStream_t s, const float* x, bf16_t* out, long count);
int k_quantize_e4m3_rows(hipStream_t s, const float* w, long ldw, int rows, int K, int Kp, unsigned char* wq, float* scale);

// fp8 engine (gemm_fp8.hip): out = sx[m] * sw[n] * (xq . wq) on the K = 128 fp8 MFMA.  epi 0: bf16 out; 1: GEGLU of the interleaved w1 rows, bf16 out
// [M][N / 2]; 2: fp32 out = resid + product
struct GemmF8Args {
    const unsigned char* X; long ldx; const float* sx;      // e4m3 [M][ldx bytes], per-row scale
    const unsigned char* W; long ldw; const float* sw;      // e4m3 [N][ldw bytes], per-row scale
    int M, N, K;                                            // K: padded contraction length, a multiple of 128
    void* out; long ldc;
    int epi;
    const float* resid; long ldr;
    int tiles_m, tiles_n;                                   // set by the launcher
};
int k_gemm_fp8(hipStream_t s, const GemmF8Args& a);
// activation rows -> e4m3 + per-row scale (max |x| / 448); columns K..Kp-1 zero.  bf16 or fp32 input
int k_quantize_act_e4m3(hipStream_t s, const void* x, int x_f32, long ldx, int rows, int K, int Kp, unsigned char* xq, float* scale);
// LayerNorm -> e4m3 rows + scales (fp8_act.hip); addvec != NULL: rows >= add_from get it added to x in place first
int k_layernorm_q8(hipStream_t s, float* x, long ldx, int rows, int D, const float* gamma, const float* beta, const float* addvec, int add_from,
                   unsigned char* q8, long ldq, float* qs);
int k_ln_inner_q8(hipStream_t s, const bf16_t* a, long lda, int rows, int F, int Fp, const float* gamma, const float* beta, unsigned char* q8, long ldq, float* qs);

struct AttnArgs {
    const bf16_t* q; long q_sb, q_sh, q_sn;      // element strides: batch, head, token (d contiguous, dh = 64)
    const bf16_t* k; long k_sb, k_sh, k_sn;
    const bf16_t* v; long v_sb, v_sh, v_sn;
    bf16_t* out; long o_sb, o_sh, o_sn;
    int B, H, nq, nk;                            // nk real keys (null key excluded)
    const uint8_t* key_mask; long km_sb;         // optional (B, nk) 1 = keep
    int normalize;                               // 1: l2norm(q)*q_scale, l2norm(k)*k_scale in-kernel
    const float* q_scale; const float* k_scale;  // [64]
    const float* null_k; const float* null_v;    // optional [H][64] fp32 (raw parameter values)
    float scale;                                 // 8
    int dh;                                      // dim_head (0 = 64).  != 64: served by the fp32-MFMA kernel of attention_f32.hip on the bf16 operands
    int kv_batch_mod;                            // > 0: k/v batch index = b % kv_batch_mod (CFG halves share one context)
    int debug;                                   // ablation bits 256 (no compute) / 512 (no staging) / 1024 (no softmax exp)
};
int k_attention(hipStream_t s, const AttnArgs& a);

// cross_fold.hip: the cross-attention block as one kernel -- q projection (LayerNorm(dim) fold, consumer side), attention with the output projection folded into
// the (step-invariant) values, residual add, fold producer epilogue
struct CrossFoldArgs {
    const bf16_t* xb_in; long ldxb_in;           // [seqs * nq][ldxb_in] bf16: raw rows of the residual stream (may alias xb: a workgroup reads its rows before it writes them)
    const float* stp_in; int in_np;              // their (sum, sum of squares) partials [rows][in_np][2] (may alias stp)
    const bf16_t* wqf;                           // k_cross_fold_pack: gain-folded q weight as fragments [8][4][16][64][8]
    const float* c1; const float* c2;            // [512] fold constants of the q projection (mm_attn_weights::ln_c1 / ln_c2; c2 may be NULL = zeros)
    const bf16_t* khat;                          // k_cross_fold_pack: K^ fragments [kv_seqs][8][3][4][64][4]
    const bf16_t* vwt;                           // k_cross_fold_pack: (V W_o^T)^T fragments [kv_seqs][32][9][64][8]
    const uint8_t* key_mask; long km_sb;         // optional [seqs][m], 1 = keep
    const float* q_scale;                        // [64]
    float* x; long ldx;                          // fp32 residual stream [seqs * nq][ldx], updated in place
    bf16_t* xb; long ldxb;                       // optional: bf16 image of the new rows ...
    float* stp; int st_np;                       // ... and their (sum, sum of squares) per 64 columns [rows][st_np][2]
    int seqs, nq, m, kv_batch_mod;
    float scale;                                 // 8
};
bool k_cross_fold_eligible(int D, int I, int H, int dh, int m);
size_t k_cross_fold_khat_elems(int kv_seqs);
size_t k_cross_fold_vwt_elems(int kv_seqs);
size_t k_cross_fold_wqf_elems();
int k_cross_fold_pack(hipStream_t s, const bf16_t* ckv, int kv_seqs, int m, int I, const float* null_k, const float* null_v, const float* k_scale,
                      const bf16_t* w_out, int ldw, const bf16_t* w_q_ln, int ldwq, bf16_t* khat, bf16_t* vwt, bf16_t* wqf);
int k_cross_fold(hipStream_t s, const CrossFoldArgs& a);
int k_cross_fold_null_row(hipStream_t s, const bf16_t* vwt, int m, float* out);      // [512] fp32: what the kernel adds to a row whose text keys are all masked

// cross_vw_x2.hip: the 'f16x2' tier's cross-attention behind its q projection (scores on the fp32 MFMA, P . (V W_o^T) as fp16 term products, residual add) as one kernel
struct CrossVwArgs {
    const float* q; long ldq;                    // [seqs * nq][ldq] fp32: the q projection's output (un-normalised)
    const float* khat;                           // k_cross_vw_x2_pack: K^ [kv_seqs][8][3][4][64][4] fp32
    const bf16_t* vwt;                           // k_cross_vw_x2_pack: the two fp16 terms of 2^8 (V W_o^T)^T as fragments [kv_seqs][2][32][9][64][8]
    const uint8_t* key_mask; long km_sb;         // optional [seqs][m], 1 = keep
    const float* q_scale;                        // [64]
    float* x; long ldx;                          // fp32 residual stream [seqs * nq][ldx], updated in place
    int seqs, nq, m, kv_batch_mod;
    float scale;                                 // 8
    // QP form (wqf != NULL): the block's LayerNorm and q projection run inside the kernel, q / ldq unused
    const bf16_t* wqf;                           // k_cross_vw_x2_wq_pack: the q weight's term planes as fragments [2][8 heads][4][16][64][8] fp16 (plane 1 read iff wq_terms == 3)
    int wq_terms;                                // 2: single fp16 weight terms (a bf16-representable checkpoint), 3: [wh | wh | wl]
    float alpha;                                 // inverse of the weight terms' power-of-two scale (mm_transformer::alpha)
    const float* ln_gamma; const float* ln_beta; // [512] the cross-attention's LayerNorm (beta may be NULL)
};
size_t k_cross_vw_x2_wqf_halves();
int k_cross_vw_x2_wq_pack(hipStream_t s, const bf16_t* w_q, int ldw, int terms, bf16_t* wqf);
bool k_cross_vw_x2_eligible(int D, int I, int H, int dh, int m);
size_t k_cross_vw_x2_khat_floats(int kv_seqs);
size_t k_cross_vw_x2_vwt_halves(int kv_seqs);
int k_cross_vw_x2_pack(hipStream_t s, const float* ckv, int kv_seqs, int m, int I, const float* null_k, const float* null_v, const float* k_scale,
                       const bf16_t* w_out, int ldw, int terms, float alpha, float* khat, bf16_t* vwt);
int k_cross_vw_x2(hipStream_t s, const CrossVwArgs& a);

// vq.hip
int k_vq_nearest(hipStream_t s, const float* x, long ldx, int N, int C, const float* cb, int K, int cosine, float* aux, int64_t* ids);
int k_vq_gather(hipStream_t s, const int64_t* ids, long N, int C, const float* cb, float* out);

// train_prep.hip: the fp32 master weights of a training step -> their bf16 operand copies (optionally padded / placed inside a concatenated operand) and the
// transposed copies, one 64 x 64 tile per workgroup, a table of jobs per launch
struct PrepJob {
    const float* src;      // fp32 [rows][cols], dense
    void* dst;             // bf16 [rows_p][ld_d] (first cols_p columns written; zeros outside the source) or nullptr;  kind 1: fp32 [cols_p]
    bf16_t* dst_t;         // bf16 [cols_p][ld_t] (first rows_p columns written) or nullptr
    int rows, cols, rows_p, cols_p, ld_d, ld_t, tile0, kind;
};
constexpr int PREP_MAX_JOBS = 64;
struct PrepArgs {
    PrepJob job[PREP_MAX_JOBS];
    int njobs;
};
struct PrepList {
    PrepArgs a;
    hipStream_t s;
    int tiles = 0, rc = 0;
    explicit PrepList(hipStream_t stream) : s(stream) { a.njobs = 0; }
    void add(const float* src, void* dst, bf16_t* dst_t, int rows, int cols, int rows_p, int cols_p, int ld_d, int ld_t, int kind = 0);
    int flush();           // launches what has been added (a full table is launched by add()); returns the first error of the list
};

// gemm_tn.hip: dW [N][K] fp32 = dY^T X for row-major dY [rows][lda], X [rows][ldb] (no transposed copies: transposing LDS reads); splits > 1: one slab per split
bool k_gemm_tn_eligible(int rows, int N, int K, long lda, long ldb);
int k_gemm_tn_splits(int rows, int N, int K);
bool k_gemm_tn_prefer(int rows, int N, int K, long lda, long ldb);      // the training step's rule: this kernel instead of transposed copies + the NT GEMM
int k_gemm_tn(hipStream_t s, const bf16_t* A, long lda, const bf16_t* B, long ldb, int rows, int N, int K, int splits, float* out_or_slabs);

// train.hip / attention_bwd.hip: backward operators
int k_transpose_bf16(hipStream_t s, const bf16_t* in, long rows, long cols, long ldi, bf16_t* out, long ldo, int zero_pad64 = 0);   // zero_pad64: also write zeros up to rows rounded to 64 (needs ldo >= that)
int k_colsum(hipStream_t s, const float* part, int nparts, long D, float* out);
long k_ln_bwd_workspace_floats(int rows, int D);
int k_layernorm_bwd(hipStream_t s, const float* x, long ldx, const bf16_t* dy, long lddy, const float* gamma, const int32_t* row_index,
                    int rows, int D, float* dx, long lddx, int accumulate, float* dgamma, float* ws, bf16_t* dxb = nullptr);
int k_ln_bwd_blocks(int rows);
int k_geglu_ln_bwd(hipStream_t s, const bf16_t* h, long ldh, const bf16_t* dz, long lddz, const float* gamma, int rows, int F, int Fp,
                   bf16_t* dh, long lddh, float* dgamma, float* ws);
int k_ce_bwd(hipStream_t s, const float* logits, long ld, int R, int V, const int64_t* labels, float scale, bf16_t* dl, long ldd, float* row_loss = nullptr);
int k_ce_finish(hipStream_t s, const float* row_loss, int R, float* out);      // sampling.hip: mean of the row losses >= 0
int k_bce_head_bwd(hipStream_t s, const bf16_t* e, long lde, const float* x, const float* y, const float* w, int rows, int D, bf16_t* de,
                   long ldde, float* dw, float* ws);
int k_embed_bwd(hipStream_t s, const int64_t* ids, int B, int n, int D, const float* dx, float* dtoken, float* dpos, void* ws = nullptr);   // ws: k_embed_bwd_workspace_bytes -> the two-level sum
size_t k_embed_bwd_workspace_bytes(int B, int n, int D);
int k_sum_parts_bf16(hipStream_t s, const bf16_t* parts, int P, long n, bf16_t* out);
int k_scatter_rows_bf16(hipStream_t s, const bf16_t* src, const int32_t* row_index, int R, int D, bf16_t* dst);
int k_attention_bwd(hipStream_t s, const bf16_t* q, long q_sb, long q_sh, long q_sn, const bf16_t* k, long k_sb, long k_sh, long k_sn,
                    const bf16_t* v, long v_sb, long v_sh, long v_sn, const bf16_t* o, long o_sb, long o_sh, long o_sn,
                    const bf16_t* dout, long do_sb, long do_sh, long do_sn, bf16_t* dqn, long dq_sb, long dq_sh, long dq_sn,
                    bf16_t* dkn, long dk_sb, long dk_sh, long dk_sn, bf16_t* dv, long dv_sb, long dv_sh, long dv_sn, float* dnk, float* dnv,
                    int B, int H, int nq, int nk, const uint8_t* key_mask, long km_sb, const float* q_scale, const float* k_scale,
                    const float* null_k, const float* null_v, float scale);
long k_qk_norm_bwd_blocks(long nvec);
int k_qk_norm_bwd(hipStream_t s, const bf16_t* x, long ldx, const float* x_f32, int H, const bf16_t* dy, long lddy, const float* dy_f32,
                  const float* scale, long rows, int heads_per_row, bf16_t* dx, long lddx, float* dx_f32, float* dscale_part);

// ---- 'bf16x3' precision tier (split.hip, attention_f32.hip): fp32 values as P bf16 segments [h|m|l|h|m|h][:P] per row
int k_split_rows(hipStream_t s, const float* x, long ldx, long rows, int K, int P, int rows_per_batch, long out_batch_stride, bf16_t* out,
                 uint8_t* nz_mask, int mask_bstride, int drop);
int k_gather_split(hipStream_t s, const float* table, int D, int P, const int64_t* idx, int B, int nc, int vocab_rows, bf16_t* ctx, uint8_t* mask,
                   int m, int L);
// out (P segments, optional) / out_f32 (optional) = LayerNorm(x[row_index ? row_index[r] : r]); addvec != NULL: rows >= add_from get addvec added in place (xw = x) first
int k_layernorm_split(hipStream_t s, const float* x, long ldx, int rows, int D, const float* gamma, const float* beta, const int32_t* row_index,
                      int P, bf16_t* out, float* out_f32, const float* addvec, int add_from, float* xw);
int k_geglu_ln_split(hipStream_t s, const float* h, long ldh, int rows, int F, int Fp, const float* gamma, const float* beta, int P, bf16_t* out);
int k_embed_f32(hipStream_t s, const int64_t* ids, long rows, int n, const float* tok, int vocab_rows, const float* pos, int D, float* x);
// e = en + (ec - en) * s on embedding rows: bf16 (P == 0, out [rows][D]) or term-segment packs (out [rows][P * D])
int k_cfg_mix(hipStream_t s, const bf16_t* ec, const bf16_t* en, long ld, long rows, int D, int P, float cond_scale, bf16_t* out);
struct AttnF32Args {
    const void* q; long q_sb, q_sh, q_sn;        // element strides: batch, head, token (d contiguous); fp32, or bf16 when io_bf16
    const void* k; long k_sb, k_sh, k_sn;
    const void* v; long v_sb, v_sh, v_sn;
    void* out; long o_sb, o_sh, o_sn;            // optional output in the operand type
    bf16_t* out_split; long os_sb, os_sn; int os_seg, P;   // optional P-segment output: element (b, token, h, d) of segment s at b*os_sb + token*os_sn + s*os_seg + h*64 + d
    int B, H, nq, nk;
    const uint8_t* key_mask; long km_sb;
    int normalize;
    const float* q_scale; const float* k_scale; const float* null_k; const float* null_v;
    float scale;
    int kv_batch_mod;
    int dh;                                      // dim_head: 32, 64 (0 = 64) or 128
    int io_bf16;                                 // q / k / v / out are bf16 (the bf16 engine's route for dim_head != 64)
};
int k_attention_f32(hipStream_t s, const AttnF32Args& a);
// attention_x2.hip (round 5): the same attention as fp16 TERM PRODUCTS on the fp16 matrix pipe (fp32 q / k / v split in registers, all keys resident): dim_head 64,
// nk in {128, 192, 256}, no key mask; fp16-term segments out (the 'f16x2' tier) or fp32 out (operator entry mm_attend_terms)
bool k_attention_x2_eligible(const AttnF32Args& a);
int k_attention_x2(hipStream_t s, const AttnF32Args& a);

int k_mask_step(hipStream_t s, float* scores, int64_t* ids, int B, int n, int k, int64_t mask_id, int32_t* rows_out);
struct SampleArgs {
    const float* logits; long ld;                // [R][V] CFG-combined logits of the gathered rows
    int R, V, k_keep;                            // keep the k largest per row
    const int32_t* rows;                         // [R] flat position b*n + pos of each gathered row
    float inv_temperature_divisor;               // unused placeholder (division is IEEE, see sampling.hip)
    float temperature;                           // already clamped to >= 1e-10
    int noise_kind;                              // MM_NOISE_*
    const float* noise; long noise_ld;           // indexed by flat position: noise[(b*n+pos)*noise_ld + v]
    uint64_t seed; uint64_t row_offset; uint32_t step;
    int64_t* ids; float* scores;                 // scattered outputs, indexed by flat position
    int64_t* pred_out; float* score_out;         // optional compact outputs [R]
    int debug;                                   // ablation bits 16 / 32 / 64 / 128 (tools/sample_bench.py)
    float z_lo;                                  // histogram lower bound in sigmas above the row mean (set by k_sample_rows)
    // per-row fallback of the fused sampler: logits row i belongs to original row src_rows[i] (whose position / compact output slot / noise
    // stream it samples for), and only the first min(R, *count_dev) rows exist
    const int32_t* src_rows; const int32_t* count_dev;
    // optional device-side Philox keys (mm_generate_params.seed_dev): {seed, row offset in SAMPLES}; when set they replace seed / row_offset (= seed_dev[1] * row_mul)
    // at execution time, so a captured launch replays with fresh noise
    const uint64_t* seed_dev; int row_mul;
};
int k_sample_rows(hipStream_t s, const SampleArgs& a);
// sampling_fused.hip: sampling from what the guidance-logits GEMM emits instead of the logits (tile statistics + candidates)
struct FusedSampleArgs {
    const float* thr;                            // [R] the lower bound the candidates were emitted with (lower edge of the select histogram)
    const float4* stats;                         // [R][V/256]: {tile max, sum exp(x - tile max), mask of the kept lanes (2 x 32 bits)}
    const float4* cand;                          // [R][V/256][FS_SLOT]: the kept lanes' 4 values each, lane-compacted
    int R, V, k_keep;
    const int32_t* rows;                         // [R] flat position of each row (default: r)
    float temperature; int noise_kind; const float* noise; long noise_ld;
    uint64_t seed; uint64_t row_offset; uint32_t step;
    int64_t* ids; float* scores; int64_t* pred_out; float* score_out;
    int* fail_flag;                              // set to 1 when a row's candidate set cannot be proven complete (caller falls back to the logits path)
    // optional on-device fallback list: a failing row is appended to fail_rows[atomicAdd(fail_count, 1)] (capacity fail_cap) instead of raising the
    // flag; the flag is raised only when the list overflows.  The caller finishes the listed rows on the logits path (model.hip).
    int32_t* fail_rows; int32_t* fail_count; int fail_cap;
    int debug;                                   // set by k_sample_fused from mm_debug_set (bit 1 << 27: test hook of the fallback)
    const uint64_t* seed_dev; int row_mul;       // optional device-side Philox keys, see SampleArgs
};
float k_fused_z(int k_keep, int V, float margin);
// distribution-free bound: thr[r] = the rank-th largest of the S sampled logits sub[r][0..S) (rank from k_fused_quantile_rank: S k / V + 4.5 standard deviations)
int k_fused_quantile_rank(int k_keep, int V, int S);
int k_fused_quantile(hipStream_t s, const float* sub, long ld, int R, int S, int rank, float* thr);
// thr[r] = mean_r + z sigma_r of row r's logits over the vocabulary; ws: k_fused_threshold_ws_bytes(R, D) bytes of scratch; wcov bf16 [D][D]
size_t k_fused_threshold_ws_bytes(int R, int D);
float* k_fused_threshold_mu(void* ws, int R, int D);      // where the rows' means live in that scratch (k_final_mix writes them; then k_fused_threshold_mixed)
int k_fused_threshold_mixed(hipStream_t s, const bf16_t* e, long ld, int R, int D, const bf16_t* wcov, float z, void* ws, float* thr);      // e = the mixed rows, means already in ws
int k_fused_threshold(hipStream_t s, const bf16_t* ec, const bf16_t* en, long ld, int R, int D, float cond_scale, const float* wmean, const bf16_t* wcov,
                      float z, void* ws, float* thr);
int k_fused_emit(hipStream_t s, const float* logits, long ld, int R, int V, const float* thr, float4* stats, float4* cand);
int k_sample_fused(hipStream_t s, const FusedSampleArgs& a);
int k_philox_fill(hipStream_t s, uint64_t seed, uint64_t row_offset, uint32_t step, int rows, int V, float* out);
int k_text_context(hipStream_t s, const float* text, int rows, int text_dim, int L, int m, bf16_t* out_bf16, long ldo,
                   uint8_t* mask, int drop_text);
int k_gather_rows_bf16(hipStream_t s, const bf16_t* table, int D, const int64_t* idx, int B, int nc, int vocab_rows,
                       bf16_t* out, long out_batch_stride, long out_row_offset, uint8_t* mask, int m, int L);

int k_ce_loss(hipStream_t s, const float* logits, long ld, int R, int V, const int64_t* labels, int64_t ignore_index,
              float* row_loss_ws, float* out);
int k_bce_loss(hipStream_t s, const float* x, const float* y, int n, float* out);

int mm_conv2d_nhwc_head(mm_stream_t stream, const void* in, int B, int Hin, int Win, int Cin, const void* w, int Cout, int TH, int TW, int stride, int off_y, int off_x,
                        int Hv, int Wv, int os, int py, int px, int Hout, int Wout, const float* bias, int act, const void* head_w, int head_ldw, const float* head_b,
                        int head_c, float* image, int half, float alpha);      // api.hip (round 6): convolution + the fused 1 x 1 head
int mm_convT2d_nhwc_4(mm_stream_t stream, const void* in, int B, int Hin, int Win, int Cin, const void* const* w4, int Cout, const float* bias, int act, void* out,
                      const void* head_w, int head_ldw, const float* head_b, int head_c, int half, float alpha);      // api.hip (round 6): the four parity classes in one launch
// vae kernels
int k_lfq_decode(hipStream_t s, const int64_t* ids, long count, int bits, int C, const float* w, const float* b, bf16_t* out, int half = 0);      // half: fp16 storage (round 6)
int k_lfq_encode(hipStream_t s, const bf16_t* x, long count, int C, int bits, const float* w, const float* b,
                 const float* wo, const float* bo, int64_t* ids, bf16_t* out);
int k_glu(hipStream_t s, const bf16_t* x, long rows, int C, bf16_t* out, int half = 0);
int k_groupnorm(hipStream_t s, const bf16_t* x, int B, int HW, int C, int groups, const float* gamma, const float* beta,
                int act, float* stats_ws, bf16_t* out, int half = 0);
int k_nchw_to_nhwc8(hipStream_t s, const float* img, int B, int C, int H, int W, bf16_t* out);
int k_nhwc_to_nchw_f32(hipStream_t s, const bf16_t* x, int B, int C, int H, int W, float* out);
