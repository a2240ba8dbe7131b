// C-ABI layer: argument validation + error plumbing around the kernel launchers.  See include/muse_hip.h.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "muse_hip_internal.h"

namespace {
thread_local char g_err[512] = "";
}

int mm_set_error(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
    return code;
}

int mm_set_hip_error(hipError_t e, const char* where) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where ? where : "hip", hipGetErrorString(e));
    return MM_ERR_HIP;
}

int mm_check_launch(const char* kernel) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mm_set_hip_error(e, kernel);
    return MM_OK;
}

#define CHK_PTR(p, name) \
    if (!(p)) return mm_set_error(MM_ERR_SHAPE, name " is NULL")
#define CHK_ALIGN16(p, name) \
    if (((uintptr_t)(p)) & 15) return mm_set_error(MM_ERR_ALIGN, name " must be 16-byte aligned")

extern "C" {

int mm_abi_version(void) { return MM_ABI_VERSION; }
int mm_debug_set(int flags) { g_mm_debug = flags; return MM_OK; }
int mm_debug_set2(int flags) { g_mm_debug2 = flags; return MM_OK; }
const char* mm_last_error(void) { return g_err; }

int mm_device_check(void) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return mm_set_hip_error(e, "hipGetDevice");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return mm_set_hip_error(e, "hipGetDeviceProperties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_err, sizeof(g_err), "libmuse_hip is built for gfx950 (MI355X); current device is %s", prop.gcnArchName);
        return MM_ERR_ARCH;
    }
    return MM_OK;
}

int mm_gemm_bf16(mm_stream_t stream, const void* x, int64_t ldx, const void* w, int64_t ldw, int M, int N, int K,
                 void* out, int64_t ldc, int out_f32, const float* resid_f32) {
    if (M == 0 || N == 0) return MM_OK;
    CHK_PTR(x, "x"); CHK_PTR(w, "w"); CHK_PTR(out, "out");
    CHK_ALIGN16(x, "x"); CHK_ALIGN16(w, "w"); CHK_ALIGN16(out, "out");
    if (M < 0 || N < 0) return mm_set_error(MM_ERR_SHAPE, "gemm: negative size");
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = MODE_DENSE;
    a.W = (const bf16_t*)w; a.N = N; a.ldw = (int)ldw; a.K = K;
    a.M = M; a.X = (const bf16_t*)x; a.ldx = (int)ldx;
    a.out = out; a.ldc = ldc; a.out_kind = out_f32 ? OUT_F32 : OUT_BF16;
    a.resid_f32 = resid_f32; a.ldr = ldc;
    return mm_gemm_launch(a, (hipStream_t)stream);
}

// precision tiers: operand rows are term-segment packs (mm_split_rows) -> fp32 out.  products = 3 / 5 / 6: bf16 terms on the bf16 MFMA (== mm_gemm_bf16);
// MM_SPLIT_F16 | 2 / 3: fp16 terms on the fp16 MFMA, the accumulators multiplied by alpha (inverse of the power-of-two scale of the packed weight terms)
int mm_gemm_split(mm_stream_t stream, const void* x, int64_t ldx, const void* w, int64_t ldw, int M, int N, int K, int products, float alpha,
                  float* out, int64_t ldc, const float* resid_f32) {
    if (M == 0 || N == 0) return MM_OK;
    CHK_PTR(x, "x"); CHK_PTR(w, "w"); CHK_PTR(out, "out");
    CHK_ALIGN16(x, "x"); CHK_ALIGN16(w, "w"); CHK_ALIGN16(out, "out");
    if (M < 0 || N < 0) return mm_set_error(MM_ERR_SHAPE, "gemm: negative size");
    const bool f16 = split_is_f16(products);
    const int cnt = split_count(products);
    if ((f16 && cnt != 2 && cnt != 3) || (!f16 && cnt != 3 && cnt != 5 && cnt != 6)) return mm_set_error(MM_ERR_SHAPE, "gemm_split: bad products code");
    if (!f16 && alpha != 0.f && alpha != 1.f) return mm_set_error(MM_ERR_SHAPE, "gemm_split: alpha applies to fp16 terms only");
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = MODE_DENSE;
    a.W = (const bf16_t*)w; a.N = N; a.ldw = (int)ldw; a.K = K;
    a.M = M; a.X = (const bf16_t*)x; a.ldx = (int)ldx;
    a.out = out; a.ldc = ldc; a.out_kind = OUT_F32;
    a.resid_f32 = resid_f32; a.ldr = ldc;
    a.f16 = f16 ? 1 : 0; a.alpha = alpha;
    a.terms = (f16 && (products & 0x200)) ? cnt : 0;      // MM_SPLIT_SHARED: genuine term-segment packs -- the term-sharing kernels may take it (gemm_terms.hip)
    return mm_gemm_launch(a, (hipStream_t)stream);
}

int mm_gemm_split_geglu(mm_stream_t stream, const void* x, int64_t ldx, const void* w, int64_t ldw, int M, int N, int K, int products, float alpha,
                        void* out_split, int64_t ldo, float* ln_part) {
    if (M == 0 || N == 0) return MM_OK;
    CHK_PTR(x, "x"); CHK_PTR(w, "w"); CHK_PTR(out_split, "out_split");
    CHK_ALIGN16(x, "x"); CHK_ALIGN16(w, "w"); CHK_ALIGN16(out_split, "out_split");
    if (M < 0 || N < 0) return mm_set_error(MM_ERR_SHAPE, "gemm: negative size");
    const int cnt = split_count(products);
    if (!split_is_f16(products) || (cnt != 2 && cnt != 3)) return mm_set_error(MM_ERR_SHAPE, "gemm_split_geglu: fp16 term operands only (MM_SPLIT_F16 | 2 / 3)");
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = MODE_DENSE; a.epi = EPI_GEGLU;
    a.W = (const bf16_t*)w; a.N = N; a.ldw = (int)ldw; a.K = K;
    a.M = M; a.X = (const bf16_t*)x; a.ldx = (int)ldx;
    a.out = out_split; a.ldc = ldo; a.out_kind = OUT_BF16;
    a.ln_part = ln_part; a.ln_np = N / 64;
    a.f16 = 1; a.alpha = alpha; a.terms = cnt;
    if (!mm_gemm_terms_eligible(a)) return mm_set_error(MM_ERR_UNSUPPORTED, "gemm_split_geglu: outside the term-sharing kernel's shape class");
    return mm_gemm_launch(a, (hipStream_t)stream);
}

int mm_gemm_cfg_logits(mm_stream_t stream, const void* x_cond, const void* x_null, int64_t ldx, const void* w,
                       int64_t ldw, int M, int N, int K, float* out, int64_t ldc, float cond_scale) {
    if (M == 0 || N == 0) return MM_OK;
    CHK_PTR(x_cond, "x_cond"); CHK_PTR(x_null, "x_null"); CHK_PTR(w, "w"); CHK_PTR(out, "out");
    CHK_ALIGN16(x_cond, "x_cond"); CHK_ALIGN16(x_null, "x_null"); CHK_ALIGN16(w, "w"); CHK_ALIGN16(out, "out");
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = MODE_CFG;
    a.W = (const bf16_t*)w; a.N = N; a.ldw = (int)ldw; a.K = K;
    a.M = M; a.X = (const bf16_t*)x_cond; a.X2 = (const bf16_t*)x_null; a.ldx = (int)ldx;
    a.out = out; a.ldc = ldc; a.out_kind = OUT_F32; a.cfg_scale = cond_scale;
    return mm_gemm_launch(a, (hipStream_t)stream);
}

float mm_fused_z(int k_keep, int V, float margin) { return k_fused_z(k_keep, V, margin); }
int mm_fused_quantile_rank(int k_keep, int V, int S) { return k_fused_quantile_rank(k_keep, V, S); }
int mm_fused_quantile(mm_stream_t stream, const float* sub, int64_t ld, int R, int S, int rank, float* thr) {
    if (R == 0) return MM_OK;
    CHK_PTR(sub, "sub"); CHK_PTR(thr, "thr");
    return k_fused_quantile((hipStream_t)stream, sub, (long)ld, R, S, rank, thr);
}
size_t mm_fused_threshold_workspace_bytes(int R, int D) { return k_fused_threshold_ws_bytes(R, D); }

int mm_fused_threshold(mm_stream_t stream, const void* emb_cond, const void* emb_null, int64_t ld, int R, int D, float cond_scale, const float* wmean,
                       const void* wcov, float z, void* ws, float* thr) {
    if (R == 0) return MM_OK;
    CHK_PTR(emb_cond, "emb_cond"); CHK_PTR(emb_null, "emb_null"); CHK_PTR(wmean, "wmean"); CHK_PTR(wcov, "wcov"); CHK_PTR(ws, "ws"); CHK_PTR(thr, "thr");
    CHK_ALIGN16(wcov, "wcov"); CHK_ALIGN16(ws, "ws");
    return k_fused_threshold((hipStream_t)stream, (const bf16_t*)emb_cond, (const bf16_t*)emb_null, (long)ld, R, D, cond_scale, wmean, (const bf16_t*)wcov, z, ws, thr);
}

int mm_gemm_cfg_logits_fused(mm_stream_t stream, const void* x_cond, const void* x_null, int64_t ldx, const void* w, int64_t ldw, int M, int N, int K,
                             float cond_scale, const float* thr, void* stats, void* cand) {
    if (M == 0 || N == 0) return MM_OK;
    CHK_PTR(x_cond, "x_cond"); CHK_PTR(w, "w"); CHK_PTR(thr, "thr"); CHK_PTR(stats, "stats"); CHK_PTR(cand, "cand");
    CHK_ALIGN16(x_cond, "x_cond"); CHK_ALIGN16(x_null, "x_null"); CHK_ALIGN16(w, "w"); CHK_ALIGN16(stats, "stats"); CHK_ALIGN16(cand, "cand");
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    // x_null == NULL: x_cond holds the already MIXED embeddings (mm_cfg_mix) -- the single-pass form mm_generate uses; cond_scale is ignored
    a.mode = x_null ? MODE_CFG : MODE_DENSE;
    a.wide_tok = x_null ? 0 : 1;
    a.W = (const bf16_t*)w; a.N = N; a.ldw = (int)ldw; a.K = K;
    a.M = M; a.X = (const bf16_t*)x_cond; a.X2 = (const bf16_t*)x_null; a.ldx = (int)ldx;
    a.out = nullptr; a.ldc = N; a.out_kind = OUT_F32; a.cfg_scale = cond_scale;
    a.fs_thr = thr; a.fs_stats = (float4*)stats; a.fs_cand = (float4*)cand;
    return mm_gemm_launch(a, (hipStream_t)stream);
}

int mm_fused_emit(mm_stream_t stream, const float* logits, int64_t ld, int R, int V, const float* thr, void* stats, void* cand) {
    if (R == 0) return MM_OK;
    CHK_PTR(logits, "logits"); CHK_PTR(thr, "thr"); CHK_PTR(stats, "stats"); CHK_PTR(cand, "cand");
    CHK_ALIGN16(logits, "logits"); CHK_ALIGN16(stats, "stats"); CHK_ALIGN16(cand, "cand");
    if (ld % 4) return mm_set_error(MM_ERR_ALIGN, "fused_emit: ld must be a multiple of 4");
    return k_fused_emit((hipStream_t)stream, logits, (long)ld, R, V, thr, (float4*)stats, (float4*)cand);
}

int mm_fused_sample(mm_stream_t stream, const float* thr, const void* stats, const void* cand, int R, int V, int k_keep, const int32_t* rows,
                    float temperature, int noise_kind, const float* noise, int64_t noise_ld, uint64_t seed, uint64_t row_offset, uint32_t step,
                    int64_t* ids, float* scores, int64_t* pred_out, float* score_out, int32_t* fail_flag, int32_t* fail_rows, int32_t* fail_count,
                    int fail_cap) {
    if (R == 0) return MM_OK;
    CHK_PTR(thr, "thr"); CHK_PTR(stats, "stats"); CHK_PTR(cand, "cand"); CHK_PTR(fail_flag, "fail_flag");
    if ((fail_rows == nullptr) != (fail_count == nullptr) || (fail_rows && fail_cap <= 0)) return mm_set_error(MM_ERR_SHAPE, "fused_sample: fail_rows / fail_count / fail_cap go together");
    FusedSampleArgs a;
    memset(&a, 0, sizeof(a));
    a.thr = thr; a.stats = (const float4*)stats; a.cand = (const float4*)cand;
    a.R = R; a.V = V; a.k_keep = k_keep; a.rows = rows; a.temperature = temperature; a.noise_kind = noise_kind; a.noise = noise; a.noise_ld = (long)noise_ld;
    a.seed = seed; a.row_offset = row_offset; a.step = step; a.ids = ids; a.scores = scores; a.pred_out = pred_out; a.score_out = score_out;
    a.fail_flag = fail_flag; a.fail_rows = fail_rows; a.fail_count = fail_count; a.fail_cap = fail_cap;
    return k_sample_fused((hipStream_t)stream, a);
}

int mm_embed(mm_stream_t stream, const int64_t* ids, int rows, int n, const void* token_emb, int vocab_rows,
             const void* pos_emb, int dim, float* x) {
    if (rows == 0) return MM_OK;
    CHK_PTR(ids, "ids"); CHK_PTR(token_emb, "token_emb"); CHK_PTR(pos_emb, "pos_emb"); CHK_PTR(x, "x");
    CHK_ALIGN16(token_emb, "token_emb"); CHK_ALIGN16(pos_emb, "pos_emb"); CHK_ALIGN16(x, "x");
    if (n <= 0) return mm_set_error(MM_ERR_SHAPE, "embed: n <= 0");
    return k_embed((hipStream_t)stream, ids, rows, n, 0, (const bf16_t*)token_emb, vocab_rows, (const bf16_t*)pos_emb, dim, x);
}

int mm_layernorm(mm_stream_t stream, const float* x, int64_t ldx, int rows, int dim, const float* gamma,
                 const float* beta, const int32_t* row_index, void* out, int64_t ldo) {
    if (rows == 0) return MM_OK;
    CHK_PTR(x, "x"); CHK_PTR(gamma, "gamma"); CHK_PTR(out, "out");
    CHK_ALIGN16(x, "x"); CHK_ALIGN16(gamma, "gamma");
    return k_layernorm((hipStream_t)stream, x, ldx, rows, dim, gamma, beta, row_index, (bf16_t*)out, ldo);
}

int mm_geglu_ln(mm_stream_t stream, const void* h, int64_t ldh, int rows, int F, int Fp, const float* gamma,
                const float* beta, void* out, int64_t ldo) {
    if (rows == 0) return MM_OK;
    CHK_PTR(h, "h"); CHK_PTR(gamma, "gamma"); CHK_PTR(out, "out");
    CHK_ALIGN16(h, "h"); CHK_ALIGN16(out, "out");
    return k_geglu_ln((hipStream_t)stream, (const bf16_t*)h, ldh, rows, F, Fp, gamma, beta, (bf16_t*)out, ldo);
}

int mm_gemm_geglu(mm_stream_t stream, const void* x, int64_t ldx, const void* w1, int64_t ldw, int M, int Fp, int K,
                  void* out, int64_t ldc) {
    if (M == 0) return MM_OK;
    CHK_PTR(x, "x"); CHK_PTR(w1, "w1"); CHK_PTR(out, "out");
    CHK_ALIGN16(x, "x"); CHK_ALIGN16(w1, "w1"); CHK_ALIGN16(out, "out");
    if (Fp <= 0 || Fp % 64) return mm_set_error(MM_ERR_SHAPE, "gemm_geglu: Fp must be a positive multiple of 64");
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = MODE_DENSE; a.epi = EPI_GEGLU;
    a.W = (const bf16_t*)w1; a.N = 2 * Fp; a.ldw = (int)ldw; a.K = K;
    a.M = M; a.X = (const bf16_t*)x; a.ldx = (int)ldx;
    a.out = out; a.ldc = ldc; a.out_kind = OUT_BF16;
    return mm_gemm_launch(a, (hipStream_t)stream);
}

int mm_layernorm_inner(mm_stream_t stream, const void* a, int64_t lda, int rows, int F, int Fp, const float* gamma,
                       const float* beta, void* out, int64_t ldo) {
    if (rows == 0) return MM_OK;
    CHK_PTR(a, "a"); CHK_PTR(gamma, "gamma"); CHK_PTR(out, "out");
    CHK_ALIGN16(a, "a"); CHK_ALIGN16(out, "out");
    return k_ln_bf16((hipStream_t)stream, (const bf16_t*)a, lda, rows, F, Fp, gamma, beta, (bf16_t*)out, ldo);
}

int mm_attend(mm_stream_t stream, const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const void* k,
              int64_t k_sb, int64_t k_sh, int64_t k_sn, const void* v, int64_t v_sb, int64_t v_sh, int64_t v_sn,
              void* out, int64_t o_sb, int64_t o_sh, int64_t o_sn, int B, int H, int nq, int nk,
              const uint8_t* key_mask, int64_t km_sb, int normalize, const float* q_scale, const float* k_scale,
              const float* null_k, const float* null_v, float scale, int dim_head) {
    CHK_PTR(q, "q"); CHK_PTR(k, "k"); CHK_PTR(v, "v"); CHK_PTR(out, "out");
    CHK_ALIGN16(q, "q"); CHK_ALIGN16(k, "k"); CHK_ALIGN16(v, "v");
    if (dim_head != 32 && dim_head != 64 && dim_head != 128) return mm_set_error(MM_ERR_UNSUPPORTED, "attend: dim_head must be 32, 64 or 128");
    if ((null_k == nullptr) != (null_v == nullptr)) return mm_set_error(MM_ERR_SHAPE, "attend: null_k and null_v go together");
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = (const bf16_t*)q; a.q_sb = q_sb; a.q_sh = q_sh; a.q_sn = q_sn;
    a.k = (const bf16_t*)k; a.k_sb = k_sb; a.k_sh = k_sh; a.k_sn = k_sn;
    a.v = (const bf16_t*)v; a.v_sb = v_sb; a.v_sh = v_sh; a.v_sn = v_sn;
    a.out = (bf16_t*)out; a.o_sb = o_sb; a.o_sh = o_sh; a.o_sn = o_sn;
    a.B = B; a.H = H; a.nq = nq; a.nk = nk;
    a.key_mask = key_mask; a.km_sb = km_sb;
    a.normalize = normalize; a.q_scale = q_scale; a.k_scale = k_scale;
    a.null_k = null_k; a.null_v = null_v; a.scale = scale; a.kv_batch_mod = 0; a.dh = dim_head;
    return k_attention((hipStream_t)stream, a);
}

int mm_mask_step(mm_stream_t stream, float* scores, int64_t* ids, int B, int n, int k, int64_t mask_id,
                 int32_t* rows_out) {
    if (B == 0) return MM_OK;
    CHK_PTR(scores, "scores"); CHK_PTR(ids, "ids");
    return k_mask_step((hipStream_t)stream, scores, ids, B, n, k, mask_id, rows_out);
}

int mm_sample_rows(mm_stream_t stream, const float* logits, int64_t ld, int R, int V, int k_keep,
                   const int32_t* rows, float temperature, int noise_kind, const float* noise, int64_t noise_ld,
                   uint64_t seed, uint64_t row_offset, uint32_t step, int64_t* ids, float* scores,
                   int64_t* pred_out, float* score_out) {
    if (R == 0) return MM_OK;
    CHK_PTR(logits, "logits"); CHK_ALIGN16(logits, "logits");
    if (noise) CHK_ALIGN16(noise, "noise");
    if (noise_kind < MM_NOISE_NONE || noise_kind > MM_NOISE_PHILOX) return mm_set_error(MM_ERR_SHAPE, "sample_rows: bad noise_kind");
    SampleArgs a;
    memset(&a, 0, sizeof(a));
    a.logits = logits; a.ld = ld; a.R = R; a.V = V; a.k_keep = k_keep; a.rows = rows;
    a.temperature = temperature; a.noise_kind = noise_kind; a.noise = noise; a.noise_ld = noise_ld;
    a.seed = seed; a.row_offset = row_offset; a.step = step;
    a.ids = ids; a.scores = scores; a.pred_out = pred_out; a.score_out = score_out;
    return k_sample_rows((hipStream_t)stream, a);
}

int mm_ce_loss(mm_stream_t stream, const float* logits, int64_t ld, int R, int V, const int64_t* labels, int64_t ignore_index,
               float* row_loss_ws, float* out) {
    CHK_PTR(logits, "logits"); CHK_PTR(labels, "labels"); CHK_PTR(row_loss_ws, "row_loss_ws"); CHK_PTR(out, "out");
    CHK_ALIGN16(logits, "logits");
    return k_ce_loss((hipStream_t)stream, logits, ld, R, V, labels, ignore_index, row_loss_ws, out);
}

int mm_bce_loss(mm_stream_t stream, const float* x, const float* y, int n, float* out) {
    CHK_PTR(x, "x"); CHK_PTR(y, "y"); CHK_PTR(out, "out");
    return k_bce_loss((hipStream_t)stream, x, y, n, out);
}

int mm_quantize_act_e4m3(mm_stream_t stream, const void* x, int x_is_f32, int64_t ldx, int rows, int K, int Kp, void* xq, float* scale) {
    if (!x || !xq || !scale) return mm_set_error(MM_ERR_SHAPE, "quantize_act_e4m3: NULL pointer");
    return k_quantize_act_e4m3((hipStream_t)stream, x, x_is_f32, ldx, rows, K, Kp, (unsigned char*)xq, scale);
}

int mm_gemm_fp8(mm_stream_t stream, const void* xq, int64_t ldx, const float* x_scale, const void* wq, int64_t ldw, const float* w_scale, int M, int N, int K,
                void* out, int64_t ldc, int epilogue, const float* resid_f32) {
    if (epilogue < 0 || epilogue > 2) return mm_set_error(MM_ERR_SHAPE, "gemm_fp8: epilogue must be 0 (bf16), 1 (GEGLU, bf16) or 2 (fp32 + residual)");
    GemmF8Args a;
    memset(&a, 0, sizeof(a));
    a.X = (const unsigned char*)xq; a.ldx = ldx; a.sx = x_scale;
    a.W = (const unsigned char*)wq; a.ldw = ldw; a.sw = w_scale;
    a.M = M; a.N = N; a.K = K; a.out = out; a.ldc = ldc; a.epi = epilogue; a.resid = resid_f32; a.ldr = ldc;
    return k_gemm_fp8((hipStream_t)stream, a);
}

int mm_quantize_e4m3_rows(mm_stream_t stream, const float* w, int64_t ldw, int rows, int K, int Kp, void* wq, float* scale) {
    if (rows == 0) return MM_OK;
    CHK_PTR(w, "w"); CHK_PTR(wq, "wq"); CHK_PTR(scale, "scale");
    return k_quantize_e4m3_rows((hipStream_t)stream, w, ldw, rows, K, Kp, (unsigned char*)wq, scale);
}


int mm_vq_nearest(mm_stream_t stream, const float* x, int64_t ldx, int N, int C, const float* codebook, int K, int cosine,
                  float* aux_ws, int64_t* ids) {
    if (N == 0) return MM_OK;
    CHK_PTR(x, "x"); CHK_PTR(codebook, "codebook"); CHK_PTR(aux_ws, "aux_ws"); CHK_PTR(ids, "ids");
    if ((((uintptr_t)codebook) & 7) || (ldx < C)) return mm_set_error(MM_ERR_ALIGN, "vq_nearest: codebook must be 8-byte aligned, ldx >= C");
    return k_vq_nearest((hipStream_t)stream, x, ldx, N, C, codebook, K, cosine, aux_ws, ids);
}

int mm_vq_gather(mm_stream_t stream, const int64_t* ids, int64_t N, int C, const float* codebook, float* out) {
    if (N == 0) return MM_OK;
    CHK_PTR(ids, "ids"); CHK_PTR(codebook, "codebook"); CHK_PTR(out, "out"); CHK_ALIGN16(codebook, "codebook"); CHK_ALIGN16(out, "out");
    return k_vq_gather((hipStream_t)stream, ids, N, C, codebook, out);
}

int mm_gemm_wgrad_splits(int M, int N, int K) {
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    const int kt = K / 64;
    int s = 1;
    if (K >= 32768 && M >= 512 && N >= 128 && (N % 4) == 0) {
        // a LONG contraction into few tiles (round 6: the training head's dX over the vocabulary): 256 x 128 tiles (gemm_big.hip, twice the flops per staged byte),
        // enough splits for >= 2 workgroups per CU, >= 4096 of K per split.  (The 128 x 128 choice below gave 344 workgroups = 1.34 rounds on 256 CUs: 0.88 ms for
        // 369 GFLOP.)
        const long tb = (long)((M + 255) / 256) * ((N + 127) / 128);
        while (tb * s < 512 && (kt % (s * 2)) == 0 && kt / (s * 2) >= 64) s *= 2;
        return s;
    }
    const long fill = 256;      // (384: 15.0-15.5 ms per C2 training step, 256: 14.3-15.1, 192: 15.0-15.6; same box -- measured through an environment override that no longer exists)
    while (tiles * s < fill && (kt % (s * 2)) == 0 && kt / (s * 2) >= 8) s *= 2;      // fill the 256 CUs, keep >= 512 of K per workgroup
    return s;
}

int mm_gemm_wgrad(mm_stream_t stream, const void* x, int64_t ldx, const void* w, int64_t ldw, int M, int N, int K, int splits,
                  float* ws, float* out) {
    if (M == 0 || N == 0) return MM_OK;
    CHK_PTR(x, "x"); CHK_PTR(w, "w"); CHK_PTR(out, "out"); CHK_ALIGN16(x, "x"); CHK_ALIGN16(w, "w"); CHK_ALIGN16(out, "out");
    if (N % 4) return mm_set_error(MM_ERR_SHAPE, "gemm_wgrad: N must be a multiple of 4");
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = MODE_DENSE;
    a.W = (const bf16_t*)w; a.N = N; a.ldw = (int)ldw; a.K = K;
    a.M = M; a.X = (const bf16_t*)x; a.ldx = (int)ldx;
    a.ldc = N; a.out_kind = OUT_F32; a.ldr = N;
    if (splits <= 1) { a.out = out; return mm_gemm_launch(a, (hipStream_t)stream); }
    CHK_PTR(ws, "ws"); CHK_ALIGN16(ws, "ws");
    a.out = ws; a.splits = splits; a.split_stride = (long)M * N;
    int rc = mm_gemm_launch(a, (hipStream_t)stream);
    if (rc) return rc;
    return k_colsum((hipStream_t)stream, ws, splits, (long)M * N, out);
}

int mm_gemm_wgrad_tn_splits(int rows, int N, int K) { return k_gemm_tn_splits(rows, N, K); }
int mm_gemm_wgrad_tn_prefer(int rows, int N, int K, int64_t ldy, int64_t ldx) { return k_gemm_tn_prefer(rows, N, K, ldy, ldx) ? 1 : 0; }

int mm_gemm_wgrad_tn(mm_stream_t stream, const void* dy, int64_t ldy, const void* x, int64_t ldx, int rows, int N, int K, float* ws, float* out) {
    if (rows == 0 || N == 0 || K == 0) return MM_OK;
    CHK_PTR(dy, "dy"); CHK_PTR(x, "x"); CHK_PTR(out, "out"); CHK_ALIGN16(dy, "dy"); CHK_ALIGN16(x, "x"); CHK_ALIGN16(out, "out");
    if (!k_gemm_tn_eligible(rows, N, K, ldy, ldx)) return mm_set_error(MM_ERR_UNSUPPORTED, "gemm_wgrad_tn: N and K must be multiples of 128 (use the transposed-copy form)");
    const int splits = k_gemm_tn_splits(rows, N, K);
    if (splits <= 1) return k_gemm_tn((hipStream_t)stream, (const bf16_t*)dy, ldy, (const bf16_t*)x, ldx, rows, N, K, 1, out);
    CHK_PTR(ws, "ws"); CHK_ALIGN16(ws, "ws");
    const int rc = k_gemm_tn((hipStream_t)stream, (const bf16_t*)dy, ldy, (const bf16_t*)x, ldx, rows, N, K, splits, ws);
    if (rc) return rc;
    return k_colsum((hipStream_t)stream, ws, splits, (long)N * K, out);
}

int mm_transpose_bf16(mm_stream_t stream, const void* in, int64_t rows, int64_t cols, int64_t ld_in, void* out, int64_t ld_out) {
    if (rows == 0 || cols == 0) return MM_OK;
    CHK_PTR(in, "in"); CHK_PTR(out, "out"); CHK_ALIGN16(in, "in"); CHK_ALIGN16(out, "out");
    return k_transpose_bf16((hipStream_t)stream, (const bf16_t*)in, rows, cols, ld_in, (bf16_t*)out, ld_out);
}

int mm_f32_to_bf16(mm_stream_t stream, const float* x, void* out, int64_t count) {
    if (count == 0) return MM_OK;
    CHK_PTR(x, "x"); CHK_PTR(out, "out");
    return k_f32_to_bf16((hipStream_t)stream, x, (bf16_t*)out, count);
}

int mm_colsum_f32(mm_stream_t stream, const float* part, int nparts, int D, float* out) {
    CHK_PTR(part, "part"); CHK_PTR(out, "out");
    return k_colsum((hipStream_t)stream, part, nparts, D, out);
}

int64_t mm_ln_bwd_workspace_floats(int rows, int D) { return k_ln_bwd_workspace_floats(rows, D); }

int mm_layernorm_bwd(mm_stream_t stream, const float* x, int64_t ldx, const void* dy, int64_t lddy, const float* gamma,
                     const int32_t* row_index, int rows, int D, float* dx, int64_t lddx, int accumulate, float* dgamma, float* ws) {
    if (rows == 0) return MM_OK;
    CHK_PTR(x, "x"); CHK_PTR(dy, "dy"); CHK_PTR(gamma, "gamma"); CHK_PTR(dx, "dx"); CHK_PTR(dgamma, "dgamma"); CHK_PTR(ws, "ws");
    CHK_ALIGN16(x, "x"); CHK_ALIGN16(dx, "dx");
    return k_layernorm_bwd((hipStream_t)stream, x, ldx, (const bf16_t*)dy, lddy, gamma, row_index, rows, D, dx, lddx, accumulate, dgamma, ws);
}

int mm_geglu_ln_bwd(mm_stream_t stream, const void* h, int64_t ldh, const void* dz, int64_t lddz, const float* gamma, int rows,
                    int F, int Fp, void* dh, int64_t lddh, float* dgamma, float* ws) {
    if (rows == 0) return MM_OK;
    CHK_PTR(h, "h"); CHK_PTR(dz, "dz"); CHK_PTR(gamma, "gamma"); CHK_PTR(dh, "dh"); CHK_PTR(dgamma, "dgamma"); CHK_PTR(ws, "ws");
    CHK_ALIGN16(h, "h"); CHK_ALIGN16(dz, "dz"); CHK_ALIGN16(dh, "dh");
    return k_geglu_ln_bwd((hipStream_t)stream, (const bf16_t*)h, ldh, (const bf16_t*)dz, lddz, gamma, rows, F, Fp, (bf16_t*)dh, lddh, dgamma, ws);
}

int mm_ce_bwd(mm_stream_t stream, const float* logits, int64_t ld, int R, int V, const int64_t* labels, float scale, void* dl,
              int64_t ldd) {
    if (R == 0) return MM_OK;
    CHK_PTR(logits, "logits"); CHK_PTR(labels, "labels"); CHK_PTR(dl, "dl"); CHK_ALIGN16(logits, "logits");
    return k_ce_bwd((hipStream_t)stream, logits, ld, R, V, labels, scale, (bf16_t*)dl, ldd);
}

int mm_bce_head_bwd(mm_stream_t stream, const void* e, int64_t lde, const float* x, const float* y, const float* w, int rows, int D,
                    void* de, int64_t ldde, float* dw, float* ws) {
    if (rows == 0) return MM_OK;
    CHK_PTR(e, "e"); CHK_PTR(x, "x"); CHK_PTR(y, "y"); CHK_PTR(w, "w"); CHK_PTR(de, "de"); CHK_PTR(dw, "dw"); CHK_PTR(ws, "ws");
    CHK_ALIGN16(e, "e"); CHK_ALIGN16(de, "de");
    return k_bce_head_bwd((hipStream_t)stream, (const bf16_t*)e, lde, x, y, w, rows, D, (bf16_t*)de, ldde, dw, ws);
}

size_t mm_embed_bwd_workspace_bytes(int B, int n, int D) { return (B <= 0 || n <= 0 || D <= 0) ? 0 : k_embed_bwd_workspace_bytes(B, n, D); }

int mm_embed_bwd(mm_stream_t stream, const int64_t* ids, int B, int n, int D, const float* dx, float* dtoken, float* dpos, void* ws, size_t ws_bytes) {
    if (B == 0) return MM_OK;
    CHK_PTR(ids, "ids"); CHK_PTR(dx, "dx"); CHK_PTR(dtoken, "dtoken"); CHK_PTR(dpos, "dpos");
    if (ws && ws_bytes < k_embed_bwd_workspace_bytes(B, n, D)) return mm_set_error(MM_ERR_WORKSPACE, "embed_bwd: workspace too small");
    if (ws) CHK_ALIGN16(ws, "ws");
    return k_embed_bwd((hipStream_t)stream, ids, B, n, D, dx, dtoken, dpos, ws);
}

int mm_scatter_rows_bf16(mm_stream_t stream, const void* src, const int32_t* row_index, int R, int D, void* dst) {
    if (R == 0) return MM_OK;
    CHK_PTR(src, "src"); CHK_PTR(row_index, "row_index"); CHK_PTR(dst, "dst"); CHK_ALIGN16(src, "src"); CHK_ALIGN16(dst, "dst");
    return k_scatter_rows_bf16((hipStream_t)stream, (const bf16_t*)src, row_index, R, D, (bf16_t*)dst);
}

int mm_sum_parts_bf16(mm_stream_t stream, const void* parts, int P, int64_t n, void* out) {
    if (n == 0) return MM_OK;
    CHK_PTR(parts, "parts"); CHK_PTR(out, "out"); CHK_ALIGN16(parts, "parts"); CHK_ALIGN16(out, "out");
    return k_sum_parts_bf16((hipStream_t)stream, (const bf16_t*)parts, P, n, (bf16_t*)out);
}

int mm_attention_bwd(mm_stream_t stream, const void* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const void* k, int64_t k_sb,
                     int64_t k_sh, int64_t k_sn, const void* v, int64_t v_sb, int64_t v_sh, int64_t v_sn, const void* o,
                     int64_t o_sb, int64_t o_sh, int64_t o_sn, const void* dout, int64_t do_sb, int64_t do_sh, int64_t do_sn,
                     void* dqn, int64_t dq_sb, int64_t dq_sh, int64_t dq_sn, void* dkn, int64_t dk_sb, int64_t dk_sh, int64_t dk_sn,
                     void* dv, int64_t dv_sb, int64_t dv_sh, int64_t dv_sn, float* dnk, float* dnv, int B, int H, int nq, int nk,
                     const uint8_t* key_mask, int64_t km_sb, const float* q_scale, const float* k_scale, const float* null_k,
                     const float* null_v, float scale) {
    CHK_PTR(q, "q"); CHK_PTR(o, "o"); CHK_PTR(dout, "dout"); CHK_PTR(dqn, "dqn"); CHK_PTR(dnk, "dnk"); CHK_PTR(dnv, "dnv");
    if (nk > 0) { CHK_PTR(k, "k"); CHK_PTR(v, "v"); CHK_PTR(dkn, "dkn"); CHK_PTR(dv, "dv"); }
    CHK_ALIGN16(q, "q"); CHK_ALIGN16(k, "k"); CHK_ALIGN16(v, "v"); CHK_ALIGN16(o, "o"); CHK_ALIGN16(dout, "dout");
    return k_attention_bwd((hipStream_t)stream, (const bf16_t*)q, q_sb, q_sh, q_sn, (const bf16_t*)k, k_sb, k_sh, k_sn, (const bf16_t*)v, v_sb,
                           v_sh, v_sn, (const bf16_t*)o, o_sb, o_sh, o_sn, (const bf16_t*)dout, do_sb, do_sh, do_sn, (bf16_t*)dqn, dq_sb, dq_sh,
                           dq_sn, (bf16_t*)dkn, dk_sb, dk_sh, dk_sn, (bf16_t*)dv, dv_sb, dv_sh, dv_sn, dnk, dnv, B, H, nq, nk, key_mask, km_sb,
                           q_scale, k_scale, null_k, null_v, scale);
}

int64_t mm_qk_norm_bwd_blocks(int64_t nvec) { return k_qk_norm_bwd_blocks(nvec); }

int mm_qk_norm_bwd(mm_stream_t stream, const void* x, int64_t ldx, const float* x_f32, int H, const void* dy, int64_t lddy,
                   const float* dy_f32, const float* scale, int64_t rows, int heads_per_row, void* dx, int64_t lddx,
                   float* dx_f32, float* dscale_part) {
    if (rows == 0) return MM_OK;
    if (!x && !x_f32) return mm_set_error(MM_ERR_SHAPE, "qk_norm_bwd: x or x_f32 is required");
    if (!dy && !dy_f32) return mm_set_error(MM_ERR_SHAPE, "qk_norm_bwd: dy or dy_f32 is required");
    if (!dx && !dx_f32) return mm_set_error(MM_ERR_SHAPE, "qk_norm_bwd: dx or dx_f32 is required");
    CHK_PTR(scale, "scale"); CHK_PTR(dscale_part, "dscale_part");
    return k_qk_norm_bwd((hipStream_t)stream, (const bf16_t*)x, ldx, x_f32, H, (const bf16_t*)dy, lddy, dy_f32, scale, rows, heads_per_row,
                         (bf16_t*)dx, lddx, dx_f32, dscale_part);
}

int mm_philox_uniform(mm_stream_t stream, uint64_t seed, uint64_t row_offset, uint32_t step, int rows, int V, float* out) {
    if (rows == 0) return MM_OK;
    CHK_PTR(out, "out"); CHK_ALIGN16(out, "out");
    return k_philox_fill((hipStream_t)stream, seed, row_offset, step, rows, V, out);
}

static int conv2d_nhwc_impl(mm_stream_t stream, const void* in, int B, int Hin, int Win, int Cin, const void* w, int Cout,
                            int TH, int TW, int stride, int off_y, int off_x, int Hv, int Wv, int os, int py, int px,
                            int Hout, int Wout, const float* bias, int act, const void* resid, void* out, int out_nchw_f32, int f16, float alpha, int terms, int half_io = 0,
                            const void* head_w = nullptr, int head_ldw = 0, const float* head_b = nullptr, int head_c = 0, const void* const* par_w = nullptr);

int mm_conv2d_nhwc(mm_stream_t stream, const void* in, int B, int Hin, int Win, int Cin, const void* w, int Cout,
                   int TH, int TW, int stride, int off_y, int off_x, int Hv, int Wv, int os, int py, int px,
                   int Hout, int Wout, const float* bias, int act, const void* resid, void* out, int out_nchw_f32) {
    return conv2d_nhwc_impl(stream, in, B, Hin, Win, Cin, w, Cout, TH, TW, stride, off_y, off_x, Hv, Wv, os, py, px, Hout, Wout, bias, act, resid, out,
                            out_nchw_f32, 0, 1.f, 0);
}

// the same convolution on fp16 TERM operands ('f16x2' tier: `in` holds MM_SPLIT_F16 | P segments per pixel, w the matching per-tap pack scaled by a power
// of two): fp16 MFMA, accumulators x alpha, fp32 output only (out_nchw_f32 = 1: NCHW, 2: NHWC with an optional fp32 NHWC residual)
int mm_conv2d_nhwc_f16(mm_stream_t stream, const void* in, int B, int Hin, int Win, int Cin, const void* w, int Cout,
                       int TH, int TW, int stride, int off_y, int off_x, int Hv, int Wv, int os, int py, int px,
                       int Hout, int Wout, const float* bias, int act, const float* resid_f32, float* out, int out_nchw_f32, float alpha) {
    if (out_nchw_f32 != 1 && out_nchw_f32 != 2) return mm_set_error(MM_ERR_SHAPE, "conv_f16: fp32 output only (out_nchw_f32 = 1 or 2)");
    return conv2d_nhwc_impl(stream, in, B, Hin, Win, Cin, w, Cout, TH, TW, stride, off_y, off_x, Hv, Wv, os, py, px, Hout, Wout, bias, act, resid_f32, out,
                            out_nchw_f32, 1, alpha, 0);
}

// the same convolution with SINGLE fp16 terms as operands and fp16 activation storage (round 6: the half-precision VAE decode): `in` NHWC fp16, w fp16 [Cout][Kp]
// scaled by 1 / alpha, out NHWC fp16 (out_nchw_f32 = 0; resid NHWC fp16) or NCHW fp32 (= 1)
int mm_conv2d_nhwc_half(mm_stream_t stream, const void* in, int B, int Hin, int Win, int Cin, const void* w, int Cout,
                        int TH, int TW, int stride, int off_y, int off_x, int Hv, int Wv, int os, int py, int px,
                        int Hout, int Wout, const float* bias, int act, const void* resid, void* out, int out_nchw_f32, float alpha) {
    if (out_nchw_f32 != 0 && out_nchw_f32 != 1) return mm_set_error(MM_ERR_SHAPE, "conv_half: out_nchw_f32 = 0 (NHWC fp16) or 1 (NCHW fp32)");
    return conv2d_nhwc_impl(stream, in, B, Hin, Win, Cin, w, Cout, TH, TW, stride, off_y, off_x, Hv, Wv, os, py, px, Hout, Wout, bias, act, resid, out,
                            out_nchw_f32, 1, alpha, 0, 1);
}

int mm_conv2d_nhwc_terms(mm_stream_t stream, const void* in, int B, int Hin, int Win, int Cin, const void* w, int Cout,
                         int TH, int TW, int stride, int off_y, int off_x, int Hv, int Wv, int os, int py, int px,
                         int Hout, int Wout, const float* bias, int act, const float* resid_f32, float* out, int out_nchw_f32, float alpha, int products) {
    if (out_nchw_f32 != 1 && out_nchw_f32 != 2) return mm_set_error(MM_ERR_SHAPE, "conv_terms: fp32 output only (out_nchw_f32 = 1 or 2)");
    const int cnt = split_count(products);
    if (!split_is_f16(products) || (cnt != 2 && cnt != 3) || (Cin % cnt)) return mm_set_error(MM_ERR_SHAPE, "conv_terms: products = MM_SPLIT_F16 | 2 / 3 segments per pixel");
    return conv2d_nhwc_impl(stream, in, B, Hin, Win, Cin, w, Cout, TH, TW, stride, off_y, off_x, Hv, Wv, os, py, px, Hout, Wout, bias, act, resid_f32, out,
                            out_nchw_f32, 1, alpha, (products & 0x200) ? cnt : 0);
}

static int conv2d_nhwc_impl(mm_stream_t stream, const void* in, int B, int Hin, int Win, int Cin, const void* w, int Cout,
                            int TH, int TW, int stride, int off_y, int off_x, int Hv, int Wv, int os, int py, int px,
                            int Hout, int Wout, const float* bias, int act, const void* resid, void* out, int out_nchw_f32, int f16, float alpha, int terms, int half_io,
                            const void* head_w, int head_ldw, const float* head_b, int head_c, const void* const* par_w) {
    if (B == 0) return MM_OK;
    CHK_PTR(in, "in"); CHK_PTR(w, "w"); CHK_PTR(out, "out");
    CHK_ALIGN16(in, "in"); CHK_ALIGN16(w, "w"); CHK_ALIGN16(out, "out");
    if (TH <= 0 || TW <= 0 || Cin <= 0 || Cout <= 0 || Hv <= 0 || Wv <= 0) return mm_set_error(MM_ERR_SHAPE, "conv: bad geometry");
    if (!out_nchw_f32 && (Cout % 8)) return mm_set_error(MM_ERR_SHAPE, "conv: NHWC bf16 output needs Cout % 8 == 0");
    if (out_nchw_f32 == 2 && (Cout % 4)) return mm_set_error(MM_ERR_SHAPE, "conv: NHWC fp32 output needs Cout % 4 == 0");
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = MODE_CONV;
    a.Ktrue = TH * TW * Cin;
    a.K = (a.Ktrue + 63) / 64 * 64;
    a.W = (const bf16_t*)w; a.N = Cout; a.ldw = a.K;
    a.M = B * Hv * Wv; a.X = (const bf16_t*)in; a.ldx = Cin;
    a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.TW = TW; a.stride = stride; a.off_y = off_y; a.off_x = off_x;
    a.Hv = Hv; a.Wv = Wv; a.os = os; a.py = py; a.px = px; a.Hout = Hout; a.Wout = Wout;
    a.out = out; a.ldc = Cout; a.out_kind = out_nchw_f32 == 2 ? OUT_F32 : (out_nchw_f32 ? OUT_NCHW_F32 : OUT_BF16);
    a.bias = bias; a.act = act ? ACT_LEAKY : ACT_NONE;
    if (out_nchw_f32 == 2) a.resid_f32 = (const float*)resid;      // fp32 NHWC in and out (the precision tier's convolutions: bf16 term segments in, fp32 out)
    else a.resid_bf16 = (const bf16_t*)resid;
    a.ldr = Cout;
    a.f16 = f16; a.alpha = alpha; a.terms = terms; a.half_io = half_io;
    a.head_w = (const bf16_t*)head_w; a.head_ldw = head_ldw; a.head_b = head_b; a.head_c = head_c;
    if (par_w) for (int i = 0; i < 4; ++i) a.par_w[i] = (const bf16_t*)par_w[i];
    return mm_gemm_launch(a, (hipStream_t)stream);
}

int mm_glu_nhwc(mm_stream_t stream, const void* x, int64_t rows, int C, void* out) {
    if (rows == 0) return MM_OK;
    CHK_PTR(x, "x"); CHK_PTR(out, "out"); CHK_ALIGN16(x, "x"); CHK_ALIGN16(out, "out");
    return k_glu((hipStream_t)stream, (const bf16_t*)x, rows, C, (bf16_t*)out);
}

int mm_groupnorm_nhwc(mm_stream_t stream, const void* x, int B, int HW, int C, int groups, const float* gamma,
                      const float* beta, int act, float* stats_ws, void* out) {
    if (B == 0) return MM_OK;
    CHK_PTR(x, "x"); CHK_PTR(out, "out"); CHK_PTR(gamma, "gamma"); CHK_PTR(beta, "beta"); CHK_PTR(stats_ws, "stats_ws");
    CHK_ALIGN16(x, "x"); CHK_ALIGN16(out, "out");
    return k_groupnorm((hipStream_t)stream, (const bf16_t*)x, B, HW, C, groups, gamma, beta, act ? ACT_LEAKY : ACT_NONE, stats_ws, (bf16_t*)out);
}

int mm_lfq_decode(mm_stream_t stream, const int64_t* ids, int64_t count, int bits, int C, const float* w,
                  const float* b, void* out) {
    if (count == 0) return MM_OK;
    CHK_PTR(ids, "ids"); CHK_PTR(out, "out");
    if (w && !b) return mm_set_error(MM_ERR_SHAPE, "lfq_decode: project_out needs its bias");
    return k_lfq_decode((hipStream_t)stream, ids, count, bits, C, w, b, (bf16_t*)out);
}

int mm_lfq_encode(mm_stream_t stream, const void* x, int64_t count, int C, int bits, const float* w_in,
                  const float* b_in, const float* w_out, const float* b_out, int64_t* ids, void* out) {
    if (count == 0) return MM_OK;
    CHK_PTR(x, "x"); CHK_PTR(ids, "ids");
    return k_lfq_encode((hipStream_t)stream, (const bf16_t*)x, count, C, bits, w_in, b_in, w_out, b_out, ids, (bf16_t*)out);
}

int mm_nchw_f32_to_nhwc8_bf16(mm_stream_t stream, const float* img, int B, int C, int H, int W, void* out) {
    CHK_PTR(img, "img"); CHK_PTR(out, "out"); CHK_ALIGN16(out, "out");
    return k_nchw_to_nhwc8((hipStream_t)stream, img, B, C, H, W, (bf16_t*)out);
}

int mm_nhwc_bf16_to_nchw_f32(mm_stream_t stream, const void* x, int B, int C, int H, int W, float* out) {
    CHK_PTR(x, "x"); CHK_PTR(out, "out");
    return k_nhwc_to_nchw_f32((hipStream_t)stream, (const bf16_t*)x, B, C, H, W, out);
}

}  // extern "C"

// internal (vae_model.hip): a 256-channel convolution with the 1 x 1 head Conv2d(256, head_c, 1) in its epilogue (gemm_wide_conv.hip): `image` = NCHW fp32
// [B][head_c][Hout][Wout]; head_w = the head's own 16-bit pack [head_c][head_ldw]; half = the fp16-storage form.  MM_ERR_UNSUPPORTED when the shape is outside
// the 256 x 256 convolution kernel's class (the caller then runs the two convolutions separately).
int mm_conv2d_nhwc_head(mm_stream_t stream, const void* in, int B, int Hin, int Win, int Cin, const void* w, int Cout, int TH, int TW, int stride, int off_y, int off_x,
                        int Hv, int Wv, int os, int py, int px, int Hout, int Wout, const float* bias, int act, const void* head_w, int head_ldw, const float* head_b,
                        int head_c, float* image, int half, float alpha) {
    if (!head_w || !head_b || !image) return mm_set_error(MM_ERR_SHAPE, "conv_head: NULL argument");
    return conv2d_nhwc_impl(stream, in, B, Hin, Win, Cin, w, Cout, TH, TW, stride, off_y, off_x, Hv, Wv, os, py, px, Hout, Wout, bias, act, nullptr, image, 1,
                            half ? 1 : 0, half ? alpha : 1.f, 0, half ? 1 : 0, head_w, head_ldw, head_b, head_c);
}

// internal (vae_model.hip): ConvTranspose2d(4, 2, 1) + bias (+ LeakyReLU) as ONE launch over its four parity classes (gemm_wide_conv.hip; w4[py * 2 + px] = the
// packed 2 x 2 weights of class (py, px)), optionally with the fused head (head_w != NULL -> image NCHW fp32, else out NHWC 16-bit [B][2 Hin][2 Win][Cout]).
// MM_ERR_UNSUPPORTED outside the 256 x 256 convolution kernel's shape class: the caller runs the four classes separately.
int mm_convT2d_nhwc_4(mm_stream_t stream, const void* in, int B, int Hin, int Win, int Cin, const void* const* w4, int Cout, const float* bias, int act, void* out,
                      const void* head_w, int head_ldw, const float* head_b, int head_c, int half, float alpha) {
    if (!w4 || !w4[0] || !w4[1] || !w4[2] || !w4[3]) return mm_set_error(MM_ERR_SHAPE, "convT2d_4: NULL weights");
    return conv2d_nhwc_impl(stream, in, B, Hin, Win, Cin, w4[0], Cout, 2, 2, 1, -1, -1, Hin, Win, 2, 0, 0, 2 * Hin, 2 * Win, bias, act, nullptr, out, head_w ? 1 : 0,
                            half ? 1 : 0, half ? alpha : 1.f, 0, half ? 1 : 0, head_w, head_ldw, head_b, head_c, w4);
}
