// mm_train_step: the training step of the transformer -- forward with saved activations, cross-entropy on the labelled rows, and the whole hand-written
// backward -- as ONE C call on one stream: no allocation, no host synchronisation, no Python between the ~1150 launches (the same operators, in the same
// order, as the operator-by-operator driver in training.py, so the loss and every gradient are bit-identical to it: tests/test_gpu_train_step.py).
// Reference: MaskGit.forward (muse_maskgit_pytorch.py:623-741) differentiating Transformer.forward (:279-348) with autograd.
// Scope of the C entry: the generator's cross-entropy path (token ids in, labels at the masked rows), optional text projection; dim_head 64 and
// n in {64, 128, 256} (the attention backward's query blocks).  Self-conditioning, conditioning ids, the critics' BCE heads and longer sequences stay on
// training.py's driver (muse_maskgit.py picks).  Linear layers: dX = dY W and dW = dY^T X are NT GEMMs on transposed bf16 copies, as there.
#include <string.h>

#include "common.h"
#include "muse_hip_internal.h"

namespace {

struct Arena {
    unsigned char* base;
    size_t off;
    template <typename T>
    T* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return r;
    }
};

inline int pad64(int v) { return (v + 63) / 64 * 64; }

#define RC(x)                \
    do {                     \
        int _rc = (x);       \
        if (_rc) return _rc; \
    } while (0)
#define HC(x)                                                       \
    do {                                                            \
        hipError_t _e = (x);                                        \
        if (_e != hipSuccess) return mm_set_hip_error(_e, "train_step"); \
    } while (0)

struct LayerBufs {
    float *x0, *x1, *x2;                                   // residual stream entering self-attention / cross-attention / feed-forward
    bf16_t *u, *qkv, *o, *u2, *q2, *kv2, *o2, *u3, *h, *z; // saved activations
    bf16_t *wqkv, *wo, *wq2, *wkv2, *wo2, *w1p, *w2p;      // bf16 operand copies of this step's parameters
    float *g2p, *b2p;                                      // LayerNorm(inner) gain / bias padded to Fp
};

struct Bufs {
    LayerBufs* L;            // host array [depth]
    float* xL;               // output of the last layer
    bf16_t *tok_b, *pos_b, *te_b, *wtp, *cx, *wl, *e;
    float *logits, *rowloss;
    // backward scratch
    float *dres, *gtmp, *wg_ws, *ln_ws, *part, *dnk, *dnv, *dnull, *pair, *dcx[2];
    bf16_t *dl, *de, *dy, *dz, *dh, *du, *dob, *dqn, *dkn, *dq2, *dkv2, *dqkv, *dcxb, *tA, *tB, *tW;
};

struct Dims { int B, n, L, R, M, D, H, I, F, Fp, V, td, Mc, depth; };

void carve(Arena& A, const mm_train_desc& d, const Dims& q, Bufs& b, LayerBufs* layers) {
    const size_t M = q.M, D = q.D, I = q.I, Fp = q.Fp, V = q.V, R = q.R, Mc = q.Mc;
    b.L = layers;
    for (int l = 0; l < q.depth; ++l) {
        LayerBufs& y = layers[l];
        y.x0 = A.take<float>(M * D); y.x1 = A.take<float>(M * D); y.x2 = A.take<float>(M * D);
        y.u = A.take<bf16_t>(M * D); y.qkv = A.take<bf16_t>(M * 3 * I); y.o = A.take<bf16_t>(M * I);
        y.u2 = A.take<bf16_t>(M * D); y.q2 = A.take<bf16_t>(M * I); y.kv2 = A.take<bf16_t>(Mc * 2 * I); y.o2 = A.take<bf16_t>(M * I);
        y.u3 = A.take<bf16_t>(M * D); y.h = A.take<bf16_t>(M * 2 * Fp); y.z = A.take<bf16_t>(M * Fp);
        y.wqkv = A.take<bf16_t>(3 * I * D); y.wo = A.take<bf16_t>(D * I); y.wq2 = A.take<bf16_t>(I * D); y.wkv2 = A.take<bf16_t>(2 * I * D);
        y.wo2 = A.take<bf16_t>(D * I); y.w1p = A.take<bf16_t>(2 * Fp * D); y.w2p = A.take<bf16_t>(D * Fp);
        y.g2p = A.take<float>(Fp); y.b2p = A.take<float>(Fp);
    }
    b.xL = A.take<float>(M * D);
    b.tok_b = A.take<bf16_t>((size_t)d.vocab_rows * D); b.pos_b = A.take<bf16_t>((size_t)d.seq_len * D);
    b.te_b = A.take<bf16_t>(Mc * q.td); b.wtp = A.take<bf16_t>(d.text_proj ? D * (size_t)q.td : 0); b.cx = d.text_proj ? A.take<bf16_t>(Mc * D) : b.te_b;
    b.wl = A.take<bf16_t>(V * D); b.e = A.take<bf16_t>(R * D); b.logits = A.take<float>(R * V); b.rowloss = A.take<float>(R);
    const size_t Rp = pad64((int)R), Mcp = pad64((int)Mc);
    const size_t wide = 3 * I > 2 * Fp ? 3 * I : 2 * Fp;
    b.dres = A.take<float>(M * D);
    size_t gmax = D * Fp; if (2 * Fp * D > gmax) gmax = 2 * Fp * D; if (3 * I * D > gmax) gmax = 3 * I * D;
    b.gtmp = A.take<float>(gmax);
    b.wg_ws = A.take<float>((size_t)384 * 128 * 128);                       // split-K slabs: mm_gemm_wgrad_splits keeps tiles x splits < 384
    size_t lnw = (size_t)mm_ln_bwd_workspace_floats((int)M, (int)D), lnw2 = (size_t)mm_ln_bwd_workspace_floats((int)M, (int)Fp);
    b.ln_ws = A.take<float>(lnw > lnw2 ? lnw : lnw2);
    const size_t nvec = (M > Mc ? M : Mc) * q.H;
    b.part = A.take<float>((size_t)mm_qk_norm_bwd_blocks((int64_t)nvec) * 64 + 128);
    b.dnk = A.take<float>((size_t)q.B * q.H * 64); b.dnv = A.take<float>((size_t)q.B * q.H * 64); b.dnull = A.take<float>((size_t)q.B * q.H * 64);
    b.pair = A.take<float>(128);
    b.dcx[0] = A.take<float>(d.text_proj ? Mc * D : 0); b.dcx[1] = A.take<float>(d.text_proj ? Mc * D : 0); b.dcxb = A.take<bf16_t>(d.text_proj ? Mc * D : 0);
    b.dl = A.take<bf16_t>(R * V); b.de = A.take<bf16_t>(R * D); b.dy = A.take<bf16_t>(M * D); b.dz = A.take<bf16_t>(M * Fp); b.dh = A.take<bf16_t>(M * 2 * Fp);
    b.du = A.take<bf16_t>(M * D); b.dob = A.take<bf16_t>(M * I); b.dqn = A.take<bf16_t>(M * I); b.dkn = A.take<bf16_t>((M > Mc ? M : Mc) * I);
    b.dq2 = A.take<bf16_t>(M * I); b.dkv2 = A.take<bf16_t>(Mc * 2 * I); b.dqkv = A.take<bf16_t>(M * 3 * I);
    // transposed copies: tA = T(gradient rows), tB = T(saved activation rows), tW = T(weight)
    size_t ta = wide * (size_t)M; if (V * Rp > ta) ta = V * Rp; if (2 * I * Mcp > ta) ta = 2 * I * Mcp;
    size_t tb = (Fp > D ? Fp : D) * (size_t)M; if (I * (size_t)M > tb) tb = I * (size_t)M; if (D * Rp > tb) tb = D * Rp; if ((size_t)(q.td > (int)D ? q.td : (int)D) * Mcp > tb) tb = (size_t)(q.td > (int)D ? q.td : (int)D) * Mcp;
    size_t tw = D * V; if (D * wide > tw) tw = D * wide; if (Fp * (size_t)pad64((int)D) > tw) tw = Fp * (size_t)pad64((int)D);
    b.tA = A.take<bf16_t>(ta); b.tB = A.take<bf16_t>(tb); b.tW = A.take<bf16_t>(tw);
}

// out [cols][Rp] = x [rows][cols]^T, Rp = rows rounded up to 64, padding columns zero   (training.py _t)
int tr64(mm_stream_t st, hipStream_t s, const bf16_t* x, long rows, long cols, long ld, bf16_t* out) {
    const long Rp = pad64((int)rows);
    if (Rp != rows) HC(hipMemsetAsync(out, 0, (size_t)cols * Rp * 2, s));
    return mm_transpose_bf16(st, x, rows, cols, ld, out, Rp);
}
// dW fp32 [N_][K_] = dY^T X for dY bf16 [rows][N_] (ld ldy), X bf16 [rows][K_] (ld ldx)   (training.py _wgrad)
int wgrad(mm_stream_t st, hipStream_t s, const Bufs& b, const bf16_t* dy, long ldy, int N_, const bf16_t* x, long ldx, int K_, long rows, float* out) {
    RC(tr64(st, s, dy, rows, N_, ldy, b.tA));
    RC(tr64(st, s, x, rows, K_, ldx, b.tB));
    const int Rp = pad64((int)rows);
    const int splits = (K_ % 4 == 0) ? mm_gemm_wgrad_splits(N_, K_, Rp) : 1;
    if (splits <= 1) return mm_gemm_bf16(st, b.tA, Rp, b.tB, Rp, N_, K_, Rp, out, K_, 1, nullptr);      // (every K_ here is a multiple of 64: rows of the gradient are dense)
    return mm_gemm_wgrad(st, b.tA, Rp, b.tB, Rp, N_, K_, Rp, splits, b.wg_ws, out);
}
// dX bf16 [rows][K_] = dY W for dY bf16 [rows][N_], W bf16 [N_][K_]   (training.py _dgrad)
int dgrad(mm_stream_t st, hipStream_t s, const Bufs& b, const bf16_t* dy, long ldy, int N_, const bf16_t* w, int K_, long rows, bf16_t* out, long ldo) {
    RC(tr64(st, s, w, N_, K_, K_, b.tW));                 // [K_][Np]
    return mm_gemm_bf16(st, dy, ldy, b.tW, pad64(N_), (int)rows, K_, pad64(N_), out, ldo, 0, nullptr);
}

}  // namespace

extern "C" {

static int train_dims(const mm_train_desc* d, int B, int n, int L, int R, Dims& q) {
    if (!d || !d->layers) return mm_set_error(MM_ERR_SHAPE, "train_step: NULL descriptor");
    q.B = B; q.n = n; q.L = L; q.R = R; q.M = B * n; q.D = d->dim; q.H = d->heads; q.I = d->heads * 64; q.F = d->ff_inner; q.Fp = pad64(d->ff_inner);
    q.V = d->dim_out; q.td = d->text_dim; q.Mc = B * L; q.depth = d->depth;
    if (B <= 0 || L <= 0 || R <= 0 || d->depth <= 0 || d->depth > 256) return mm_set_error(MM_ERR_SHAPE, "train_step: bad sizes");
    if (n != 64 && n != 128 && n != 256) return mm_set_error(MM_ERR_UNSUPPORTED, "train_step: n must be 64, 128 or 256 (longer / other lengths: training.py's driver)");
    if ((q.M % 64) || (q.D % 64) || (q.V % 64) || (q.td % 64) || n > d->seq_len) return mm_set_error(MM_ERR_SHAPE, "train_step: batch * n, dim, dim_out and text_dim must be multiples of 64");
    if (!d->text_proj && q.td != q.D) return mm_set_error(MM_ERR_SHAPE, "train_step: text_proj is NULL but text_dim != dim");
    return MM_OK;
}

size_t mm_train_step_workspace_bytes(const mm_train_desc* desc, int B, int n, int L, int R) {
    Dims q;
    if (train_dims(desc, B, n, L, R, q)) return 0;
    Arena A{nullptr, 0};
    Bufs b;
    LayerBufs layers[256];
    carve(A, *desc, q, b, layers);
    return A.off + 512;
}

int mm_train_step(const mm_train_desc* desc, mm_stream_t stream, const int64_t* ids, int B, int n, const float* text_embeds, int L, const uint8_t* ctx_mask,
                  const int32_t* row_index, const int64_t* labels_rows, int R, float* loss_out, float* logits_rows_out, void* workspace, size_t workspace_bytes) {
    Dims q;
    RC(train_dims(desc, B, n, L, R, q));
    const mm_train_desc& d = *desc;
    if (!ids || !text_embeds || !ctx_mask || !row_index || !labels_rows || !loss_out || !workspace) return mm_set_error(MM_ERR_SHAPE, "train_step: NULL argument");
    if (workspace_bytes < mm_train_step_workspace_bytes(desc, B, n, L, R)) return mm_set_error(MM_ERR_WORKSPACE, "train_step: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Arena A{(unsigned char*)workspace, 0};
    Bufs b;
    LayerBufs layers[256];
    carve(A, d, q, b, layers);
    const int M = q.M, D = q.D, H = q.H, I = q.I, F = q.F, Fp = q.Fp, V = q.V, td = q.td, Mc = q.Mc;

    // ================================================================ forward (training.py TransformerTrainFn.forward)
    RC(mm_f32_to_bf16(stream, d.token_emb, b.tok_b, (int64_t)d.vocab_rows * D));
    RC(mm_f32_to_bf16(stream, d.pos_emb, b.pos_b, (int64_t)d.seq_len * D));
    RC(mm_embed(stream, ids, M, n, b.tok_b, d.vocab_rows, b.pos_b, D, layers[0].x0));                                   // mmp.py:322-323
    RC(mm_f32_to_bf16(stream, text_embeds, b.te_b, (int64_t)Mc * td));
    if (d.text_proj) {
        RC(mm_f32_to_bf16(stream, d.text_proj, b.wtp, (int64_t)D * td));
        RC(mm_gemm_bf16(stream, b.te_b, td, b.wtp, td, Mc, D, td, b.cx, D, 0, nullptr));                               // mmp.py:302
    }
    for (int l = 0; l < q.depth; ++l) {
        const mm_train_layer& w = d.layers[l];
        LayerBufs& y = layers[l];
        float* xn = l + 1 < q.depth ? layers[l + 1].x0 : b.xL;
        // ---- self attention (mmp.py:137-162, 186)
        RC(mm_f32_to_bf16(stream, w.sa.to_q, y.wqkv, (int64_t)I * D));
        RC(mm_f32_to_bf16(stream, w.sa.to_kv, y.wqkv + (size_t)I * D, (int64_t)2 * I * D));
        RC(mm_f32_to_bf16(stream, w.sa.to_out, y.wo, (int64_t)D * I));
        RC(mm_layernorm(stream, y.x0, D, M, D, w.sa.gamma, w.sa.beta, nullptr, y.u, D));
        RC(mm_gemm_bf16(stream, y.u, D, y.wqkv, D, M, 3 * I, D, y.qkv, 3 * I, 0, nullptr));
        RC(mm_attend(stream, y.qkv, (int64_t)n * 3 * I, 64, 3 * I, y.qkv + I, (int64_t)n * 3 * I, 64, 3 * I, y.qkv + 2 * I, (int64_t)n * 3 * I, 64, 3 * I,
                     y.o, (int64_t)n * I, 64, I, B, H, n, n, nullptr, 0, 1, w.sa.q_scale, w.sa.k_scale, w.sa.null_kv, w.sa.null_kv + (size_t)H * 64, 8.f, 64));
        RC(mm_gemm_bf16(stream, y.o, I, y.wo, I, M, D, I, y.x1, D, 1, y.x0));
        // ---- cross attention (mmp.py:139-141, 155-157, 187)
        RC(mm_f32_to_bf16(stream, w.ca.to_q, y.wq2, (int64_t)I * D));
        RC(mm_f32_to_bf16(stream, w.ca.to_kv, y.wkv2, (int64_t)2 * I * D));
        RC(mm_f32_to_bf16(stream, w.ca.to_out, y.wo2, (int64_t)D * I));
        RC(mm_layernorm(stream, y.x1, D, M, D, w.ca.gamma, w.ca.beta, nullptr, y.u2, D));
        RC(mm_gemm_bf16(stream, y.u2, D, y.wq2, D, M, I, D, y.q2, I, 0, nullptr));
        RC(mm_gemm_bf16(stream, b.cx, D, y.wkv2, D, Mc, 2 * I, D, y.kv2, 2 * I, 0, nullptr));
        RC(mm_attend(stream, y.q2, (int64_t)n * I, 64, I, y.kv2, (int64_t)L * 2 * I, 64, 2 * I, y.kv2 + I, (int64_t)L * 2 * I, 64, 2 * I,
                     y.o2, (int64_t)n * I, 64, I, B, H, n, L, ctx_mask, L, 1, w.ca.q_scale, w.ca.k_scale, w.ca.null_kv, w.ca.null_kv + (size_t)H * 64, 8.f, 64));
        RC(mm_gemm_bf16(stream, y.o2, I, y.wo2, I, M, D, I, y.x2, D, 1, y.x1));
        // ---- feed forward (mmp.py:79-89, 188): plain [x | gate] halves padded to Fp (the backward recomputes GEGLU from the saved pre-activation)
        HC(hipMemsetAsync(y.w1p, 0, (size_t)2 * Fp * D * 2, s));
        RC(mm_f32_to_bf16(stream, w.ff.w1, y.w1p, (int64_t)F * D));
        RC(mm_f32_to_bf16(stream, w.ff.w1 + (size_t)F * D, y.w1p + (size_t)Fp * D, (int64_t)F * D));
        if (Fp != F) {      // w2 [D][F] -> bf16 [D][Fp], zero columns: row by row through a dense bf16 copy in tW
            HC(hipMemsetAsync(y.w2p, 0, (size_t)D * Fp * 2, s));
            RC(mm_f32_to_bf16(stream, w.ff.w2, b.tW, (int64_t)D * F));
            HC(hipMemcpy2DAsync(y.w2p, (size_t)Fp * 2, b.tW, (size_t)F * 2, (size_t)F * 2, D, hipMemcpyDeviceToDevice, s));
        } else {
            RC(mm_f32_to_bf16(stream, w.ff.w2, y.w2p, (int64_t)D * F));
        }
        HC(hipMemsetAsync(y.g2p, 0, (size_t)Fp * 4, s));
        HC(hipMemsetAsync(y.b2p, 0, (size_t)Fp * 4, s));
        HC(hipMemcpyAsync(y.g2p, w.ff.g2, (size_t)F * 4, hipMemcpyDeviceToDevice, s));
        if (w.ff.b2) HC(hipMemcpyAsync(y.b2p, w.ff.b2, (size_t)F * 4, hipMemcpyDeviceToDevice, s));
        RC(mm_layernorm(stream, y.x2, D, M, D, w.ff.g1, w.ff.b1, nullptr, y.u3, D));
        RC(mm_gemm_bf16(stream, y.u3, D, y.w1p, D, M, 2 * Fp, D, y.h, 2 * Fp, 0, nullptr));
        RC(mm_geglu_ln(stream, y.h, 2 * Fp, M, F, Fp, y.g2p, w.ff.b2 ? y.b2p : nullptr, y.z, Fp));
        RC(mm_gemm_bf16(stream, y.z, Fp, y.w2p, Fp, M, D, Fp, xn, D, 1, y.x2));
    }
    // ---- head on the rows that carry a label (mmp.py:330-343)
    RC(mm_f32_to_bf16(stream, d.to_logits, b.wl, (int64_t)V * D));
    RC(mm_layernorm(stream, b.xL, D, R, D, d.final_gamma, d.final_beta, row_index, b.e, D));
    RC(mm_gemm_bf16(stream, b.e, D, b.wl, D, R, V, D, b.logits, V, 1, nullptr));
    RC(mm_ce_loss(stream, b.logits, V, R, V, labels_rows, -100, b.rowloss, loss_out));
    if (logits_rows_out) HC(hipMemcpyAsync(logits_rows_out, b.logits, (size_t)R * V * 4, hipMemcpyDeviceToDevice, s));

    // ================================================================ backward (training.py TransformerTrainFn.backward, gloss == 1: the caller scales)
    HC(hipMemsetAsync(b.dres, 0, (size_t)M * D * 4, s));
    RC(mm_ce_bwd(stream, b.logits, V, R, V, labels_rows, 1.0f / (float)R, b.dl, V));
    RC(wgrad(stream, s, b, b.dl, V, V, b.e, D, D, R, d.d_to_logits));
    RC(dgrad(stream, s, b, b.dl, V, V, b.wl, D, R, b.de, D));
    RC(mm_layernorm_bwd(stream, b.xL, D, b.de, D, d.final_gamma, row_index, R, D, b.dres, D, 0, d.d_final_gamma, b.ln_ws));
    int dcx_i = -1;      // ping-pong buffer holding the context gradient so far (text projection only)
    for (int l = q.depth - 1; l >= 0; --l) {
        const mm_train_layer& w = d.layers[l];
        LayerBufs& y = layers[l];
        // ---- feed forward (training.py _ff_backward)
        RC(mm_f32_to_bf16(stream, b.dres, b.dy, (int64_t)M * D));
        RC(dgrad(stream, s, b, b.dy, D, D, y.w2p, Fp, M, b.dz, Fp));
        RC(wgrad(stream, s, b, b.dy, D, D, y.z, Fp, Fp, M, b.gtmp));
        HC(hipMemcpy2DAsync(w.ff.d_w2, (size_t)F * 4, b.gtmp, (size_t)Fp * 4, (size_t)F * 4, D, hipMemcpyDeviceToDevice, s));
        RC(mm_geglu_ln_bwd(stream, y.h, 2 * Fp, b.dz, Fp, y.g2p, M, F, Fp, b.dh, 2 * Fp, b.gtmp, b.ln_ws));
        HC(hipMemcpyAsync(w.ff.d_g2, b.gtmp, (size_t)F * 4, hipMemcpyDeviceToDevice, s));
        RC(wgrad(stream, s, b, b.dh, 2 * Fp, 2 * Fp, y.u3, D, D, M, b.gtmp));
        HC(hipMemcpyAsync(w.ff.d_w1, b.gtmp, (size_t)F * D * 4, hipMemcpyDeviceToDevice, s));
        HC(hipMemcpyAsync(w.ff.d_w1 + (size_t)F * D, b.gtmp + (size_t)Fp * D, (size_t)F * D * 4, hipMemcpyDeviceToDevice, s));
        RC(dgrad(stream, s, b, b.dh, 2 * Fp, 2 * Fp, y.w1p, D, M, b.du, D));
        RC(mm_layernorm_bwd(stream, y.x2, D, b.du, D, w.ff.g1, nullptr, M, D, b.dres, D, 1, w.ff.d_g1, b.ln_ws));
        // ---- cross attention
        RC(mm_f32_to_bf16(stream, b.dres, b.dy, (int64_t)M * D));
        RC(dgrad(stream, s, b, b.dy, D, D, y.wo2, I, M, b.dob, I));
        RC(wgrad(stream, s, b, b.dy, D, D, y.o2, I, I, M, w.ca.d_to_out));
        const float* nk = w.ca.null_kv;
        const float* nv = w.ca.null_kv + (size_t)H * 64;
        RC(mm_attention_bwd(stream, y.q2, (int64_t)n * I, 64, I, y.kv2, (int64_t)L * 2 * I, 64, 2 * I, y.kv2 + I, (int64_t)L * 2 * I, 64, 2 * I,
                            y.o2, (int64_t)n * I, 64, I, b.dob, (int64_t)n * I, 64, I, b.dqn, (int64_t)n * I, 64, I, b.dkn, (int64_t)L * I, 64, I,
                            b.dkv2 + I, (int64_t)L * 2 * I, 64, 2 * I, b.dnk, b.dnv, B, H, n, L, ctx_mask, L, w.ca.q_scale, w.ca.k_scale, nk, nv, 8.f));
        RC(mm_qk_norm_bwd(stream, y.q2, I, nullptr, H, b.dqn, I, nullptr, w.ca.q_scale, M, H, b.dq2, I, nullptr, b.part));
        RC(mm_colsum_f32(stream, b.part, (int)mm_qk_norm_bwd_blocks((int64_t)M * H), 64, w.ca.d_q_scale));
        RC(mm_qk_norm_bwd(stream, y.kv2, 2 * I, nullptr, H, b.dkn, I, nullptr, w.ca.k_scale, Mc, H, b.dkv2, 2 * I, nullptr, b.part));
        RC(mm_colsum_f32(stream, b.part, (int)mm_qk_norm_bwd_blocks((int64_t)Mc * H), 64, b.pair));
        RC(mm_qk_norm_bwd(stream, nullptr, 0, nk, H, nullptr, 0, b.dnk, w.ca.k_scale, (int64_t)B * H, 1, nullptr, 0, b.dnull, b.part));
        RC(mm_colsum_f32(stream, b.part, (int)mm_qk_norm_bwd_blocks((int64_t)B * H), 64, b.pair + 64));
        RC(mm_colsum_f32(stream, b.pair, 2, 64, w.ca.d_k_scale));                                                        // dks + dks_n
        RC(mm_colsum_f32(stream, b.dnull, B, H * 64, w.ca.d_null_kv));
        RC(mm_colsum_f32(stream, b.dnv, B, H * 64, w.ca.d_null_kv + (size_t)H * 64));
        RC(dgrad(stream, s, b, b.dq2, I, I, y.wq2, D, M, b.du, D));
        RC(wgrad(stream, s, b, b.dq2, I, I, y.u2, D, D, M, w.ca.d_to_q));
        RC(wgrad(stream, s, b, b.dkv2, 2 * I, 2 * I, b.cx, D, D, Mc, w.ca.d_to_kv));
        if (d.text_proj) {      // the context's gradient (only the projection needs it): dcx = dkv2 W_kv (+ what the layers above left)
            RC(tr64(stream, s, y.wkv2, 2 * I, D, D, b.tW));
            const int nxt = dcx_i < 0 ? 0 : 1 - dcx_i;
            RC(mm_gemm_bf16(stream, b.dkv2, 2 * I, b.tW, pad64(2 * I), Mc, D, pad64(2 * I), b.dcx[nxt], D, 1, dcx_i < 0 ? nullptr : b.dcx[dcx_i]));
            dcx_i = nxt;
        }
        RC(mm_layernorm_bwd(stream, y.x1, D, b.du, D, w.ca.gamma, nullptr, M, D, b.dres, D, 1, w.ca.d_gamma, b.ln_ws));
        // ---- self attention
        RC(mm_f32_to_bf16(stream, b.dres, b.dy, (int64_t)M * D));
        RC(dgrad(stream, s, b, b.dy, D, D, y.wo, I, M, b.dob, I));
        RC(wgrad(stream, s, b, b.dy, D, D, y.o, I, I, M, w.sa.d_to_out));
        nk = w.sa.null_kv;
        nv = w.sa.null_kv + (size_t)H * 64;
        RC(mm_attention_bwd(stream, y.qkv, (int64_t)n * 3 * I, 64, 3 * I, y.qkv + I, (int64_t)n * 3 * I, 64, 3 * I, y.qkv + 2 * I, (int64_t)n * 3 * I, 64, 3 * I,
                            y.o, (int64_t)n * I, 64, I, b.dob, (int64_t)n * I, 64, I, b.dqn, (int64_t)n * I, 64, I, b.dkn, (int64_t)n * I, 64, I,
                            b.dqkv + 2 * I, (int64_t)n * 3 * I, 64, 3 * I, b.dnk, b.dnv, B, H, n, n, nullptr, 0, w.sa.q_scale, w.sa.k_scale, nk, nv, 8.f));
        RC(mm_qk_norm_bwd(stream, y.qkv, 3 * I, nullptr, H, b.dqn, I, nullptr, w.sa.q_scale, M, H, b.dqkv, 3 * I, nullptr, b.part));
        RC(mm_colsum_f32(stream, b.part, (int)mm_qk_norm_bwd_blocks((int64_t)M * H), 64, w.sa.d_q_scale));
        RC(mm_qk_norm_bwd(stream, y.qkv + I, 3 * I, nullptr, H, b.dkn, I, nullptr, w.sa.k_scale, M, H, b.dqkv + I, 3 * I, nullptr, b.part));
        RC(mm_colsum_f32(stream, b.part, (int)mm_qk_norm_bwd_blocks((int64_t)M * H), 64, b.pair));
        RC(mm_qk_norm_bwd(stream, nullptr, 0, nk, H, nullptr, 0, b.dnk, w.sa.k_scale, (int64_t)B * H, 1, nullptr, 0, b.dnull, b.part));
        RC(mm_colsum_f32(stream, b.part, (int)mm_qk_norm_bwd_blocks((int64_t)B * H), 64, b.pair + 64));
        RC(mm_colsum_f32(stream, b.pair, 2, 64, w.sa.d_k_scale));
        RC(mm_colsum_f32(stream, b.dnull, B, H * 64, w.sa.d_null_kv));
        RC(mm_colsum_f32(stream, b.dnv, B, H * 64, w.sa.d_null_kv + (size_t)H * 64));
        RC(dgrad(stream, s, b, b.dqkv, 3 * I, 3 * I, y.wqkv, D, M, b.du, D));
        RC(wgrad(stream, s, b, b.dqkv, 3 * I, 3 * I, y.u, D, D, M, b.gtmp));
        HC(hipMemcpyAsync(w.sa.d_to_q, b.gtmp, (size_t)I * D * 4, hipMemcpyDeviceToDevice, s));
        HC(hipMemcpyAsync(w.sa.d_to_kv, b.gtmp + (size_t)I * D, (size_t)2 * I * D * 4, hipMemcpyDeviceToDevice, s));
        RC(mm_layernorm_bwd(stream, y.x0, D, b.du, D, w.sa.gamma, nullptr, M, D, b.dres, D, 1, w.sa.d_gamma, b.ln_ws));
    }
    // ---- embeddings / text projection
    HC(hipMemsetAsync(d.d_token_emb, 0, (size_t)d.vocab_rows * D * 4, s));
    if (n < d.seq_len) HC(hipMemsetAsync(d.d_pos_emb, 0, (size_t)d.seq_len * D * 4, s));
    RC(mm_embed_bwd(stream, ids, B, n, D, b.dres, d.d_token_emb, d.d_pos_emb));
    if (d.text_proj) {
        RC(mm_f32_to_bf16(stream, b.dcx[dcx_i], b.dcxb, (int64_t)Mc * D));
        RC(wgrad(stream, s, b, b.dcxb, D, D, b.te_b, td, td, Mc, d.d_text_proj));
    }
    return MM_OK;
}

}  // extern "C"
