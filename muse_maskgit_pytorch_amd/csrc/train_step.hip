// mm_train_step: the training step of the transformer -- forward with saved activations, cross-entropy on the labelled rows, and the whole hand-written
// backward -- as ONE C call: no allocation, no host synchronisation, no Python between the ~1100 launches.  The operators and their arithmetic are those of
// the operator-by-operator driver in training.py, so the loss and every gradient are bit-identical to it (tests/test_gpu_train_step.py).
// Reference: MaskGit.forward (muse_maskgit_pytorch.py:623-741) differentiating Transformer.forward (:279-348) with autograd.
// Scope of the C entry: the generator's cross-entropy path (token ids in, labels at the masked rows), optional text projection; dim_head 64 and
// n in {64, 128, 256} (the attention backward's query blocks).  Self-conditioning, conditioning ids, the critics' BCE heads and longer sequences stay on
// training.py's driver (muse_maskgit.py picks).  Linear layers: dX = dY W and dW = dY^T X are NT GEMMs on transposed bf16 copies, as there.
//
// Two streams.  The caller's stream carries the dependent chain only: forward, loss, and of the backward what the next operator needs (dX GEMMs, attention /
// LayerNorm / GEGLU backward).  Everything else is enqueued on a second stream of the library, forked from and joined to the caller's stream with events
// inside this call: (1) what depends on the parameters alone -- their bf16 operand copies and the transposes the dX GEMMs read -- runs ahead of the forward;
// (2) the transposes of a layer's saved activations follow that layer's forward; (3) the LEAVES of the backward -- every dW GEMM with its gradient-row
// transpose, split-K reduction and copies, and the reductions of the dgamma / scale / null key-value partials: nothing reads them before the call returns --
// follow the operator that produced their input.  Buffers a leaf reads are per layer (the chain moves on while the leaf is pending).  Same kernels, same
// inputs, same values; mm_debug_set2(4) keeps everything on the caller's stream (A/B: 14.7-15.1 vs 15.7-16.3 ms per C2 step, same box).
// Three launches of the driver are fused away here with identical results: the bf16 image of the residual-stream gradient is written by the LayerNorm backward
// that produced it (no f32->bf16 pass), the cross-entropy value comes out of its backward kernel (no forward pass over the logits), and the head's
// dX = dlogits . W contracts over the vocabulary with split-K (training.py does the same: _dgrad_long_k).
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

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
    bf16_t *twqkv, *two, *twq2, *twkv2, *two2, *tw1p, *tw2p;  // transposed weights  [K][pad64(N)]  (B operand of dX = dY W)
    bf16_t *tu, *to, *tu2, *to2, *tu3, *tz;                // transposed saved activations [K][M]  (B operand of dW = dY^T X)
    float *part[2][3], *pair[2], *dnull[2], *dnv[2];       // leaf-reduction inputs of the cross- (0) / self- (1) attention backward, read by the side stream
    float *lnw[3], *glw;                                   // dgamma partials of the three LayerNorm(dim) backwards / of the GEGLU + LayerNorm(inner) backward
    bf16_t *dy[3], *dh, *dq2, *dkv2, *dqkv;                // this layer's gradient rows (A operand of its dW GEMMs, which run on the side stream while the chain moves on)
};

struct Bufs {
    LayerBufs* L;            // host array [depth]
    float* xL;               // output of the last layer
    bf16_t *tok_b, *pos_b, *te_b, *wtp, *cx, *wl, *e;
    float *logits, *rowloss;
    // backward scratch
    float *dres, *gtmp, *wg_ws, *ln_ws, *dnk, *dcx[2];
    bf16_t *dl, *de, *dz, *du, *dob, *dqn, *dkn, *dcxb, *tA, *tB, *tW;
    float *hd_ws, *de32;                                   // the head's split-K dX: slabs, fp32 sum
    int hd_splits;
    float* gtmp2;                                          // LayerNorm(inner) gain gradient padded to Fp (caller's stream; gtmp belongs to the dW GEMMs)
    unsigned char* emb_ws;                                 // workspace of the embedding backward's two-level sum
    bf16_t *twl, *tcx;                                     // transposed to_logits weight [D][V], transposed context [D][Mcp]
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
        y.twqkv = A.take<bf16_t>(D * (size_t)pad64((int)(3 * I))); y.two = A.take<bf16_t>(I * (size_t)pad64((int)D)); y.twq2 = A.take<bf16_t>(D * (size_t)pad64((int)I));
        y.twkv2 = A.take<bf16_t>(D * (size_t)pad64((int)(2 * I))); y.two2 = A.take<bf16_t>(I * (size_t)pad64((int)D));
        y.tw1p = A.take<bf16_t>(D * (size_t)pad64((int)(2 * Fp))); y.tw2p = A.take<bf16_t>(Fp * (size_t)pad64((int)D));
        y.tu = A.take<bf16_t>(D * M); y.to = A.take<bf16_t>(I * M); y.tu2 = A.take<bf16_t>(D * M); y.to2 = A.take<bf16_t>(I * M); y.tu3 = A.take<bf16_t>(D * M); y.tz = A.take<bf16_t>(Fp * M);
        for (int i = 0; i < 3; ++i) y.dy[i] = A.take<bf16_t>(M * D);
        for (int i = 0; i < 3; ++i) y.lnw[i] = A.take<float>((size_t)mm_ln_bwd_workspace_floats((int)M, (int)D));
        y.glw = A.take<float>((size_t)mm_ln_bwd_workspace_floats((int)M, (int)Fp));
        y.dh = A.take<bf16_t>(M * 2 * Fp); y.dq2 = A.take<bf16_t>(M * I); y.dkv2 = A.take<bf16_t>(Mc * 2 * I); y.dqkv = A.take<bf16_t>(M * 3 * I);
        const size_t nvec_l = (M > Mc ? M : Mc) * q.H;
        for (int a = 0; a < 2; ++a) {
            for (int i = 0; i < 3; ++i) y.part[a][i] = A.take<float>((size_t)mm_qk_norm_bwd_blocks((int64_t)nvec_l) * 64 + 128);
            y.pair[a] = A.take<float>(128); y.dnull[a] = A.take<float>((size_t)q.B * q.H * 64); y.dnv[a] = A.take<float>((size_t)q.B * q.H * 64);
        }
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
    // split-K slabs of the dW GEMMs: splits x N_ x K_ floats, the largest over the shapes wgrad() is called with below
    auto slabs = [](long N_, long K_, long rows) -> size_t {
        if (k_gemm_tn_prefer((int)rows, (int)N_, (int)K_, N_, K_)) {      // (the rule looks at sizes only for dense rows; strides of the step's operands are multiples of 8)
            const int st_ = k_gemm_tn_splits((int)rows, (int)N_, (int)K_);
            return st_ > 1 ? (size_t)st_ * N_ * K_ : 0;
        }
        const int s_ = (K_ % 4 == 0) ? mm_gemm_wgrad_splits((int)N_, (int)K_, pad64((int)rows)) : 1;
        return s_ > 1 ? (size_t)s_ * N_ * K_ : 0;
    };
    size_t wmax = slabs(V, D, R);
    for (size_t c : {slabs(D, Fp, M), slabs(2 * Fp, D, M), slabs(D, I, M), slabs(I, D, M), slabs(2 * I, D, Mc), slabs(3 * I, D, M), slabs(D, q.td, Mc)}) wmax = c > wmax ? c : wmax;
    b.wg_ws = A.take<float>(wmax + 64);
    // the head's dX = dl W_logits contracts over the vocabulary into only R/128 x D/128 tiles: split-K as well (slabs + fp32 sum on the caller's stream)
    b.hd_splits = mm_gemm_wgrad_splits((int)R, (int)D, (int)V);
    b.hd_ws = A.take<float>(b.hd_splits > 1 ? (size_t)b.hd_splits * R * D : 0);
    b.de32 = A.take<float>(b.hd_splits > 1 ? R * D : 0);
    size_t lnw = (size_t)mm_ln_bwd_workspace_floats((int)M, (int)D), lnw2 = (size_t)mm_ln_bwd_workspace_floats((int)M, (int)Fp);
    b.ln_ws = A.take<float>(lnw > lnw2 ? lnw : lnw2);
    b.dnk = A.take<float>((size_t)q.B * q.H * 64);
    b.dcx[0] = A.take<float>(d.text_proj ? Mc * D : 0); b.dcx[1] = A.take<float>(d.text_proj ? Mc * D : 0); b.dcxb = A.take<bf16_t>(d.text_proj ? Mc * D : 0);
    b.dl = A.take<bf16_t>(R * V); b.de = A.take<bf16_t>(R * D); b.dz = A.take<bf16_t>(M * Fp);
    b.du = A.take<bf16_t>(M * D); b.dob = A.take<bf16_t>(M * I); b.dqn = A.take<bf16_t>(M * I); b.dkn = A.take<bf16_t>((M > Mc ? M : Mc) * I);
    // transposed copies: tA = T(gradient rows), tB = T(saved activation rows), tW = T(weight)
    size_t ta = wide * (size_t)M; if (V * Rp > ta) ta = V * Rp; if (2 * I * Mcp > ta) ta = 2 * I * Mcp;
    size_t tb = (Fp > D ? Fp : D) * (size_t)M; if (I * (size_t)M > tb) tb = I * (size_t)M; if (D * Rp > tb) tb = D * Rp; if ((size_t)(q.td > (int)D ? q.td : (int)D) * Mcp > tb) tb = (size_t)(q.td > (int)D ? q.td : (int)D) * Mcp;
    size_t tw = D * V; if (D * wide > tw) tw = D * wide; if (Fp * (size_t)pad64((int)D) > tw) tw = Fp * (size_t)pad64((int)D);
    b.tA = A.take<bf16_t>(ta); b.tB = A.take<bf16_t>(tb); b.tW = A.take<bf16_t>(tw);
    b.gtmp2 = A.take<float>(Fp);
    b.twl = A.take<bf16_t>(D * V); b.tcx = A.take<bf16_t>(D * Mcp);
    b.emb_ws = A.take<unsigned char>(k_embed_bwd_workspace_bytes(q.B, q.n, q.D));
}

// out [cols][Rp] = x [rows][cols]^T, Rp = rows rounded up to 64, padding columns zero   (training.py _t)
int tr64(mm_stream_t st, hipStream_t s, const bf16_t* x, long rows, long cols, long ld, bf16_t* out) {
    // (round 6: the zero padding of the last 64-row block is written by the transpose itself -- in front of the head's dW GEMM the memset this replaces cleared
    //  vocabulary x Rp x 2 bytes = 730 MB per step)
    (void)st;
    return k_transpose_bf16(s, x, rows, cols, ld, out, pad64((int)rows), 1);
}
// dW fp32 [N_][K_] = dY^T X for dY bf16 [rows][N_] (ld ldy), X bf16 [rows][K_] (ld ldx)   (training.py _wgrad)
// xt: the transposed activation [K_][Rp] when the side stream has already made it (else nullptr: made here into tB)
int wgrad(mm_stream_t st, hipStream_t s, const Bufs& b, const bf16_t* dy, long ldy, int N_, const bf16_t* x, long ldx, int K_, long rows, float* out, const bf16_t* xt = nullptr) {
    if (k_gemm_tn_prefer((int)rows, N_, K_, ldy, ldx)) {      // round 6: the operands as they are (gemm_tn.hip) -- no transposed copy of dy (nor of x: the caller skipped it)
        const int sp = k_gemm_tn_splits((int)rows, N_, K_);
        if (sp <= 1) return k_gemm_tn(s, dy, ldy, x, ldx, (int)rows, N_, K_, 1, out);
        RC(k_gemm_tn(s, dy, ldy, x, ldx, (int)rows, N_, K_, sp, b.wg_ws));
        return k_colsum(s, b.wg_ws, sp, (long)N_ * K_, out);
    }
    RC(tr64(st, s, dy, rows, N_, ldy, b.tA));
    if (!xt) RC(tr64(st, s, x, rows, K_, ldx, b.tB));
    const bf16_t* tB = xt ? xt : b.tB;
    const int Rp = pad64((int)rows);
    const int splits = (K_ % 4 == 0) ? mm_gemm_wgrad_splits(N_, K_, Rp) : 1;
    if (splits <= 1) return mm_gemm_bf16(st, b.tA, Rp, tB, Rp, N_, K_, Rp, out, K_, 1, nullptr);      // (every K_ here is a multiple of 64: rows of the gradient are dense)
    return mm_gemm_wgrad(st, b.tA, Rp, tB, Rp, N_, K_, Rp, splits, b.wg_ws, out);
}
// dX bf16 [rows][K_] = dY W for dY bf16 [rows][N_], W bf16 [N_][K_]   (training.py _dgrad)
// wt: the transposed weight [K_][Np] when the side stream has already made it (else nullptr: made here into tW)
int dgrad(mm_stream_t st, hipStream_t s, const Bufs& b, const bf16_t* dy, long ldy, int N_, const bf16_t* w, int K_, long rows, bf16_t* out, long ldo, const bf16_t* wt = nullptr) {
    if (!wt) RC(tr64(st, s, w, N_, K_, K_, b.tW));       // [K_][Np]
    return mm_gemm_bf16(st, dy, ldy, wt ? wt : b.tW, pad64(N_), (int)rows, K_, pad64(N_), out, ldo, 0, nullptr);
}

// The second stream and its events: one per device, created on first use, reused by every step (a step joins the side stream before it returns, so an event is
// never re-recorded while a wait on its previous record is still pending).
struct Side {
    hipStream_t s2 = nullptr;
    std::vector<hipEvent_t> ev;
    size_t used = 0;
};
constexpr int SIDE_DEVS = 16;
Side g_side[SIDE_DEVS];
std::mutex g_side_mu[SIDE_DEVS];      // a device's side stream and event pool serve ONE step at a time: two host threads training on one device take turns

// *out = the device's side state with `lock` held until the caller drops it, or nullptr (one-stream step) for a device ordinal beyond the table
int side_get(Side** out, std::unique_lock<std::mutex>& lock) {
    int dev = 0;
    *out = nullptr;
    HC(hipGetDevice(&dev));
    if (dev < 0 || dev >= SIDE_DEVS) return MM_OK;
    lock = std::unique_lock<std::mutex>(g_side_mu[dev]);
    Side& sd = g_side[dev];
    if (!sd.s2) HC(hipStreamCreateWithFlags(&sd.s2, hipStreamNonBlocking));
    sd.used = 0;
    *out = &sd;
    return MM_OK;
}
// mark(): an event recorded at the current tail of `from`; await(): `to` continues only after that point.  Both are no-ops when the step runs on one stream.
int mark(Side* sd, hipStream_t from, hipEvent_t* out) {
    *out = nullptr;
    if (!sd) return MM_OK;
    if (sd->used == sd->ev.size()) {
        hipEvent_t e;
        HC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        sd->ev.push_back(e);
    }
    *out = sd->ev[sd->used++];
    HC(hipEventRecord(*out, from));
    return MM_OK;
}
int await(hipStream_t to, hipEvent_t e) {
    if (e) HC(hipStreamWaitEvent(to, e, 0));
    return MM_OK;
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

    // ---- the side stream (see the head of this file); sd == nullptr: one stream, mark / await are no-ops, s2 == s
    // mm_debug_set2 bit 4 (A/B and the bit-identity test): everything on the caller's stream.  (Round 6: an explicit call instead of the MM_TRAIN_SIDE
    // environment variable -- the library reads no environment variable any more; tests/test_host_logic.py checks that it does not even import getenv.)
    Side* sd = nullptr;
    std::unique_lock<std::mutex> side_lock;      // (released when the call returns: the step joins the side stream before that)
    if (!(g_mm_debug2 & 4)) RC(side_get(&sd, side_lock));
    hipStream_t s2 = sd ? sd->s2 : s;
    mm_stream_t stream2 = (mm_stream_t)s2;
    hipEvent_t e_f;
    const int lnb = k_ln_bwd_blocks(M);
    // FORK(): the side stream may read what the caller's stream has produced so far.  What follows a FORK on the side stream are leaves of the step -- the dW
    // GEMMs (nothing reads a weight gradient before the call returns) and the reductions of dgamma / scale / null key-value partials.
#define FORK()                      \
    do {                            \
        RC(mark(sd, s, &e_f));      \
        RC(await(s2, e_f));         \
    } while (0)
    hipEvent_t e_in, e_w[256], e_wl, e_tw[256], e_twl, e_a[256], e_ta[256], e_tcx = nullptr, e_out;
    RC(mark(sd, s, &e_in));
    RC(await(s2, e_in));                                   // the parameters (and this workspace) as the caller's stream leaves them

    // ================================================================ side stream, part 1: bf16 operand copies of the parameters and their transposes
    // (round 6: a job table per launch -- train_prep.hip -- instead of ~190 conversions, memsets, strided copies and transposes: layer 0 first, so that the forward
    //  starts behind ONE small launch; the other layers and the head follow while it runs)
    {
        // the zero-initialised gradient tables of the embeddings (the embedding backward at the very end accumulates into them): cleared here, ahead of everything --
        // the caller's stream waits for e_w[0] (recorded behind this) in front of layer 0, long before it reaches the embedding backward
        HC(hipMemsetAsync(d.d_token_emb, 0, (size_t)d.vocab_rows * D * 4, s2));
        if (n < d.seq_len) HC(hipMemsetAsync(d.d_pos_emb, 0, (size_t)d.seq_len * D * 4, s2));
        PrepList P(s2);
        int marked = 0;
        for (int l = 0; l < q.depth; ++l) {
            const mm_train_layer& w = d.layers[l];
            LayerBufs& y = layers[l];
            P.add(w.sa.to_q, y.wqkv, y.twqkv, I, D, I, D, D, pad64(3 * I));                                      // [to_q ; to_kv] -> wqkv [3I][D], twqkv [D][3I]
            P.add(w.sa.to_kv, y.wqkv + (size_t)I * D, y.twqkv + I, 2 * I, D, 2 * I, D, D, pad64(3 * I));
            P.add(w.sa.to_out, y.wo, y.two, D, I, D, I, I, pad64(D));
            P.add(w.ca.to_q, y.wq2, y.twq2, I, D, I, D, D, pad64(I));
            P.add(w.ca.to_kv, y.wkv2, d.text_proj ? y.twkv2 : nullptr, 2 * I, D, 2 * I, D, D, pad64(2 * I));
            P.add(w.ca.to_out, y.wo2, y.two2, D, I, D, I, I, pad64(D));
            // feed forward (mmp.py:79-89): plain [x | gate] halves padded to Fp (the backward recomputes GEGLU from the saved pre-activation)
            P.add(w.ff.w1, y.w1p, y.tw1p, F, D, Fp, D, D, pad64(2 * Fp));
            P.add(w.ff.w1 + (size_t)F * D, y.w1p + (size_t)Fp * D, y.tw1p + Fp, F, D, Fp, D, D, pad64(2 * Fp));
            P.add(w.ff.w2, y.w2p, y.tw2p, D, F, D, Fp, Fp, pad64(D));                                            // w2 [D][F] -> [D][Fp], zero columns
            P.add(w.ff.g2, y.g2p, nullptr, 1, F, 1, Fp, 0, 0, 1);
            if (w.ff.b2) P.add(w.ff.b2, y.b2p, nullptr, 1, F, 1, Fp, 0, 0, 1);
            if (l == 0 || l == q.depth - 1) {
                RC(P.flush());
                hipEvent_t e;
                RC(mark(sd, s2, &e));
                for (; marked <= l; ++marked) e_w[marked] = e_tw[marked] = e;
            }
        }
        P.add(d.to_logits, b.wl, b.twl, V, D, V, D, D, pad64(V));
        RC(P.flush());
        RC(mark(sd, s2, &e_wl));
        e_twl = e_wl;
    }

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
        RC(await(s, e_w[l]));
        // ---- self attention (mmp.py:137-162, 186)
        RC(mm_layernorm(stream, y.x0, D, M, D, w.sa.gamma, w.sa.beta, nullptr, y.u, D));
        RC(mm_gemm_bf16(stream, y.u, D, y.wqkv, D, M, 3 * I, D, y.qkv, 3 * I, 0, nullptr));
        RC(mm_attend(stream, y.qkv, (int64_t)n * 3 * I, 64, 3 * I, y.qkv + I, (int64_t)n * 3 * I, 64, 3 * I, y.qkv + 2 * I, (int64_t)n * 3 * I, 64, 3 * I,
                     y.o, (int64_t)n * I, 64, I, B, H, n, n, nullptr, 0, 1, w.sa.q_scale, w.sa.k_scale, w.sa.null_kv, w.sa.null_kv + (size_t)H * 64, 8.f, 64));
        RC(mm_gemm_bf16(stream, y.o, I, y.wo, I, M, D, I, y.x1, D, 1, y.x0));
        // ---- cross attention (mmp.py:139-141, 155-157, 187)
        RC(mm_layernorm(stream, y.x1, D, M, D, w.ca.gamma, w.ca.beta, nullptr, y.u2, D));
        RC(mm_gemm_bf16(stream, y.u2, D, y.wq2, D, M, I, D, y.q2, I, 0, nullptr));
        RC(mm_gemm_bf16(stream, b.cx, D, y.wkv2, D, Mc, 2 * I, D, y.kv2, 2 * I, 0, nullptr));
        RC(mm_attend(stream, y.q2, (int64_t)n * I, 64, I, y.kv2, (int64_t)L * 2 * I, 64, 2 * I, y.kv2 + I, (int64_t)L * 2 * I, 64, 2 * I,
                     y.o2, (int64_t)n * I, 64, I, B, H, n, L, ctx_mask, L, 1, w.ca.q_scale, w.ca.k_scale, w.ca.null_kv, w.ca.null_kv + (size_t)H * 64, 8.f, 64));
        RC(mm_gemm_bf16(stream, y.o2, I, y.wo2, I, M, D, I, y.x2, D, 1, y.x1));
        // ---- feed forward (mmp.py:79-89, 188)
        RC(mm_layernorm(stream, y.x2, D, M, D, w.ff.g1, w.ff.b1, nullptr, y.u3, D));
        RC(mm_gemm_bf16(stream, y.u3, D, y.w1p, D, M, 2 * Fp, D, y.h, 2 * Fp, 0, nullptr));
        RC(mm_geglu_ln(stream, y.h, 2 * Fp, M, F, Fp, y.g2p, w.ff.b2 ? y.b2p : nullptr, y.z, Fp));
        RC(mm_gemm_bf16(stream, y.z, Fp, y.w2p, Fp, M, D, Fp, xn, D, 1, y.x2));
        // ---- side stream, part 2: this layer's saved activations transposed for the dW GEMMs of the backward
        RC(mark(sd, s, &e_a[l]));
        RC(await(s2, e_a[l]));
        // (round 6: a dW that reads its operands as they are -- gemm_tn.hip, wgrad() asks the same rule -- needs no transposed activation)
        if (l == 0) {
            if (!k_gemm_tn_prefer(Mc, 2 * I, D, 2 * I, D)) RC(tr64(stream2, s2, b.cx, Mc, D, D, b.tcx));
            RC(mark(sd, s2, &e_tcx));
        }
        if (!k_gemm_tn_prefer(M, D, Fp, D, Fp)) RC(tr64(stream2, s2, y.z, M, Fp, Fp, y.tz));
        if (!k_gemm_tn_prefer(M, 2 * Fp, D, 2 * Fp, D)) RC(tr64(stream2, s2, y.u3, M, D, D, y.tu3));
        if (!k_gemm_tn_prefer(M, D, I, D, I)) RC(tr64(stream2, s2, y.o2, M, I, I, y.to2));
        if (!k_gemm_tn_prefer(M, I, D, I, D)) RC(tr64(stream2, s2, y.u2, M, D, D, y.tu2));
        if (!k_gemm_tn_prefer(M, D, I, D, I)) RC(tr64(stream2, s2, y.o, M, I, I, y.to));
        if (!k_gemm_tn_prefer(M, 3 * I, D, 3 * I, D)) RC(tr64(stream2, s2, y.u, M, D, D, y.tu));
        RC(mark(sd, s2, &e_ta[l]));
    }
    // ---- head on the rows that carry a label (mmp.py:330-343)
    RC(await(s, e_wl));
    RC(mm_layernorm(stream, b.xL, D, R, D, d.final_gamma, d.final_beta, row_index, b.e, D));
    RC(mm_gemm_bf16(stream, b.e, D, b.wl, D, R, V, D, b.logits, V, 1, nullptr));
    // (the cross-entropy itself comes out of the backward kernel below: it recomputes the row's max and denominator anyway)
    if (logits_rows_out) HC(hipMemcpyAsync(logits_rows_out, b.logits, (size_t)R * V * 4, hipMemcpyDeviceToDevice, s));

    // ================================================================ backward (training.py TransformerTrainFn.backward, gloss == 1: the caller scales)
    HC(hipMemsetAsync(b.dres, 0, (size_t)M * D * 4, s));
    RC(k_ce_bwd(s, b.logits, V, R, V, labels_rows, 1.0f / (float)R, b.dl, V, b.rowloss));                               // mmp.py:343 and its gradient
    RC(k_ce_finish(s, b.rowloss, R, loss_out));
    // the head's dW is a leaf, but it streams the same vocabulary-wide dl as the head's dX on the caller's stream: side by side each slowed the other (dX 0.80 -> 1.12 ms once
    // the dW no longer waited behind its transposes).  It goes out BEHIND the dX (bit 8192: at once, A/B)
    const bool head_dw_late = !(g_mm_debug2 & 8192);
    if (!head_dw_late) {
        FORK();
        RC(wgrad(stream2, s2, b, b.dl, V, V, b.e, D, D, R, d.d_to_logits));
    }
    RC(await(s, e_twl));
    if (b.hd_splits > 1) {                                 // (training.py _dgrad_long_k)
        RC(mm_gemm_wgrad(stream, b.dl, V, b.twl, V, R, D, V, b.hd_splits, b.hd_ws, b.de32));
        RC(mm_f32_to_bf16(stream, b.de32, b.de, (int64_t)R * D));
    } else {
        RC(dgrad(stream, s, b, b.dl, V, V, b.wl, D, R, b.de, D, b.twl));
    }
    if (head_dw_late) {
        FORK();
        RC(wgrad(stream2, s2, b, b.dl, V, V, b.e, D, D, R, d.d_to_logits));
    }
    RC(mm_layernorm_bwd(stream, b.xL, D, b.de, D, d.final_gamma, row_index, R, D, b.dres, D, 0, d.d_final_gamma, b.ln_ws));
    RC(await(s, e_tcx));
    int dcx_i = -1;      // ping-pong buffer holding the context gradient so far (text projection only)
    for (int l = q.depth - 1; l >= 0; --l) {
        const mm_train_layer& w = d.layers[l];
        LayerBufs& y = layers[l];
        RC(await(s, e_tw[l]));
        RC(await(s, e_ta[l]));
        // ---- feed forward (training.py _ff_backward)
        // dy[0] = bf16(dres): written by the LayerNorm backward that produced dres (the layer above's), except under the head (its backward touches labelled rows only)
        if (l == q.depth - 1) {
            RC(mm_f32_to_bf16(stream, b.dres, y.dy[0], (int64_t)M * D));
            FORK();
        }
        RC(wgrad(stream2, s2, b, y.dy[0], D, D, y.z, Fp, Fp, M, b.gtmp, y.tz));
        HC(hipMemcpy2DAsync(w.ff.d_w2, (size_t)F * 4, b.gtmp, (size_t)Fp * 4, (size_t)F * 4, D, hipMemcpyDeviceToDevice, s2));
        RC(dgrad(stream, s, b, y.dy[0], D, D, y.w2p, Fp, M, b.dz, Fp, y.tw2p));
        RC(k_geglu_ln_bwd(s, y.h, 2 * Fp, b.dz, Fp, y.g2p, M, F, Fp, y.dh, 2 * Fp, nullptr, y.glw));
        FORK();
        RC(k_colsum(s2, y.glw, lnb, Fp, b.gtmp2));
        HC(hipMemcpyAsync(w.ff.d_g2, b.gtmp2, (size_t)F * 4, hipMemcpyDeviceToDevice, s2));
        RC(wgrad(stream2, s2, b, y.dh, 2 * Fp, 2 * Fp, y.u3, D, D, M, b.gtmp, y.tu3));
        HC(hipMemcpyAsync(w.ff.d_w1, b.gtmp, (size_t)F * D * 4, hipMemcpyDeviceToDevice, s2));
        HC(hipMemcpyAsync(w.ff.d_w1 + (size_t)F * D, b.gtmp + (size_t)Fp * D, (size_t)F * D * 4, hipMemcpyDeviceToDevice, s2));
        RC(dgrad(stream, s, b, y.dh, 2 * Fp, 2 * Fp, y.w1p, D, M, b.du, D, y.tw1p));
        RC(k_layernorm_bwd(s, y.x2, D, b.du, D, w.ff.g1, nullptr, M, D, b.dres, D, 1, nullptr, y.lnw[0], y.dy[1]));
        FORK();
        RC(k_colsum(s2, y.lnw[0], lnb, D, w.ff.d_g1));
        // ---- cross attention
        RC(wgrad(stream2, s2, b, y.dy[1], D, D, y.o2, I, I, M, w.ca.d_to_out, y.to2));
        RC(dgrad(stream, s, b, y.dy[1], D, D, y.wo2, I, M, b.dob, I, y.two2));
        const float* nk = w.ca.null_kv;
        const float* nv = w.ca.null_kv + (size_t)H * 64;
        RC(mm_attention_bwd(stream, y.q2, (int64_t)n * I, 64, I, y.kv2, (int64_t)L * 2 * I, 64, 2 * I, y.kv2 + I, (int64_t)L * 2 * I, 64, 2 * I,
                            y.o2, (int64_t)n * I, 64, I, b.dob, (int64_t)n * I, 64, I, b.dqn, (int64_t)n * I, 64, I, b.dkn, (int64_t)L * I, 64, I,
                            y.dkv2 + I, (int64_t)L * 2 * I, 64, 2 * I, b.dnk, y.dnv[0], B, H, n, L, ctx_mask, L, w.ca.q_scale, w.ca.k_scale, nk, nv, 8.f));
        RC(mm_qk_norm_bwd(stream, y.q2, I, nullptr, H, b.dqn, I, nullptr, w.ca.q_scale, M, H, y.dq2, I, nullptr, y.part[0][0]));
        RC(mm_qk_norm_bwd(stream, y.kv2, 2 * I, nullptr, H, b.dkn, I, nullptr, w.ca.k_scale, Mc, H, y.dkv2, 2 * I, nullptr, y.part[0][1]));
        RC(mm_qk_norm_bwd(stream, nullptr, 0, nk, H, nullptr, 0, b.dnk, w.ca.k_scale, (int64_t)B * H, 1, nullptr, 0, y.dnull[0], y.part[0][2]));
        // the leaf reductions (scale and null key / value gradients) leave the chain: side stream, same kernels in the same order
        FORK();
        RC(mm_colsum_f32(stream2, y.part[0][0], (int)mm_qk_norm_bwd_blocks((int64_t)M * H), 64, w.ca.d_q_scale));
        RC(mm_colsum_f32(stream2, y.part[0][1], (int)mm_qk_norm_bwd_blocks((int64_t)Mc * H), 64, y.pair[0]));
        RC(mm_colsum_f32(stream2, y.part[0][2], (int)mm_qk_norm_bwd_blocks((int64_t)B * H), 64, y.pair[0] + 64));
        RC(mm_colsum_f32(stream2, y.pair[0], 2, 64, w.ca.d_k_scale));                                                     // dks + dks_n
        RC(mm_colsum_f32(stream2, y.dnull[0], B, H * 64, w.ca.d_null_kv));
        RC(mm_colsum_f32(stream2, y.dnv[0], B, H * 64, w.ca.d_null_kv + (size_t)H * 64));
        RC(wgrad(stream2, s2, b, y.dq2, I, I, y.u2, D, D, M, w.ca.d_to_q, y.tu2));
        RC(wgrad(stream2, s2, b, y.dkv2, 2 * I, 2 * I, b.cx, D, D, Mc, w.ca.d_to_kv, b.tcx));
        RC(dgrad(stream, s, b, y.dq2, I, I, y.wq2, D, M, b.du, D, y.twq2));
        if (d.text_proj) {      // the context's gradient (only the projection needs it): dcx = dkv2 W_kv (+ what the layers above left)
            const int nxt = dcx_i < 0 ? 0 : 1 - dcx_i;
            RC(mm_gemm_bf16(stream, y.dkv2, 2 * I, y.twkv2, pad64(2 * I), Mc, D, pad64(2 * I), b.dcx[nxt], D, 1, dcx_i < 0 ? nullptr : b.dcx[dcx_i]));
            dcx_i = nxt;
        }
        RC(k_layernorm_bwd(s, y.x1, D, b.du, D, w.ca.gamma, nullptr, M, D, b.dres, D, 1, nullptr, y.lnw[1], y.dy[2]));
        FORK();
        RC(k_colsum(s2, y.lnw[1], lnb, D, w.ca.d_gamma));
        // ---- self attention
        RC(wgrad(stream2, s2, b, y.dy[2], D, D, y.o, I, I, M, w.sa.d_to_out, y.to));
        RC(dgrad(stream, s, b, y.dy[2], D, D, y.wo, I, M, b.dob, I, y.two));
        nk = w.sa.null_kv;
        nv = w.sa.null_kv + (size_t)H * 64;
        RC(mm_attention_bwd(stream, y.qkv, (int64_t)n * 3 * I, 64, 3 * I, y.qkv + I, (int64_t)n * 3 * I, 64, 3 * I, y.qkv + 2 * I, (int64_t)n * 3 * I, 64, 3 * I,
                            y.o, (int64_t)n * I, 64, I, b.dob, (int64_t)n * I, 64, I, b.dqn, (int64_t)n * I, 64, I, b.dkn, (int64_t)n * I, 64, I,
                            y.dqkv + 2 * I, (int64_t)n * 3 * I, 64, 3 * I, b.dnk, y.dnv[1], B, H, n, n, nullptr, 0, w.sa.q_scale, w.sa.k_scale, nk, nv, 8.f));
        RC(mm_qk_norm_bwd(stream, y.qkv, 3 * I, nullptr, H, b.dqn, I, nullptr, w.sa.q_scale, M, H, y.dqkv, 3 * I, nullptr, y.part[1][0]));
        RC(mm_qk_norm_bwd(stream, y.qkv + I, 3 * I, nullptr, H, b.dkn, I, nullptr, w.sa.k_scale, M, H, y.dqkv + I, 3 * I, nullptr, y.part[1][1]));
        RC(mm_qk_norm_bwd(stream, nullptr, 0, nk, H, nullptr, 0, b.dnk, w.sa.k_scale, (int64_t)B * H, 1, nullptr, 0, y.dnull[1], y.part[1][2]));
        FORK();
        RC(mm_colsum_f32(stream2, y.part[1][0], (int)mm_qk_norm_bwd_blocks((int64_t)M * H), 64, w.sa.d_q_scale));
        RC(mm_colsum_f32(stream2, y.part[1][1], (int)mm_qk_norm_bwd_blocks((int64_t)M * H), 64, y.pair[1]));
        RC(mm_colsum_f32(stream2, y.part[1][2], (int)mm_qk_norm_bwd_blocks((int64_t)B * H), 64, y.pair[1] + 64));
        RC(mm_colsum_f32(stream2, y.pair[1], 2, 64, w.sa.d_k_scale));
        RC(mm_colsum_f32(stream2, y.dnull[1], B, H * 64, w.sa.d_null_kv));
        RC(mm_colsum_f32(stream2, y.dnv[1], B, H * 64, w.sa.d_null_kv + (size_t)H * 64));
        RC(wgrad(stream2, s2, b, y.dqkv, 3 * I, 3 * I, y.u, D, D, M, b.gtmp, y.tu));
        HC(hipMemcpyAsync(w.sa.d_to_q, b.gtmp, (size_t)I * D * 4, hipMemcpyDeviceToDevice, s2));
        HC(hipMemcpyAsync(w.sa.d_to_kv, b.gtmp + (size_t)I * D, (size_t)2 * I * D * 4, hipMemcpyDeviceToDevice, s2));
        RC(dgrad(stream, s, b, y.dqkv, 3 * I, 3 * I, y.wqkv, D, M, b.du, D, y.twqkv));
        RC(k_layernorm_bwd(s, y.x0, D, b.du, D, w.sa.gamma, nullptr, M, D, b.dres, D, 1, nullptr, y.lnw[2], l > 0 ? layers[l - 1].dy[0] : nullptr));
        FORK();                                            // (also hands layers[l - 1].dy[0] to the dW GEMM that opens the next iteration)
        RC(k_colsum(s2, y.lnw[2], lnb, D, w.sa.d_gamma));
    }
    // ---- embeddings / text projection
    // (the embedding gradients' tables were zeroed on the side stream at the start of the call: 134 MB at the base size, off the dependent chain)
    RC(k_embed_bwd(s, ids, B, n, D, b.dres, d.d_token_emb, d.d_pos_emb, b.emb_ws));      // (two levels: train.hip embed_token_bwd2_kernel)
    if (d.text_proj) {
        RC(mm_f32_to_bf16(stream, b.dcx[dcx_i], b.dcxb, (int64_t)Mc * D));
        FORK();
        RC(wgrad(stream2, s2, b, b.dcxb, D, D, b.te_b, td, td, Mc, d.d_text_proj));
    }
#undef FORK
    RC(mark(sd, s2, &e_out));                              // join: the caller's stream owns every gradient when this call's work on it has run
    RC(await(s, e_out));
    return MM_OK;
}

}  // extern "C"
