// Host-side orchestration of the transformer forward and the MaskGit decode loop on one HIP stream.
// Pure launch sequencing: no allocation, no synchronisation, no host<->device copies inside the timed path
// (so the whole of mm_generate is hipGraph-capturable).  Reference: muse_maskgit_pytorch.py:187-195 (blocks),
// :279-335 (forward), :491-615 (generate).
#include <new>
#include <string.h>
#include <vector>

#include "common.h"
#include "muse_hip_internal.h"

// ---- optional event timing of the two dominant kernels of mm_generate (see mm_profile_* in muse_hip.h)
namespace prof {
struct Rec { hipEvent_t a, b; double work; };
bool enabled = false;
std::vector<Rec> recs[MM_PROF_SLOTS];
std::vector<Rec> pool;
inline Rec begin(hipStream_t s, double work) {
    Rec r;
    if (!pool.empty()) { r = pool.back(); pool.pop_back(); }
    else { (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b); }
    r.work = work;
    (void)hipEventRecord(r.a, s);
    return r;
}
inline void end(hipStream_t s, int slot, Rec& r) { (void)hipEventRecord(r.b, s); recs[slot].push_back(r); }
}  // namespace prof

// ---- debug trace (mm_debug_trace, tools/determinism_stress.py): when a device buffer is registered, the transformer forward hashes every
// intermediate it produces (one 64-bit position-sensitive integer sum per operator output, integer atomics: order-independent) so that
// two runs can be compared operator by operator.  Off (one pointer test per operator) in the product path.
namespace trace {
uint64_t* buf = nullptr;
int cap = 0, idx = 0;
unsigned char* cap_buf = nullptr;      // optional: raw copies of the operator outputs #first, #first + step, ... (cap_stride bytes apart)
size_t cap_stride = 0;
int cap_first = 0, cap_step = 1;
__global__ __launch_bounds__(256) void hash_kernel(const uint32_t* __restrict__ p, long nwords, unsigned long long* out) {
    unsigned long long acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (long)gridDim.x * blockDim.x)
        acc += ((unsigned long long)p[i] + 0x9E3779B97F4A7C15ull) * (2ull * (unsigned long long)i + 1ull);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
inline void point(hipStream_t s, const void* p, size_t bytes) {
    if (!buf || idx >= cap) return;
    (void)hipMemsetAsync(buf + idx, 0, 8, s);
    const long nwords = (long)(bytes / 4);
    int blocks = (int)((nwords + 256 * 8 - 1) / (256 * 8));
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(hash_kernel, dim3(blocks), dim3(256), 0, s, (const uint32_t*)p, nwords, (unsigned long long*)(buf + idx));
    if (cap_buf && idx >= cap_first && (idx - cap_first) % cap_step == 0 && bytes <= cap_stride)
        (void)hipMemcpyAsync(cap_buf + (size_t)((idx - cap_first) / cap_step) * cap_stride, p, bytes, hipMemcpyDeviceToDevice, s);
    ++idx;
}
}  // namespace trace
#define TR(ptr_, bytes_) trace::point(s, (ptr_), (size_t)(bytes_))

struct mm_transformer {
    mm_transformer_desc d;
    std::vector<mm_layer_weights> layers;
    int I;    // heads * dim_head
    int Fp;
};

namespace {

// ------------------------------------------------------------------------------------------------ small kernels
// text_embeds fp32 [B][L][text_dim] -> bf16 rows + key mask  (mmp.py:304: mask = (text_embeds != 0).any(-1))
__global__ __launch_bounds__(256) void text_context_kernel(const float* __restrict__ text, int B, int L, int text_dim,
                                                           bf16_t* __restrict__ out, int out_rows_per_batch, long ldo,
                                                           uint8_t* __restrict__ mask, int m, int drop_text) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * L) return;
    const int b = row / L, j = row - b * L;
    const float* tr = text + (size_t)row * text_dim;
    bf16_t* orow = out + ((size_t)b * out_rows_per_batch + j) * ldo;
    bool nz = false;
    for (int c = lane; c < text_dim; c += 64) {
        const float v = tr[c];
        nz |= (v != 0.f);
        orow[c] = f32_to_bf16(v);
    }
    const bool any = __ballot(nz) != 0ull;
    if (lane == 0 && mask) mask[(size_t)b * m + j] = (any && !drop_text) ? 1 : 0;
}

// ctx[b][L + c][:] = token_emb[cond_ids[b][c]][:]; mask = 1   (mmp.py:314-318)
__global__ __launch_bounds__(256) void gather_cond_kernel(const bf16_t* __restrict__ table, int D, const int64_t* __restrict__ idx,
                                                          int B, int nc, int vocab_rows, bf16_t* __restrict__ ctx,
                                                          uint8_t* __restrict__ mask, int m, int L) {
    const int chunks = D >> 3;
    const long total = (long)B * nc * chunks;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / chunks;
        const int c = (int)(i - r * chunks);
        const int b = (int)(r / nc), j = (int)(r - (long)b * nc);
        long id = idx[r];
        id = id < 0 ? 0 : (id >= vocab_rows ? vocab_rows - 1 : id);
        *reinterpret_cast<uint4*>(ctx + ((size_t)b * m + L + j) * D + c * 8) =
            *reinterpret_cast<const uint4*>(table + id * D + c * 8);
        if (c == 0 && mask) mask[(size_t)b * m + L + j] = 1;
    }
}

__global__ void fill_i64_kernel(int64_t* p, long n, int64_t v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

// ------------------------------------------------------------------------------------------------ workspace carving
struct Carver {
    unsigned char* base;
    size_t off;
    explicit Carver(void* p) : base((unsigned char*)p), off(0) {}
    template <typename T>
    T* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return r;
    }
    size_t used() const { return (off + 255) & ~(size_t)255; }
};

#define RC(x)                \
    do {                     \
        int _rc = (x);       \
        if (_rc) return _rc; \
    } while (0)

int gemm_dense(hipStream_t s, const bf16_t* X, int ldx, const bf16_t* W, int ldw, int M, int N, int K, void* out, long ldc,
               int out_kind, const float* resid) {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = MODE_DENSE;
    a.W = W; a.N = N; a.ldw = ldw; a.K = K; a.M = M; a.X = X; a.ldx = ldx;
    a.out = out; a.ldc = ldc; a.out_kind = out_kind; a.resid_f32 = resid; a.ldr = ldc;
    return mm_gemm_launch(a, s);
}

struct Bufs {       // activation scratch for `rows` token rows
    float* x;       // [rows][D] residual stream fp32
    bf16_t* xn;     // [rows][D]
    bf16_t* qkv;    // [rows][3I]
    bf16_t* att;    // [rows][I]
    bf16_t* h;      // [rows][2Fp]
    bf16_t* a;      // [rows][Fp]
    float* lnp;     // [rows][Fp / 32][2]: LayerNorm(inner) partial sums of the folded feed-forward
};

void carve_bufs(Carver& c, const mm_transformer* t, size_t rows, Bufs& b) {
    const int D = t->d.dim, I = t->I, Fp = t->Fp;
    b.x = c.take<float>(rows * D);
    b.xn = c.take<bf16_t>(rows * D);
    b.qkv = c.take<bf16_t>(rows * 3 * I);
    b.att = c.take<bf16_t>(rows * I);
    b.h = c.take<bf16_t>(rows * 2 * Fp);
    b.a = c.take<bf16_t>(rows * Fp);
    b.lnp = c.take<float>(rows * (Fp / 32) * 2);
}

// dst += FF(src)   (mmp.py:79-89 with the residual of :193 / the self-cond add of :328)
// addvec != NULL (src == dst): rows [add_from, rows) first get the row vector added in place, inside the first LayerNorm's pass
int ff_block(const mm_transformer* t, hipStream_t s, const mm_ff_weights& w, const float* src, float* dst, int rows, Bufs& b,
             const float* addvec = nullptr, int add_from = 0) {
    const int D = t->d.dim, F = t->d.ff_inner, Fp = t->Fp;
    if (addvec) RC(k_layernorm_addvec(s, dst, D, rows, D, w.ln1_gamma, w.ln1_beta, addvec, add_from, b.xn, D));
    else RC(k_layernorm(s, src, D, rows, D, w.ln1_gamma, w.ln1_beta, nullptr, b.xn, D));
    GemmArgs a1;      // Linear(D, 2F) with the GEGLU fused into the epilogue: w1 is packed GEGLU-interleaved, the GEMM emits gate*gelu(x)
    memset(&a1, 0, sizeof(a1));
    a1.mode = MODE_DENSE; a1.epi = EPI_GEGLU;
    a1.W = (const bf16_t*)w.w1; a1.N = 2 * Fp; a1.ldw = D; a1.K = D; a1.M = rows; a1.X = b.xn; a1.ldx = D;
    a1.out = b.h; a1.ldc = Fp; a1.out_kind = OUT_BF16;
    GemmArgs a2;      // Linear(F, D) + residual
    memset(&a2, 0, sizeof(a2));
    a2.mode = MODE_DENSE;
    a2.W = (const bf16_t*)w.w2; a2.N = D; a2.ldw = Fp; a2.K = Fp; a2.M = rows; a2.ldx = Fp;
    a2.out = dst; a2.ldc = D; a2.out_kind = OUT_F32; a2.resid_f32 = dst; a2.ldr = D;
    // LayerNorm(inner) folded into the GEMM pair (mmp.py:86-88): w1's epilogue emits per-row partial sums of its bf16 output, w2 runs on
    // that output directly with the gains folded into its weights and applies mean / rstd / bias in its epilogue -- the LayerNorm's
    // own pass over the [rows][Fp] activation (read + write, 2.8 ms per generate at the base config) disappears.  Every kernel of the
    // GEMM family implements both halves identically, so the result does not depend on which kernel a shape is dispatched to.
    const bool fold = w.w2_folded && w.ln2_c1 && w.ln2_c2 && !(g_mm_debug & (1 << 24)) && (D % 4) == 0;
    TR(b.xn, (size_t)rows * D * 2);
    if (fold) {
        a1.ln_part = b.lnp;
        RC(mm_gemm_launch(a1, s));
        TR(b.h, (size_t)rows * Fp * 2);
        TR(b.lnp, (size_t)rows * (Fp / 64) * 8);
        a2.W = (const bf16_t*)w.w2_folded; a2.X = b.h;
        a2.ln_part = b.lnp; a2.ln_np = 2 * Fp / 128; a2.ln_F = F; a2.ln_c1 = w.ln2_c1; a2.ln_c2 = w.ln2_c2;
        RC(mm_gemm_launch(a2, s));
        TR(dst, (size_t)rows * D * 4);
        return MM_OK;
    }
    RC(mm_gemm_launch(a1, s));
    TR(b.h, (size_t)rows * Fp * 2);
    RC(k_ln_bf16(s, b.h, Fp, rows, F, Fp, w.ln2_gamma, w.ln2_beta, b.a, Fp));
    a2.X = b.a;
    RC(mm_gemm_launch(a2, s));
    TR(dst, (size_t)rows * D * 4);
    return MM_OK;
}

// b.att = heads of SelfAttention(LN(x)) over `seqs` sequences of n tokens, before the output projection  (mmp.py:126-159, context = None)
int self_attn_core(const mm_transformer* t, hipStream_t s, const mm_attn_weights& w, int seqs, int n, Bufs& b) {
    const int D = t->d.dim, I = t->I, H = t->d.heads;
    const int rows = seqs * n;
    RC(k_layernorm(s, b.x, D, rows, D, w.ln_gamma, w.ln_beta, nullptr, b.xn, D));
    TR(b.xn, (size_t)rows * D * 2);
    const bf16_t* wq = (const bf16_t*)w.w_q;
    const bf16_t* wkv = (const bf16_t*)w.w_kv;
    if (wkv == wq + (size_t)I * D) {
        RC(gemm_dense(s, b.xn, D, wq, D, rows, 3 * I, D, b.qkv, 3 * I, OUT_BF16, nullptr));
    } else {
        RC(gemm_dense(s, b.xn, D, wq, D, rows, I, D, b.qkv, 3 * I, OUT_BF16, nullptr));
        RC(gemm_dense(s, b.xn, D, wkv, D, rows, 2 * I, D, b.qkv + I, 3 * I, OUT_BF16, nullptr));
    }
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = b.qkv; a.q_sb = (long)n * 3 * I; a.q_sh = 64; a.q_sn = 3 * I;
    a.k = b.qkv + I; a.k_sb = a.q_sb; a.k_sh = 64; a.k_sn = 3 * I;
    a.v = b.qkv + 2 * I; a.v_sb = a.q_sb; a.v_sh = 64; a.v_sn = 3 * I;
    a.out = b.att; a.o_sb = (long)n * I; a.o_sh = 64; a.o_sn = I;
    a.B = seqs; a.H = H; a.nq = n; a.nk = n;
    a.normalize = 1; a.q_scale = w.q_scale; a.k_scale = w.k_scale; a.null_k = w.null_k; a.null_v = w.null_v;
    a.scale = 8.f;
    TR(b.qkv, (size_t)rows * 3 * I * 2);
    RC(k_attention(s, a));
    TR(b.att, (size_t)rows * I * 2);
    return MM_OK;
}

// x += SelfAttention(x) over `seqs` sequences of n tokens  (mmp.py:126-162 with context = None, :189)
int self_attn_block(const mm_transformer* t, hipStream_t s, const mm_attn_weights& w, int seqs, int n, Bufs& b) {
    RC(self_attn_core(t, s, w, seqs, n, b));
    RC(gemm_dense(s, b.att, t->I, (const bf16_t*)w.w_out, t->I, seqs * n, t->d.dim, t->I, b.x, t->d.dim, OUT_F32, b.x));
    TR(b.x, (size_t)seqs * n * t->d.dim * 4);
    return MM_OK;
}

// x += CrossAttention(x, ctx) for `seqs` sequences; ckv = ctx @ to_kv^T given as [kv_seqs*m][2I]  (mmp.py:191)
int cross_attn_block(const mm_transformer* t, hipStream_t s, const mm_attn_weights& w, int seqs, int n, const bf16_t* ckv,
                     int m, int kv_batch_mod, const uint8_t* key_mask, Bufs& b) {
    const int D = t->d.dim, I = t->I, H = t->d.heads;
    const int rows = seqs * n;
    RC(k_layernorm(s, b.x, D, rows, D, w.ln_gamma, w.ln_beta, nullptr, b.xn, D));
    TR(b.xn, (size_t)rows * D * 2);
    RC(gemm_dense(s, b.xn, D, (const bf16_t*)w.w_q, D, rows, I, D, b.qkv, I, OUT_BF16, nullptr));
    TR(b.qkv, (size_t)rows * I * 2);
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = b.qkv; a.q_sb = (long)n * I; a.q_sh = 64; a.q_sn = I;
    a.k = ckv; a.k_sb = (long)m * 2 * I; a.k_sh = 64; a.k_sn = 2 * I;
    a.v = ckv + I; a.v_sb = a.k_sb; a.v_sh = 64; a.v_sn = 2 * I;
    a.out = b.att; a.o_sb = (long)n * I; a.o_sh = 64; a.o_sn = I;
    a.B = seqs; a.H = H; a.nq = n; a.nk = m;
    a.key_mask = key_mask; a.km_sb = m;
    a.normalize = 1; a.q_scale = w.q_scale; a.k_scale = w.k_scale; a.null_k = w.null_k; a.null_v = w.null_v;
    a.scale = 8.f; a.kv_batch_mod = kv_batch_mod;
    RC(k_attention(s, a));
    TR(b.att, (size_t)rows * I * 2);
    RC(gemm_dense(s, b.att, I, (const bf16_t*)w.w_out, I, rows, D, I, b.x, D, OUT_F32, b.x));
    TR(b.x, (size_t)rows * D * 4);
    return MM_OK;
}

int check_model(const mm_transformer* t) {
    if (!t) return mm_set_error(MM_ERR_SHAPE, "model handle is NULL");
    return MM_OK;
}

}  // namespace

extern "C" {

int mm_transformer_create(const mm_transformer_desc* desc, mm_transformer_t** out) {
    if (!desc || !out) return mm_set_error(MM_ERR_SHAPE, "transformer_create: NULL argument");
    const mm_transformer_desc& d = *desc;
    if (d.dim_head != 64) return mm_set_error(MM_ERR_UNSUPPORTED, "transformer: dim_head must be 64");
    if (d.dim <= 0 || d.dim % 64 || d.dim > 2048) return mm_set_error(MM_ERR_SHAPE, "transformer: dim must be a multiple of 64, <= 2048");
    if (d.depth <= 0 || d.heads <= 0 || !d.layers) return mm_set_error(MM_ERR_SHAPE, "transformer: depth/heads/layers");
    if (d.ff_inner_padded % 64 || d.ff_inner_padded < d.ff_inner) return mm_set_error(MM_ERR_SHAPE, "transformer: ff_inner_padded must be a multiple of 64 >= ff_inner");
    if (d.text_proj && (d.text_dim % 64)) return mm_set_error(MM_ERR_SHAPE, "transformer: text_dim must be a multiple of 64 when projected");
    if (!d.text_proj && d.text_dim != d.dim) return mm_set_error(MM_ERR_SHAPE, "transformer: text_proj is NULL but text_dim != dim");
    if (!d.token_emb || !d.pos_emb || !d.to_logits || !d.final_gamma) return mm_set_error(MM_ERR_SHAPE, "transformer: missing weight pointer");
    mm_transformer* t = new (std::nothrow) mm_transformer();
    if (!t) return mm_set_error(MM_ERR_HIP, "out of host memory");
    t->d = d;
    t->layers.assign(d.layers, d.layers + d.depth);
    t->d.layers = t->layers.data();
    t->I = d.heads * d.dim_head;
    t->Fp = d.ff_inner_padded;
    *out = t;
    return MM_OK;
}

void mm_transformer_destroy(mm_transformer_t* model) { delete model; }

size_t mm_context_workspace_bytes(const mm_transformer_t* t, int B, int L) {
    if (!t) return 0;
    Carver c(nullptr);
    if (t->d.text_proj) {
        c.take<bf16_t>((size_t)B * L * t->d.text_dim);
        c.take<bf16_t>((size_t)B * L * t->d.dim);
    }
    return c.used() + 256;
}

int mm_transformer_context(const mm_transformer_t* t, mm_stream_t stream, const float* text_embeds, int B, int L,
                           const int64_t* cond_ids, int nc, int drop_text, void* ctx, uint8_t* key_mask,
                           void* workspace, size_t workspace_bytes) {
    RC(check_model(t));
    hipStream_t s = (hipStream_t)stream;
    const int D = t->d.dim, m = L + nc;
    if (B <= 0 || L < 0 || nc < 0 || m <= 0) return mm_set_error(MM_ERR_SHAPE, "context: bad sizes");
    if (L > 0 && !text_embeds) return mm_set_error(MM_ERR_SHAPE, "context: text_embeds is NULL");
    if (nc > 0 && !cond_ids) return mm_set_error(MM_ERR_SHAPE, "context: cond_ids is NULL");
    if (workspace_bytes < mm_context_workspace_bytes(t, B, L)) return mm_set_error(MM_ERR_WORKSPACE, "context: workspace too small");
    bf16_t* ctxp = (bf16_t*)ctx;
    if (L > 0) {
        if (!t->d.text_proj) {
            hipLaunchKernelGGL(text_context_kernel, dim3((B * L + 3) / 4), dim3(256), 0, s, text_embeds, B, L, t->d.text_dim, ctxp,
                               m, (long)D, key_mask, m, drop_text);
            RC(mm_check_launch("text_context_kernel"));
        } else {
            Carver c(workspace);
            bf16_t* tb = c.take<bf16_t>((size_t)B * L * t->d.text_dim);
            bf16_t* proj = c.take<bf16_t>((size_t)B * L * D);
            hipLaunchKernelGGL(text_context_kernel, dim3((B * L + 3) / 4), dim3(256), 0, s, text_embeds, B, L, t->d.text_dim, tb,
                               L, (long)t->d.text_dim, key_mask, m, drop_text);
            RC(mm_check_launch("text_context_kernel"));
            RC(gemm_dense(s, tb, t->d.text_dim, (const bf16_t*)t->d.text_proj, t->d.text_dim, B * L, D, t->d.text_dim, proj, D,
                          OUT_BF16, nullptr));
            const hipError_t e = hipMemcpy2DAsync(ctxp, (size_t)m * D * 2, proj, (size_t)L * D * 2, (size_t)L * D * 2, B,
                                                  hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return mm_set_hip_error(e, "context: hipMemcpy2DAsync");
        }
    }
    if (nc > 0) {
        long chunks = (long)B * nc * (D / 8);
        int blocks = (int)((chunks + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(gather_cond_kernel, dim3(blocks), dim3(256), 0, s, (const bf16_t*)t->d.token_emb, D, cond_ids, B, nc,
                           t->d.vocab_rows, ctxp, key_mask, m, L);
        RC(mm_check_launch("gather_cond_kernel"));
    }
    return MM_OK;
}

size_t mm_transformer_workspace_bytes(const mm_transformer_t* t, int B, int n, int m) {
    if (!t) return 0;
    Carver c(nullptr);
    Bufs b;
    carve_bufs(c, t, (size_t)B * n, b);
    c.take<bf16_t>((size_t)B * m * 2 * t->I);    // cross K/V of one layer
    c.take<bf16_t>((size_t)B * n * t->d.dim);    // embed when the caller does not want it
    return c.used() + 256;
}

int mm_transformer_forward(const mm_transformer_t* t, mm_stream_t stream, const int64_t* ids, int B, int n,
                           const void* ctx, const uint8_t* key_mask, int m, const float* self_cond_embed,
                           void* embed_out, float* logits_out, void* workspace, size_t workspace_bytes) {
    RC(check_model(t));
    hipStream_t s = (hipStream_t)stream;
    const int D = t->d.dim, I = t->I;
    if (B <= 0 || n <= 0 || n > t->d.seq_len) return mm_set_error(MM_ERR_SHAPE, "forward: need 0 < n <= seq_len");   // mmp.py:293
    if (m <= 0 || !ctx || !key_mask || !ids) return mm_set_error(MM_ERR_SHAPE, "forward: ids/ctx/key_mask required");
    if (workspace_bytes < mm_transformer_workspace_bytes(t, B, n, m)) return mm_set_error(MM_ERR_WORKSPACE, "forward: workspace too small");
    const int rows = B * n;
    Carver c(workspace);
    Bufs b;
    carve_bufs(c, t, (size_t)rows, b);
    bf16_t* ckv = c.take<bf16_t>((size_t)B * m * 2 * I);
    bf16_t* emb = c.take<bf16_t>((size_t)rows * D);
    if (embed_out) emb = (bf16_t*)embed_out;

    trace::idx = 0;
    RC(k_embed(s, ids, rows, n, 0, (const bf16_t*)t->d.token_emb, t->d.vocab_rows, (const bf16_t*)t->d.pos_emb, D, b.x));
    TR(b.x, (size_t)rows * D * 4);
    if (t->d.self_cond && self_cond_embed)       // mmp.py:325-328 (zeros when absent: FF(0) still adds LN-beta terms = 0)
        RC(ff_block(t, s, t->d.self_cond_ff, self_cond_embed, b.x, rows, b));
    for (int l = 0; l < t->d.depth; ++l) {
        const mm_layer_weights& w = t->layers[l];
        RC(self_attn_block(t, s, w.self_attn, B, n, b));
        RC(gemm_dense(s, (const bf16_t*)ctx, D, (const bf16_t*)w.cross_attn.w_kv, D, B * m, 2 * I, D, ckv, 2 * I, OUT_BF16, nullptr));
        TR(ckv, (size_t)B * m * 2 * I * 2);
        RC(cross_attn_block(t, s, w.cross_attn, B, n, ckv, m, 0, key_mask, b));
        RC(ff_block(t, s, w.ff, b.x, b.x, rows, b));
    }
    RC(k_layernorm(s, b.x, D, rows, D, t->d.final_gamma, t->d.final_beta, nullptr, emb, D));
    TR(emb, (size_t)rows * D * 2);
    if (logits_out)
        RC(gemm_dense(s, emb, D, (const bf16_t*)t->d.to_logits, D, rows, t->d.dim_out, D, logits_out, t->d.dim_out, OUT_F32, nullptr));
    return MM_OK;
}

// ------------------------------------------------------------------------------------------------ generate
namespace {
constexpr float FS_MARGIN = 0.20f;    // the bound sits this many sigmas below the Gaussian quantile of the kept fraction: ~14 % instead of 10 % pass
struct GenBufs {
    Bufs b;                 // 2B sequences
    bf16_t* ctx;            // [B][m][D]
    uint8_t* masks;         // [2B][m]: cond masks then null masks
    bf16_t* ckv;            // [depth][B*m][2I]
    float* cvec;            // [depth][D]: to_out(null_v) of each cross-attention (base model null pass)
    bf16_t* nullv;          // [I] scratch
    int32_t* rows;          // [B*n]
    bf16_t* embc;           // [B*n][D]
    bf16_t* embn;           // [B*n][D]
    float* logits;          // [B*n][V]
    float* xc;              // [2*B*n][D]: the last layer's residual stream, compacted to the sampled rows
    bf16_t* attc;           // [2*B*n][I]
    // fused sampling (sampling_fused.hip): what the guidance-logits GEMM emits instead of the logits
    float* fs_thr;          // [B*n]
    float4* fs_stats;       // [B*n][V/256]
    float4* fs_cand;        // [B*n][V/256][FS_SLOT]
    void* fs_ws;            // scratch of the bound estimate (k_fused_threshold)
    void* ctx_ws; size_t ctx_ws_bytes;
};
void carve_gen(Carver& c, const mm_transformer* t, int B, int n, int L, int nc, GenBufs& g) {
    const int D = t->d.dim, I = t->I, m = L + nc;
    carve_bufs(c, t, (size_t)2 * B * n, g.b);
    g.ctx = c.take<bf16_t>((size_t)B * m * D);
    g.masks = c.take<uint8_t>((size_t)2 * B * m);
    g.ckv = c.take<bf16_t>((size_t)t->d.depth * B * m * 2 * I);
    g.cvec = c.take<float>((size_t)t->d.depth * D);
    g.nullv = c.take<bf16_t>((size_t)I + 64);
    g.rows = c.take<int32_t>((size_t)B * n);
    g.embc = c.take<bf16_t>((size_t)B * n * D);
    g.embn = c.take<bf16_t>((size_t)B * n * D);
    g.logits = c.take<float>((size_t)B * n * t->d.dim_out);
    g.xc = c.take<float>((size_t)2 * B * n * D);
    g.attc = c.take<bf16_t>((size_t)2 * B * n * I);
    const bool fs = t->d.logits_wcov && (t->d.dim_out % 256) == 0 && (D % 64) == 0;
    const size_t NT = fs ? (size_t)t->d.dim_out / 256 : 0;
    g.fs_thr = c.take<float>(fs ? (size_t)B * n : 0);
    g.fs_stats = c.take<float4>((size_t)B * n * NT);
    g.fs_cand = c.take<float4>((size_t)B * n * NT * FS_SLOT);
    g.fs_ws = c.take<unsigned char>(fs ? k_fused_threshold_ws_bytes(B * n, D) : 0);
    g.ctx_ws_bytes = mm_context_workspace_bytes(t, B, L);
    g.ctx_ws = c.take<unsigned char>(g.ctx_ws_bytes);
}
}  // namespace

int mm_debug_trace(uint64_t* device_buf, int capacity) {
    trace::buf = device_buf; trace::cap = device_buf ? capacity : 0; trace::idx = 0;
    return MM_OK;
}
int mm_debug_trace_count(void) { return trace::idx; }
int mm_debug_capture(void* device_buf, size_t stride_bytes, int first, int step) {
    trace::cap_buf = (unsigned char*)device_buf; trace::cap_stride = stride_bytes; trace::cap_first = first; trace::cap_step = step > 0 ? step : 1;
    return MM_OK;
}

int mm_profile_enable(int enable) {
    prof::enabled = enable != 0;
    return MM_OK;
}

int mm_profile_read(int slot, int64_t* launches, double* total_ms, double* total_work) {
    if (slot < 0 || slot >= MM_PROF_SLOTS) return mm_set_error(MM_ERR_SHAPE, "profile_read: bad slot");
    double ms = 0.0, work = 0.0;
    int64_t n = 0;
    for (prof::Rec& r : prof::recs[slot]) {
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return mm_set_hip_error(e, "profile_read: hipEventSynchronize");
        float t = 0.f;
        e = hipEventElapsedTime(&t, r.a, r.b);
        if (e != hipSuccess) return mm_set_hip_error(e, "profile_read: hipEventElapsedTime");
        ms += t; work += r.work; ++n;
        prof::pool.push_back(r);
    }
    prof::recs[slot].clear();
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (total_work) *total_work = work;
    return MM_OK;
}

size_t mm_generate_workspace_bytes(const mm_transformer_t* t, int B, int n, int L, int nc) {
    if (!t) return 0;
    Carver c(nullptr);
    GenBufs g;
    carve_gen(c, t, B, n, L, nc, g);
    return c.used() + 256;
}

int mm_generate(const mm_transformer_t* t, mm_stream_t stream, const mm_generate_params* p, void* workspace, size_t workspace_bytes) {
    RC(check_model(t));
    if (!p) return mm_set_error(MM_ERR_SHAPE, "generate: params is NULL");
    hipStream_t s = (hipStream_t)stream;
    const int B = p->batch, n = p->n, T = p->timesteps, L = p->L, nc = p->nc, m = L + nc;
    const int D = t->d.dim, I = t->I, V = t->d.dim_out;
    if (B <= 0 || n <= 0 || n > t->d.seq_len || T <= 0) return mm_set_error(MM_ERR_SHAPE, "generate: bad batch/n/timesteps");
    if (t->d.vocab_rows != t->d.num_tokens + 1) return mm_set_error(MM_ERR_SHAPE, "generate: transformer has no mask id (MaskGitTransformer required)");
    if (t->d.self_cond) return mm_set_error(MM_ERR_UNSUPPORTED, "generate: self-conditioning transformers use the stepwise path");
    if (!p->mask_counts || !p->temperatures || !p->ids || !p->scores) return mm_set_error(MM_ERR_SHAPE, "generate: schedule / outputs required");
    if (m <= 0) return mm_set_error(MM_ERR_SHAPE, "generate: empty context");
    if (p->cond_scale == 1.f) return mm_set_error(MM_ERR_UNSUPPORTED, "generate: cond_scale == 1 (single pass) uses the stepwise path");
    if ((p->noise_kind == MM_NOISE_GUMBEL || p->noise_kind == MM_NOISE_UNIFORM) && !p->noise)
        return mm_set_error(MM_ERR_SHAPE, "generate: noise tensor required for this noise_kind");
    if (workspace_bytes < mm_generate_workspace_bytes(t, B, n, L, nc)) return mm_set_error(MM_ERR_WORKSPACE, "generate: workspace too small");
    for (int i = 0; i < T; ++i)
        if (p->mask_counts[i] < 1 || p->mask_counts[i] > n || (i > 0 && p->mask_counts[i] > p->mask_counts[i - 1]))
            return mm_set_error(MM_ERR_SHAPE, "generate: mask_counts must be non-increasing within [1, n]");
    if (p->mask_counts[0] != n) return mm_set_error(MM_ERR_SHAPE, "generate: the first step must mask every token (mmp.py:519-563)");

    Carver c(workspace);
    GenBufs g;
    carve_gen(c, t, B, n, L, nc, g);
    const int M = B * n;            // rows of one CFG half
    const int64_t mask_id = t->d.num_tokens;

    // ---- step-invariant work: context, masks, cross-attention K/V of every layer, null-pass constants
    RC(mm_transformer_context(t, stream, p->text_embeds, B, L, p->cond_ids, nc, 0, g.ctx, g.masks, g.ctx_ws, g.ctx_ws_bytes));
    {   // null masks: text keys off, cond-id keys on (mmp.py:308-310, 318)
        hipError_t e = hipMemsetAsync(g.masks + (size_t)B * m, 0, (size_t)B * m, s);
        if (e != hipSuccess) return mm_set_hip_error(e, "generate: memset masks");
        if (nc > 0) {
            e = hipMemset2DAsync(g.masks + (size_t)B * m + L, m, 1, nc, B, s);
            if (e != hipSuccess) return mm_set_hip_error(e, "generate: memset2d masks");
        }
    }
    for (int l = 0; l < t->d.depth; ++l) {
        const mm_attn_weights& w = t->layers[l].cross_attn;
        bf16_t* ckv_l = g.ckv + (size_t)l * B * m * 2 * I;
        RC(gemm_dense(s, g.ctx, D, (const bf16_t*)w.w_kv, D, B * m, 2 * I, D, ckv_l, 2 * I, OUT_BF16, nullptr));
        if (nc == 0) {
            // softmax over the single unmasked (null) key is exactly 1 -> attention out = bf16(null_v) for every
            // query, so the null pass's cross-attention is the constant row to_out(null_v)
            RC(k_f32_to_bf16(s, w.null_v, g.nullv, I));
            RC(gemm_dense(s, g.nullv, I, (const bf16_t*)w.w_out, I, 1, D, I, g.cvec + (size_t)l * D, D, OUT_F32, nullptr));
        }
    }
    {
        hipLaunchKernelGGL(fill_i64_kernel, dim3(64), dim3(256), 0, s, p->ids, (long)M, mask_id);   // mmp.py:519
        RC(mm_check_launch("fill_i64_kernel"));
        const hipError_t e = hipMemsetAsync(p->scores, 0, (size_t)M * 4, s);                          // mmp.py:520
        if (e != hipSuccess) return mm_set_hip_error(e, "generate: memset scores");
    }

    Bufs& b = g.b;
    for (int step = 0; step < T; ++step) {
        const int k = p->mask_counts[step];
        const int R = B * k;
        RC(k_mask_step(s, p->scores, p->ids, B, n, k, mask_id, g.rows));                            // mmp.py:558-563
        if (p->trace_masked_ids) {
            const hipError_t e = hipMemcpyAsync(p->trace_masked_ids + (size_t)step * M, p->ids, (size_t)M * 8, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return mm_set_hip_error(e, "generate: trace copy");
        }
        // both CFG halves see the same ids: rows [0, M) = cond pass, [M, 2M) = null pass (mmp.py:250-252)
        RC(k_embed(s, p->ids, M, n, 0, (const bf16_t*)t->d.token_emb, t->d.vocab_rows, (const bf16_t*)t->d.pos_emb, D, b.x));
        {
            const hipError_t e = hipMemcpyAsync(b.x + (size_t)M * D, b.x, (size_t)M * D * 4, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return mm_set_hip_error(e, "generate: x copy");
        }
        // Only the R = B*k rows sampled at this step are read after the last layer (final norm + to_logits, below), and after the last
        // self-attention has mixed the tokens every operator is row-wise: its output projection, the cross-attention (its queries) and
        // the feed-forward run on the COMPACTED rows [cond R | null R] (every sample has exactly k of them, position-sorted, so a
        // sample's queries stay contiguous).  Identical values for the rows that matter; (1 - k/n) of that work is skipped.
        const bool compact_last = k < n && !(g_mm_debug & 16384);
        for (int l = 0; l < t->d.depth; ++l) {
            const mm_layer_weights& w = t->layers[l];
            const bf16_t* ckv_l = g.ckv + (size_t)l * B * m * 2 * I;
            if (l == t->d.depth - 1 && compact_last) {
                RC(self_attn_core(t, s, w.self_attn, 2 * B, n, b));
                RC(k_gather_rows16(s, b.x, (long)D * 4, g.rows, R, 0, D * 4, g.xc));
                RC(k_gather_rows16(s, b.x, (long)D * 4, g.rows, R, M, D * 4, g.xc + (size_t)R * D));
                RC(k_gather_rows16(s, b.att, (long)I * 2, g.rows, R, 0, I * 2, g.attc));
                RC(k_gather_rows16(s, b.att, (long)I * 2, g.rows, R, M, I * 2, g.attc + (size_t)R * I));
                RC(gemm_dense(s, g.attc, I, (const bf16_t*)w.self_attn.w_out, I, 2 * R, D, I, g.xc, D, OUT_F32, g.xc));
                Bufs bc = b;
                bc.x = g.xc; bc.att = g.attc;
                if (nc == 0) {
                    RC(cross_attn_block(t, s, w.cross_attn, B, k, ckv_l, m, 0, g.masks, bc));
                    RC(ff_block(t, s, w.ff, g.xc, g.xc, 2 * R, bc, g.cvec + (size_t)l * D, R));      // null rows += to_out(null_v)
                } else {
                    RC(cross_attn_block(t, s, w.cross_attn, 2 * B, k, ckv_l, m, B, g.masks, bc));
                    RC(ff_block(t, s, w.ff, g.xc, g.xc, 2 * R, bc));
                }
                break;
            }
            RC(self_attn_block(t, s, w.self_attn, 2 * B, n, b));
            if (nc == 0) {
                RC(cross_attn_block(t, s, w.cross_attn, B, n, ckv_l, m, 0, g.masks, b));
                RC(ff_block(t, s, w.ff, b.x, b.x, 2 * M, b, g.cvec + (size_t)l * D, M));      // null rows += to_out(null_v) (the constant cross-attention)
            } else {
                RC(cross_attn_block(t, s, w.cross_attn, 2 * B, n, ckv_l, m, B, g.masks, b));
                RC(ff_block(t, s, w.ff, b.x, b.x, 2 * M, b));
            }
        }
        // final norm + to_logits + CFG only at the rows that are sampled this step
        if (compact_last) {
            RC(k_layernorm(s, g.xc, D, R, D, t->d.final_gamma, t->d.final_beta, nullptr, g.embc, D));
            RC(k_layernorm(s, g.xc + (size_t)R * D, D, R, D, t->d.final_gamma, t->d.final_beta, nullptr, g.embn, D));
        } else {
            RC(k_layernorm(s, b.x, D, R, D, t->d.final_gamma, t->d.final_beta, g.rows, g.embc, D));
            RC(k_layernorm(s, b.x + (size_t)M * D, D, R, D, t->d.final_gamma, t->d.final_beta, g.rows, g.embn, D));
        }
        GemmArgs a;
        memset(&a, 0, sizeof(a));
        a.mode = MODE_CFG;
        a.W = (const bf16_t*)t->d.to_logits; a.N = V; a.ldw = D; a.K = D;
        a.M = R; a.X = g.embc; a.X2 = g.embn; a.ldx = D;
        a.out = g.logits; a.ldc = V; a.out_kind = OUT_F32; a.cfg_scale = p->cond_scale;
        a.debug = g_mm_debug;
        // Sampling without the logits round trip: the GEMM emits tile statistics + the candidates above a per-row lower bound of the k-th largest
        // logit (estimated from the row's embeddings and the vocabulary statistics of to_logits), the finishing kernel verifies the bound.
        const bool fused = t->d.logits_wcov && t->d.logits_wmean && p->status && !(p->flags & MM_GEN_NO_FUSED_SAMPLING) && (V % 256) == 0 &&
                           !(g_mm_debug & (8 | 4096 | 8192 | (1 << 25))) && mm_gemm_cfg2_eligible(a);
        const double gemm_flops = 2.0 * 2.0 * (double)R * (double)V * (double)D;      // cond + null rows
        if (fused) {
            RC(k_fused_threshold(s, g.embc, g.embn, D, R, D, p->cond_scale, t->d.logits_wmean, (const bf16_t*)t->d.logits_wcov, k_fused_z(p->k_keep, V, FS_MARGIN),
                                 g.fs_ws, g.fs_thr));
            a.out = nullptr;
            a.fs_thr = g.fs_thr; a.fs_stats = g.fs_stats; a.fs_cand = g.fs_cand;
            prof::Rec pr;
            if (prof::enabled) pr = prof::begin(s, gemm_flops);
            RC(mm_gemm_launch(a, s));
            if (prof::enabled) prof::end(s, 0, pr);
            FusedSampleArgs fa;
            memset(&fa, 0, sizeof(fa));
            fa.thr = g.fs_thr; fa.stats = g.fs_stats; fa.cand = g.fs_cand;
            fa.R = R; fa.V = V; fa.k_keep = p->k_keep; fa.rows = g.rows;
            fa.temperature = p->temperatures[step]; fa.noise_kind = p->noise_kind;
            fa.noise = p->noise ? p->noise + (size_t)step * M * V : nullptr; fa.noise_ld = V;
            fa.seed = p->seed; fa.row_offset = p->row_offset * (uint64_t)n; fa.step = (uint32_t)step;
            fa.ids = p->ids; fa.scores = p->scores; fa.fail_flag = p->status;
            if (prof::enabled) pr = prof::begin(s, 4.0 * (double)R * (double)V);      // logits-equivalent bytes (what a logits-reading sampler reads)
            RC(k_sample_fused(s, fa));                                                                  // mmp.py:576-609
            if (prof::enabled) prof::end(s, 1, pr);
        } else {
            {
                prof::Rec pr;
                if (prof::enabled) pr = prof::begin(s, gemm_flops);
                RC(mm_gemm_launch(a, s));
                if (prof::enabled) prof::end(s, 0, pr);
            }
            SampleArgs sa;
            memset(&sa, 0, sizeof(sa));
            sa.logits = g.logits; sa.ld = V; sa.R = R; sa.V = V; sa.k_keep = p->k_keep; sa.rows = g.rows;
            sa.temperature = p->temperatures[step]; sa.noise_kind = p->noise_kind;
            sa.noise = p->noise ? p->noise + (size_t)step * M * V : nullptr; sa.noise_ld = V;
            sa.seed = p->seed; sa.row_offset = p->row_offset * (uint64_t)n; sa.step = (uint32_t)step;
            sa.ids = p->ids; sa.scores = p->scores;
            prof::Rec pr;
            if (prof::enabled) pr = prof::begin(s, 4.0 * (double)R * (double)V);      // one fp32 read of each row
            RC(k_sample_rows(s, sa));                                                                // mmp.py:576-609
            if (prof::enabled) prof::end(s, 1, pr);
        }
        if (p->trace_ids) {
            const hipError_t e = hipMemcpyAsync(p->trace_ids + (size_t)step * M, p->ids, (size_t)M * 8, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return mm_set_hip_error(e, "generate: trace copy");
        }
        if (p->trace_scores) {
            const hipError_t e = hipMemcpyAsync(p->trace_scores + (size_t)step * M, p->scores, (size_t)M * 4, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return mm_set_hip_error(e, "generate: trace copy");
        }
    }
    return MM_OK;
}

}  // extern "C"
