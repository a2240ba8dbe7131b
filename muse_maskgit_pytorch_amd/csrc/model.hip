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
    int F8;   // fp8 engine (mm_transformer_desc.fp8): the layers' Linear weights are e4m3 rows + per-row scales, their activations are quantised per token
              // row by the producing kernel (fp8_act.hip), the products run on the K = 128 fp8 MFMA (gemm_fp8.hip); attention, to_logits, sampling: bf16 engine
    int P;    // 0: bf16 engine.  3 / 5 / 6: the 'bf16x3' precision tier (split.hip) -- every GEMM operand is P bf16 segments of an fp32 value, the
              // weights are packed to match ([N][P*K]), tables / q|k|v / GEMM outputs are fp32, attention runs on the fp32 MFMA (attention_f32.hip).
              // 2 / 3 with F16: the 'f16x2' tier -- the same engine on fp16 terms (two per value) and the fp16 MFMA
    int PC;   // the operand code the row producers take: P, or MM_SPLIT_F16 | P
    int F16;  // fp16 terms
    float alpha;   // fp16 terms: inverse of the power-of-two scale of the packed weight terms (mm_transformer_desc.split_alpha), applied to every accumulator
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

// self-conditioning input of the next step: the fp32 view of the bf16 cond-pass embed (what the reference hands back as `embed`, mmp.py:574)
__global__ void bf16_to_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = bf16_to_f32(x[i]);
}

// can_remask_prev_masked (mmp.py:582-588, 603-609): every position was sampled; ids only change where the token was masked, the scores are the
// sampled token's 1 - p at EVERY position
__global__ void merge_pred_kernel(int64_t* __restrict__ ids, const int64_t* __restrict__ pred, float* __restrict__ scores, const float* __restrict__ conf,
                                  int64_t mask_id, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        if (ids[i] == mask_id) ids[i] = pred[i];
        scores[i] = conf[i];
    }
}

// token-critic scores with the annealed noise (mmp.py:590-601): scores = critic + (u - 0.5) * critic_noise_scale * (steps_until_x0 / timesteps),
// each product rounded to fp32 like the reference's tensor-times-scalar chain
__global__ void critic_scores_kernel(const float* __restrict__ critic, const float* __restrict__ u, float noise_scale, float ratio,
                                     float* __restrict__ scores, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        scores[i] = __fadd_rn(critic[i], __fmul_rn(__fmul_rn(__fadd_rn(u[i], -0.5f), noise_scale), ratio));
}

// 'f16x2' tier: bf16 view of the leading (h) fp16 term of operand rows -- what the fused sampler's bound estimate (k_fused_threshold, a bf16 MFMA product) reads;
// an estimate that the finishing kernel verifies, so bf16 is all it needs
__global__ void f16_rows_to_bf16_kernel(const bf16_t* __restrict__ src, long ld, long rows, int D, bf16_t* __restrict__ out) {
    const int chunks = D >> 2;
    const long total = rows * chunks;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / chunks;
        const int c = (int)(i - r * chunks) * 4;
        const uint2 a = *reinterpret_cast<const uint2*>(src + r * ld + c);
        *reinterpret_cast<uint2*>(out + r * D + c) = make_uint2(pack_bf16x2(f16_bits_to_f32((uint16_t)a.x), f16_bits_to_f32((uint16_t)(a.x >> 16))),
                                                                 pack_bf16x2(f16_bits_to_f32((uint16_t)a.y), f16_bits_to_f32((uint16_t)(a.y >> 16))));
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

// add_row (with resid, fp32 out): a row vector added to rows >= add_row_from behind the residual, (acc + resid) + add_row -- the precision tier's null half takes its
// constant cross-attention row here (round 6), as the bf16 engine's fold producer does
int gemm_dense(const mm_transformer* t, hipStream_t s, const bf16_t* X, int ldx, const bf16_t* W, int ldw, int M, int N, int K, void* out, long ldc,
               int out_kind, const float* resid, const float* add_row = nullptr, int add_row_from = 0) {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    if (add_row && resid && out_kind == OUT_F32 && (N % 4) == 0) { a.add_row = add_row; a.add_row_from = add_row_from; }
    else if (add_row) return mm_set_error(MM_ERR_UNSUPPORTED, "gemm_dense: add_row rides on the fp32-residual epilogue");
    a.mode = MODE_DENSE;
    a.f16 = t->F16; a.alpha = t->alpha;      // 'f16x2' tier: fp16 term operands (every GEMM of such a model)
    a.terms = t->F16 ? t->P : 0;             // ... as equal-length term segments: gemm_terms.hip stages every term plane once where its shape class applies
    a.W = W; a.N = N; a.ldw = ldw; a.K = K; a.M = M; a.X = X; a.ldx = ldx;
    a.out = out; a.ldc = ldc; a.out_kind = out_kind; a.resid_f32 = resid; a.ldr = ldc;
    return mm_gemm_launch(a, s);
}

// operand code of the block-internal term-segment buffers (b.xn, b.att, b.a) of an 'f16x2' model with three products: their only readers are GEMMs that state the
// term count (gemm_dense: GemmArgs::terms), i.e. term-sharing k-loops that stage the h plane once -- so the producers skip the repeated h segment (a third of
// their writes; common.h MM_SPLIT_NODUP_BIT).  Off with the term-sharing kernels (mm_debug_set2 bit 2) or a k-loop ablation bit.
int pc_scratch(const mm_transformer* t) {
    return (t->F16 && t->P == 3 && !(g_mm_debug2 & 2) && !(g_mm_debug & 7)) ? (t->PC | MM_SPLIT_NODUP_BIT) : t->PC;
}

struct Bufs {       // activation scratch for `rows` token rows
    float* x;       // [rows][D] residual stream fp32
    bf16_t* xn;     // [rows][D]
    bf16_t* qkv;    // [rows][3I]
    bf16_t* att;    // [rows][I]
    bf16_t* h;      // [rows][2Fp]
    bf16_t* a;      // [rows][Fp]
    float* lnp;     // [rows][Fp / 32][2]: LayerNorm(inner) partial sums of the folded feed-forward
    bf16_t* xb;     // [rows][D]: the residual stream as bf16, written by the residual-adding epilogues (LayerNorm(dim) fold, bf16 engine)
    float* stp;     // [rows][2 ceil(D / 128)][2]: ... with the rows' (sum, sum of squares) per 64 columns
    unsigned char* q8;   // fp8 engine: the e4m3 rows of the activation that feeds the next Linear, [rows][max(D, I, Fp)]
    float* q8s;          // ... and their per-row scales [rows]
};

// precision tier: GEMM operands (xn, att, a) are P segments wide, GEMM outputs (qkv, h) are fp32
void carve_bufs(Carver& c, const mm_transformer* t, size_t rows, Bufs& b) {
    const int D = t->d.dim, I = t->I, Fp = t->Fp;
    const size_t seg = t->P ? (size_t)t->P : 1, f32 = t->P ? 2 : 1;
    b.x = c.take<float>(rows * D);
    b.xn = c.take<bf16_t>(rows * D * seg);
    b.qkv = c.take<bf16_t>(rows * 3 * I * f32);
    b.att = c.take<bf16_t>(rows * I * seg);
    b.h = c.take<bf16_t>(rows * 2 * Fp * f32);
    b.a = c.take<bf16_t>(rows * Fp * seg);
    b.lnp = c.take<float>(rows * (Fp / 32) * 2);
    b.xb = c.take<bf16_t>((t->P || t->F8) ? 0 : rows * D);
    b.stp = c.take<float>((t->P || t->F8) ? 0 : rows * 2 * ((D + 127) / 128) * 2);      // (>= rows * ln_fold_np(D) * 2)
    b.q8 = nullptr; b.q8s = nullptr;
    if (t->F8) {
        const size_t w = (size_t)(D > I ? (D > Fp ? D : Fp) : (I > Fp ? I : Fp));
        b.q8 = c.take<unsigned char>(rows * w);
        b.q8s = c.take<float>(rows);
    }
}

// fp8 engine: out = act . W^T on the fp8 MFMA; act given as e4m3 rows + scales (b.q8 / b.q8s, written by the producer in front of this call)
int f8_linear(hipStream_t s, const Bufs& b, int K, const void* w, const float* w_scale, int M, int N, void* out, long ldc, int epi, const float* resid) {
    GemmF8Args a;
    memset(&a, 0, sizeof(a));
    a.X = b.q8; a.ldx = K; a.sx = b.q8s;
    a.W = (const unsigned char*)w; a.ldw = K; a.sw = w_scale;
    a.M = M; a.N = N; a.K = K; a.out = out; a.ldc = ldc; a.epi = epi; a.resid = resid; a.ldr = ldc;
    return k_gemm_fp8(s, a);
}
// ... with a bf16 activation (attention output): quantise its rows first
int f8_linear_bf16(hipStream_t s, Bufs& b, const bf16_t* act, long lda, int K, const void* w, const float* w_scale, int M, int N, void* out, long ldc, int epi,
                   const float* resid) {
    RC(k_quantize_act_e4m3(s, act, 0, lda, M, K, K, b.q8, b.q8s));
    return f8_linear(s, b, K, w, w_scale, M, N, out, ldc, epi, resid);
}

// LayerNorm(dim) folded into the GEMMs around it (round 4, bf16 engine; GemmArgs::xb_out / in_c1): the residual-adding epilogues (attention output
// projections, FF w2) also write the new residual row as bf16 + its (sum, sum of squares) per 128-column tile, and the projection behind the next
// LayerNorm (q|k|v, cross-attention q, FF w1) multiplies those RAW rows by gain-folded weights and applies rstd * (acc - mean * c1) + c2 in its
// epilogue: the LayerNorm's own pass over the stream (468 launches, 4.5 ms per generate at the base config) disappears.  Layer 0 keeps the unfolded
// self-attention / feed-forward LayerNorms (its inputs come from the embedding kernel and, in the decode loop, from a row copy); debug bit 1 << 29 turns
// the fold off (A/B and the closeness test).
int ln_fold_np(int D) { return D <= 512 ? (D + 63) / 64 : (D + 127) / 128; }      // statistics partials per row: per 64 columns up to dim 512, per 128 beyond (== mm_gemm_launch's st_np)
bool ln_fold_on(const mm_transformer* t) {
    return !t->P && !t->F8 && !t->d.ln_fold_off && !(g_mm_debug & (1 << 29)) && (t->d.dim % 4) == 0;
}
// safety probe of the fold (mm_transformer_desc.ln_probe): called in front of every fold consumer's launch
int fold_probe(const mm_transformer* t, hipStream_t s, const Bufs& b, int rows) {
    if (!t->d.ln_probe) return MM_OK;
    return k_ln_fold_ratio(s, b.stp, rows, ln_fold_np(t->d.dim), t->d.dim, t->d.ln_probe);
}
void fold_consume(GemmArgs& a, const mm_transformer* t, const Bufs& b, const void* w_ln, const float* c1, const float* c2) {
    a.X = b.xb; a.ldx = t->d.dim; a.W = (const bf16_t*)w_ln;
    a.in_part = b.stp; a.in_np = ln_fold_np(t->d.dim); a.in_F = t->d.dim; a.in_c1 = c1; a.in_c2 = c2;
}
void fold_produce(GemmArgs& a, const mm_transformer* t, const Bufs& b, const float* add_row, int add_row_from) {
    a.xb_out = b.xb; a.ldxb = t->d.dim; a.st_part = b.stp; a.add_row = add_row; a.add_row_from = add_row_from;
}
// out (fp32, in place over resid) = resid + X . W^T, optionally producing the fold data of the new rows
int gemm_resid(const mm_transformer* t, hipStream_t s, const bf16_t* X, int ldx, const bf16_t* W, int ldw, int M, int N, int K, float* out, const Bufs& b,
               bool fold_out, const float* add_row = nullptr, int add_row_from = 0) {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = MODE_DENSE;
    a.W = W; a.N = N; a.ldw = ldw; a.K = K; a.M = M; a.X = X; a.ldx = ldx;
    a.out = out; a.ldc = N; a.out_kind = OUT_F32; a.resid_f32 = out; a.ldr = N;
    if (fold_out) fold_produce(a, t, b, add_row, add_row_from);
    return mm_gemm_launch(a, s);
}

// dst += FF(src)   (mmp.py:79-89 with the residual of :193 / the self-cond add of :328)
// addvec != NULL (src == dst): rows [add_from, rows) first get the row vector added in place, inside the first LayerNorm's pass
// fold_in: the rows' bf16 copy + statistics are in b.xb / b.stp (src == dst, no addvec: the producer added it); fold_out: the w2 epilogue produces them
int ff_block(const mm_transformer* t, hipStream_t s, const mm_ff_weights& w, const float* src, float* dst, int rows, Bufs& b,
             const float* addvec = nullptr, int add_from = 0, bool fold_in = false, bool fold_out = false) {
    const int D = t->d.dim, F = t->d.ff_inner, Fp = t->Fp;
    if (t->P) {      // precision tier: LN -> P segments -> w1 (fp32 out, plain [x | gate] halves) -> GEGLU + LN(inner) -> P segments -> w2 + residual
        const int P = t->P;
        RC(k_layernorm_split(s, addvec ? dst : src, D, rows, D, w.ln1_gamma, w.ln1_beta, nullptr, pc_scratch(t), b.xn, nullptr, addvec, add_from, addvec ? dst : nullptr));
        if (t->F16 && w.w1_terms_geglu && w.w2_folded && w.ln2_c1 && w.ln2_c2 && !(g_mm_debug & (1 << 24))) {
            // round 5: w1 on the term-sharing kernel with GEGLU + the term split of its output + the LayerNorm(inner) partial sums in the epilogue
            // (gemm_terms.hip), LayerNorm(inner) folded into w2 as in the bf16 engine (mmp.py:85-88): the fp32 [rows][2 Fp] intermediate and the
            // GEGLU / LayerNorm / split pass over it disappear.  Shapes outside the kernel's class (small batches) take the three-kernel form below.
            GemmArgs a1;
            memset(&a1, 0, sizeof(a1));
            a1.mode = MODE_DENSE; a1.epi = EPI_GEGLU; a1.f16 = 1; a1.alpha = t->alpha; a1.terms = P;
            a1.W = (const bf16_t*)w.w1_terms_geglu; a1.N = 2 * Fp; a1.ldw = P * D; a1.K = P * D; a1.M = rows; a1.X = b.xn; a1.ldx = P * D;
            a1.out = b.a; a1.ldc = (long)P * Fp; a1.out_kind = OUT_BF16; a1.ln_part = b.lnp; a1.ln_np = 2 * Fp / 64;
            a1.terms_nodup = (pc_scratch(t) & MM_SPLIT_NODUP_BIT) ? 1 : 0;
            if (mm_gemm_terms_eligible(a1)) {
                RC(mm_gemm_launch(a1, s));
                GemmArgs a2;
                memset(&a2, 0, sizeof(a2));
                a2.mode = MODE_DENSE; a2.f16 = 1; a2.alpha = t->alpha; a2.terms = P;
                a2.W = (const bf16_t*)w.w2_folded; a2.N = D; a2.ldw = P * Fp; a2.K = P * Fp; a2.M = rows; a2.X = b.a; a2.ldx = P * Fp;
                a2.out = dst; a2.ldc = D; a2.out_kind = OUT_F32; a2.resid_f32 = dst; a2.ldr = D;
                a2.ln_part = b.lnp; a2.ln_np = Fp / 32; a2.ln_F = F; a2.ln_c1 = w.ln2_c1; a2.ln_c2 = w.ln2_c2;
                RC(mm_gemm_launch(a2, s));
                TR(dst, (size_t)rows * D * 4);
                return MM_OK;
            }
        }
        float* hf = reinterpret_cast<float*>(b.h);
        RC(gemm_dense(t, s, b.xn, P * D, (const bf16_t*)w.w1, P * D, rows, 2 * Fp, P * D, hf, 2 * Fp, OUT_F32, nullptr));
        RC(k_geglu_ln_split(s, hf, 2 * Fp, rows, F, Fp, w.ln2_gamma, w.ln2_beta, pc_scratch(t), b.a));
        RC(gemm_dense(t, s, b.a, P * Fp, (const bf16_t*)w.w2, P * Fp, rows, D, P * Fp, dst, D, OUT_F32, dst));
        TR(dst, (size_t)rows * D * 4);
        return MM_OK;
    }
    if (t->F8) {      // fp8 engine: LN -> e4m3 | w1 + GEGLU (bf16) | LN(inner) -> e4m3 | w2 + residual
        RC(k_layernorm_q8(s, const_cast<float*>(addvec ? dst : src), D, rows, D, w.ln1_gamma, w.ln1_beta, addvec, add_from, b.q8, D, b.q8s));
        RC(f8_linear(s, b, D, w.w1, w.w1_scale, rows, 2 * Fp, b.h, Fp, 1, nullptr));
        TR(b.h, (size_t)rows * Fp * 2);
        RC(k_ln_inner_q8(s, b.h, Fp, rows, F, Fp, w.ln2_gamma, w.ln2_beta, b.q8, Fp, b.q8s));
        RC(f8_linear(s, b, Fp, w.w2, w.w2_scale, rows, D, dst, D, 2, dst));
        TR(dst, (size_t)rows * D * 4);
        return MM_OK;
    }
    fold_in = fold_in && w.w1_ln && w.ln1_c1 && !addvec && src == dst;
    fold_out = fold_out && src == dst;
    if (fold_in) {}      // LayerNorm(dim) rides in w1's epilogue
    else if (addvec) RC(k_layernorm_addvec(s, dst, D, rows, D, w.ln1_gamma, w.ln1_beta, addvec, add_from, b.xn, D));
    else RC(k_layernorm(s, src, D, rows, D, w.ln1_gamma, w.ln1_beta, nullptr, b.xn, D));
    GemmArgs a1;      // Linear(D, 2F) with the GEGLU fused into the epilogue: w1 is packed GEGLU-interleaved, the GEMM emits gate*gelu(x)
    memset(&a1, 0, sizeof(a1));
    a1.mode = MODE_DENSE; a1.epi = EPI_GEGLU;
    a1.W = (const bf16_t*)w.w1; a1.N = 2 * Fp; a1.ldw = D; a1.K = D; a1.M = rows; a1.X = b.xn; a1.ldx = D;
    a1.out = b.h; a1.ldc = Fp; a1.out_kind = OUT_BF16;
    if (fold_in) { fold_consume(a1, t, b, w.w1_ln, w.ln1_c1, w.ln1_c2); RC(fold_probe(t, s, b, rows)); }
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
    TR(fold_in ? b.xb : b.xn, (size_t)rows * D * 2);      // (fold: the raw bf16 rows the producer's epilogue left; their statistics show in w1's output)
    if (fold_out) fold_produce(a2, t, b, nullptr, 0);
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
// fold_in: the rows' bf16 copy + statistics are in b.xb / b.stp (LayerNorm(dim) fold): q|k|v runs on them with the gain-folded weights, no LayerNorm pass
int self_attn_core(const mm_transformer* t, hipStream_t s, const mm_attn_weights& w, int seqs, int n, Bufs& b, bool fold_in = false) {
    const int D = t->d.dim, I = t->I, H = t->d.heads, dh = t->d.dim_head;
    const int rows = seqs * n;
    if (t->P) {      // precision tier: q|k|v stay fp32, the attention runs on the fp32 MFMA and writes its output as P segments
        const int P = t->P;
        RC(k_layernorm_split(s, b.x, D, rows, D, w.ln_gamma, w.ln_beta, nullptr, pc_scratch(t), b.xn, nullptr, nullptr, 0, nullptr));
        const bf16_t* wq = (const bf16_t*)w.w_q;
        const bf16_t* wkv = (const bf16_t*)w.w_kv;
        float* qkv = reinterpret_cast<float*>(b.qkv);
        if (wkv == wq + (size_t)I * P * D) {
            RC(gemm_dense(t, s, b.xn, P * D, wq, P * D, rows, 3 * I, P * D, qkv, 3 * I, OUT_F32, nullptr));
        } else {
            RC(gemm_dense(t, s, b.xn, P * D, wq, P * D, rows, I, P * D, qkv, 3 * I, OUT_F32, nullptr));
            RC(gemm_dense(t, s, b.xn, P * D, wkv, P * D, rows, 2 * I, P * D, qkv + I, 3 * I, OUT_F32, nullptr));
        }
        AttnF32Args a;
        memset(&a, 0, sizeof(a));
        a.q = qkv; a.q_sb = (long)n * 3 * I; a.q_sh = dh; a.q_sn = 3 * I;
        a.k = qkv + I; a.k_sb = a.q_sb; a.k_sh = dh; a.k_sn = 3 * I;
        a.v = qkv + 2 * I; a.v_sb = a.q_sb; a.v_sh = dh; a.v_sn = 3 * I;
        a.out_split = b.att; a.os_sb = (long)n * P * I; a.os_sn = (long)P * I; a.os_seg = I; a.P = pc_scratch(t);
        a.B = seqs; a.H = H; a.nq = n; a.nk = n;
        a.normalize = 1; a.q_scale = w.q_scale; a.k_scale = w.k_scale; a.null_k = w.null_k; a.null_v = w.null_v;
        a.scale = 8.f; a.dh = dh;
        return k_attention_f32(s, a);
    }
    const bf16_t* wq = (const bf16_t*)w.w_q;
    const bf16_t* wkv = (const bf16_t*)w.w_kv;
    if (t->F8) {      // fp8 engine: LN -> e4m3, q|k|v on the fp8 MFMA (bf16 out); the attention itself is the bf16 engine's
        RC(k_layernorm_q8(s, b.x, D, rows, D, w.ln_gamma, w.ln_beta, nullptr, 0, b.q8, D, b.q8s));
        const unsigned char* q8w = (const unsigned char*)w.w_q;
        if ((const unsigned char*)w.w_kv == q8w + (size_t)I * D && w.w_kv_scale == w.w_q_scale + I) {
            RC(f8_linear(s, b, D, w.w_q, w.w_q_scale, rows, 3 * I, b.qkv, 3 * I, 0, nullptr));
        } else {
            RC(f8_linear(s, b, D, w.w_q, w.w_q_scale, rows, I, b.qkv, 3 * I, 0, nullptr));
            RC(f8_linear(s, b, D, w.w_kv, w.w_kv_scale, rows, 2 * I, b.qkv + I, 3 * I, 0, nullptr));
        }
    } else {
    if (fold_in && w.w_q_ln && w.ln_c1 && wkv == wq + (size_t)I * D) {
        GemmArgs a;
        memset(&a, 0, sizeof(a));
        a.mode = MODE_DENSE; a.N = 3 * I; a.ldw = D; a.K = D; a.M = rows;
        a.out = b.qkv; a.ldc = 3 * I; a.out_kind = OUT_BF16;
        fold_consume(a, t, b, w.w_q_ln, w.ln_c1, w.ln_c2);
        RC(fold_probe(t, s, b, rows));
        TR(b.xb, (size_t)rows * D * 2);
        RC(mm_gemm_launch(a, s));
    } else {
    RC(k_layernorm(s, b.x, D, rows, D, w.ln_gamma, w.ln_beta, nullptr, b.xn, D));
    TR(b.xn, (size_t)rows * D * 2);
    if (wkv == wq + (size_t)I * D) {
        RC(gemm_dense(t, s, b.xn, D, wq, D, rows, 3 * I, D, b.qkv, 3 * I, OUT_BF16, nullptr));
    } else {
        RC(gemm_dense(t, s, b.xn, D, wq, D, rows, I, D, b.qkv, 3 * I, OUT_BF16, nullptr));
        RC(gemm_dense(t, s, b.xn, D, wkv, D, rows, 2 * I, D, b.qkv + I, 3 * I, OUT_BF16, nullptr));
    }
    }
    }
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = b.qkv; a.q_sb = (long)n * 3 * I; a.q_sh = dh; a.q_sn = 3 * I;
    a.k = b.qkv + I; a.k_sb = a.q_sb; a.k_sh = dh; a.k_sn = 3 * I;
    a.v = b.qkv + 2 * I; a.v_sb = a.q_sb; a.v_sh = dh; a.v_sn = 3 * I;
    a.out = b.att; a.o_sb = (long)n * I; a.o_sh = dh; a.o_sn = I;
    a.B = seqs; a.H = H; a.nq = n; a.nk = n;
    a.normalize = 1; a.q_scale = w.q_scale; a.k_scale = w.k_scale; a.null_k = w.null_k; a.null_v = w.null_v;
    a.scale = 8.f; a.dh = dh;
    TR(b.qkv, (size_t)rows * 3 * I * 2);
    RC(k_attention(s, a));
    TR(b.att, (size_t)rows * I * 2);
    return MM_OK;
}

// x += SelfAttention(x) over `seqs` sequences of n tokens  (mmp.py:126-162 with context = None, :189)
// fold_out: the output projection's epilogue also writes the new rows as bf16 + statistics (b.xb / b.stp); add_row (fold_out only): a row vector added to
// rows >= add_row_from behind the residual (the null half's constant cross-attention, which otherwise rides in the feed-forward's LayerNorm pass)
int self_attn_block(const mm_transformer* t, hipStream_t s, const mm_attn_weights& w, int seqs, int n, Bufs& b, bool fold_in = false, bool fold_out = false,
                    const float* add_row = nullptr, int add_row_from = 0) {
    RC(self_attn_core(t, s, w, seqs, n, b, fold_in));
    const int KI = (t->P ? t->P : 1) * t->I;      // precision tier: P segments per operand row
    if (t->F8) RC(f8_linear_bf16(s, b, b.att, t->I, t->I, w.w_out, w.w_out_scale, seqs * n, t->d.dim, b.x, t->d.dim, 2, b.x));
    else if (fold_out) RC(gemm_resid(t, s, b.att, KI, (const bf16_t*)w.w_out, KI, seqs * n, t->d.dim, KI, b.x, b, true, add_row, add_row_from));
    else RC(gemm_dense(t, s, b.att, KI, (const bf16_t*)w.w_out, KI, seqs * n, t->d.dim, KI, b.x, t->d.dim, OUT_F32, b.x, add_row, add_row_from));
    TR(b.x, (size_t)seqs * n * t->d.dim * 4);
    return MM_OK;
}

// The cross-attention block as one kernel (cross_fold.hip: q projection on the LayerNorm(dim) fold's consumer side, attention with the output projection folded into
// the step-invariant values, residual, fold producer outputs): bf16 engine with the LayerNorm(dim) fold on, the headline shape class (dim = inner = 512, 8 heads x 64,
// <= 35 context tokens); debug bit 1 << 31 turns it off (A/B and the closeness test; bits 16 .. 23 are attention.hip's repetition count)
bool cross_fold_on(const mm_transformer* t, const mm_attn_weights& w, int m) {
    return ln_fold_on(t) && !(g_mm_debug & (int)0x80000000) && k_cross_fold_eligible(t->d.dim, t->I, t->d.heads, t->d.dim_head, m) && w.null_k && w.null_v &&
           w.q_scale && w.k_scale && w.w_q_ln && w.ln_c1;      // (ln_c2 is NULL for the reference's LayerNorm: its beta is a zeros buffer)
}
// the packed operands of one layer: K^ and (V W_o^T)^T of every kv sequence (from ckv = ctx @ to_kv^T) and the gain-folded q weight as MFMA fragments
struct CrossFoldPack { bf16_t* khat; bf16_t* vwt; bf16_t* wqf; bool x2 = false; };      // x2: the 'f16x2' tier's pack (cross_vw_x2.hip: khat = fp32 K^, vwt = two fp16 term planes, no wqf)
// the tier's cross-attention behind its q projection as one kernel (cross_vw_x2.hip); mm_debug_set2 bit 64 keeps the attention + output projection launches (A/B, tests)
bool cross_vw_on(const mm_transformer* t, const mm_attn_weights& w, int m) {
    return t->F16 && (t->P == 2 || t->P == 3) && !(g_mm_debug2 & 64) && !(g_mm_debug & 7) && k_cross_vw_x2_eligible(t->d.dim, t->I, t->d.heads, t->d.dim_head, m) &&
           w.null_k && w.null_v && w.q_scale && w.k_scale;
}
int cross_fold_pack(const mm_transformer* t, hipStream_t s, const mm_attn_weights& w, const bf16_t* ckv, int kv_seqs, int m, const CrossFoldPack& pk) {
    return k_cross_fold_pack(s, ckv, kv_seqs, m, t->I, w.null_k, w.null_v, w.k_scale, (const bf16_t*)w.w_out, t->I, (const bf16_t*)w.w_q_ln, t->d.dim, pk.khat, pk.vwt, pk.wqf);
}

// bf16_t elements of one layer's K^ pack, whichever of the two one-kernel forms a model takes (the tier's is fp32: 12288 floats per sequence)
size_t xpack_khat_elems(const mm_transformer* t, int seqs) {
    const size_t a = k_cross_fold_khat_elems(seqs), b = t->F16 ? k_cross_vw_x2_khat_floats(seqs) * 2 : 0;
    return a > b ? a : b;
}

size_t xpack_wqf_elems(const mm_transformer* t) {
    const size_t a = k_cross_fold_wqf_elems(), b = t->F16 ? k_cross_vw_x2_wqf_halves() : 0;
    return a > b ? a : b;
}

// x += CrossAttention(x, ctx) for `seqs` sequences; ckv = ctx @ to_kv^T given as [kv_seqs*m][2I]  (mmp.py:191)
// pk (cross_fold_pack of this layer, or NULL) with fold_in: the whole block is ONE kernel (cross_fold.hip)
int cross_attn_block(const mm_transformer* t, hipStream_t s, const mm_attn_weights& w, int seqs, int n, const bf16_t* ckv,
                     int m, int kv_batch_mod, const uint8_t* key_mask, Bufs& b, bool fold_in = false, bool fold_out = false,
                     const CrossFoldPack* pk = nullptr) {
    const int D = t->d.dim, I = t->I, H = t->d.heads, dh = t->d.dim_head;
    const int rows = seqs * n;
    if (pk && fold_in && !t->P && !t->F8 && w.w_q_ln && w.ln_c1) {
        CrossFoldArgs f;
        memset(&f, 0, sizeof(f));
        f.xb_in = b.xb; f.ldxb_in = D; f.stp_in = b.stp; f.in_np = ln_fold_np(D);
        f.wqf = pk->wqf; f.c1 = w.ln_c1; f.c2 = w.ln_c2; f.khat = pk->khat; f.vwt = pk->vwt;
        f.key_mask = key_mask; f.km_sb = m; f.q_scale = w.q_scale;
        f.x = b.x; f.ldx = D;
        if (fold_out) { f.xb = b.xb; f.ldxb = D; f.stp = b.stp; f.st_np = ln_fold_np(D); }
        f.seqs = seqs; f.nq = n; f.m = m; f.kv_batch_mod = kv_batch_mod; f.scale = 8.f;
        RC(fold_probe(t, s, b, rows));
        TR(b.xb, (size_t)rows * D * 2);
        RC(k_cross_fold(s, f));
        TR(b.x, (size_t)rows * D * 4);
        return MM_OK;
    }
    if (t->P) {      // precision tier: ckv is fp32 [kv_seqs*m][2I]
        const int P = t->P;
        const bool qp = pk && pk->x2 && pk->wqf && !(g_mm_debug2 & 128);      // the LayerNorm and the q projection inside the kernel as well (bit 128, A/B: as launches in front of it)
        float* q = reinterpret_cast<float*>(b.qkv);
        if (!qp) {
            RC(k_layernorm_split(s, b.x, D, rows, D, w.ln_gamma, w.ln_beta, nullptr, pc_scratch(t), b.xn, nullptr, nullptr, 0, nullptr));
            RC(gemm_dense(t, s, b.xn, P * D, (const bf16_t*)w.w_q, P * D, rows, I, P * D, q, I, OUT_F32, nullptr));
        }
        if (pk && pk->x2) {      // round 6: scores, softmax, P . (V W_o^T) on fp16 terms and the residual add in ONE kernel (the pack: once per generate and layer)
            CrossVwArgs v;
            memset(&v, 0, sizeof(v));
            if (qp) { v.wqf = pk->wqf; v.wq_terms = P; v.alpha = t->alpha; v.ln_gamma = w.ln_gamma; v.ln_beta = w.ln_beta; }
            v.q = q; v.ldq = I; v.khat = reinterpret_cast<const float*>(pk->khat); v.vwt = pk->vwt;
            v.key_mask = key_mask; v.km_sb = m; v.q_scale = w.q_scale;
            v.x = b.x; v.ldx = D; v.seqs = seqs; v.nq = n; v.m = m; v.kv_batch_mod = kv_batch_mod; v.scale = 8.f;
            RC(k_cross_vw_x2(s, v));
            TR(b.x, (size_t)rows * D * 4);
            return MM_OK;
        }
        const float* kv = reinterpret_cast<const float*>(ckv);
        AttnF32Args a;
        memset(&a, 0, sizeof(a));
        a.q = q; a.q_sb = (long)n * I; a.q_sh = dh; a.q_sn = I;
        a.k = kv; a.k_sb = (long)m * 2 * I; a.k_sh = dh; a.k_sn = 2 * I;
        a.v = kv + I; a.v_sb = a.k_sb; a.v_sh = dh; a.v_sn = 2 * I;
        a.out_split = b.att; a.os_sb = (long)n * P * I; a.os_sn = (long)P * I; a.os_seg = I; a.P = pc_scratch(t);
        a.B = seqs; a.H = H; a.nq = n; a.nk = m;
        a.key_mask = key_mask; a.km_sb = m;
        a.normalize = 1; a.q_scale = w.q_scale; a.k_scale = w.k_scale; a.null_k = w.null_k; a.null_v = w.null_v;
        a.scale = 8.f; a.dh = dh; a.kv_batch_mod = kv_batch_mod;
        RC(k_attention_f32(s, a));
        RC(gemm_dense(t, s, b.att, P * I, (const bf16_t*)w.w_out, P * I, rows, D, P * I, b.x, D, OUT_F32, b.x));
        TR(b.x, (size_t)rows * D * 4);
        return MM_OK;
    }
    if (t->F8) {      // fp8 engine: LN -> e4m3, q projection on the fp8 MFMA (the context's k | v projection stays bf16: once per generate)
        RC(k_layernorm_q8(s, b.x, D, rows, D, w.ln_gamma, w.ln_beta, nullptr, 0, b.q8, D, b.q8s));
        RC(f8_linear(s, b, D, w.w_q, w.w_q_scale, rows, I, b.qkv, I, 0, nullptr));
    } else if (fold_in && w.w_q_ln && w.ln_c1) {      // LayerNorm(dim) fold: q from the raw bf16 rows + statistics the self-attention's output projection left
        GemmArgs a;
        memset(&a, 0, sizeof(a));
        a.mode = MODE_DENSE; a.N = I; a.ldw = D; a.K = D; a.M = rows;
        a.out = b.qkv; a.ldc = I; a.out_kind = OUT_BF16;
        fold_consume(a, t, b, w.w_q_ln, w.ln_c1, w.ln_c2);
        RC(fold_probe(t, s, b, rows));
        TR(b.xb, (size_t)rows * D * 2);
        RC(mm_gemm_launch(a, s));
    } else {
        RC(k_layernorm(s, b.x, D, rows, D, w.ln_gamma, w.ln_beta, nullptr, b.xn, D));
        TR(b.xn, (size_t)rows * D * 2);
        RC(gemm_dense(t, s, b.xn, D, (const bf16_t*)w.w_q, D, rows, I, D, b.qkv, I, OUT_BF16, nullptr));
    }
    TR(b.qkv, (size_t)rows * I * 2);
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = b.qkv; a.q_sb = (long)n * I; a.q_sh = dh; a.q_sn = I;
    a.k = ckv; a.k_sb = (long)m * 2 * I; a.k_sh = dh; a.k_sn = 2 * I;
    a.v = ckv + I; a.v_sb = a.k_sb; a.v_sh = dh; a.v_sn = 2 * I;
    a.out = b.att; a.o_sb = (long)n * I; a.o_sh = dh; a.o_sn = I;
    a.B = seqs; a.H = H; a.nq = n; a.nk = m;
    a.key_mask = key_mask; a.km_sb = m;
    a.normalize = 1; a.q_scale = w.q_scale; a.k_scale = w.k_scale; a.null_k = w.null_k; a.null_v = w.null_v;
    a.scale = 8.f; a.dh = dh; a.kv_batch_mod = kv_batch_mod;
    RC(k_attention(s, a));
    TR(b.att, (size_t)rows * I * 2);
    if (t->F8) RC(f8_linear_bf16(s, b, b.att, I, I, w.w_out, w.w_out_scale, rows, D, b.x, D, 2, b.x));
    else if (fold_out) RC(gemm_resid(t, s, b.att, I, (const bf16_t*)w.w_out, I, rows, D, I, b.x, b, true));
    else RC(gemm_dense(t, s, b.att, I, (const bf16_t*)w.w_out, I, rows, D, I, b.x, D, OUT_F32, b.x));
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
    if (d.dim_head != 32 && d.dim_head != 64 && d.dim_head != 128) return mm_set_error(MM_ERR_UNSUPPORTED, "transformer: dim_head must be 32, 64 or 128");
    if ((d.heads * d.dim_head) % 64) return mm_set_error(MM_ERR_SHAPE, "transformer: heads * dim_head must be a multiple of 64");
    if (d.dim <= 0 || d.dim % 64 || d.dim > 2048) return mm_set_error(MM_ERR_SHAPE, "transformer: dim must be a multiple of 64, <= 2048");
    if (d.depth <= 0 || d.heads <= 0 || !d.layers) return mm_set_error(MM_ERR_SHAPE, "transformer: depth/heads/layers");
    if (d.ff_inner_padded % 64 || d.ff_inner_padded < d.ff_inner) return mm_set_error(MM_ERR_SHAPE, "transformer: ff_inner_padded must be a multiple of 64 >= ff_inner");
    if (d.text_proj && (d.text_dim % 64)) return mm_set_error(MM_ERR_SHAPE, "transformer: text_dim must be a multiple of 64 when projected");
    if (!d.text_proj && d.text_dim != d.dim) return mm_set_error(MM_ERR_SHAPE, "transformer: text_proj is NULL but text_dim != dim");
    if (!d.token_emb || !d.pos_emb || !d.to_logits || !d.final_gamma) return mm_set_error(MM_ERR_SHAPE, "transformer: missing weight pointer");
    if (d.split_products != 0 && d.split_products != 3 && d.split_products != 5 && d.split_products != 6 && d.split_products != (MM_SPLIT_F16 | 2) &&
        d.split_products != (MM_SPLIT_F16 | 3))
        return mm_set_error(MM_ERR_SHAPE, "transformer: split_products must be 0 (bf16 engine), 3, 5, 6 (bf16 terms) or MM_SPLIT_F16 | 2, 3 (fp16 terms)");
    mm_transformer* t = new (std::nothrow) mm_transformer();
    if (!t) return mm_set_error(MM_ERR_HIP, "out of host memory");
    t->d = d;
    t->layers.assign(d.layers, d.layers + d.depth);
    t->d.layers = t->layers.data();
    t->I = d.heads * d.dim_head;
    t->Fp = d.ff_inner_padded;
    t->P = split_count(d.split_products);
    t->PC = d.split_products;
    t->F16 = split_is_f16(d.split_products) ? 1 : 0;
    t->alpha = (t->F16 && d.split_alpha != 0.f) ? d.split_alpha : 1.f;
    t->F8 = d.fp8 ? 1 : 0;
    if (t->F8) {
        bool ok = t->P == 0 && (d.dim % 128) == 0 && (t->I % 128) == 0 && (t->Fp % 128) == 0;
        for (int l = 0; ok && l < d.depth; ++l) {
            const mm_layer_weights& w = t->layers[l];
            ok = w.self_attn.w_q_scale && w.self_attn.w_kv_scale && w.self_attn.w_out_scale && w.cross_attn.w_q_scale && w.cross_attn.w_out_scale &&
                 w.ff.w1_scale && w.ff.w2_scale;
        }
        if (ok && d.self_cond) ok = d.self_cond_ff.w1_scale && d.self_cond_ff.w2_scale;
        if (!ok) {
            delete t;
            return mm_set_error(MM_ERR_SHAPE, "transformer: the fp8 engine needs dim, heads * dim_head and ff_inner_padded to be multiples of 128, split_products == 0 and a scale vector for every e4m3 weight");
        }
    }
    *out = t;
    return MM_OK;
}

void mm_transformer_destroy(mm_transformer_t* model) { delete model; }

size_t mm_context_workspace_bytes(const mm_transformer_t* t, int B, int L) {
    if (!t) return 0;
    Carver c(nullptr);
    if (t->d.text_proj) {
        c.take<bf16_t>((size_t)B * L * t->d.text_dim * (t->P ? t->P : 1));
        c.take<bf16_t>((size_t)B * L * t->d.dim * (t->P ? 2 : 1));      // precision tier: the projection leaves its GEMM as fp32
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
    if (t->P) {      // precision tier: ctx bf16 [B][m][P*D] (P segments per row), token table fp32
        const int P = t->P, td = t->d.text_dim;
        if (L > 0) {
            if (!t->d.text_proj) {
                RC(k_split_rows(s, text_embeds, td, (long)B * L, td, t->PC, L, (long)m * P * D, ctxp, key_mask, m, drop_text));
            } else {
                Carver c(workspace);
                bf16_t* tb = c.take<bf16_t>((size_t)B * L * td * P);
                float* proj = reinterpret_cast<float*>(c.take<bf16_t>((size_t)B * L * D * 2));
                RC(k_split_rows(s, text_embeds, td, (long)B * L, td, t->PC, L, (long)L * P * td, tb, key_mask, m, drop_text));
                RC(gemm_dense(t, s, tb, P * td, (const bf16_t*)t->d.text_proj, P * td, B * L, D, P * td, proj, D, OUT_F32, nullptr));
                RC(k_split_rows(s, proj, D, (long)B * L, D, t->PC, L, (long)m * P * D, ctxp, nullptr, 0, 0));
            }
        }
        if (nc > 0) RC(k_gather_split(s, (const float*)t->d.token_emb, D, t->PC, cond_ids, B, nc, t->d.vocab_rows, ctxp, key_mask, m, L));
        return MM_OK;
    }
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
            RC(gemm_dense(t, s, tb, t->d.text_dim, (const bf16_t*)t->d.text_proj, t->d.text_dim, B * L, D, t->d.text_dim, proj, D,
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
    c.take<bf16_t>((size_t)B * m * 2 * t->I * (t->P ? 2 : 1));    // cross K/V of one layer (fp32 in the precision tier)
    c.take<bf16_t>((size_t)B * n * t->d.dim * (t->P ? t->P : 1));  // embed when the caller does not want it
    c.take<bf16_t>(xpack_khat_elems(t, B));                        // packed cross-attention operands of one layer (cross_fold.hip / the tier's cross_vw_x2.hip; reserved whatever the shape)
    c.take<bf16_t>(k_cross_fold_vwt_elems(B));
    c.take<bf16_t>(xpack_wqf_elems(t));
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
    const int P = t->P, KD = (P ? P : 1) * D;      // operand row width of the GEMMs that read [.][D] activations
    bf16_t* ckv = c.take<bf16_t>((size_t)B * m * 2 * I * (P ? 2 : 1));
    bf16_t* emb = c.take<bf16_t>((size_t)rows * KD);
    CrossFoldPack pk;
    pk.khat = c.take<bf16_t>(xpack_khat_elems(t, B));
    pk.vwt = c.take<bf16_t>(k_cross_fold_vwt_elems(B));
    pk.wqf = c.take<bf16_t>(xpack_wqf_elems(t));
    if (embed_out) emb = (bf16_t*)embed_out;

    trace::idx = 0;
    if (P) RC(k_embed_f32(s, ids, rows, n, (const float*)t->d.token_emb, t->d.vocab_rows, (const float*)t->d.pos_emb, D, b.x));
    else RC(k_embed(s, ids, rows, n, 0, (const bf16_t*)t->d.token_emb, t->d.vocab_rows, (const bf16_t*)t->d.pos_emb, D, b.x));
    TR(b.x, (size_t)rows * D * 4);
    if (t->d.self_cond && self_cond_embed)       // mmp.py:325-328 (zeros when absent: FF(0) still adds LN-beta terms = 0)
        RC(ff_block(t, s, t->d.self_cond_ff, self_cond_embed, b.x, rows, b));
    const bool fold = ln_fold_on(t);      // LayerNorm(dim) fold: layer 0's self-attention / feed-forward keep their own LayerNorm pass (the same rule as mm_generate)
    for (int l = 0; l < t->d.depth; ++l) {
        const mm_layer_weights& w = t->layers[l];
        RC(self_attn_block(t, s, w.self_attn, B, n, b, fold && l > 0, fold));
        RC(gemm_dense(t, s, (const bf16_t*)ctx, KD, (const bf16_t*)w.cross_attn.w_kv, KD, B * m, 2 * I, KD, ckv, 2 * I, P ? OUT_F32 : OUT_BF16, nullptr));
        TR(ckv, (size_t)B * m * 2 * I * 2);
        const bool xf = cross_fold_on(t, w.cross_attn, m);
        const bool xv = !xf && cross_vw_on(t, w.cross_attn, m);      // the tier's one-kernel form behind its q projection (cross_vw_x2.hip)
        pk.x2 = xv;
        if (xf) RC(cross_fold_pack(t, s, w.cross_attn, ckv, B, m, pk));
        else if (xv) {
            RC(k_cross_vw_x2_pack(s, reinterpret_cast<const float*>(ckv), B, m, I, w.cross_attn.null_k, w.cross_attn.null_v, w.cross_attn.k_scale,
                                  (const bf16_t*)w.cross_attn.w_out, P * I, P, t->alpha, reinterpret_cast<float*>(pk.khat), pk.vwt));
            RC(k_cross_vw_x2_wq_pack(s, (const bf16_t*)w.cross_attn.w_q, P * D, P, pk.wqf));
        }
        RC(cross_attn_block(t, s, w.cross_attn, B, n, ckv, m, 0, key_mask, b, fold, fold && l > 0, (xf || xv) ? &pk : nullptr));
        RC(ff_block(t, s, w.ff, b.x, b.x, rows, b, nullptr, 0, fold && l > 0, fold && l + 1 < t->d.depth));
    }
    if (P) RC(k_layernorm_split(s, b.x, D, rows, D, t->d.final_gamma, t->d.final_beta, nullptr, t->PC, emb, nullptr, nullptr, 0, nullptr));
    else RC(k_layernorm(s, b.x, D, rows, D, t->d.final_gamma, t->d.final_beta, nullptr, emb, D));
    TR(emb, (size_t)rows * D * 2);
    if (logits_out)
        RC(gemm_dense(t, s, emb, KD, (const bf16_t*)t->d.to_logits, KD, rows, t->d.dim_out, KD, logits_out, t->d.dim_out, OUT_F32, nullptr));
    return MM_OK;
}

// ------------------------------------------------------------------------------------------------ the cross-attention block as an operator
// x += CrossAttention(LayerNorm(x), context) of one layer (mmp.py:139-162, 191) on a caller-given residual stream, exactly as mm_transformer_forward runs it for this
// model -- on the bf16 engine's headline shape class the one-kernel form (csrc/cross_fold.hip), otherwise (or with mm_debug_set bit 1 << 31) q projection +
// attention + output projection.  Exists so that the block can be tested against the oracle as an OPERATOR (tests/test_gpu_ops.py), not only through a model.
size_t mm_cross_attention_block_workspace_bytes(const mm_transformer_t* t, int seqs, int n, int m) {
    if (!t || seqs <= 0 || n <= 0 || m <= 0) return 0;
    Carver c(nullptr);
    Bufs b;
    carve_bufs(c, t, (size_t)seqs * n, b);
    const size_t f32 = t->P ? 2 : 1;
    c.take<bf16_t>((size_t)seqs * m * 2 * t->I * f32);
    c.take<bf16_t>(k_cross_fold_khat_elems(seqs)); c.take<bf16_t>(k_cross_fold_vwt_elems(seqs)); c.take<bf16_t>(k_cross_fold_wqf_elems());
    return c.used() + 256;
}
int mm_cross_attention_block(const mm_transformer_t* t, mm_stream_t stream, int layer, float* x, int seqs, int n, const void* ctx, const uint8_t* key_mask, int m,
                             void* workspace, size_t workspace_bytes) {
    if (!t || !x || !ctx || !workspace) return mm_set_error(MM_ERR_SHAPE, "cross_attention_block: null argument");
    if (layer < 0 || layer >= t->d.depth || seqs <= 0 || n <= 0 || m <= 0) return mm_set_error(MM_ERR_SHAPE, "cross_attention_block: bad layer / shape");
    if (t->P || t->F8) return mm_set_error(MM_ERR_UNSUPPORTED, "cross_attention_block: bf16 engine only");
    if (workspace_bytes < mm_cross_attention_block_workspace_bytes(t, seqs, n, m)) return mm_set_error(MM_ERR_WORKSPACE, "cross_attention_block: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int D = t->d.dim, I = t->I;
    const size_t rows = (size_t)seqs * n;
    Carver c(workspace);
    Bufs b;
    carve_bufs(c, t, rows, b);
    b.x = x;      // the caller's residual stream, updated in place
    bf16_t* ckv = c.take<bf16_t>((size_t)seqs * m * 2 * I);
    CrossFoldPack pk;
    pk.khat = c.take<bf16_t>(k_cross_fold_khat_elems(seqs)); pk.vwt = c.take<bf16_t>(k_cross_fold_vwt_elems(seqs)); pk.wqf = c.take<bf16_t>(k_cross_fold_wqf_elems());
    const mm_attn_weights& w = t->layers[layer].cross_attn;
    const bool fold = ln_fold_on(t);
    if (fold) RC(k_fold_image(s, x, D, (int)rows, D, b.xb, D, b.stp, ln_fold_np(D)));      // what the self-attention's output projection leaves in the model
    RC(gemm_dense(t, s, (const bf16_t*)ctx, D, (const bf16_t*)w.w_kv, D, seqs * m, 2 * I, D, ckv, 2 * I, OUT_BF16, nullptr));
    const bool xf = cross_fold_on(t, w, m);
    if (xf) RC(cross_fold_pack(t, s, w, ckv, seqs, m, pk));
    return cross_attn_block(t, s, w, seqs, n, ckv, m, 0, key_mask, b, fold, false, xf ? &pk : nullptr);
}

// ------------------------------------------------------------------------------------------------ generate
namespace {
#ifndef MM_FS_MARGIN
#define MM_FS_MARGIN 0.12f
#endif
constexpr int FB_CAP = 128;      // rows per step the on-device fallback can finish (one row tile of the 128 x 128 GEMM); more raise the status flag
constexpr int FB_STEPS = 1024;   // per-step counters carved (timesteps beyond this run without the list: flag only)
constexpr float FS_MARGIN = MM_FS_MARGIN;    // the bound sits this many sigmas below the Gaussian quantile of the kept fraction: ~14 % instead of 10 % pass
struct GenBufs {
    Bufs b;                 // 2B sequences
    bf16_t* ctx;            // [B][m][D]
    uint8_t* masks;         // [2B][m]: cond masks then null masks
    bf16_t* ckv;            // [depth][B*m][2I]
    bf16_t* khat;           // [depth][...] packed cross-attention operands (cross_fold.hip): K^ fragments ...
    bf16_t* vwt;            // ... (V W_o^T)^T fragments ...
    bf16_t* wqf;            // ... and gain-folded q weight fragments of every layer, or NULL (shape class not covered)
    float* cvec;            // [depth][D]: to_out(null_v) of each cross-attention (base model null pass)
    bf16_t* nullv;          // [I] scratch
    int32_t* rows;          // [B*n]
    bf16_t* embc;           // [B*n][D]
    bf16_t* embn;           // [B*n][D]
    bf16_t* embm;           // [B*n][D]: null + (cond - null) * cond_scale of the two passes' embeddings -- the ONE operand of the guidance-logits GEMM
    bf16_t* embb;           // [B*n][D] ('f16x2' tier only): bf16 view of embm's leading term for the bound estimate
    float* logits;          // [B*n][V]
    float* xc;              // [2*B*n][D]: the last layer's residual stream, compacted to the sampled rows
    bf16_t* attc;           // [2*B*n][I]
    // fused sampling (sampling_fused.hip): what the guidance-logits GEMM emits instead of the logits
    float* fs_thr;          // [B*n]
    float4* fs_stats;       // [B*n][V/256][FS_REC]
    float4* fs_cand;        // [B*n][V/256][FS_SLOT]
    void* fs_ws;            // scratch of the bound estimate (k_fused_threshold)
    float* fs_sub;          // [B*n][logits_wsub_rows]: the rows' logits at the sampled vocabulary columns (distribution-free bound)
    // on-device per-row fallback of the fused sampler: rows whose bound could not be verified are listed by the finishing kernel and finished
    // on the logits path inside the same step (their logits recomputed by a small dense GEMM, sampled by sample_kernel)
    int32_t* fb_rows;       // [FB_CAP] rows listed this step
    int32_t* fb_cnt;        // [timesteps <= FB_STEPS] one counter per step (zeroed at the start of the call)
    bf16_t* fb_x;           // [FB_CAP][KD] their (mixed) embeddings
    float* fb_logits;       // [FB_CAP][V]
    void* ctx_ws; size_t ctx_ws_bytes;
    // decode variants
    float* sce;             // [B*n][D] fp32: the previous step's cond-pass embed (self-conditioning transformers)
    bf16_t* emb_all;        // [B*n][D] the cond-pass embed of every position, as the final LayerNorm emits it
    int64_t* pred;          // [B*n] compact sampler outputs (can_remask_prev_masked)
    float* conf;            // [B*n]
};
// what a critic (mmp.py:590-601) needs per decode: its own context (a TokenCritic has its own embeddings / text projection), one forward workspace,
// the embeds of its passes and the raw scores
struct CriticBufs {
    bf16_t* ctx; uint8_t* masks; void* ctx_ws; size_t ctx_ws_bytes;
    bf16_t* embc; bf16_t* embn; float* sc; void* fwd_ws; size_t fwd_ws_bytes;
};
void carve_critic(Carver& c, const mm_transformer* ct, int B, int n, int L, int nc, CriticBufs& k) {
    const int m = L + nc;
    const size_t seg = ct->P ? (size_t)ct->P : 1;
    k.ctx = c.take<bf16_t>((size_t)B * m * ct->d.dim * seg);
    k.masks = c.take<uint8_t>((size_t)2 * B * m);
    k.ctx_ws_bytes = mm_context_workspace_bytes(ct, B, L);
    k.ctx_ws = c.take<unsigned char>(k.ctx_ws_bytes);
    k.embc = c.take<bf16_t>((size_t)B * n * ct->d.dim * seg);
    k.embn = c.take<bf16_t>((size_t)B * n * ct->d.dim * seg);
    k.sc = c.take<float>((size_t)B * n + 64);
    k.fwd_ws_bytes = mm_transformer_workspace_bytes(ct, B, n, m);
    k.fwd_ws = c.take<unsigned char>(k.fwd_ws_bytes);
}
void carve_gen(Carver& c, const mm_transformer* t, int B, int n, int L, int nc, GenBufs& g) {
    const int D = t->d.dim, I = t->I, m = L + nc;
    const size_t seg = t->P ? (size_t)t->P : 1, f32 = t->P ? 2 : 1;      // precision tier: operands P segments wide, cross K/V fp32
    carve_bufs(c, t, (size_t)2 * B * n, g.b);
    g.ctx = c.take<bf16_t>((size_t)B * m * D * seg);
    g.masks = c.take<uint8_t>((size_t)2 * B * m);
    g.ckv = c.take<bf16_t>((size_t)t->d.depth * B * m * 2 * I * f32);
    const bool xf = !t->P && !t->F8 && k_cross_fold_eligible(D, I, t->d.heads, t->d.dim_head, m);      // (sized by shape only: the debug switch must not change the workspace)
    const bool xv = t->F16 && k_cross_vw_x2_eligible(D, I, t->d.heads, t->d.dim_head, m);              // the tier's pack (cross_vw_x2.hip): fp32 K^, two fp16 planes of V W_o^T
    g.khat = c.take<bf16_t>(xf ? (size_t)t->d.depth * k_cross_fold_khat_elems(B) : xv ? (size_t)t->d.depth * k_cross_vw_x2_khat_floats(B) * 2 : 0);
    g.vwt = c.take<bf16_t>(xf ? (size_t)t->d.depth * k_cross_fold_vwt_elems(B) : xv ? (size_t)t->d.depth * k_cross_vw_x2_vwt_halves(B) : 0);
    g.wqf = c.take<bf16_t>(xf ? (size_t)t->d.depth * k_cross_fold_wqf_elems() : xv ? (size_t)t->d.depth * k_cross_vw_x2_wqf_halves() : 0);
    if (!xf && !xv) { g.khat = nullptr; g.vwt = nullptr; g.wqf = nullptr; }
    g.cvec = c.take<float>((size_t)t->d.depth * D);
    g.nullv = c.take<bf16_t>((size_t)I * seg + 64);
    g.rows = c.take<int32_t>((size_t)B * n);
    g.embc = c.take<bf16_t>((size_t)B * n * D * seg);
    g.embn = c.take<bf16_t>((size_t)B * n * D * seg);
    g.embm = c.take<bf16_t>((size_t)B * n * D * seg);
    g.embb = c.take<bf16_t>(t->F16 ? (size_t)B * n * D : 0);
    g.logits = c.take<float>((size_t)B * n * t->d.dim_out);
    g.xc = c.take<float>((size_t)2 * B * n * D);
    g.attc = c.take<bf16_t>((size_t)2 * B * n * I * seg);
    const bool fs = (t->d.logits_wcov || t->d.logits_wsub) && (t->d.dim_out % 256) == 0 && (D % 64) == 0;
    const size_t NT = fs ? (size_t)t->d.dim_out / 256 : 0;
    g.fs_thr = c.take<float>(fs ? (size_t)B * n : 0);
    g.fs_stats = c.take<float4>((size_t)B * n * NT * FS_REC);
    g.fs_cand = c.take<float4>((size_t)B * n * NT * FS_SLOT);
    g.fs_ws = c.take<unsigned char>(fs ? k_fused_threshold_ws_bytes(B * n, D) : 0);
    g.fs_sub = c.take<float>(fs && t->d.logits_wsub ? (size_t)B * n * t->d.logits_wsub_rows : 0);
    g.fb_rows = c.take<int32_t>(fs ? FB_CAP : 0);
    g.fb_cnt = c.take<int32_t>(fs ? FB_STEPS : 0);
    g.fb_x = c.take<bf16_t>(fs ? (size_t)FB_CAP * D * seg : 0);
    g.fb_logits = c.take<float>(fs ? (size_t)FB_CAP * t->d.dim_out : 0);
    if (!fs) {      // Carver::take(0) still hands out the (non-NULL) cursor: nothing was reserved, so nothing may be written through these
        g.fs_thr = nullptr; g.fs_stats = nullptr; g.fs_cand = nullptr; g.fs_ws = nullptr;
        g.fb_rows = nullptr; g.fb_cnt = nullptr; g.fb_x = nullptr; g.fb_logits = nullptr;
    }
    g.ctx_ws_bytes = mm_context_workspace_bytes(t, B, L);
    g.ctx_ws = c.take<unsigned char>(g.ctx_ws_bytes);
    g.sce = c.take<float>(t->d.self_cond ? (size_t)B * n * D : 0);
    g.emb_all = c.take<bf16_t>(t->d.self_cond ? (size_t)B * n * D : 0);
    g.pred = c.take<int64_t>((size_t)B * n);
    g.conf = c.take<float>((size_t)B * n);
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

size_t mm_generate_critic_workspace_bytes(const mm_transformer_t* critic, int B, int n, int L, int nc) {
    if (!critic) return 0;
    Carver c(nullptr);
    CriticBufs k;
    carve_critic(c, critic, B, n, L, nc, k);
    return c.used() + 256;
}

int mm_generate(const mm_transformer_t* t, mm_stream_t stream, const mm_generate_params* p, void* workspace, size_t workspace_bytes) {
    RC(check_model(t));
    if (!p) return mm_set_error(MM_ERR_SHAPE, "generate: params is NULL");
    hipStream_t s = (hipStream_t)stream;
    const int B = p->batch, n = p->n, T = p->timesteps, L = p->L, nc = p->nc, m = L + nc;
    const int D = t->d.dim, I = t->I, V = t->d.dim_out;
    const int PT = t->P, KD = (PT ? PT : 1) * D, KI = (PT ? PT : 1) * I;      // precision tier: operand rows are P segments wide
    if (B <= 0 || n <= 0 || n > t->d.seq_len || T <= 0) return mm_set_error(MM_ERR_SHAPE, "generate: bad batch/n/timesteps");
    if (t->d.vocab_rows != t->d.num_tokens + 1) return mm_set_error(MM_ERR_SHAPE, "generate: transformer has no mask id (MaskGitTransformer required)");
    if (!p->mask_counts || !p->temperatures || !p->ids || !p->scores) return mm_set_error(MM_ERR_SHAPE, "generate: schedule / outputs required");
    if (m <= 0) return mm_set_error(MM_ERR_SHAPE, "generate: empty context");
    // decode variants (mmp.py:556-609): single pass at cond_scale == 1 (mmp.py:247-248), self-conditioning (:325-328, 574), re-masking of
    // previously unmasked tokens (:608-609), token critic / self critic scores (:590-601)
    const bool single = p->cond_scale == 1.f;
    const int P = single ? 1 : 2;                                  // passes of the transformer per step
    const bool can_remask = (p->flags & MM_GEN_CAN_REMASK) != 0;
    const bool self_cond = t->d.self_cond != 0;
    const mm_transformer* critic = p->critic;
    const bool self_critic = p->critic_head_w != nullptr;
    const mm_transformer* cmodel = critic ? critic : (self_critic ? t : nullptr);      // the network the critic scores come from
    if (critic && self_critic) return mm_set_error(MM_ERR_SHAPE, "generate: token critic and self critic are exclusive (mmp.py:456)");
    if (critic && critic->d.dim_out != 1) return mm_set_error(MM_ERR_SHAPE, "generate: the token critic must have dim_out == 1");
    if (critic && critic->PC != t->PC) return mm_set_error(MM_ERR_SHAPE, "generate: the token critic must be packed for the same precision tier as the generator");
    if (cmodel) {
        if (!p->critic_noise) return mm_set_error(MM_ERR_SHAPE, "generate: critic_noise [timesteps][B][n] required with a critic");
        if (self_critic && !p->critic_head_b) return mm_set_error(MM_ERR_SHAPE, "generate: critic_head_b required with critic_head_w");
        if (!p->critic_workspace || p->critic_workspace_bytes < mm_generate_critic_workspace_bytes(cmodel, B, n, L, nc))
            return mm_set_error(MM_ERR_WORKSPACE, "generate: critic workspace too small (mm_generate_critic_workspace_bytes)");
    }
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
        bf16_t* ckv_l = g.ckv + (size_t)l * B * m * 2 * I * (PT ? 2 : 1);
        RC(gemm_dense(t, s, g.ctx, KD, (const bf16_t*)w.w_kv, KD, B * m, 2 * I, KD, ckv_l, 2 * I, PT ? OUT_F32 : OUT_BF16, nullptr));
        if (g.khat && cross_fold_on(t, w, m)) {
            const CrossFoldPack pk = {g.khat + (size_t)l * k_cross_fold_khat_elems(B), g.vwt + (size_t)l * k_cross_fold_vwt_elems(B), g.wqf + (size_t)l * k_cross_fold_wqf_elems()};
            RC(cross_fold_pack(t, s, w, ckv_l, B, m, pk));
        } else if (g.khat && cross_vw_on(t, w, m)) {
            RC(k_cross_vw_x2_pack(s, reinterpret_cast<const float*>(ckv_l), B, m, I, w.null_k, w.null_v, w.k_scale, (const bf16_t*)w.w_out, KI, PT, t->alpha,
                                  reinterpret_cast<float*>(g.khat) + (size_t)l * k_cross_vw_x2_khat_floats(B), g.vwt + (size_t)l * k_cross_vw_x2_vwt_halves(B)));
            RC(k_cross_vw_x2_wq_pack(s, (const bf16_t*)w.w_q, KD, PT, g.wqf + (size_t)l * k_cross_vw_x2_wqf_halves()));
        }
        if (nc == 0 && P == 2) {
            // softmax over the single unmasked (null) key is exactly 1 -> attention out = bf16(null_v) for every
            // query, so the null pass's cross-attention is the constant row to_out(null_v)
            if (PT) RC(k_split_rows(s, w.null_v, I, 1, I, t->PC, 0, 0, g.nullv, nullptr, 0, 0));
            else RC(k_f32_to_bf16(s, w.null_v, g.nullv, I));
            if (t->F8) RC(f8_linear_bf16(s, g.b, g.nullv, I, I, w.w_out, w.w_out_scale, 1, D, g.cvec + (size_t)l * D, D, 2, nullptr));
            else if (g.khat && cross_fold_on(t, w, m))      // the value the folded kernel adds on an all-masked row (the general path's null pass runs that kernel)
                RC(k_cross_fold_null_row(s, g.vwt + (size_t)l * k_cross_fold_vwt_elems(B), m, g.cvec + (size_t)l * D));
            else RC(gemm_dense(t, s, g.nullv, KI, (const bf16_t*)w.w_out, KI, 1, D, KI, g.cvec + (size_t)l * D, D, OUT_F32, nullptr));
        }
    }
    {
        hipLaunchKernelGGL(fill_i64_kernel, dim3(64), dim3(256), 0, s, p->ids, (long)M, mask_id);   // mmp.py:519
        RC(mm_check_launch("fill_i64_kernel"));
        const hipError_t e = hipMemsetAsync(p->scores, 0, (size_t)M * 4, s);                          // mmp.py:520
        if (e != hipSuccess) return mm_set_hip_error(e, "generate: memset scores");
    }
    // the critic's step-invariant context: a TokenCritic embeds / projects the text and the conditioning ids with its own weights; the self
    // critic reads the generator's cond-pass context
    CriticBufs cb;
    memset(&cb, 0, sizeof(cb));
    if (cmodel) {
        Carver cc(p->critic_workspace);
        carve_critic(cc, cmodel, B, n, L, nc, cb);
        RC(mm_transformer_context(cmodel, stream, p->text_embeds, B, L, p->cond_ids, nc, 0, cb.ctx, cb.masks, cb.ctx_ws, cb.ctx_ws_bytes));
        hipError_t e = hipMemsetAsync(cb.masks + (size_t)B * m, 0, (size_t)B * m, s);
        if (e != hipSuccess) return mm_set_hip_error(e, "generate: memset critic masks");
        if (nc > 0) {
            e = hipMemset2DAsync(cb.masks + (size_t)B * m + L, m, 1, nc, B, s);
            if (e != hipSuccess) return mm_set_hip_error(e, "generate: memset2d critic masks");
        }
    }

    if (g.fb_cnt && p->status) {
        const hipError_t e = hipMemsetAsync(g.fb_cnt, 0, (size_t)FB_STEPS * 4, s);
        if (e != hipSuccess) return mm_set_hip_error(e, "generate: memset fallback counters");
    }
    Bufs& b = g.b;
    const int seqs = P * B;          // sequences of one transformer pass over the batch: [cond B | null B]
    for (int step = 0; step < T; ++step) {
        const int k = p->mask_counts[step];
        const int R = can_remask ? M : B * k;                                                     // rows sampled at this step
        const int32_t* rows = can_remask ? nullptr : g.rows;
        RC(k_mask_step(s, p->scores, p->ids, B, n, k, mask_id, g.rows));                            // mmp.py:558-563
        if (p->trace_masked_ids) {
            const hipError_t e = hipMemcpyAsync(p->trace_masked_ids + (size_t)step * M, p->ids, (size_t)M * 8, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return mm_set_hip_error(e, "generate: trace copy");
        }
        // both CFG halves see the same ids: rows [0, M) = cond pass, [M, 2M) = null pass (mmp.py:250-252)
        if (PT) RC(k_embed_f32(s, p->ids, M, n, (const float*)t->d.token_emb, t->d.vocab_rows, (const float*)t->d.pos_emb, D, b.x));
        else RC(k_embed(s, p->ids, M, n, 0, (const bf16_t*)t->d.token_emb, t->d.vocab_rows, (const bf16_t*)t->d.pos_emb, D, b.x));
        if (self_cond && step > 0)                      // x += self_cond_to_init_embed(previous cond embed), mmp.py:325-328 (FF(zeros) = 0 at step 0)
            RC(ff_block(t, s, t->d.self_cond_ff, g.sce, b.x, M, b));
        // Both guidance halves start from the same ids, so their residual streams are identical until the first cross-attention: layer 0's
        // self-attention runs on the cond half only and the stream is duplicated behind it (below) instead of in front of it.
        // Round 6: compaction only where it pays.  The compacted row counts miss the tile classes of the wide kernels (FF w1 of 15232 rows ran on the 128 x 128
        // kernel: 75 us for 93 % of the rows the 256 x 256 kernel does in 70), so while more than 72 % of the positions are still masked (the first 9 of 18 steps) the
        // last layer runs on all rows like the others: 202 -> 162 us at step 4, loop 50.7 -> 50.4 ms per generate (same box, two runs each).  Same values either way.
        const bool compact_last0 = p->mask_counts[step] < n && p->mask_counts[step] * 100 <= n * 72 && !(g_mm_debug & 16384) && !self_cond && !can_remask;
        const bool share0 = P == 2 && !(t->d.depth == 1 && compact_last0);
        if (P == 2 && !share0) {
            const hipError_t e = hipMemcpyAsync(b.x + (size_t)M * D, b.x, (size_t)M * D * 4, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return mm_set_hip_error(e, "generate: x copy");
        }
        // Only the R = B*k rows sampled at this step are read after the last layer (final norm + to_logits, below), and after the last
        // self-attention has mixed the tokens every operator is row-wise: its output projection, the cross-attention (its queries) and
        // the feed-forward run on the COMPACTED rows [cond R | null R] (every sample has exactly k of them, position-sorted, so a
        // sample's queries stay contiguous).  Identical values for the rows that matter; (1 - k/n) of that work is skipped.
        // (Self-conditioning needs the embed of every position for the next step, re-masking samples every position: no compaction.)
        const bool compact_last = compact_last0;
        const bool fold = ln_fold_on(t);
        for (int l = 0; l < t->d.depth; ++l) {
            const mm_layer_weights& w = t->layers[l];
            const bf16_t* ckv_l = g.ckv + (size_t)l * B * m * 2 * I * (PT ? 2 : 1);
            const bool last_compact = l == t->d.depth - 1 && compact_last;
            const int nq = last_compact ? k : n;             // queries per sequence from here on
            const int Mq = B * nq;
            Bufs bc = b;
            // LayerNorm(dim) fold (see ln_fold_on): from layer 1 on every LayerNorm of the blocks rides in the GEMMs around it; layer 0 keeps the self-attention's
            // and the feed-forward's (its rows come from the embedding kernel and from the row copy below).  The null half's constant cross-attention row is then
            // added by the self-attention's output projection ((acc + x) + c: the same two roundings as the feed-forward LayerNorm's in-place add).
            const bool fold_l = fold && l > 0;
            const bool null_const = P == 2 && nc == 0;
            const float* cvec_l = g.cvec + (size_t)l * D;
            // precision tier (round 6): the null half's constant cross-attention row goes in with the self-attention's output projection as well ((acc + x) + c: the same two
            // roundings as the feed-forward LayerNorm's in-place add it replaces -- that pass then neither adds nor writes x back); layer 0 of a shared first layer keeps the add
            const bool early_c = (fold_l || (PT && !(g_mm_debug2 & 256) && !(l == 0 && share0))) && null_const;
            const bool xf = g.khat && cross_fold_on(t, w.cross_attn, m);
            const bool xv = !xf && g.khat && cross_vw_on(t, w.cross_attn, m);
            const CrossFoldPack pk_l = xv ? CrossFoldPack{reinterpret_cast<bf16_t*>(reinterpret_cast<float*>(g.khat) + (size_t)l * k_cross_vw_x2_khat_floats(B)),
                                                          g.vwt + (size_t)l * k_cross_vw_x2_vwt_halves(B), g.wqf + (size_t)l * k_cross_vw_x2_wqf_halves(), true}
                                          : CrossFoldPack{g.khat + (size_t)l * k_cross_fold_khat_elems(B), g.vwt + (size_t)l * k_cross_fold_vwt_elems(B),
                                                          g.wqf + (size_t)l * k_cross_fold_wqf_elems()};
            const CrossFoldPack* pkp = (xf || xv) ? &pk_l : nullptr;
            if (last_compact) {
                RC(self_attn_core(t, s, w.self_attn, seqs, n, b, fold_l));
                {      // residual stream and attention output of both guidance halves: one launch
                    const void* gsrc[4]; void* gdst[4]; long gpitch[4]; int gadd[4], gbytes[4];
                    int nj = 0;
                    for (int h = 0; h < P; ++h) {
                        gsrc[nj] = b.x; gpitch[nj] = (long)D * 4; gadd[nj] = h * M; gbytes[nj] = D * 4; gdst[nj] = g.xc + (size_t)h * R * D; ++nj;
                        gsrc[nj] = b.att; gpitch[nj] = (long)KI * 2; gadd[nj] = h * M; gbytes[nj] = KI * 2; gdst[nj] = g.attc + (size_t)h * R * KI; ++nj;
                    }
                    RC(k_gather_rows16_multi(s, nj, gsrc, gpitch, gadd, gbytes, gdst, g.rows, R));
                }
                bc.x = g.xc; bc.att = g.attc;
                if (t->F8) RC(f8_linear_bf16(s, b, g.attc, I, I, w.self_attn.w_out, w.self_attn.w_out_scale, P * R, D, g.xc, D, 2, g.xc));
                else if (fold) RC(gemm_resid(t, s, g.attc, KI, (const bf16_t*)w.self_attn.w_out, KI, P * R, D, KI, g.xc, bc, true, (fold_l && null_const) ? cvec_l : nullptr, R));
                else RC(gemm_dense(t, s, g.attc, KI, (const bf16_t*)w.self_attn.w_out, KI, P * R, D, KI, g.xc, D, OUT_F32, g.xc, (PT && early_c) ? cvec_l : nullptr, R));
            } else if (l == 0 && share0) {
                RC(self_attn_block(t, s, w.self_attn, B, n, b, false, fold));
                hipError_t e = hipMemcpyAsync(b.x + (size_t)M * D, b.x, (size_t)M * D * 4, hipMemcpyDeviceToDevice, s);
                if (e == hipSuccess && fold && !null_const) {      // the null half's cross-attention runs (condition ids stay attended): it reads the fold data of its rows too
                    const size_t np_ = (size_t)ln_fold_np(D);
                    e = hipMemcpyAsync(b.xb + (size_t)M * D, b.xb, (size_t)M * D * 2, hipMemcpyDeviceToDevice, s);
                    if (e == hipSuccess) e = hipMemcpyAsync(b.stp + (size_t)M * np_ * 2, b.stp, (size_t)M * np_ * 8, hipMemcpyDeviceToDevice, s);
                }
                if (e != hipSuccess) return mm_set_hip_error(e, "generate: x copy");
            } else {
                RC(self_attn_block(t, s, w.self_attn, seqs, n, b, fold_l, fold, early_c ? cvec_l : nullptr, M));
            }
            if (null_const) {
                RC(cross_attn_block(t, s, w.cross_attn, B, nq, ckv_l, m, 0, g.masks, bc, fold, fold_l, pkp));
                if (fold_l) RC(ff_block(t, s, w.ff, bc.x, bc.x, 2 * Mq, bc, nullptr, 0, true, l + 1 < t->d.depth));      // (the constant row went in with the output projection above)
                else if (early_c) RC(ff_block(t, s, w.ff, bc.x, bc.x, 2 * Mq, bc, nullptr, 0, false, false));             // (tier: likewise)
                else RC(ff_block(t, s, w.ff, bc.x, bc.x, 2 * Mq, bc, cvec_l, Mq, false, fold && l + 1 < t->d.depth));      // null rows += to_out(null_v) (the constant cross-attention)
            } else {
                RC(cross_attn_block(t, s, w.cross_attn, seqs, nq, ckv_l, m, P == 2 ? B : 0, g.masks, bc, fold, fold_l, pkp));
                RC(ff_block(t, s, w.ff, bc.x, bc.x, P * Mq, bc, nullptr, 0, fold_l, fold && l + 1 < t->d.depth));
            }
        }
        int64_t* ids_out = can_remask ? nullptr : p->ids;
        float* scores_out = can_remask ? nullptr : p->scores;
        int64_t* pred_out = can_remask ? g.pred : nullptr;
        float* conf_out = can_remask ? g.conf : nullptr;
        GemmArgs a;
        memset(&a, 0, sizeof(a));
        a.mode = MODE_DENSE; a.wide_tok = 1;
        a.W = (const bf16_t*)t->d.to_logits; a.N = V; a.ldw = KD; a.K = KD;
        a.M = R; a.X = nullptr; a.ldx = KD;      // (X: the mixed rows, set below)
        a.out = g.logits; a.ldc = V; a.out_kind = OUT_F32;
        a.debug = g_mm_debug;
        a.f16 = t->F16; a.alpha = t->alpha; a.terms = t->F16 ? t->P : 0;
        // Sampling without the logits round trip: the GEMM emits tile statistics + the candidates above a per-row lower bound of the k-th largest
        // logit (estimated from the row's embeddings and the vocabulary statistics of to_logits), the finishing kernel verifies the bound.
        const bool qbound = t->d.logits_wsub && t->d.logits_wsub_rows > 0;      // distribution-free bound from sampled vocabulary columns (else: the Gaussian estimate)
        bool fused = (qbound || (t->d.logits_wcov && t->d.logits_wmean)) && p->status && !(p->flags & MM_GEN_NO_FUSED_SAMPLING) && (V % 256) == 0 &&
                     !(g_mm_debug & (8 | 4096 | 8192 | (1 << 25))) && mm_gemm_cfg2_eligible(a);
        if (fused && t->F16) {      // fp16 terms: the emission exists on the 256 x 256 kernel only (>= 1024 rows); smaller steps take the logits path
            GemmArgs probe = a;
            probe.fs_thr = g.fs_thr; probe.fs_stats = g.fs_stats; probe.fs_cand = g.fs_cand;
            fused = mm_gemm_wide_fused_eligible(probe) && !(g_mm_debug & ((1 << 26) | (1 << 28) | (1 << 30)));
        }
        // The step tail of the plain decode (bf16 engine, two guidance passes, fused sampling): final LayerNorm of both passes' sampled rows, the guidance mix
        // and the bound estimate's row means in ONE pass (k_final_mix) -- the same values as the separate LayerNorm / k_cfg_mix / fused_combine kernels below
        const bool fmix = fused && !PT && !t->F16 && !t->F8 && !single && !self_cond && P == 2 && (t->d.logits_wmean || qbound) && KD == D;
        const bf16_t* emb_in = g.embc;
        if (fmix) {
            const float* xc_ = compact_last ? g.xc : b.x;
            const float* xn_ = compact_last ? g.xc + (size_t)R * D : b.x + (size_t)M * D;
            RC(k_final_mix(s, xc_, xn_, D, R, D, t->d.final_gamma, t->d.final_beta, compact_last ? nullptr : rows, p->cond_scale, g.embm, qbound ? nullptr : t->d.logits_wmean,
                           k_fused_threshold_mu(g.fs_ws, R, D)));
            emb_in = g.embm;
        } else {
            // final norm + to_logits + CFG only at the rows that are sampled this step
            if (PT) {
                const float* fg = t->d.final_gamma;
                const float* fb = t->d.final_beta;
                if (compact_last) {
                    RC(k_layernorm_split(s, g.xc, D, R, D, fg, fb, nullptr, t->PC, g.embc, nullptr, nullptr, 0, nullptr));
                    if (P == 2) RC(k_layernorm_split(s, g.xc + (size_t)R * D, D, R, D, fg, fb, nullptr, t->PC, g.embn, nullptr, nullptr, 0, nullptr));
                } else {
                    if (self_cond)      // the cond pass's fp32 embed at every position is the next step's self-conditioning input (mmp.py:574)
                        RC(k_layernorm_split(s, b.x, D, M, D, fg, fb, nullptr, t->PC, nullptr, g.sce, nullptr, 0, nullptr));
                    RC(k_layernorm_split(s, b.x, D, R, D, fg, fb, rows, t->PC, g.embc, nullptr, nullptr, 0, nullptr));
                    if (P == 2) RC(k_layernorm_split(s, b.x + (size_t)M * D, D, R, D, fg, fb, rows, t->PC, g.embn, nullptr, nullptr, 0, nullptr));
                }
            } else if (compact_last) {
                RC(k_layernorm(s, g.xc, D, R, D, t->d.final_gamma, t->d.final_beta, nullptr, g.embc, D));
                if (P == 2) RC(k_layernorm(s, g.xc + (size_t)R * D, D, R, D, t->d.final_gamma, t->d.final_beta, nullptr, g.embn, D));
            } else {
                if (self_cond) {      // the cond pass's embed at EVERY position is the next step's self-conditioning input (mmp.py:574)
                    RC(k_layernorm(s, b.x, D, M, D, t->d.final_gamma, t->d.final_beta, nullptr, g.emb_all, D));
                    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(1024), dim3(256), 0, s, g.emb_all, g.sce, (long)M * D);
                    RC(mm_check_launch("bf16_to_f32_kernel"));
                }
                RC(k_layernorm(s, b.x, D, R, D, t->d.final_gamma, t->d.final_beta, rows, g.embc, D));
                if (P == 2) RC(k_layernorm(s, b.x + (size_t)M * D, D, R, D, t->d.final_gamma, t->d.final_beta, rows, g.embn, D));
            }
            // Guidance in the embedding (round 3): to_logits is linear (mmp.py:332), so null + (cond - null) * s of the logits (mmp.py:254) is to_logits of
            // e = e_null + (e_cond - e_null) * s.  The two passes' final embeddings are mixed first (k_cfg_mix) and multiplied ONCE: half the flops of the
            // loop's dominant GEMM; the general path (Transformer.forward_with_cond_scale) does the same, so the two stay bit-identical.
            if (!single) {
                RC(k_cfg_mix(s, g.embc, g.embn, KD, R, D, t->PC, p->cond_scale, g.embm));
                emb_in = g.embm;
            }
        }
        a.X = emb_in;
        const double gemm_flops = 2.0 * (double)R * (double)V * (double)KD;      // EXECUTED bf16 MFMA flops: one pass over the mixed rows (x the term products in the precision tier)
        if (fused) {
            const bf16_t* emb_est = emb_in;      // the rows the bound is estimated from: bf16 (the leading bf16 term in the 'bf16x3' tier)
            long ld_est = KD;
            if (t->F16) {
                hipLaunchKernelGGL(f16_rows_to_bf16_kernel, dim3(1024), dim3(256), 0, s, emb_in, (long)KD, (long)R, D, g.embb);
                RC(mm_check_launch("f16_rows_to_bf16_kernel"));
                emb_est = g.embb; ld_est = D;
            }
            if (qbound) {
                // the rows' logits at the sampled columns (a plain bf16 GEMM, 3 % of the step's flops) and their rank-th largest
                const int S = t->d.logits_wsub_rows;
                GemmArgs q;
                memset(&q, 0, sizeof(q));
                q.mode = MODE_DENSE;
                q.W = (const bf16_t*)t->d.logits_wsub; q.N = S; q.ldw = D; q.K = D; q.M = R; q.X = emb_est; q.ldx = (int)ld_est;
                q.out = g.fs_sub; q.ldc = S; q.out_kind = OUT_F32; q.debug = g_mm_debug;
                RC(mm_gemm_launch(q, s));
                RC(k_fused_quantile(s, g.fs_sub, S, R, S, k_fused_quantile_rank(p->k_keep, V, S), g.fs_thr));
            } else if (fmix) RC(k_fused_threshold_mixed(s, emb_in, D, R, D, (const bf16_t*)t->d.logits_wcov, k_fused_z(p->k_keep, V, FS_MARGIN), g.fs_ws, g.fs_thr));
            else RC(k_fused_threshold(s, emb_est, emb_est, ld_est, R, D, 1.f, t->d.logits_wmean, (const bf16_t*)t->d.logits_wcov, k_fused_z(p->k_keep, V, FS_MARGIN),
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
            fa.R = R; fa.V = V; fa.k_keep = p->k_keep; fa.rows = rows;
            fa.temperature = p->temperatures[step]; fa.noise_kind = p->noise_kind;
            fa.noise = p->noise ? p->noise + (size_t)step * M * V : nullptr; fa.noise_ld = V;
            fa.seed = p->seed; fa.row_offset = p->row_offset * (uint64_t)n; fa.step = (uint32_t)step; fa.seed_dev = p->seed_dev; fa.row_mul = n;
            fa.ids = ids_out; fa.scores = scores_out; fa.pred_out = pred_out; fa.score_out = conf_out; fa.fail_flag = p->status;
            const bool row_fallback = step < FB_STEPS;
            if (row_fallback) { fa.fail_rows = g.fb_rows; fa.fail_count = g.fb_cnt + step; fa.fail_cap = FB_CAP; }
            if (prof::enabled) pr = prof::begin(s, 4.0 * (double)R * (double)V);      // logits-equivalent bytes (what a logits-reading sampler reads)
            RC(k_sample_fused(s, fa));                                                                  // mmp.py:576-609
            if (prof::enabled) prof::end(s, 1, pr);
            if (row_fallback) {
                // Rows whose candidate bound could not be verified (heavy-tailed logits) are finished HERE, on the logits path: gather their
                // embeddings, recompute their logits with the 128 x 128 dense kernel (same MFMA order as the persistent kernel: the same
                // values), sample them with sample_kernel (same per-tile softmax statistics: the same confidences).  The row count lives on
                // the device; with no failed row the three launches exit at once.  Only more than FB_CAP rows in one step raise *status.
                RC(k_gather_rows16_counted(s, emb_in, (long)KD * 2, g.fb_rows, g.fb_cnt + step, FB_CAP, KD * 2, g.fb_x, p->status + 1));
                GemmArgs fg;
                memset(&fg, 0, sizeof(fg));
                fg.mode = MODE_DENSE;
                fg.W = (const bf16_t*)t->d.to_logits; fg.N = V; fg.ldw = KD; fg.K = KD; fg.M = FB_CAP; fg.X = g.fb_x; fg.ldx = KD;
                fg.out = g.fb_logits; fg.ldc = V; fg.out_kind = OUT_F32; fg.m_dev = g.fb_cnt + step;
                fg.f16 = t->F16; fg.alpha = t->alpha;
                RC(mm_gemm_launch(fg, s));
                SampleArgs fs_;
                memset(&fs_, 0, sizeof(fs_));
                fs_.logits = g.fb_logits; fs_.ld = V; fs_.R = FB_CAP; fs_.V = V; fs_.k_keep = p->k_keep; fs_.rows = rows;
                fs_.temperature = p->temperatures[step]; fs_.noise_kind = p->noise_kind;
                fs_.noise = p->noise ? p->noise + (size_t)step * M * V : nullptr; fs_.noise_ld = V;
                fs_.seed = p->seed; fs_.row_offset = p->row_offset * (uint64_t)n; fs_.step = (uint32_t)step; fs_.seed_dev = p->seed_dev; fs_.row_mul = n;
                fs_.ids = ids_out; fs_.scores = scores_out; fs_.pred_out = pred_out; fs_.score_out = conf_out;
                fs_.src_rows = g.fb_rows; fs_.count_dev = g.fb_cnt + step;
                RC(k_sample_rows(s, fs_));
            }
        } else {
            {
                prof::Rec pr;
                if (prof::enabled) pr = prof::begin(s, gemm_flops);
                RC(mm_gemm_launch(a, s));      // (cond_scale == 1: the plain to_logits of the one pass, mmp.py:247-248, 332)
                if (prof::enabled) prof::end(s, 0, pr);
            }
            SampleArgs sa;
            memset(&sa, 0, sizeof(sa));
            sa.logits = g.logits; sa.ld = V; sa.R = R; sa.V = V; sa.k_keep = p->k_keep; sa.rows = rows;
            sa.temperature = p->temperatures[step]; sa.noise_kind = p->noise_kind;
            sa.noise = p->noise ? p->noise + (size_t)step * M * V : nullptr; sa.noise_ld = V;
            sa.seed = p->seed; sa.row_offset = p->row_offset * (uint64_t)n; sa.step = (uint32_t)step; sa.seed_dev = p->seed_dev; sa.row_mul = n;
            sa.ids = ids_out; sa.scores = scores_out; sa.pred_out = pred_out; sa.score_out = conf_out;
            prof::Rec pr;
            if (prof::enabled) pr = prof::begin(s, 4.0 * (double)R * (double)V);      // one fp32 read of each row
            RC(k_sample_rows(s, sa));                                                                // mmp.py:576-609
            if (prof::enabled) prof::end(s, 1, pr);
        }
        if (can_remask) {
            hipLaunchKernelGGL(merge_pred_kernel, dim3((M + 255) / 256), dim3(256), 0, s, p->ids, g.pred, p->scores, g.conf, mask_id, (long)M);
            RC(mm_check_launch("merge_pred_kernel"));
        }
        if (cmodel) {
            // critic scores of the freshly sampled ids at every position (mmp.py:590-601)
            if (critic) {        // TokenCritic.forward_with_cond_scale: both passes + the guidance combine of its 1-wide head
                RC(mm_transformer_forward(critic, stream, p->ids, B, n, cb.ctx, cb.masks, m, nullptr, cb.embc, nullptr, cb.fwd_ws, cb.fwd_ws_bytes));
                const int Dc = (critic->P ? critic->P : 1) * critic->d.dim;      // operand row width of the critic's head
                if (single) {
                    RC(gemm_dense(critic, s, cb.embc, Dc, (const bf16_t*)critic->d.to_logits, Dc, M, 1, Dc, cb.sc, 1, OUT_F32, nullptr));
                } else {
                    RC(mm_transformer_forward(critic, stream, p->ids, B, n, cb.ctx, cb.masks + (size_t)B * m, m, nullptr, cb.embn, nullptr, cb.fwd_ws,
                                              cb.fwd_ws_bytes));
                    // guidance in the embedding, like the generator's logits (and like TokenCritic.forward_with_cond_scale on the general path):
                    // mix the two passes' embeddings (in place over the null pass's), then the 1-wide head once
                    RC(k_cfg_mix(s, cb.embc, cb.embn, Dc, M, critic->d.dim, critic->PC, p->cond_scale, cb.embn));
                    RC(gemm_dense(critic, s, cb.embn, Dc, (const bf16_t*)critic->d.to_logits, Dc, M, 1, Dc, cb.sc, 1, OUT_F32, nullptr));
                }
            } else {             // SelfCritic (mmp.py:352-374): Linear(dim, 1) on the generator's cond-pass embed of the new ids (no self-conditioning input)
                RC(mm_transformer_forward(t, stream, p->ids, B, n, cb.ctx, cb.masks, m, nullptr, cb.embc, nullptr, cb.fwd_ws, cb.fwd_ws_bytes));
                if (PT) {      // precision tier: critic_head_w is the [1][P*D] segment pack of the Linear(dim, 1) weight
                    GemmArgs ha;
                    memset(&ha, 0, sizeof(ha));
                    ha.mode = MODE_DENSE;
                    ha.W = (const bf16_t*)p->critic_head_w; ha.N = 1; ha.ldw = KD; ha.K = KD; ha.M = M; ha.X = cb.embc; ha.ldx = KD;
                    ha.out = cb.sc; ha.ldc = 1; ha.out_kind = OUT_F32; ha.bias = p->critic_head_b;
                    ha.f16 = t->F16; ha.alpha = t->alpha;
                    RC(mm_gemm_launch(ha, s));
                } else {
                    RC(mm_conv2d_nhwc(stream, cb.embc, M, 1, 1, D, p->critic_head_w, 1, 1, 1, 1, 0, 0, 1, 1, 1, 0, 0, 1, 1, p->critic_head_b, 0, nullptr, cb.sc, 1));
                }
            }
            const float ratio = (float)((double)(T - 1 - step) / (double)T);
            hipLaunchKernelGGL(critic_scores_kernel, dim3((M + 255) / 256), dim3(256), 0, s, cb.sc, p->critic_noise + (size_t)step * M, p->critic_noise_scale, ratio,
                               p->scores, (long)M);
            RC(mm_check_launch("critic_scores_kernel"));
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
