// Row kernels of the 'bf16x3' precision tier (model.hip): fp32-grade results on the bf16 matrix pipe.
//
// An fp32 value is the exact sum of three bf16 terms, x = h + m + l (8 + 8 + 8 significant bits: h = rn(x), m = rn(x - h),
// l = x - h - m, every difference exact in fp32).  A product of two such sums keeps fp32 accuracy when all term pairs of
// level i + j <= 2 are accumulated (h.h, m.h, l.h, h.m, m.m, h.l): the dropped pairs are < 2^-27 of the leading one.  How
// many of them exist depends on the WEIGHT: a checkpoint whose values are bf16-representable (trained or stored in bf16,
// and every fixture of tests/golden) has m = l = 0 -> 3 products; two-term weights -> 5; general fp32 weights -> 6.
//
// The term pairs ride on the UNCHANGED bf16 GEMM kernels (gemm*.hip) as a K-concatenation:  X' = [xh|xm|xl|xh|xm|xh] (first P
// segments, each K wide), W' = [wh|wh|wh|wm|wm|wl], so that X'.W'^T = sum of the P kept pairs with fp32 accumulation inside the
// MFMA -- same tiles, same pipelines, P x the flops of the bf16 engine and none of the 1/16-rate fp32 MFMA.  What is new is only
// the producers of X' below: every kernel that feeds a GEMM writes its fp32 result as P bf16 segments (reference operators:
// LayerNorm mmp.py:63-70, GEGLU + LayerNorm(inner) :72-88, embeddings :322-323, text context :302-318).
#include "common.h"
#include "muse_hip_internal.h"

namespace {

// out[(r / rpb) * out_bstride + (r % rpb) * P*K ...] = split(x[r][0..K))        (rows grouped per batch so a [B][L] block can land inside [B][m])
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, long ldx, long rows, int K, int P, int rpb, long out_bstride,
                                                         bf16_t* __restrict__ out, uint8_t* __restrict__ nz_mask, int mask_bstride, int drop) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long b = row / rpb, j = row - b * rpb;
    const float* xr = x + row * ldx;
    bf16_t* orow = out + b * out_bstride + j * (long)split_count(P) * K;
    bool nz = false;
    for (int c = lane * 4; c < K; c += 256) {
        const float4 q = *reinterpret_cast<const float4*>(xr + c);
        const float v[4] = {q.x, q.y, q.z, q.w};
        nz |= (q.x != 0.f) | (q.y != 0.f) | (q.z != 0.f) | (q.w != 0.f);
        store_split4(orow, K, P, c, v);
    }
    if (nz_mask) {                                     // mmp.py:304: mask = (text_embeds != 0).any(-1)
        const bool any = __ballot(nz) != 0ull;
        if (lane == 0) nz_mask[b * mask_bstride + j] = (any && !drop) ? 1 : 0;
    }
}

// ctx[b][L + c] = split(token_emb[cond_ids[b][c]]), mask = 1       (mmp.py:314-318)
__global__ __launch_bounds__(256) void gather_split_kernel(const float* __restrict__ table, int D, int P, const int64_t* __restrict__ idx, int B, int nc,
                                                           int vocab_rows, bf16_t* __restrict__ ctx, uint8_t* __restrict__ mask, int m, int L) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= (long)B * nc) return;
    const int b = (int)(r / nc), j = (int)(r - (long)b * nc);
    long id = idx[r];
    id = id < 0 ? 0 : (id >= vocab_rows ? vocab_rows - 1 : id);
    const float* src = table + id * D;
    bf16_t* orow = ctx + ((long)b * m + L + j) * (long)split_count(P) * D;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 q = *reinterpret_cast<const float4*>(src + c);
        const float v[4] = {q.x, q.y, q.z, q.w};
        store_split4(orow, D, P, c, v);
    }
    if (lane == 0 && mask) mask[(long)b * m + L + j] = 1;
}

// LayerNorm (F.layer_norm, eps 1e-5, mmp.py:63-70) of fp32 rows, two passes over the row in registers like the fp32 engine (parity.hip);
// the result leaves as P bf16 segments and / or as fp32.  Optional row gather; ADD: rows >= add_from get addvec added in place first.
template <int NIT, bool ADD>
__global__ __launch_bounds__(256) void layernorm_split_kernel(const float* __restrict__ x, long ldx, int rows, int D, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const int32_t* __restrict__ row_index, int P,
                                                              bf16_t* __restrict__ out, float* __restrict__ out_f32, const float* __restrict__ addvec,
                                                              int add_from, float* xw) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long src = row_index ? (long)row_index[row] : (long)row;
    const float* xr = x + src * ldx;
    const int nvec = D >> 2;
    float4 v[NIT];
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nvec) {
            v[it] = *reinterpret_cast<const float4*>(xr + c * 4);
            if constexpr (ADD) {
                if (row >= add_from) {
                    const float4 av = *reinterpret_cast<const float4*>(addvec + c * 4);
                    v[it].x += av.x; v[it].y += av.y; v[it].z += av.z; v[it].w += av.w;
                    *reinterpret_cast<float4*>(xw + src * ldx + c * 4) = v[it];
                }
            }
            sum += (v[it].x + v[it].y) + (v[it].z + v[it].w);
        }
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nvec) {
            const float a = v[it].x - mean, b = v[it].y - mean, cc = v[it].z - mean, d = v[it].w - mean;
            sq += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(sq) / (float)D + 1e-5f);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nvec) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c * 4);
            float4 bt = make_float4(0.f, 0.f, 0.f, 0.f);
            if (beta) bt = *reinterpret_cast<const float4*>(beta + c * 4);
            const float o[4] = {(v[it].x - mean) * rstd * g.x + bt.x, (v[it].y - mean) * rstd * g.y + bt.y,
                                (v[it].z - mean) * rstd * g.z + bt.z, (v[it].w - mean) * rstd * g.w + bt.w};
            if (out) store_split4(out + (long)row * split_count(P) * D, D, P, c * 4, o);
            if (out_f32) *reinterpret_cast<float4*>(out_f32 + (long)row * D + c * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// GEGLU + LayerNorm(inner) (mmp.py:72-77, 86-88) on the fp32 output of the w1 GEMM: h = [x (Fp wide, F valid) | gate (Fp wide)],
// a = gate * gelu_erf(x) with libm's erff (the fp32 engine's form), LN over the F valid columns, P segments out (columns >= F zero).
template <int NIT>
__global__ __launch_bounds__(256) void geglu_ln_split_kernel(const float* __restrict__ h, long ldh, int rows, int F, int Fp, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int P, bf16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* hr = h + (long)row * ldh;
    const int nvec = Fp >> 2;
    float a[NIT][4];
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nvec) {
            const float4 xv = *reinterpret_cast<const float4*>(hr + c * 4);
            const float4 gv = *reinterpret_cast<const float4*>(hr + Fp + c * 4);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float act = gs[j] * (0.5f * xs[j] * (1.f + erff(xs[j] * 0.70710678118654752440f)));
                a[it][j] = (c * 4 + j < F) ? act : 0.f;
                sum += a[it][j];
            }
        }
    }
    const float mean = wave_sum(sum) / (float)F;
    float sq = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nvec) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c * 4 + j < F) { const float d = a[it][j] - mean; sq += d * d; }
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(sq) / (float)F + 1e-5f);
    bf16_t* orow = out + (long)row * split_count(P) * Fp;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nvec) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c * 4);      // gamma / beta padded to Fp floats by the caller
            float4 bt = make_float4(0.f, 0.f, 0.f, 0.f);
            if (beta) bt = *reinterpret_cast<const float4*>(beta + c * 4);
            const float gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {bt.x, bt.y, bt.z, bt.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (c * 4 + j < F) ? (a[it][j] - mean) * rstd * gg[j] + bb[j] : 0.f;
            store_split4(orow, Fp, P, c * 4, o);
        }
    }
}

// x[row] = tok[ids[row]] + pos[row % n], fp32 tables (mmp.py:322-323)
__global__ __launch_bounds__(256) void embed_f32v_kernel(const int64_t* __restrict__ ids, long rows, int n, const float* __restrict__ tok, int vocab_rows,
                                                         const float* __restrict__ pos, int D, float* __restrict__ x) {
    const int chunks = D >> 2;
    const long total = rows * chunks;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / chunks;
        const int c = (int)(i - r * chunks);
        long id = ids[r];
        id = id < 0 ? 0 : (id >= vocab_rows ? vocab_rows - 1 : id);
        const float4 a = *reinterpret_cast<const float4*>(tok + id * D + c * 4);
        const float4 b = *reinterpret_cast<const float4*>(pos + (long)(r % n) * D + c * 4);
        *reinterpret_cast<float4*>(x + r * D + c * 4) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

// The two guidance passes combined in the EMBEDDING: logits = embed @ W^T (mmp.py:332) is linear, so null + (cond - null) * s (mmp.py:254) of the
// logits equals the logits of e = e_null + (e_cond - e_null) * s -- one [R x V x D] product instead of two.  P == 0: bf16 rows in, the mix
// rounded to bf16 once; P > 0 (precision tier): the rows are term-segment packs, mixed as exact fp32 values (h + m) + l and re-split.
__global__ __launch_bounds__(256) void cfg_mix_kernel(const bf16_t* __restrict__ ec, const bf16_t* __restrict__ en, long ld, long rows, int D, int P, float s,
                                                      bf16_t* __restrict__ out) {
    const int chunks = D >> 2;
    const long total = rows * chunks;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / chunks;
        const int c = (int)(i - r * chunks) * 4;
        float cv[4], nv[4], o[4];
        if (P == 0) {
            const uint2 a = *reinterpret_cast<const uint2*>(ec + r * ld + c), b = *reinterpret_cast<const uint2*>(en + r * ld + c);
            cv[0] = bf16lo(a.x); cv[1] = bf16hi(a.x); cv[2] = bf16lo(a.y); cv[3] = bf16hi(a.y);
            nv[0] = bf16lo(b.x); nv[1] = bf16hi(b.x); nv[2] = bf16lo(b.y); nv[3] = bf16hi(b.y);
        } else if (split_is_f16(P)) {      // two fp16 terms per value: segments 0 (h) and 1 (l)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint2 a = *reinterpret_cast<const uint2*>(ec + r * ld + (long)k * D + c), b = *reinterpret_cast<const uint2*>(en + r * ld + (long)k * D + c);
                const uint32_t aw[2] = {a.x, a.y}, bw[2] = {b.x, b.y};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float av = f16_bits_to_f32((uint16_t)(aw[j >> 1] >> (16 * (j & 1)))), bv = f16_bits_to_f32((uint16_t)(bw[j >> 1] >> (16 * (j & 1))));
                    cv[j] = k == 0 ? av : cv[j] + av;
                    nv[j] = k == 0 ? bv : nv[j] + bv;
                }
            }
        } else {
            float t[2][3][4];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint2 a = *reinterpret_cast<const uint2*>(ec + r * ld + (long)k * D + c), b = *reinterpret_cast<const uint2*>(en + r * ld + (long)k * D + c);
                t[0][k][0] = bf16lo(a.x); t[0][k][1] = bf16hi(a.x); t[0][k][2] = bf16lo(a.y); t[0][k][3] = bf16hi(a.y);
                t[1][k][0] = bf16lo(b.x); t[1][k][1] = bf16hi(b.x); t[1][k][2] = bf16lo(b.y); t[1][k][3] = bf16hi(b.y);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { cv[j] = (t[0][0][j] + t[0][1][j]) + t[0][2][j]; nv[j] = (t[1][0][j] + t[1][1][j]) + t[1][2][j]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = nv[j] + (cv[j] - nv[j]) * s;
        if (P == 0) *reinterpret_cast<uint2*>(out + r * D + c) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        else store_split4(out + r * (long)split_count(P) * D, D, P, c, o);
    }
}

inline bool bad_p(int P) {      // (fp16 terms: the producer-side MM_SPLIT_NODUP_BIT may ride on the code)
    if (P & MM_SPLIT_NODUP_BIT) return P != (MM_SPLIT_NODUP_BIT | MM_SPLIT_F16_BIT | 3);
    return P != 3 && P != 5 && P != 6 && P != (MM_SPLIT_F16_BIT | 2) && P != (MM_SPLIT_F16_BIT | 3);
}      // bf16 terms: 3 / 5 / 6; fp16 terms: MM_SPLIT_F16 | 2 / 3

}  // namespace

int k_split_rows(hipStream_t s, const float* x, long ldx, long rows, int K, int P, int rows_per_batch, long out_batch_stride, bf16_t* out,
                 uint8_t* nz_mask, int mask_bstride, int drop) {
    if (rows <= 0) return MM_OK;
    if (bad_p(P)) return mm_set_error(MM_ERR_SHAPE, "split_rows: products must be 3, 5, 6 (bf16 terms) or MM_SPLIT_F16 | 2, 3 (fp16 terms)");
    if ((K % 4) || (ldx % 4)) return mm_set_error(MM_ERR_ALIGN, "split_rows: K and the row stride must be multiples of 4");
    if (rows_per_batch <= 0) { rows_per_batch = (int)(rows > 0x7fffffff ? 0x7fffffff : rows); out_batch_stride = 0; }
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, ldx, rows, K, P, rows_per_batch, out_batch_stride, out,
                       nz_mask, mask_bstride, drop);
    return mm_check_launch("split_rows_kernel");
}

int k_gather_split(hipStream_t s, const float* table, int D, int P, const int64_t* idx, int B, int nc, int vocab_rows, bf16_t* ctx, uint8_t* mask,
                   int m, int L) {
    if (B <= 0 || nc <= 0) return MM_OK;
    if (bad_p(P) || (D % 4)) return mm_set_error(MM_ERR_SHAPE, "gather_split: bad products / dim");
    hipLaunchKernelGGL(gather_split_kernel, dim3((unsigned)(((long)B * nc + 3) / 4)), dim3(256), 0, s, table, D, P, idx, B, nc, vocab_rows, ctx, mask, m, L);
    return mm_check_launch("gather_split_kernel");
}

int k_layernorm_split(hipStream_t s, const float* x, long ldx, int rows, int D, const float* gamma, const float* beta, const int32_t* row_index,
                      int P, bf16_t* out, float* out_f32, const float* addvec, int add_from, float* xw) {
    if (rows <= 0) return MM_OK;
    if (out && bad_p(P)) return mm_set_error(MM_ERR_SHAPE, "layernorm_split: products must be 3, 5 or 6");
    if (D % 4 || D > 2048 || (ldx % 4)) return mm_set_error(MM_ERR_SHAPE, "layernorm_split: dim must be a multiple of 4 and <= 2048");
    const int nit = (D / 4 + 63) / 64;
    const dim3 grid((rows + 3) / 4), block(256);
#define MM_LNS(N_, A_) hipLaunchKernelGGL((layernorm_split_kernel<N_, A_>), grid, block, 0, s, x, ldx, rows, D, gamma, beta, row_index, P, out, out_f32, addvec, add_from, xw)
    if (addvec) {
        if (nit <= 2) MM_LNS(2, true); else if (nit <= 4) MM_LNS(4, true); else MM_LNS(8, true);
    } else {
        if (nit <= 2) MM_LNS(2, false); else if (nit <= 4) MM_LNS(4, false); else MM_LNS(8, false);
    }
#undef MM_LNS
    return mm_check_launch("layernorm_split_kernel");
}

int k_geglu_ln_split(hipStream_t s, const float* h, long ldh, int rows, int F, int Fp, const float* gamma, const float* beta, int P, bf16_t* out) {
    if (rows <= 0) return MM_OK;
    if (bad_p(P)) return mm_set_error(MM_ERR_SHAPE, "geglu_ln_split: products must be 3, 5 or 6");
    if (Fp % 4 || Fp < F || Fp > 64 * 4 * 24 || (ldh % 4)) return mm_set_error(MM_ERR_SHAPE, "geglu_ln_split: padded inner dim must be a multiple of 4, >= F and <= 6144");
    const int nit = (Fp / 4 + 63) / 64;
    const dim3 grid((rows + 3) / 4), block(256);
    if (nit <= 6) hipLaunchKernelGGL((geglu_ln_split_kernel<6>), grid, block, 0, s, h, ldh, rows, F, Fp, gamma, beta, P, out);
    else if (nit <= 12) hipLaunchKernelGGL((geglu_ln_split_kernel<12>), grid, block, 0, s, h, ldh, rows, F, Fp, gamma, beta, P, out);
    else hipLaunchKernelGGL((geglu_ln_split_kernel<24>), grid, block, 0, s, h, ldh, rows, F, Fp, gamma, beta, P, out);
    return mm_check_launch("geglu_ln_split_kernel");
}

int k_embed_f32(hipStream_t s, const int64_t* ids, long rows, int n, const float* tok, int vocab_rows, const float* pos, int D, float* x) {
    if (rows <= 0) return MM_OK;
    if (D % 4) return mm_set_error(MM_ERR_SHAPE, "embed: dim must be a multiple of 4");
    long blocks = (rows * (D / 4) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(embed_f32v_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ids, rows, n, tok, vocab_rows, pos, D, x);
    return mm_check_launch("embed_f32v_kernel");
}

int k_cfg_mix(hipStream_t s, const bf16_t* ec, const bf16_t* en, long ld, long rows, int D, int P, float cond_scale, bf16_t* out) {
    if (rows <= 0) return MM_OK;
    if (P != 0 && bad_p(P)) return mm_set_error(MM_ERR_SHAPE, "cfg_mix: products must be 0 (bf16 rows), 3, 5 or 6");
    if ((D % 4) || (ld % 4)) return mm_set_error(MM_ERR_ALIGN, "cfg_mix: dim and the row stride must be multiples of 4");
    long blocks = (rows * (D / 4) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(cfg_mix_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ec, en, ld, rows, D, P, cond_scale, out);
    return mm_check_launch("cfg_mix_kernel");
}

extern "C" {

// e = e_null + (e_cond - e_null) * cond_scale on the final embeddings of the two guidance passes (see cfg_mix_kernel): bf16 rows [R][ld] -> bf16
// [R][D] (products == 0), or term-segment packs [R][ld >= products * D] -> [R][products * D] (precision tier)
int mm_cfg_mix(mm_stream_t stream, const void* emb_cond, const void* emb_null, int64_t ld, int64_t rows, int D, int products, float cond_scale, void* out) {
    if (rows > 0 && (!emb_cond || !emb_null || !out)) return mm_set_error(MM_ERR_SHAPE, "cfg_mix: NULL pointer");
    return k_cfg_mix((hipStream_t)stream, (const bf16_t*)emb_cond, (const bf16_t*)emb_null, ld, rows, D, products, cond_scale, (bf16_t*)out);
}

// fp32 rows -> the P-segment bf16 operand form of the 'bf16x3' tier (X' of the header comment); out [rows][P * K]
int mm_split_rows(mm_stream_t stream, const float* x, int64_t ldx, int64_t rows, int K, int products, void* out) {
    if (rows > 0 && (!x || !out)) return mm_set_error(MM_ERR_SHAPE, "split_rows: NULL pointer");
    return k_split_rows((hipStream_t)stream, x, ldx, rows, K, products, 0, 0, (bf16_t*)out, nullptr, 0, 0);
}

}  // extern "C"
