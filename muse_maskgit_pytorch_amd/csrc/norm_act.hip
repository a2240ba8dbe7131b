// HBM-bound row kernels of the transformer (gfx950): token+position embedding, LayerNorm, GEGLU+LayerNorm,
// row-vector add.  One 64-lane wave per row, 16-byte vector accesses, fp32 statistics via wave butterflies.
#include <string.h>

#include "common.h"
#include "muse_hip_internal.h"

namespace {

// x[row][:] = token_emb[ids[row]] + pos_emb[row % n]           (muse_maskgit_pytorch.py:322-323)
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids, int rows, int n, int pos_offset,
                                                    const bf16_t* __restrict__ tok, int vocab_rows,
                                                    const bf16_t* __restrict__ pos, int D, float* __restrict__ x) {
    const int chunks = D >> 3;
    const long total = (long)rows * chunks;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int row = (int)(i / chunks), c = (int)(i - (long)row * chunks);
        long id = ids[row];
        id = id < 0 ? 0 : (id >= vocab_rows ? vocab_rows - 1 : id);
        const uint4 tv = *reinterpret_cast<const uint4*>(tok + id * D + c * 8);
        const uint4 pv = *reinterpret_cast<const uint4*>(pos + (long)(row % n + pos_offset) * D + c * 8);
        float a[8], b[8];
        unpack8(tv, a); unpack8(pv, b);
        float* xo = x + (long)row * D + c * 8;
        *reinterpret_cast<float4*>(xo) = make_float4(a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]);
        *reinterpret_cast<float4*>(xo + 4) = make_float4(a[4] + b[4], a[5] + b[5], a[6] + b[6], a[7] + b[7]);
    }
}

// F.layer_norm(x, (D,), gamma, beta), eps 1e-5 (muse_maskgit_pytorch.py:63-70); fp32 in, bf16 out.
// Optional row gather (row_index) so the final norm only touches the rows that are sampled.
constexpr int LN_MAX_IT = 8;   // D <= 64 lanes * 4 * 8 = 2048
// ADD: rows >= add_from first get the row vector `addvec` added IN PLACE (x is then also an output) -- the null pass's constant
// cross-attention output (model.hip) rides on the LayerNorm that follows it instead of costing its own pass over the stream.
template <int NIT, bool ADD = false>             // float4 iterations per lane: instantiated for 2 / 4 / 8 so small dims keep registers (and occupancy)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long ldx, int rows, int D,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const int32_t* __restrict__ row_index, bf16_t* __restrict__ out, long ldo,
                                                        const float* __restrict__ addvec = nullptr, int add_from = 0, float* xw = nullptr) {
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
    bf16_t* orow = out + (long)row * ldo;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nvec) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c * 4);
            float4 bt = make_float4(0.f, 0.f, 0.f, 0.f);
            if (beta) bt = *reinterpret_cast<const float4*>(beta + c * 4);
            const float o0 = (v[it].x - mean) * rstd * g.x + bt.x, o1 = (v[it].y - mean) * rstd * g.y + bt.y;
            const float o2 = (v[it].z - mean) * rstd * g.z + bt.z, o3 = (v[it].w - mean) * rstd * g.w + bt.w;
            *reinterpret_cast<uint2*>(orow + c * 4) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
        }
    }
}

// The decode loop's step tail in one pass over the sampled rows (round 4): final LayerNorm of the conditional and of the null pass (the arithmetic of
// layernorm_kernel, each result rounded to bf16 as that kernel stores it), the guidance mix in the embedding e = e_null + (e_cond - e_null) * s rounded to
// bf16 once (the arithmetic of cfg_mix_kernel), and the row's mean of the logits <e, mean_w> for the fused sampler's bound -- the values the three kernels
// (+ fused_combine_kernel) produced with four launches and three round trips of the rows through HBM.
template <int NIT>
__global__ __launch_bounds__(256) void final_mix_kernel(const float* __restrict__ xc, const float* __restrict__ xn, long ldx, int rows, int D,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, const int32_t* __restrict__ row_index,
                                                        float s, bf16_t* __restrict__ out, const float* __restrict__ wmean, float* __restrict__ mu) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long src = row_index ? (long)row_index[row] : (long)row;
    const int nvec = D >> 2;
    float4 e[2][NIT];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float* xr = (h == 0 ? xc : xn) + src * ldx;
        float4 v[NIT];
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane;
            if (c < nvec) {
                v[it] = *reinterpret_cast<const float4*>(xr + c * 4);
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
                const float o0 = (v[it].x - mean) * rstd * g.x + bt.x, o1 = (v[it].y - mean) * rstd * g.y + bt.y;
                const float o2 = (v[it].z - mean) * rstd * g.z + bt.z, o3 = (v[it].w - mean) * rstd * g.w + bt.w;
                const uint32_t w0 = pack_bf16x2(o0, o1), w1 = pack_bf16x2(o2, o3);      // the bf16 values layernorm_kernel stores
                e[h][it] = make_float4(bf16lo(w0), bf16hi(w0), bf16lo(w1), bf16hi(w1));
            }
        }
    }
    float m = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nvec) {
            const float4 cv = e[0][it], nv = e[1][it];
            const uint32_t w0 = pack_bf16x2(nv.x + (cv.x - nv.x) * s, nv.y + (cv.y - nv.y) * s), w1 = pack_bf16x2(nv.z + (cv.z - nv.z) * s, nv.w + (cv.w - nv.w) * s);
            *reinterpret_cast<uint2*>(out + (long)row * D + c * 4) = make_uint2(w0, w1);
            if (wmean) {
                const float4 wm = *reinterpret_cast<const float4*>(wmean + c * 4);
                m += (bf16lo(w0) * wm.x + bf16hi(w0) * wm.y) + (bf16lo(w1) * wm.z + bf16hi(w1) * wm.w);
            }
        }
    }
    if (mu) {
        m = wave_sum(m);
        if (lane == 0) mu[row] = m;
    }
}

// several row gathers with one row list in one launch (the last layer's compaction: residual stream and attention output of both guidance halves)
struct GatherJobs { const unsigned char* src[4]; unsigned char* dst[4]; long pitch[4]; int row_add[4]; int chunks[4]; int n; };
__global__ __launch_bounds__(256) void gather_rows16_multi_kernel(const GatherJobs j, const int32_t* __restrict__ rows, int R) {
    const int job = blockIdx.y;
    const int chunks = j.chunks[job];
    const long total = (long)R * chunks;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / chunks), c = (int)(i - (long)r * chunks);
        *reinterpret_cast<uint4*>(j.dst[job] + ((long)r * chunks + c) * 16) =
            *reinterpret_cast<const uint4*>(j.src[job] + (long)(rows[r] + j.row_add[job]) * j.pitch[job] + (long)c * 16);
    }
}

// GEGLU + LayerNorm(inner)  (muse_maskgit_pytorch.py:72-77, 86-87): h = [x | gate] (each Fp wide, F valid),
// a = gate * gelu_erf(x); out = LN(a) over the F valid columns; columns F..Fp-1 are written as zeros so the
// following GEMM can run on the padded K.
constexpr int GG_MAX_IT = 12;  // Fp <= 64 * 8 * 12 = 6144
template <int NIT, bool FUSED_IN>   // 16-byte iterations per lane: 3 (F <= 1536) / 6 / 12; FUSED_IN: h already holds gate*gelu(x) [rows][Fp]
__global__ __launch_bounds__(256) void geglu_ln_kernel(const bf16_t* __restrict__ h, long ldh, int rows, int F, int Fp,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       bf16_t* __restrict__ out, long ldo) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* hr = h + (long)row * ldh;
    const int nch = Fp >> 3;
    float a[NIT][8];
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nch) {
            float xv[8], gv[8];
            unpack8(*reinterpret_cast<const uint4*>(hr + c * 8), xv);
            if (!FUSED_IN) unpack8(*reinterpret_cast<const uint4*>(hr + Fp + c * 8), gv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float act = FUSED_IN ? xv[j] : geglu_f(xv[j], gv[j]);
                const float val = (c * 8 + j < F) ? act : 0.f;
                a[it][j] = val;
                sum += val;
            }
        }
    }
    const float mean = wave_sum(sum) / (float)F;
    float sq = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nch) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (c * 8 + j < F) { const float d = a[it][j] - mean; sq += d * d; }
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(sq) / (float)F + 1e-5f);
    bf16_t* orow = out + (long)row * ldo;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nch) {
            // gamma / beta are padded to Fp floats by the caller: two 16-byte loads instead of 16 scalar ones (the scalar
            // form made this kernel issue-bound at ~1 TB/s)
            const float4 g0 = *reinterpret_cast<const float4*>(gamma + c * 8), g1 = *reinterpret_cast<const float4*>(gamma + c * 8 + 4);
            float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
            if (beta) { b0 = *reinterpret_cast<const float4*>(beta + c * 8); b1 = *reinterpret_cast<const float4*>(beta + c * 8 + 4); }
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = c * 8 + j;
                o[j] = (col < F) ? (a[it][j] - mean) * rstd * gg[j] + bb[j] : 0.f;
            }
            *reinterpret_cast<uint4*>(orow + c * 8) = pack8(o);
        }
    }
}

// dst[r][0..bytes) = src[rows[r] + row_add][0..bytes) for 16-byte aligned rows (the last layer's compaction to the sampled rows)
__global__ __launch_bounds__(256) void gather_rows16_kernel(const unsigned char* __restrict__ src, long src_pitch, const int32_t* __restrict__ rows,
                                                            int R, int row_add, int chunks, unsigned char* __restrict__ dst) {
    const long total = (long)R * chunks;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / chunks), c = (int)(i - (long)r * chunks);
        *reinterpret_cast<uint4*>(dst + ((long)r * chunks + c) * 16) =
            *reinterpret_cast<const uint4*>(src + (long)(rows[r] + row_add) * src_pitch + (long)c * 16);
    }
}

// the same gather for a row list whose length lives on the device (the fused sampler's fallback list): dst[i] = src[rows[i]] for i < min(*count, cap);
// thread 0 adds that number to *total (telemetry) when given
__global__ __launch_bounds__(256) void gather_rows16_counted_kernel(const unsigned char* __restrict__ src, long src_pitch, const int32_t* __restrict__ rows,
                                                                    const int32_t* __restrict__ count, int cap, int chunks, unsigned char* __restrict__ dst,
                                                                    int32_t* total) {
    const int n = min(*count, cap);
    if (total && n > 0 && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(total, n);
    const long items = (long)n * chunks;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / chunks), c = (int)(i - (long)r * chunks);
        *reinterpret_cast<uint4*>(dst + ((long)r * chunks + c) * 16) = *reinterpret_cast<const uint4*>(src + (long)rows[r] * src_pitch + (long)c * 16);
    }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, long count) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        out[i] = f32_to_bf16(x[i]);
}

inline int grid_for(long items, int per_block = 256, int cap = 256 * 8) {
    long b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

}  // namespace

int k_embed(hipStream_t s, const int64_t* ids, int rows, int n, int pos_offset, const bf16_t* tok, int vocab_rows,
            const bf16_t* pos, int D, float* x) {
    if (rows <= 0) return MM_OK;
    if (D % 8) return mm_set_error(MM_ERR_SHAPE, "embed: dim must be a multiple of 8");
    hipLaunchKernelGGL(embed_kernel, dim3(grid_for((long)rows * (D / 8))), dim3(256), 0, s, ids, rows, n, pos_offset, tok,
                       vocab_rows, pos, D, x);
    return mm_check_launch("embed_kernel");
}

int k_layernorm(hipStream_t s, const float* x, long ldx, int rows, int D, const float* gamma, const float* beta,
                const int32_t* row_index, bf16_t* out, long ldo) {
    if (rows <= 0) return MM_OK;
    if (D % 4 || D > 64 * 4 * LN_MAX_IT) return mm_set_error(MM_ERR_SHAPE, "layernorm: dim must be a multiple of 4 and <= 2048");
    if (ldx % 4 || ldo % 4) return mm_set_error(MM_ERR_ALIGN, "layernorm: strides must be multiples of 4 elements");
    const int nit = (D / 4 + 63) / 64;
    if (nit <= 2) hipLaunchKernelGGL((layernorm_kernel<2, false>), dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, rows, D, gamma, beta, row_index, out, ldo, nullptr, 0, nullptr);
    else if (nit <= 4) hipLaunchKernelGGL((layernorm_kernel<4, false>), dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, rows, D, gamma, beta, row_index, out, ldo, nullptr, 0, nullptr);
    else hipLaunchKernelGGL((layernorm_kernel<8, false>), dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, rows, D, gamma, beta, row_index, out, ldo, nullptr, 0, nullptr);
    return mm_check_launch("layernorm_kernel");
}

// LayerNorm(dim) fold, producer outputs of a residual stream that no GEMM epilogue has just written (operator-level entry mm_cross_attention_block): the bf16 image
// of every row and its (sum, sum of squares) per `gran` columns in the canonical order of common.h row_stats16 (16 adjacent lanes x 4 columns = 64 columns; two such
// groups combined as (first + second) for 128-column partials) -- the bits a fp32-residual GEMM epilogue would have left.
__global__ __launch_bounds__(256) void fold_image_kernel(const float* __restrict__ x, long ldx, int rows, int D, int gran, bf16_t* __restrict__ xb, long ldxb, float* __restrict__ stp, int np) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    for (int c0 = 0; c0 < D; c0 += 256) {      // a wave covers 256 columns per sweep: 4 groups of 16 lanes x 4 columns
        const int col = c0 + lane * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < D) v = *reinterpret_cast<const float4*>(x + (size_t)row * ldx + col);
        if (col < D) *reinterpret_cast<uint2*>(xb + (size_t)row * ldxb + col) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        float2 st = row_stats16((v.x + v.y) + (v.z + v.w), (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
        if (gran == 128) {      // (first 64 columns + second 64 columns)
            st.x += __shfl_xor(st.x, 16, 64); st.y += __shfl_xor(st.y, 16, 64);
            if ((lane & 31) == 0 && col < D) *reinterpret_cast<float2*>(stp + ((size_t)row * np + col / 128) * 2) = st;
        } else if ((lane & 15) == 0 && col < D) {
            *reinterpret_cast<float2*>(stp + ((size_t)row * np + col / 64) * 2) = st;
        }
    }
}
int k_fold_image(hipStream_t s, const float* x, long ldx, int rows, int D, bf16_t* xb, long ldxb, float* stp, int np) {
    if (rows <= 0) return MM_OK;
    if ((D % 64) || (ldx % 4) || (ldxb % 4)) return mm_set_error(MM_ERR_SHAPE, "fold_image: dim must be a multiple of 64, rows 16-byte aligned");
    const int gran = D <= 512 ? 64 : 128;
    if (np != (D + gran - 1) / gran) return mm_set_error(MM_ERR_SHAPE, "fold_image: partial count");
    hipLaunchKernelGGL(fold_image_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, rows, D, gran, xb, ldxb, stp, np);
    return mm_check_launch("fold_image_kernel");
}

// LayerNorm(dim) fold, safety probe (mm_transformer_desc.ln_probe): max over rows of |mean| * rstd of the rows a fold consumer is about to read, from the
// statistics partials their producer left (common.h ln_rstd_negmean: the value the consumer itself computes).  The fold multiplies the bf16 image of the RAW
// row, whose rounding error in normalised units grows like |x^ + mean / sigma| instead of |x^|: harmless while the row mean is small against its spread
// (random init: ~0.05), degrading for a checkpoint whose residual stream carries a DC offset.  One thread per row, atomicMax on the float's bits (>= 0).
__global__ __launch_bounds__(256) void ln_fold_ratio_kernel(const float* __restrict__ stp, int rows, int np, float inv_F, float* __restrict__ out) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    float r = 0.f;
    if (row < rows) {
        const float2 st = ln_rstd_negmean(reinterpret_cast<const float2*>(stp) + (size_t)row * np, np, inv_F);
        r = fabsf(st.y) * st.x;
    }
    r = wave_max(r);
    if ((threadIdx.x & 63) == 0 && r > 0.f) atomicMax(reinterpret_cast<int*>(out), __float_as_int(r));
}
int k_ln_fold_ratio(hipStream_t s, const float* stp, int rows, int np, int F, float* out) {
    if (rows <= 0) return MM_OK;
    hipLaunchKernelGGL(ln_fold_ratio_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, stp, rows, np, 1.f / (float)F, out);
    return mm_check_launch("ln_fold_ratio_kernel");
}

// x[add_from..rows) += addvec (in place), then out = LayerNorm(x) for all rows
int k_layernorm_addvec(hipStream_t s, float* x, long ldx, int rows, int D, const float* gamma, const float* beta, const float* addvec,
                       int add_from, bf16_t* out, long ldo) {
    if (rows <= 0) return MM_OK;
    if (D % 4 || D > 64 * 4 * LN_MAX_IT) return mm_set_error(MM_ERR_SHAPE, "layernorm: dim must be a multiple of 4 and <= 2048");
    if (ldx % 4 || ldo % 4) return mm_set_error(MM_ERR_ALIGN, "layernorm: strides must be multiples of 4 elements");
    const int nit = (D / 4 + 63) / 64;
    const dim3 grid((rows + 3) / 4), block(256);
    if (nit <= 2) hipLaunchKernelGGL((layernorm_kernel<2, true>), grid, block, 0, s, x, ldx, rows, D, gamma, beta, nullptr, out, ldo, addvec, add_from, x);
    else if (nit <= 4) hipLaunchKernelGGL((layernorm_kernel<4, true>), grid, block, 0, s, x, ldx, rows, D, gamma, beta, nullptr, out, ldo, addvec, add_from, x);
    else hipLaunchKernelGGL((layernorm_kernel<8, true>), grid, block, 0, s, x, ldx, rows, D, gamma, beta, nullptr, out, ldo, addvec, add_from, x);
    return mm_check_launch("layernorm_kernel");
}

int k_geglu_ln(hipStream_t s, const bf16_t* h, long ldh, int rows, int F, int Fp, const float* gamma, const float* beta,
               bf16_t* out, long ldo) {
    if (rows <= 0) return MM_OK;
    if (Fp % 8 || Fp < F || Fp > 64 * 8 * GG_MAX_IT) return mm_set_error(MM_ERR_SHAPE, "geglu_ln: padded inner dim must be a multiple of 8, >= F and <= 6144");
    if (ldh % 8 || ldo % 8) return mm_set_error(MM_ERR_ALIGN, "geglu_ln: strides must be multiples of 8 elements");
    const int nit = (Fp / 8 + 63) / 64;
    if (nit <= 3) hipLaunchKernelGGL((geglu_ln_kernel<3, false>), dim3((rows + 3) / 4), dim3(256), 0, s, h, ldh, rows, F, Fp, gamma, beta, out, ldo);
    else if (nit <= 6) hipLaunchKernelGGL((geglu_ln_kernel<6, false>), dim3((rows + 3) / 4), dim3(256), 0, s, h, ldh, rows, F, Fp, gamma, beta, out, ldo);
    else hipLaunchKernelGGL((geglu_ln_kernel<12, false>), dim3((rows + 3) / 4), dim3(256), 0, s, h, ldh, rows, F, Fp, gamma, beta, out, ldo);
    return mm_check_launch("geglu_ln_kernel");
}

// LayerNorm(inner) on the already-activated GEGLU output a [rows][Fp] bf16 (columns >= F are zero / ignored)
int k_ln_bf16(hipStream_t s, const bf16_t* a, long lda, int rows, int F, int Fp, const float* gamma, const float* beta,
              bf16_t* out, long ldo) {
    if (rows <= 0) return MM_OK;
    if (Fp % 8 || Fp < F || Fp > 64 * 8 * GG_MAX_IT) return mm_set_error(MM_ERR_SHAPE, "ln_bf16: padded width must be a multiple of 8, >= F and <= 6144");
    if (lda % 8 || ldo % 8) return mm_set_error(MM_ERR_ALIGN, "ln_bf16: strides must be multiples of 8 elements");
    const int nit = (Fp / 8 + 63) / 64;
    if (nit <= 3) hipLaunchKernelGGL((geglu_ln_kernel<3, true>), dim3((rows + 3) / 4), dim3(256), 0, s, a, lda, rows, F, Fp, gamma, beta, out, ldo);
    else if (nit <= 6) hipLaunchKernelGGL((geglu_ln_kernel<6, true>), dim3((rows + 3) / 4), dim3(256), 0, s, a, lda, rows, F, Fp, gamma, beta, out, ldo);
    else hipLaunchKernelGGL((geglu_ln_kernel<12, true>), dim3((rows + 3) / 4), dim3(256), 0, s, a, lda, rows, F, Fp, gamma, beta, out, ldo);
    return mm_check_launch("ln_bf16_kernel");
}

int k_gather_rows16(hipStream_t s, const void* src, long src_pitch_bytes, const int32_t* rows, int R, int row_add, int row_bytes, void* dst) {
    if (R <= 0) return MM_OK;
    if ((row_bytes % 16) || (src_pitch_bytes % 16)) return mm_set_error(MM_ERR_ALIGN, "gather_rows: rows must be multiples of 16 bytes");
    hipLaunchKernelGGL(gather_rows16_kernel, dim3(grid_for((long)R * (row_bytes / 16))), dim3(256), 0, s, (const unsigned char*)src,
                       src_pitch_bytes, rows, R, row_add, row_bytes / 16, (unsigned char*)dst);
    return mm_check_launch("gather_rows16_kernel");
}

int k_final_mix(hipStream_t s, const float* xc, const float* xn, long ldx, int rows, int D, const float* gamma, const float* beta, const int32_t* row_index,
                float cond_scale, bf16_t* out, const float* wmean, float* mu) {
    if (rows <= 0) return MM_OK;
    if (D % 4 || D > 64 * 4 * LN_MAX_IT || (ldx % 4)) return mm_set_error(MM_ERR_SHAPE, "final_mix: dim must be a multiple of 4 and <= 2048");
    const int nit = (D / 4 + 63) / 64;
    const dim3 grid((rows + 3) / 4), block(256);
    if (nit <= 2) hipLaunchKernelGGL((final_mix_kernel<2>), grid, block, 0, s, xc, xn, ldx, rows, D, gamma, beta, row_index, cond_scale, out, wmean, mu);
    else if (nit <= 4) hipLaunchKernelGGL((final_mix_kernel<4>), grid, block, 0, s, xc, xn, ldx, rows, D, gamma, beta, row_index, cond_scale, out, wmean, mu);
    else hipLaunchKernelGGL((final_mix_kernel<8>), grid, block, 0, s, xc, xn, ldx, rows, D, gamma, beta, row_index, cond_scale, out, wmean, mu);
    return mm_check_launch("final_mix_kernel");
}

// up to four gathers with the same row list in one launch: dst_j[r] = src_j[rows[r] + row_add_j] (16-byte aligned rows)
int k_gather_rows16_multi(hipStream_t s, int njobs, const void* const* src, const long* src_pitch_bytes, const int* row_add, const int* row_bytes, void* const* dst,
                          const int32_t* rows, int R) {
    if (R <= 0 || njobs <= 0) return MM_OK;
    if (njobs > 4) return mm_set_error(MM_ERR_SHAPE, "gather_rows_multi: at most 4 jobs");
    GatherJobs j;
    memset(&j, 0, sizeof(j));
    int maxc = 0;
    for (int i = 0; i < njobs; ++i) {
        if ((row_bytes[i] % 16) || (src_pitch_bytes[i] % 16)) return mm_set_error(MM_ERR_ALIGN, "gather_rows: rows must be multiples of 16 bytes");
        j.src[i] = (const unsigned char*)src[i]; j.dst[i] = (unsigned char*)dst[i]; j.pitch[i] = src_pitch_bytes[i]; j.row_add[i] = row_add[i];
        j.chunks[i] = row_bytes[i] / 16;
        if (j.chunks[i] > maxc) maxc = j.chunks[i];
    }
    j.n = njobs;
    hipLaunchKernelGGL(gather_rows16_multi_kernel, dim3(grid_for((long)R * maxc), njobs), dim3(256), 0, s, j, rows, R);
    return mm_check_launch("gather_rows16_multi_kernel");
}

int k_gather_rows16_counted(hipStream_t s, const void* src, long src_pitch_bytes, const int32_t* rows, const int32_t* count, int cap, int row_bytes, void* dst,
                            int32_t* total) {
    if (cap <= 0) return MM_OK;
    if ((row_bytes % 16) || (src_pitch_bytes % 16)) return mm_set_error(MM_ERR_ALIGN, "gather_rows: rows must be multiples of 16 bytes");
    hipLaunchKernelGGL(gather_rows16_counted_kernel, dim3(grid_for((long)cap * (row_bytes / 16))), dim3(256), 0, s, (const unsigned char*)src, src_pitch_bytes,
                       rows, count, cap, row_bytes / 16, (unsigned char*)dst, total);
    return mm_check_launch("gather_rows16_counted_kernel");
}

// per-row symmetric quantisation to OCP fp8 e4m3: scale = max|w| / 448, wq = rne(w / scale); columns K..Kp-1 are zero
__global__ __launch_bounds__(256) void quantize_e4m3_rows_kernel(const float* __restrict__ w, long ldw, int rows, int K, int Kp,
                                                                 unsigned char* __restrict__ wq, float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* wr = w + (long)row * ldw;
    float m = 0.f;
    for (int c = lane; c < K; c += 64) m = fmaxf(m, fabsf(wr[c]));
    m = wave_max(m);
    const float sc = m > 0.f ? m / 448.f : 1.f;
    if (lane == 0) scale[row] = sc;
    for (int c = lane * 4; c < Kp; c += 256) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (c + j < K) ? wr[c + j] / sc : 0.f;      // IEEE division: the same value torch's w / scale rounds
        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
        *reinterpret_cast<int*>(wq + (long)row * Kp + c) = pk;
    }
}

// activation rows (bf16 or fp32) -> e4m3 + per-row scale for the fp8 engine (gemm_fp8.hip): same rule as the weights (scale = max |x| / 448, IEEE division,
// round to nearest even), one wave per row, 8 values per lane and pass
template <bool F32>
__global__ __launch_bounds__(256) void quantize_act_e4m3_kernel(const void* __restrict__ xin, long ldx, int rows, int K, int Kp, unsigned char* __restrict__ xq,
                                                                float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float m = 0.f;
    for (int c = lane * 8; c < K; c += 512) {
        float v[8];
        if constexpr (F32) {
            const float* xr = reinterpret_cast<const float*>(xin) + (long)row * ldx + c;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (c + j < K) ? xr[j] : 0.f;
        } else {
            const bf16_t* xr = reinterpret_cast<const bf16_t*>(xin) + (long)row * ldx + c;
            if (c + 8 <= K) unpack8(*reinterpret_cast<const uint4*>(xr), v);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (c + j < K) ? bf16_to_f32(xr[j]) : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));
    }
    m = wave_max(m);
    const float sc = m > 0.f ? m / 448.f : 1.f;
    if (lane == 0) scale[row] = sc;
    for (int c = lane * 8; c < Kp; c += 512) {
        float v[8];
        if constexpr (F32) {
            const float* xr = reinterpret_cast<const float*>(xin) + (long)row * ldx + c;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (c + j < K) ? xr[j] / sc : 0.f;
        } else {
            const bf16_t* xr = reinterpret_cast<const bf16_t*>(xin) + (long)row * ldx + c;
            if (c + 8 <= K) {
                unpack8(*reinterpret_cast<const uint4*>(xr), v);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = v[j] / sc;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (c + j < K) ? bf16_to_f32(xr[j]) / sc : 0.f;
            }
        }
        int p0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        p0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], p0, true);
        int p1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
        p1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], p1, true);
        *reinterpret_cast<int2*>(xq + (long)row * Kp + c) = make_int2(p0, p1);
    }
}

int k_quantize_act_e4m3(hipStream_t s, const void* x, int x_f32, long ldx, int rows, int K, int Kp, unsigned char* xq, float* scale) {
    if (rows <= 0) return MM_OK;
    if (Kp % 8 || Kp < K) return mm_set_error(MM_ERR_SHAPE, "quantize_act_e4m3: padded width must be a multiple of 8 and >= K");
    if (!x_f32 && ((ldx % 8) || (((uintptr_t)x) & 15))) return mm_set_error(MM_ERR_ALIGN, "quantize_act_e4m3: bf16 rows must be 16-byte aligned");
    if (x_f32) hipLaunchKernelGGL(quantize_act_e4m3_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, rows, K, Kp, xq, scale);
    else hipLaunchKernelGGL(quantize_act_e4m3_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, rows, K, Kp, xq, scale);
    return mm_check_launch("quantize_act_e4m3_kernel");
}

int k_quantize_e4m3_rows(hipStream_t s, const float* w, long ldw, int rows, int K, int Kp, unsigned char* wq, float* scale) {
    if (rows <= 0) return MM_OK;
    if (Kp % 4 || Kp < K) return mm_set_error(MM_ERR_SHAPE, "quantize_e4m3_rows: padded width must be a multiple of 4 and >= K");
    hipLaunchKernelGGL(quantize_e4m3_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, w, ldw, rows, K, Kp, wq, scale);
    return mm_check_launch("quantize_e4m3_rows_kernel");
}

int k_f32_to_bf16(hipStream_t s, const float* x, bf16_t* out, long count) {
    if (count <= 0) return MM_OK;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(count)), dim3(256), 0, s, x, out, count);
    return mm_check_launch("f32_to_bf16_kernel");
}
