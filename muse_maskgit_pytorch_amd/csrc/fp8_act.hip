// Activation producers of the fp8 engine (gemm_fp8.hip): the two LayerNorms that feed Linear layers write their rows as OCP e4m3 with one scale per row
// (scale = max |y| / 448, y / scale rounded to nearest even: the rule of mm_quantize_e4m3_rows / mm_quantize_act_e4m3) instead of bf16 -- the quantisation
// costs no pass of its own.  Same statistics and affine arithmetic, in the same order, as layernorm_kernel / geglu_ln_kernel (norm_act.hip): the fp32 values
// that get quantised here are the ones the bf16 engine rounds to bf16.  One wave per row.
#include "common.h"
#include "muse_hip_internal.h"

namespace {

__device__ __forceinline__ int pack4_e4m3(float a, float b, float c, float d) {
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
}

// F.layer_norm(x, (D,), gamma, beta), eps 1e-5 (mmp.py:63-70) -> e4m3 [rows][ldq] + scale[rows].  ADD: rows >= add_from first get `addvec` added in place
// (the null pass's constant cross-attention output, model.hip).  D % 4 == 0; columns D..ldq-1 are not written (the GEMM's K is D: D % 128 == 0).
template <int NIT, bool ADD>
__global__ __launch_bounds__(256) void layernorm_q8_kernel(const float* __restrict__ x, long ldx, int rows, int D, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, unsigned char* __restrict__ q8, long ldq, float* __restrict__ qs,
                                                           const float* __restrict__ addvec, int add_from, float* xw) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * ldx;
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
                    *reinterpret_cast<float4*>(xw + (long)row * ldx + c * 4) = v[it];
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
    float amax = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nvec) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c * 4);
            float4 bt = make_float4(0.f, 0.f, 0.f, 0.f);
            if (beta) bt = *reinterpret_cast<const float4*>(beta + c * 4);
            v[it].x = (v[it].x - mean) * rstd * g.x + bt.x; v[it].y = (v[it].y - mean) * rstd * g.y + bt.y;
            v[it].z = (v[it].z - mean) * rstd * g.z + bt.z; v[it].w = (v[it].w - mean) * rstd * g.w + bt.w;
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[it].x), fabsf(v[it].y))), fmaxf(fabsf(v[it].z), fabsf(v[it].w)));
        }
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / 448.f : 1.f;
    if (lane == 0) qs[row] = sc;
    unsigned char* qr = q8 + (long)row * ldq;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nvec) *reinterpret_cast<int*>(qr + c * 4) = pack4_e4m3(v[it].x / sc, v[it].y / sc, v[it].z / sc, v[it].w / sc);
    }
}

// LayerNorm(inner) (mmp.py:86-87) of the activated GEGLU output a = gate * gelu(x), bf16 [rows][lda] with F valid columns -> e4m3 [rows][Fp] (columns
// F..Fp-1 zero: the following GEMM contracts over the padded width) + scale[rows].  gamma / beta padded to Fp floats.
template <int NIT>
__global__ __launch_bounds__(256) void ln_inner_q8_kernel(const bf16_t* __restrict__ a_in, long lda, int rows, int F, int Fp, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, unsigned char* __restrict__ q8, long ldq, float* __restrict__ qs) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* ar = a_in + (long)row * lda;
    const int nch = Fp >> 3;
    float a[NIT][8];
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nch) {
            float xv[8];
            unpack8(*reinterpret_cast<const uint4*>(ar + c * 8), xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float val = (c * 8 + j < F) ? xv[j] : 0.f;
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
    float amax = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nch) {
            const float4 g0 = *reinterpret_cast<const float4*>(gamma + c * 8), g1 = *reinterpret_cast<const float4*>(gamma + c * 8 + 4);
            float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
            if (beta) { b0 = *reinterpret_cast<const float4*>(beta + c * 8); b1 = *reinterpret_cast<const float4*>(beta + c * 8 + 4); }
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float o = (c * 8 + j < F) ? (a[it][j] - mean) * rstd * gg[j] + bb[j] : 0.f;
                a[it][j] = o;
                amax = fmaxf(amax, fabsf(o));
            }
        }
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / 448.f : 1.f;
    if (lane == 0) qs[row] = sc;
    unsigned char* qr = q8 + (long)row * ldq;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        if (c < nch)
            *reinterpret_cast<int2*>(qr + c * 8) = make_int2(pack4_e4m3(a[it][0] / sc, a[it][1] / sc, a[it][2] / sc, a[it][3] / sc),
                                                             pack4_e4m3(a[it][4] / sc, a[it][5] / sc, a[it][6] / sc, a[it][7] / sc));
    }
}

}  // namespace

int k_layernorm_q8(hipStream_t s, float* x, long ldx, int rows, int D, const float* gamma, const float* beta, const float* addvec, int add_from,
                   unsigned char* q8, long ldq, float* qs) {
    if (rows <= 0) return MM_OK;
    if (D % 4 || D > 64 * 4 * 8) return mm_set_error(MM_ERR_SHAPE, "layernorm_q8: dim must be a multiple of 4, <= 2048");
    if ((ldx % 4) || (ldq % 4) || ldq < D) return mm_set_error(MM_ERR_ALIGN, "layernorm_q8: strides");
    const int nit = (D / 4 + 63) / 64;
    const dim3 grid((rows + 3) / 4), block(256);
#define LNQ(NIT_)                                                                                                                                   \
    if (addvec) hipLaunchKernelGGL((layernorm_q8_kernel<NIT_, true>), grid, block, 0, s, x, ldx, rows, D, gamma, beta, q8, ldq, qs, addvec, add_from, x); \
    else hipLaunchKernelGGL((layernorm_q8_kernel<NIT_, false>), grid, block, 0, s, x, ldx, rows, D, gamma, beta, q8, ldq, qs, nullptr, 0, nullptr)
    if (nit <= 2) { LNQ(2); } else if (nit <= 4) { LNQ(4); } else { LNQ(8); }
#undef LNQ
    return mm_check_launch("layernorm_q8_kernel");
}

int k_ln_inner_q8(hipStream_t s, const bf16_t* a, long lda, int rows, int F, int Fp, const float* gamma, const float* beta, unsigned char* q8, long ldq, float* qs) {
    if (rows <= 0) return MM_OK;
    if (Fp % 8 || Fp < F || Fp > 64 * 8 * 12) return mm_set_error(MM_ERR_SHAPE, "ln_inner_q8: padded width must be a multiple of 8, >= F and <= 6144");
    if ((lda % 8) || (ldq % 8) || ldq < Fp) return mm_set_error(MM_ERR_ALIGN, "ln_inner_q8: strides");
    const int nit = (Fp / 8 + 63) / 64;
    const dim3 grid((rows + 3) / 4), block(256);
    if (nit <= 3) hipLaunchKernelGGL((ln_inner_q8_kernel<3>), grid, block, 0, s, a, lda, rows, F, Fp, gamma, beta, q8, ldq, qs);
    else if (nit <= 6) hipLaunchKernelGGL((ln_inner_q8_kernel<6>), grid, block, 0, s, a, lda, rows, F, Fp, gamma, beta, q8, ldq, qs);
    else hipLaunchKernelGGL((ln_inner_q8_kernel<12>), grid, block, 0, s, a, lda, rows, F, Fp, gamma, beta, q8, ldq, qs);
    return mm_check_launch("ln_inner_q8_kernel");
}
