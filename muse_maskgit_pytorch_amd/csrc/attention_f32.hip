// fp32 attention on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate) for the 'bf16x3'
// precision tier and the fp32 verification engine: null key / value, l2norm * learned scale, key mask, softmax, P.V
// (muse_maskgit_pytorch.py:137-162, attend.py:109-140).  Both operands of QK^T and of PV are activations, so the split-bf16
// trick of the tier's GEMMs would need six products plus the splitting VALU work per tile; the fp32 MFMA does the same job at
// 1/16 of the bf16 rate with no conversion at all, and attention is 7 % of a pass's flops.
//
// Templated on dim_head (32 / 64 / 128) and on the operand type: fp32 q / k / v (the precision tier, the fp32 engine) or bf16 q / k / v with a bf16
// result -- the route the bf16 engine takes for dim_head != 64, which its tuned bf16 attention kernels (attention.hip) do not cover
// (muse_maskgit_pytorch.py:165-174 accepts any dim_head).
//
// One 256-thread workgroup = 64 queries of one (sequence, head); wave w owns queries 16w .. 16w+15.  K and V are staged per
// 64-key tile as fp32 rows in LDS (K normalised and scaled while staged), online softmax across tiles, the null key / value
// initialises the softmax state (m = s_null, l = 1, O = v_null) so every tile holds real keys only.
//   S^T tile:  D[key][query] += A[key][d] * B[d][query], 16 steps of 4 d each; the Q fragment lives in 16 registers.
//   A lane then holds S[key = 16 kb + 4 (lane >> 4) + r][query = lane & 15] in register r of block kb -- which is exactly the B
//   operand layout of a 16x16x4 step contracting over the keys {16 kb + 4 g + r : g = 0..3}, so exp(S - max) feeds
//   O^T[d][query] += V^T[d][key] * P[key][query] straight from the accumulator registers: no P buffer, no transposes.
#include <float.h>

#include "common.h"
#include "muse_hip_internal.h"

namespace {

constexpr int AQ = 64, AKT = 64;      // queries per workgroup, keys per tile (LDS row stride: DH + 4 floats, 16-byte aligned rows)

// 4 consecutive operand values as fp32 (bf16 operands: one 8-byte load)
template <bool IO16>
__device__ __forceinline__ float4 ld4(const void* base, size_t idx) {
    if constexpr (IO16) {
        const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(base) + idx);
        return make_float4(bf16lo(w.x), bf16hi(w.x), bf16lo(w.y), bf16hi(w.y));
    } else {
        return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
    }
}

__device__ __forceinline__ f32x4_t mfma4(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int DH, bool IO16>
__global__ __launch_bounds__(256) void attention_f32_mfma_kernel(const AttnF32Args p) {
    constexpr int ALD = DH + 4, NJ = DH / 16, NE = DH / 4;      // LDS row stride, 16-dim blocks, Q-fragment registers
    __shared__ __attribute__((aligned(16))) float Ks[AKT * ALD];
    __shared__ __attribute__((aligned(16))) float Vs[AKT * ALD];
    __shared__ unsigned char valid[AKT];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int kvb = p.kv_batch_mod > 0 ? b % p.kv_batch_mod : b;
    const int qg = blockIdx.x * AQ + wid * 16 + fr;          // this lane's query
    const bool qok = qg < p.nq;

    // ---- Q fragment: lane (fr, fg) holds dims 16 j + 4 fg + i of query fr in qf[4 j + i]
    float qf[NE];
    {
        const size_t qo = (size_t)b * p.q_sb + (size_t)h * p.q_sh + (size_t)(qok ? qg : 0) * p.q_sn;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qok) v = ld4<IO16>(p.q, qo + 16 * j + 4 * fg);
            qf[4 * j] = v.x; qf[4 * j + 1] = v.y; qf[4 * j + 2] = v.z; qf[4 * j + 3] = v.w;
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        if (p.normalize) {      // F.normalize(q, dim = -1) * q_scale  (mmp.py:151-153)
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) qf[4 * j + i] = qf[4 * j + i] / den * p.q_scale[16 * j + 4 * fg + i];
        }
    }

    // ---- softmax state; the null key / value (mmp.py:145-149) is extended key 0 and is never masked (:155-157)
    float m_run = -FLT_MAX, l_run = 0.f;
    f32x4_t acc_o[NJ];
#pragma unroll
    for (int db = 0; db < NJ; ++db) acc_o[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (p.null_k) {
        float kn[NE];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) { kn[4 * j + i] = p.null_k[h * DH + 16 * j + 4 * fg + i]; ss += kn[4 * j + i] * kn[4 * j + i]; }
        if (p.normalize) {
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) kn[4 * j + i] = kn[4 * j + i] / den * p.k_scale[16 * j + 4 * fg + i];
        }
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) dot += qf[e] * kn[e];
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
        m_run = dot * p.scale;
        l_run = 1.f;
#pragma unroll
        for (int db = 0; db < NJ; ++db)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_o[db][r] = p.null_v[h * DH + db * 16 + 4 * fg + r];
    }

    for (int kt0 = 0; kt0 < p.nk; kt0 += AKT) {
        __syncthreads();          // the previous tile has been consumed
        {   // K / V rows: thread (row = t / 4, quarter = t % 4) stages DH / 4 dims of one key
            constexpr int QW = DH / 4, NV = QW / 4;      // dims per thread, float4s per thread
            const int row = t >> 2, qd = (t & 3) * QW;
            const int kr = kt0 + row;
            bool ok = kr < p.nk;
            float4 kv[NV], vv[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) { kv[j] = make_float4(0.f, 0.f, 0.f, 0.f); vv[j] = kv[j]; }
            if (ok) {
                const size_t ko = (size_t)kvb * p.k_sb + (size_t)h * p.k_sh + (size_t)kr * p.k_sn + qd;
                const size_t vo = (size_t)kvb * p.v_sb + (size_t)h * p.v_sh + (size_t)kr * p.v_sn + qd;
#pragma unroll
                for (int j = 0; j < NV; ++j) { kv[j] = ld4<IO16>(p.k, ko + 4 * j); vv[j] = ld4<IO16>(p.v, vo + 4 * j); }
                if (p.key_mask && !p.key_mask[(size_t)b * p.km_sb + kr]) ok = false;
            }
            if (p.normalize) {
                float ss = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j) ss += (kv[j].x * kv[j].x + kv[j].y * kv[j].y) + (kv[j].z * kv[j].z + kv[j].w * kv[j].w);
                ss += __shfl_xor(ss, 1, 64);
                ss += __shfl_xor(ss, 2, 64);
                const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const float4 sc = *reinterpret_cast<const float4*>(p.k_scale + qd + 4 * j);
                    kv[j].x = kv[j].x / den * sc.x; kv[j].y = kv[j].y / den * sc.y; kv[j].z = kv[j].z / den * sc.z; kv[j].w = kv[j].w / den * sc.w;
                }
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                *reinterpret_cast<float4*>(Ks + row * ALD + qd + 4 * j) = kv[j];
                *reinterpret_cast<float4*>(Vs + row * ALD + qd + 4 * j) = vv[j];
            }
            if ((t & 3) == 0) valid[row] = ok ? 1 : 0;
        }
        __syncthreads();

        // ---- S^T = K^ Q^T (raw dot products), 64 keys x 16 queries per wave
        f32x4_t acc_s[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) acc_s[kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int dcol = 16 * (e >> 2) + 4 * fg + (e & 3);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc_s[kb] = mfma4(Ks[(kb * 16 + fr) * ALD + dcol], qf[e], acc_s[kb]);
        }
        // ---- online softmax over this lane's 16 keys of its query (+ the three other lane groups of the query)
        float sv[4][4];
        float tmax = -FLT_MAX;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sv[kb][r] = valid[kb * 16 + 4 * fg + r] ? acc_s[kb][r] * p.scale : -FLT_MAX;      // attend.py:126-131: masked_fill(-finfo.max): such weights are exactly 0
                tmax = fmaxf(tmax, sv[kb][r]);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = (m_run == -FLT_MAX) ? 0.f : expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sv[kb][r] = (sv[kb][r] == -FLT_MAX) ? 0.f : expf(sv[kb][r] - m_new);
                psum += sv[kb][r];
            }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < NJ; ++db)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_o[db][r] *= alpha;
        // ---- O^T += V^T P: step (kb, r) contracts over the keys 16 kb + 4 g + r, g = lane group
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vrow = Vs + (kb * 16 + 4 * fg + r) * ALD + fr;
#pragma unroll
                for (int db = 0; db < NJ; ++db) acc_o[db] = mfma4(vrow[db * 16], sv[kb][r], acc_o[db]);
            }
    }

    if (!qok) return;
#pragma unroll
    for (int db = 0; db < NJ; ++db) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = acc_o[db][r] / l_run;
        const int d0 = db * 16 + 4 * fg;
        const size_t oo = (size_t)b * p.o_sb + (size_t)h * p.o_sh + (size_t)qg * p.o_sn + d0;
        if (p.out) {
            if constexpr (IO16) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + oo) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
            else *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + oo) = make_float4(o[0], o[1], o[2], o[3]);
        }
        if (p.out_split) store_split4(p.out_split + (size_t)b * p.os_sb + (size_t)qg * p.os_sn, p.os_seg, p.P, h * DH + d0, o);
    }
}

}  // namespace

int k_attention_f32(hipStream_t s, const AttnF32Args& a) {
    if (a.B <= 0 || a.H <= 0 || a.nq <= 0) return MM_OK;
    const int dh = a.dh ? a.dh : 64;
    if (dh != 32 && dh != 64 && dh != 128) return mm_set_error(MM_ERR_UNSUPPORTED, "attention: dim_head must be 32, 64 or 128");
    if (a.nk < 0 || (a.nk == 0 && !a.null_k)) return mm_set_error(MM_ERR_SHAPE, "attention_f32: no keys");
    const int al = a.io_bf16 ? 8 : 4;      // elements per 16 bytes of the operand type (rows are read 8 / 16 bytes at a time: 4-element alignment suffices)
    (void)al;
    if ((a.q_sn % 4) || (a.k_sn % 4) || (a.v_sn % 4) || (a.q_sh % 4) || (a.k_sh % 4) || (a.v_sh % 4) || (a.q_sb % 4) || (a.k_sb % 4) || (a.v_sb % 4) ||
        (((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v) & (a.io_bf16 ? 7 : 15)))
        return mm_set_error(MM_ERR_ALIGN, "attention_f32: q / k / v rows must be aligned to 4 elements");
    if (a.out && ((a.o_sn % 4) || (a.o_sh % 4) || (a.o_sb % 4) || (((uintptr_t)a.out) & (a.io_bf16 ? 7 : 15)))) return mm_set_error(MM_ERR_ALIGN, "attention_f32: output rows must be aligned to 4 elements");
    if (a.out_split && (a.P != 3 && a.P != 5 && a.P != 6 && a.P != (MM_SPLIT_F16_BIT | 2) && a.P != (MM_SPLIT_F16_BIT | 3) && a.P != (MM_SPLIT_NODUP_BIT | MM_SPLIT_F16_BIT | 3)))
        return mm_set_error(MM_ERR_SHAPE, "attention_f32: products must be 3, 5, 6 or MM_SPLIT_F16 | 2, 3");
    if (a.normalize && (!a.q_scale || !a.k_scale)) return mm_set_error(MM_ERR_SHAPE, "attention_f32: q_scale / k_scale required with normalize");
    if ((a.null_k == nullptr) != (a.null_v == nullptr)) return mm_set_error(MM_ERR_SHAPE, "attention_f32: null_k and null_v go together");
    // 'f16x2' tier, self-attention of the 256-token configs: fp16 term products on the fp16 matrix pipe (attention_x2.hip) instead of the 1/16-rate fp32 MFMA;
    // debug bit 32768 ("no resident-key attention kernel") keeps this kernel for A/B
    if (a.out_split && !a.out && split_is_f16(a.P) && !(g_mm_debug & 32768) && k_attention_x2_eligible(a)) return k_attention_x2(s, a);
    const dim3 grid((a.nq + AQ - 1) / AQ, a.H, a.B), block(256);
#define MM_AF(D_, B_) hipLaunchKernelGGL((attention_f32_mfma_kernel<D_, B_>), grid, block, 0, s, a)
    if (a.io_bf16) { if (dh == 32) MM_AF(32, true); else if (dh == 64) MM_AF(64, true); else MM_AF(128, true); }
    else { if (dh == 32) MM_AF(32, false); else if (dh == 64) MM_AF(64, false); else MM_AF(128, false); }
#undef MM_AF
    return mm_check_launch("attention_f32_mfma_kernel");
}
