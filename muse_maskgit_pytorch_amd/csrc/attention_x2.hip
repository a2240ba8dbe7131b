// Self-attention of the 'f16x2' precision tier as fp16 TERM PRODUCTS on the fp16 matrix pipe (round 5) -- muse_maskgit_pytorch.py:137-162, attend.py:109-140.
//
// Rounds 3-4 ran the tier's attention on v_mfma_f32_16x16x4_f32 (attention_f32.hip): exact fp32 products, but at 1/16 of the fp16 rate -- 21.7 ms of the tier's
// 155 ms per generate (profiles/r05_f16x2_fp32w_kstats.txt).  Both operands of QK^T and of PV are activations, so the GEMMs' weight-side shortcut does not apply;
// but the SAME two-term split does: x = xh + xl with xh = fp16(x), xl = fp16(x - xh) carries 22 significand bits, and
//     a . b ~ ah . bh + al . bh + ah . bl        (the dropped al . bl term is below 2^-22 relative)
// is three v_mfma_f32_16x16x32_f16 per block instead of one -- 3/16 of the fp32 MFMA's time per product.  Applied to S^T = K^ Q^T (q^, k^ = l2-normalised x learned
// scale, |.| <= ~1: no range issue) and to O = P V (P = exp(s - max) in (0, 1]; V = a projection output of O(1)).  Softmax, the null key / value and the
// normalisation stay fp32.
//
// Structure = attention_full_kernel (attention.hip): ONE 512-thread workgroup per (sequence, head) takes all (up to 256) queries, all keys resident in LDS:
//   K^ as two fp16 images (high / low terms) of 128-byte rows, chunk XOR (row & 7): conflict-free ds_read_b128;
//   V  as two fp16 images [4 d-blocks][256 keys][16 d] (32-byte rows) read with ds_read_b64_tr_b16 -- the transposing read only cares about 16-bit elements;
//   scores of all 16 key blocks in registers, two packed 16-key accumulator blocks ARE an A operand of the P V MFMA (as two fp16 terms each).
// q / k / v arrive as fp32 (the tier's projection outputs) and are split in registers; the output leaves as the P term segments [h | l | h] the output projection's
// GEMM multiplies (common.h store_split4's layout).  LDS: 4 x 32 KiB + 8 x 1 KiB output staging = 136.5 KiB -> one workgroup per CU.
// Shape class: dim_head 64, nk in {128, 192, 256} (= the self-attention of the 256-token configs), no key mask, fp16-term output; everything else stays on
// attention_f32.hip (k_attention_f32 dispatches).
#include "common.h"
#include "muse_hip_internal.h"

namespace {

constexpr int X2_NK = 256, X2_DH = 64;
constexpr int X2_IMG = X2_NK * 128;                              // one 32 KiB image
constexpr int X2_SMEM = 4 * X2_IMG + 8 * 1024 + 512;
constexpr float X2_NEG_BIG = -3.0e38f;

__device__ __forceinline__ int x2_sw(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

typedef short x2_v4i16_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 x2_read_tr16(const unsigned char* ptr) {
    typedef __attribute__((address_space(3))) x2_v4i16_t* lds_v4_t;
    const x2_v4i16_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)ptr);
    return __builtin_bit_cast(uint2, r);
}
// 8 fp32 values -> their high and low fp16 terms, 8 x 16 bit each
__device__ __forceinline__ void x2_split8(const float (&f)[8], uint4& h, uint4& l) {
    uint16_t hh[8], ll[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split2_f16(f[j], hh[j], ll[j]);
    h = make_uint4((uint32_t)hh[0] | ((uint32_t)hh[1] << 16), (uint32_t)hh[2] | ((uint32_t)hh[3] << 16), (uint32_t)hh[4] | ((uint32_t)hh[5] << 16), (uint32_t)hh[6] | ((uint32_t)hh[7] << 16));
    l = make_uint4((uint32_t)ll[0] | ((uint32_t)ll[1] << 16), (uint32_t)ll[2] | ((uint32_t)ll[3] << 16), (uint32_t)ll[4] | ((uint32_t)ll[5] << 16), (uint32_t)ll[6] | ((uint32_t)ll[7] << 16));
}
__device__ __forceinline__ f32x4_t x2_mfma(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

template <int NKB>      // 16-key blocks: nk = 16 * NKB exactly (128, 192 or 256 keys)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_x2_kernel(const AttnF32Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    unsigned char* Kh = xsm;
    unsigned char* Kl = xsm + X2_IMG;
    unsigned char* Vh = xsm + 2 * X2_IMG;
    unsigned char* Vl = xsm + 3 * X2_IMG;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    unsigned char* Os = xsm + 4 * X2_IMG + w * 1024;      // per wave: 512 B of high terms, 512 B of low terms (4 queries x 64 d)
    float* qs_row = reinterpret_cast<float*>(xsm + 4 * X2_IMG + 8 * 1024);
    float* nk_row = qs_row + 64;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int kb_ = p.kv_batch_mod > 0 ? b % p.kv_batch_mod : b;
    const float* qg = reinterpret_cast<const float*>(p.q);
    const float* kbase = reinterpret_cast<const float*>(p.k) + (size_t)kb_ * p.k_sb + (size_t)h * p.k_sh;
    const float* vbase = reinterpret_cast<const float*>(p.v) + (size_t)kb_ * p.v_sb + (size_t)h * p.v_sh;
    constexpr int nk = NKB * 16;

    // ---- V -> two fp16 images [4 d-blocks][256 keys][16 d]: item = (key, d-block), two items per thread
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int item = t + 512 * i, key = item >> 2, dt = item & 3;
        float f[2][8];
        const float* src = vbase + (size_t)(key < nk ? key : 0) * p.v_sn + dt * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(src + 4 * c);
            f[c >> 1][(c & 1) * 4 + 0] = v.x; f[c >> 1][(c & 1) * 4 + 1] = v.y; f[c >> 1][(c & 1) * 4 + 2] = v.z; f[c >> 1][(c & 1) * 4 + 3] = v.w;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint4 hv, lv;
            x2_split8(f[c], hv, lv);
            *reinterpret_cast<uint4*>(Vh + dt * 8192 + key * 32 + c * 16) = hv;
            *reinterpret_cast<uint4*>(Vl + dt * 8192 + key * 32 + c * 16) = lv;
        }
    }
    // ---- K: half a key row (32 d) per thread -> normalise, scale, split, both images
    {
        const int s_key = t >> 1, s_half = t & 1;
        float f[4][8];
        float ss = 0.f;
        const float* kp = kbase + (size_t)(s_key < nk ? s_key : 0) * p.k_sn + s_half * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 a = *reinterpret_cast<const float4*>(kp + c * 8), b4 = *reinterpret_cast<const float4*>(kp + c * 8 + 4);
            f[c][0] = a.x; f[c][1] = a.y; f[c][2] = a.z; f[c][3] = a.w; f[c][4] = b4.x; f[c][5] = b4.y; f[c][6] = b4.z; f[c][7] = b4.w;
            if (s_key >= nk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[c][j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += f[c][j] * f[c][j];
        }
        float inv = 1.f;
        if (p.normalize) {
            ss += __shfl_xor(ss, 1, 64);
            inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);      // F.normalize eps (mmp.py:41-42)
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (p.normalize) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[c][j] = f[c][j] * inv * p.k_scale[s_half * 32 + c * 8 + j];
            }
            uint4 hv, lv;
            x2_split8(f[c], hv, lv);
            *reinterpret_cast<uint4*>(Kh + x2_sw(s_key, s_half * 4 + c)) = hv;
            *reinterpret_cast<uint4*>(Kl + x2_sw(s_key, s_half * 4 + c)) = lv;
        }
    }
    // ---- the per-dim query scales and the normalised null key (mmp.py:145-149) as fp32 rows in LDS
    if (w == 0) {
        qs_row[lane] = p.normalize ? p.q_scale[lane] : 1.f;
        float nkv = 0.f;
        if (p.null_k) {
            const float nkl = p.null_k[h * X2_DH + lane];
            const float ninv = p.normalize ? 1.f / fmaxf(sqrtf(wave_sum(nkl * nkl)), 1e-12f) : 1.f;
            nkv = nkl * ninv * (p.normalize ? p.k_scale[lane] : 1.f);
        }
        nk_row[lane] = nkv;
    }
    __syncthreads();

    const float c1 = p.scale * 1.4426950408889634f;
    float nvv[4];                        // this lane's dims of the null value
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) nvv[dt] = p.null_k ? p.null_v[h * X2_DH + dt * 16 + fr] : 0.f;
#pragma unroll 1
    for (int qb = 0; qb < 2; ++qb) {
        // ---- Q fragments of this wave's 16-query block (B operand of S^T): query fr, d = ks * 32 + 8 fg .. + 7, as high / low fp16 terms
        uint4 qh[2], ql[2];
        float s_null;
        {
            const int qi = blockIdx.x * 256 + w * 32 + qb * 16 + fr;
            const bool q_ok = qi < p.nq;
            const float* qp = qg + (size_t)b * p.q_sb + (size_t)h * p.q_sh + (size_t)(q_ok ? qi : 0) * p.q_sn;
            float qv[2][8];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const float4 a = *reinterpret_cast<const float4*>(qp + ks * 32 + 8 * fg), b4 = *reinterpret_cast<const float4*>(qp + ks * 32 + 8 * fg + 4);
                qv[ks][0] = a.x; qv[ks][1] = a.y; qv[ks][2] = a.z; qv[ks][3] = a.w; qv[ks][4] = b4.x; qv[ks][5] = b4.y; qv[ks][6] = b4.z; qv[ks][7] = b4.w;
                if (!q_ok) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) qv[ks][j] = 0.f;
                }
            }
            if (p.normalize) {
                float ss = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += qv[0][j] * qv[0][j] + qv[1][j] * qv[1][j];
                ss += __shfl_xor(ss, 16, 64);
                ss += __shfl_xor(ss, 32, 64);
                const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    qv[0][j] = qv[0][j] * inv * qs_row[8 * fg + j];
                    qv[1][j] = qv[1][j] * inv * qs_row[32 + 8 * fg + j];
                }
            }
            x2_split8(qv[0], qh[0], ql[0]);
            x2_split8(qv[1], qh[1], ql[1]);
            s_null = X2_NEG_BIG;
            if (p.null_k) {
                float part = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) part += qv[0][j] * nk_row[8 * fg + j] + qv[1][j] * nk_row[32 + 8 * fg + j];
                part += __shfl_xor(part, 16, 64);
                part += __shfl_xor(part, 32, 64);
                s_null = part;      // raw units (q^ . k^), like the scores
            }
        }
        // ---- S^T = K^ Q^T as three term products: acc_s[kb][r] -> key kb * 16 + 4 fg + r, query fr
        f32x4_t acc_s[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            acc_s[kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 kh = *reinterpret_cast<const uint4*>(Kh + x2_sw(kb * 16 + fr, ks * 4 + fg));
                const uint4 kl = *reinterpret_cast<const uint4*>(Kl + x2_sw(kb * 16 + fr, ks * 4 + fg));
                acc_s[kb] = x2_mfma(kl, qh[ks], acc_s[kb]);      // (the small terms first)
                acc_s[kb] = x2_mfma(kh, ql[ks], acc_s[kb]);
                acc_s[kb] = x2_mfma(kh, qh[ks], acc_s[kb]);
            }
        }
        float m = s_null;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            m = __builtin_fmaxf(__builtin_fmaxf(m, acc_s[kb][0]), acc_s[kb][1]);
            m = __builtin_fmaxf(__builtin_fmaxf(m, acc_s[kb][2]), acc_s[kb][3]);
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f((acc_s[kb][r] - m) * c1);      // (v_exp_f32: 1 ulp; the subtraction first -- exact at the maximum)
                acc_s[kb][r] = e;
                psum += e;
            }
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        const float p_null = p.null_k ? __builtin_amdgcn_exp2f((s_null - m) * c1) : 0.f;
        // Round 6: the probabilities enter the P V product multiplied by 2^12 (exact), the inverse rides in linv.  Unscaled, the LOW fp16 term of a
        // probability below ~2^-3 is subnormal (absolute floor 2^-25 per key): diffuse attention over 256 keys kept ~17 bits.  With P in (0, 4096] the low term
        // is normal down to p ~ 2^-15 and the floor is 2^-37 -- the weight side of the tier scales by a power of two for the same reason (ops.f16_weight_scale).
        constexpr float PSC = 4096.f;
        const float linv = (1.f / PSC) / (psum + p_null);
        // ---- O = p_null * v_null + P V as three term products: acc_o[dt][r] -> query 4 fg + r, d = dt * 16 + fr
        f32x4_t acc_o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pn = __shfl(p_null, 4 * fg + r, 64) * PSC;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc_o[dt][r] = pn * nvv[dt];
        }
#pragma unroll
        for (int ks = 0; ks < NKB / 2; ++ks) {
            const float pe[8] = {acc_s[2 * ks][0] * PSC, acc_s[2 * ks][1] * PSC, acc_s[2 * ks][2] * PSC, acc_s[2 * ks][3] * PSC,
                                 acc_s[2 * ks + 1][0] * PSC, acc_s[2 * ks + 1][1] * PSC, acc_s[2 * ks + 1][2] * PSC, acc_s[2 * ks + 1][3] * PSC};
            uint4 ph, pl;
            x2_split8(pe, ph, pl);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const uint2 hlo = x2_read_tr16(Vh + dt * 8192 + ks * 1024 + lane * 8), hhi = x2_read_tr16(Vh + dt * 8192 + ks * 1024 + 512 + lane * 8);
                const uint2 llo = x2_read_tr16(Vl + dt * 8192 + ks * 1024 + lane * 8), lhi = x2_read_tr16(Vl + dt * 8192 + ks * 1024 + 512 + lane * 8);
                const uint4 vh = make_uint4(hlo.x, hlo.y, hhi.x, hhi.y), vl = make_uint4(llo.x, llo.y, lhi.x, lhi.y);
                acc_o[dt] = x2_mfma(ph, vl, acc_o[dt]);
                acc_o[dt] = x2_mfma(pl, vh, acc_o[dt]);
                acc_o[dt] = x2_mfma(ph, vh, acc_o[dt]);
            }
        }
        // ---- O / l -> fp16 terms, transposed through LDS per wave (4 queries = one accumulator register index r per pass) so that a lane stores 16 B of one output
        //      row per segment: segments [h | l | h][:P] of the output projection's operand (common.h store_split4)
        const int nseg = (p.P & MM_SPLIT_NODUP_BIT) ? 2 : split_count(p.P);      // (NODUP: the repeated h segment is not written -- its readers stage the h plane once)
        if (p.out) {      // (operator-level entry mm_attend_terms: the fp32 result itself, same staging area: 4 queries x 256 B)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float lr = __shfl(linv, 4 * fg + r, 64);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<float*>(Os + fg * 256 + (dt * 16 + fr) * 4) = acc_o[dt][r] * lr;
                __builtin_amdgcn_wave_barrier();
                const int qo = blockIdx.x * 256 + w * 32 + qb * 16 + 4 * (lane >> 4) + r;      // row lane >> 4, 16-byte chunk lane & 15
                if (qo < p.nq)
                    *reinterpret_cast<uint4*>(reinterpret_cast<float*>(p.out) + (size_t)b * p.o_sb + (size_t)h * p.o_sh + (size_t)qo * p.o_sn + (lane & 15) * 4) =
                        *reinterpret_cast<const uint4*>(Os + lane * 16);
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lr = __shfl(linv, 4 * fg + r, 64);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint16_t oh, ol;
                split2_f16(acc_o[dt][r] * lr, oh, ol);
                *reinterpret_cast<uint16_t*>(Os + fg * 128 + (dt * 16 + fr) * 2) = oh;
                *reinterpret_cast<uint16_t*>(Os + 512 + fg * 128 + (dt * 16 + fr) * 2) = ol;
            }
            __builtin_amdgcn_wave_barrier();
            const int qo = blockIdx.x * 256 + w * 32 + qb * 16 + 4 * (lane >> 3) + r;      // lanes 0 .. 31: row lane >> 3, chunk lane & 7
            if (lane < 32 && qo < p.nq) {
                const uint4 vh = *reinterpret_cast<const uint4*>(Os + lane * 16), vl = *reinterpret_cast<const uint4*>(Os + 512 + lane * 16);
                bf16_t* orow = p.out_split + (size_t)b * p.os_sb + (size_t)qo * p.os_sn + (size_t)h * X2_DH + (lane & 7) * 8;
                *reinterpret_cast<uint4*>(orow) = vh;
                *reinterpret_cast<uint4*>(orow + p.os_seg) = vl;
                if (nseg > 2) *reinterpret_cast<uint4*>(orow + 2 * (size_t)p.os_seg) = vh;
            }
        }
    }
}

}  // namespace

bool k_attention_x2_eligible(const AttnF32Args& a) {
    const int dh = a.dh ? a.dh : 64;
    if (dh != X2_DH || a.io_bf16 || a.key_mask || a.nq < 128 || !(a.nk == 128 || a.nk == 192 || a.nk == 256) || !(a.scale > 0.f)) return false;
    if (a.out) return !a.out_split;      // fp32 result (mm_attend_terms)
    return a.out_split && split_is_f16(a.P) && (a.os_seg % 8) == 0 && (a.os_sn % 8) == 0 && (a.os_sb % 8) == 0 && ((uintptr_t)a.out_split & 15) == 0;
}

int k_attention_x2(hipStream_t s, const AttnF32Args& a) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_x2_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, X2_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_x2_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize, X2_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_x2_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, X2_SMEM);
        if (e != hipSuccess) return mm_set_hip_error(e, "attention_x2 hipFuncSetAttribute");
        attr_set = true;
    }
    const dim3 grid((a.nq + 255) / 256, a.H, a.B);
    if (a.nk == 256) hipLaunchKernelGGL(attention_x2_kernel<16>, grid, dim3(512), X2_SMEM, s, a);
    else if (a.nk == 192) hipLaunchKernelGGL(attention_x2_kernel<12>, grid, dim3(512), X2_SMEM, s, a);
    else hipLaunchKernelGGL(attention_x2_kernel<8>, grid, dim3(512), X2_SMEM, s, a);
    return mm_check_launch("attention_x2_kernel");
}
