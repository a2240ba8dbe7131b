// Muse attention for gfx950:   softmax(8 * q^ . k^ + key mask) @ v   with a learned null key/value
// (muse_maskgit_pytorch.py:143-159 + attend.py:123-140; the default flash branch attend.py:66-107 computes the
// same function).  dim_head is 64 (the reference default and every BASELINE config).
//
// One 256-thread workgroup = 4 waves = 64 queries of one (batch, head); each wave owns 16 queries.
//   * optional prologue fusion (normalize=1): q / k rows come straight out of the projection GEMMs
//     ([tokens][heads*64] bf16); F.normalize + the learned per-dim scales are applied while staging, so
//     the reference's rearrange / cat / l2norm / scale passes (mmp.py:143-153) never touch HBM.
//   * the null key/value (mmp.py:145-149) is not a 257th key: it INITIALISES the online-softmax state
//     (m = s_null, l = 1, O = v_null), so the key loop runs over exactly nk real keys in tiles of 64.
//   * S^T = K Q^T on MFMA (16x16x32 bf16): a lane then owns 4 consecutive keys of ONE query, so the
//     row max / row sum need two cross-lane steps and P packs into 8-byte LDS writes.
//   * O = P V on MFMA; V is transposed while it is staged into LDS (keys become the contiguous axis).
//   LDS: K tile 8 KiB + V^T tile 8 KiB + 4 x 2 KiB P, all in 128-byte rows with the 16-byte chunk index
//   XOR (row & 7)  -> conflict-free ds_read_b128 fragment reads.
#include <string.h>

#include "common.h"
#include "muse_hip_internal.h"

namespace {

constexpr int DH = 64;
constexpr int KT = 64;       // keys per tile
constexpr float NEG_BIG = -3.0e38f;

__device__ __forceinline__ int sw_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <int NWV>      // waves per workgroup: 4 (64 queries) or 8 (128 queries: K/V staging amortised over twice the queries)
__global__ __launch_bounds__(NWV * 64) void attention_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[8192 + 8192 + NWV * 2048];
    unsigned char* Ks = smem;
    unsigned char* Vt = smem + 8192;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    unsigned char* Ps = smem + 16384 + w * 2048;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q_wave0 = blockIdx.x * (NWV * 16) + w * 16;

    // (the first K/V tile's loads are issued here, ahead of the Q / null-key prologue, so the latencies overlap)
    const int kb = p.kv_batch_mod > 0 ? b % p.kv_batch_mod : b;
    const bf16_t* kbase = p.k + (size_t)kb * p.k_sb + (size_t)h * p.k_sh;
    const bf16_t* vbase = p.v + (size_t)kb * p.v_sb + (size_t)h * p.v_sh;
    const uint8_t* kmask = p.key_mask ? p.key_mask + (size_t)b * p.km_sb : nullptr;

    // ---- K/V staging is split (issue the global loads early, write LDS late): the next tile's loads are in flight while
    //      the current tile's MFMAs and softmax run
    // K: 512 16-byte chunks per tile: NWV=4 -> 2 adjacent chunks per thread, NWV=8 -> 1 chunk per thread
    constexpr int KCH = 512 / (NWV * 64);                   // chunks per thread (2 or 1)
    const int s_key = t / (8 / KCH), s_dpart = (t % (8 / KCH)) * (8 * KCH);
    const int s_kp2 = t & 31, s_dc = (t >> 5) & 7;         // V: threads 0..255 -> key pair t&31, d chunk t>>5 (written transposed)
    const bool v_thread = t < 256;
    uint4 kr0, kr1, vr0, vr1;
    bool kok, vok0, vok1;
#define LOAD_KV(kt0_)                                                                                              \
    {                                                                                                              \
        const int kg_ = (kt0_) + s_key;                                                                            \
        kok = kg_ < p.nk;                                                                                          \
        const bf16_t* kp_ = kbase + (size_t)(kok ? kg_ : 0) * p.k_sn + s_dpart;                                    \
        kr0 = *reinterpret_cast<const uint4*>(kp_);                                                                \
        if (KCH == 2) kr1 = *reinterpret_cast<const uint4*>(kp_ + 8);                                              \
        const int vg_ = (kt0_) + 2 * s_kp2;                                                                        \
        vok0 = vg_ < p.nk; vok1 = vg_ + 1 < p.nk;                                                                  \
        vr0 = *reinterpret_cast<const uint4*>(vbase + (size_t)(vok0 ? vg_ : 0) * p.v_sn + s_dc * 8);               \
        vr1 = *reinterpret_cast<const uint4*>(vbase + (size_t)(vok1 ? vg_ + 1 : 0) * p.v_sn + s_dc * 8);           \
    }
    if (p.nk > 0) LOAD_KV(0);


    // ---- Q fragment (B operand of S^T): query = q_wave0 + fr, d = ks*32 + 8*fg .. +7
    const int qi = q_wave0 + fr;
    const bool q_ok = qi < p.nq;
    uint4 qf[2];
    float qv0[8], qv1[8];     // the (bf16-rounded) q^ values this lane holds, for the null-key score
    {
        const bf16_t* qp = p.q + (size_t)b * p.q_sb + (size_t)h * p.q_sh + (size_t)(q_ok ? qi : 0) * p.q_sn;
        const uint4 z = make_uint4(0, 0, 0, 0);
        const uint4 l0 = *reinterpret_cast<const uint4*>(qp + 8 * fg);        // row index clamped: always in bounds
        const uint4 l1 = *reinterpret_cast<const uint4*>(qp + 32 + 8 * fg);
        qf[0] = q_ok ? l0 : z;
        qf[1] = q_ok ? l1 : z;
        unpack8(qf[0], qv0);
        unpack8(qf[1], qv1);
        if (p.normalize) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += qv0[j] * qv0[j] + qv1[j] * qv1[j];
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);      // F.normalize eps (mmp.py:41-42)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                qv0[j] = qv0[j] * inv * p.q_scale[8 * fg + j];
                qv1[j] = qv1[j] * inv * p.q_scale[32 + 8 * fg + j];
            }
            qf[0] = pack8(qv0);
            qf[1] = pack8(qv1);
            unpack8(qf[0], qv0);   // what the MFMA sees
            unpack8(qf[1], qv1);
        }
    }

    // ---- online-softmax state.  Scores are per query = per lane column fr (replicated over fg);
    //      acc_o[dt][r] holds query 4*fg + r, d = dt*16 + fr.
    float m_run = -1.0e30f, l_run = 0.f;
    f32x4_t acc_o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc_o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (p.null_k) {
        const float* nk = p.null_k + h * DH;
        const float* nv = p.null_v + h * DH;
        float nss = 0.f;
        for (int d = 0; d < DH; ++d) nss += nk[d] * nk[d];
        const float ninv = p.normalize ? 1.f / fmaxf(sqrtf(nss), 1e-12f) : 1.f;
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d0 = 8 * fg + j, d1 = 32 + 8 * fg + j;
            const float ks0 = p.normalize ? p.k_scale[d0] : 1.f, ks1 = p.normalize ? p.k_scale[d1] : 1.f;
            const float k0 = bf16_to_f32(f32_to_bf16(nk[d0] * ninv * ks0));
            const float k1 = bf16_to_f32(f32_to_bf16(nk[d1] * ninv * ks1));
            part += qv0[j] * k0 + qv1[j] * k1;
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        m_run = part * p.scale;
        l_run = 1.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const float v0 = bf16_to_f32(f32_to_bf16(nv[dt * 16 + fr]));
            acc_o[dt] = f32x4_t{v0, v0, v0, v0};
        }
    }

    for (int kt0 = 0; kt0 < p.nk; kt0 += KT) {
        // ---- write the staged K tile (normalised) and V tile (transposed) to LDS
        if (!(p.debug & 512)) {
            const uint4 z = make_uint4(0, 0, 0, 0);
            uint4 r0 = kok ? kr0 : z;
            uint4 r1 = (KCH == 2 && kok) ? kr1 : z;
            if (p.normalize) {
                float f0[8], f1[8];
                unpack8(r0, f0); unpack8(r1, f1);
                float ss = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += f0[j] * f0[j] + f1[j] * f1[j];
                ss += __shfl_xor(ss, 1, 64);
                ss += __shfl_xor(ss, 2, 64);
                if (KCH == 1) ss += __shfl_xor(ss, 4, 64);
                const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f0[j] = f0[j] * inv * p.k_scale[s_dpart + j];
                    if (KCH == 2) f1[j] = f1[j] * inv * p.k_scale[s_dpart + 8 + j];
                }
                r0 = pack8(f0); r1 = pack8(f1);
            }
            *reinterpret_cast<uint4*>(Ks + sw_off(s_key, s_dpart >> 3)) = r0;
            if (KCH == 2) *reinterpret_cast<uint4*>(Ks + sw_off(s_key, (s_dpart >> 3) + 1)) = r1;
            if (v_thread) {
            const uint4 a0 = vok0 ? vr0 : z;
            const uint4 a1 = vok1 ? vr1 : z;
            const uint32_t w0[4] = {a0.x, a0.y, a0.z, a0.w};
            const uint32_t w1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t lo = (i & 1) ? (w0[i >> 1] >> 16) : (w0[i >> 1] & 0xFFFFu);
                const uint32_t hi = (i & 1) ? (w1[i >> 1] >> 16) : (w1[i >> 1] & 0xFFFFu);
                const int d = s_dc * 8 + i;
                *reinterpret_cast<uint32_t*>(Vt + sw_off(d, s_kp2 >> 2) + (s_kp2 & 3) * 4) = lo | (hi << 16);
            }
            }
        }
        __syncthreads();
        if (kt0 + KT < p.nk) LOAD_KV(kt0 + KT);      // in flight during this tile's compute

        if (p.debug & 256) { __syncthreads(); continue; }
        // ---- S^T = K Q^T : acc_s[kt4][r] -> key kt0 + kt4*16 + 4*fg + r, query fr
        f32x4_t acc_s[4];
#pragma unroll
        for (int kt4 = 0; kt4 < 4; ++kt4) {
            acc_s[kt4] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 kf = *reinterpret_cast<const uint4*>(Ks + sw_off(kt4 * 16 + fr, ks * 4 + fg));
                acc_s[kt4] = mfma16(kf, qf[ks], acc_s[kt4]);
            }
        }
        float tmax = NEG_BIG;
        // interior tiles without a key mask need no per-score validity test (wave-uniform branch)
        const bool edge_tile = (kt0 + KT > p.nk) || (kmask != nullptr);
        if (!edge_tile) {
#pragma unroll
            for (int kt4 = 0; kt4 < 4; ++kt4)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s = acc_s[kt4][r] * p.scale;
                    acc_s[kt4][r] = s;
                    tmax = fmaxf(tmax, s);
                }
        } else {
#pragma unroll
            for (int kt4 = 0; kt4 < 4; ++kt4)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kg = kt0 + kt4 * 16 + 4 * fg + r;
                    bool ok = kg < p.nk;
                    if (ok && kmask) ok = kmask[kg] != 0;
                    const float s = ok ? acc_s[kt4][r] * p.scale : NEG_BIG;
                    acc_s[kt4][r] = s;
                    tmax = fmaxf(tmax, s);
                }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        // the weights are rounded to bf16 for the PV MFMA, so the hardware exp2 path (~1e-6 relative) is exact enough
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int kt4 = 0; kt4 < 4; ++kt4) {
            float pv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pv[r] = (p.debug & 1024) ? acc_s[kt4][r] - m_new : __expf(acc_s[kt4][r] - m_new);
                psum += pv[r];
            }
            // P[query fr][keys kt4*16 + 4fg .. +3] -> 8-byte LDS write
            const int chunk = kt4 * 2 + (fg >> 1);
            *reinterpret_cast<uint2*>(Ps + sw_off(fr, chunk) + (fg & 1) * 8) =
                make_uint2(pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3]));
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        // rescale O: row 4*fg + r needs that query's alpha (held by lane 4*fg + r)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ar = __shfl(alpha, 4 * fg + r, 64);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc_o[dt][r] *= ar;
        }
        __builtin_amdgcn_wave_barrier();   // P writes (this wave) before P reads (this wave)
        // ---- O += P V
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 pf = *reinterpret_cast<const uint4*>(Ps + sw_off(fr, ks * 4 + fg));
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const uint4 vf = *reinterpret_cast<const uint4*>(Vt + sw_off(dt * 16 + fr, ks * 4 + fg));
                acc_o[dt] = mfma16(pf, vf, acc_o[dt]);
            }
        }
        __syncthreads();   // all waves done with Ks / Vt before the next tile is staged
    }

    // ---- epilogue: O / l -> bf16, transposed through this wave's P buffer (16 queries x 64 d = 2 KiB) so that each lane
    //      stores 2 x 16 B of one 128-byte output row instead of 16 scattered 2-byte values
    const float linv = 1.f / l_run;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float lr = __shfl(linv, 4 * fg + r, 64);
        const int row = 4 * fg + r;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<bf16_t*>(Ps + row * 128 + (dt * 16 + fr) * 2) = f32_to_bf16(acc_o[dt][r] * lr);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = i * 64 + lane;            // 128 chunks of 16 B: row c>>3, chunk c&7
        const int qo = q_wave0 + (c >> 3);
        if (qo < p.nq) {
            const uint4 val = *reinterpret_cast<const uint4*>(Ps + (c >> 3) * 128 + (c & 7) * 16);
            *reinterpret_cast<uint4*>(p.out + (size_t)b * p.o_sb + (size_t)h * p.o_sh + (size_t)qo * p.o_sn + (c & 7) * 8) = val;
        }
    }
}


// ------------------------------------------------------------------------------------------------ all keys resident (nk <= 256)
// Self-attention of the 256-token configs: ONE 512-thread workgroup per (batch, head) takes all (up to 256) queries, so K and V
// are fetched, and K is normalised, once per (batch, head) instead of once per 128 queries, and the whole key axis sits in LDS:
//   * K^ (l2-normalised * k_scale, bf16) goes through registers into 128-byte rows with the chunk XOR (row & 7) as above;
//   * V is NOT transposed by the VALU: LDS-DMA (global_load_lds) copies it as [4 d-blocks][256 keys][16 d] (32-byte rows) and the
//     P.V B-operand is fetched with ds_read_b64_tr_b16, gfx950's transposing LDS read -- within a 16-lane group, lane c receives
//     element (c & 3) of the 8 bytes addressed by lanes (c >> 2), 4 + (c >> 2), 8 + .., 12 + ..; with lane l addressing byte 8*l of
//     a 16-key x 16-d block that is V[4 consecutive keys 4*fg ..][d0 + c].  (A row-major [key][64 d] image with an XOR swizzle is
//     conflict-free by the ordinary bank rule and still ran this kernel at 53 us instead of 34: the transposing read has its own
//     conflict classes, MI355X_MICROARCH.md LDS table.)
//   * no online softmax and no P buffer: S^T = K^ Q^T for all 16 key blocks stays in registers (a lane owns 4 keys x 16 blocks of
//     ONE query), softmax is two cross-lane steps, and two 16-key accumulator blocks, packed to bf16, ARE an A operand of the
//     P.V MFMA (contraction slot 8*fg + j <-> key block (j >> 2), key 4*fg + (j & 3)) as long as V's fragment uses the same slot
//     order -- which the two transposing reads per fragment do;
//   * the null key/value enters the softmax as one more score per query and O starts at p_null * v_null.
// LDS: 32 KiB K^ + 32 KiB V + 8 x 512 B output transposition = 68 KiB -> two workgroups per CU, 128 VGPRs per wave.
constexpr int FULL_NK = 256;
constexpr int FULL_SMEM = 2 * FULL_NK * 128 + 8 * 512 + 512;

typedef short v4i16_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint2 lds_read_tr16(const unsigned char* ptr) {
    typedef __attribute__((address_space(3))) v4i16_t* lds_v4_t;
    const v4i16_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)ptr);
    return __builtin_bit_cast(uint2, r);
}

template <int NKB>      // 16-key blocks: nk = 16 * NKB exactly (128, 192 or 256 keys) -- no per-score validity tests anywhere
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void attention_full_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    unsigned char* Ks = fsm;
    unsigned char* Vs = fsm + FULL_NK * 128;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    unsigned char* Os = fsm + 2 * FULL_NK * 128 + w * 512;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int kb_ = p.kv_batch_mod > 0 ? b % p.kv_batch_mod : b;
    const bf16_t* kbase = p.k + (size_t)kb_ * p.k_sb + (size_t)h * p.k_sh;
    const bf16_t* vbase = p.v + (size_t)kb_ * p.v_sb + (size_t)h * p.v_sh;
    constexpr int nk = NKB * 16;

    // ---- V: 4 LDS-DMA instructions per wave, each 32 keys x one 16-d block = 1 KiB of the [4 d-blocks][256 keys][16 d] image
    //      (rows of the image beyond nk are never read)
    {
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = w * 32 + (lane >> 1);
            const bf16_t* src = vbase + (size_t)(key < nk ? key : 0) * p.v_sn + i * 16 + (lane & 1) * 8;
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(Vs + i * 8192 + w * 1024), 16, 0, 0);
        }
    }
    // ---- K: half a key row (32 d) per thread
    const int s_key = t >> 1, s_half = t & 1;
    uint4 kr[4];
    {
        const bf16_t* kp = kbase + (size_t)(s_key < nk ? s_key : 0) * p.k_sn + s_half * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) kr[c] = *reinterpret_cast<const uint4*>(kp + c * 8);
    }
    // ---- the per-dim query scales and the normalised null key (mmp.py:145-149; bf16-rounded like a real key), as fp32 rows in LDS:
    //      every wave reads its 16 dims of both per query block with four 16-byte reads instead of 32 scalar global loads
    float* qs_row = reinterpret_cast<float*>(fsm + 2 * FULL_NK * 128 + 8 * 512);
    float* nk_row = qs_row + 64;
    if (w == 0) {
        qs_row[lane] = p.normalize ? p.q_scale[lane] : 1.f;
        float nkv = 0.f;
        if (p.null_k) {
            const float nkl = p.null_k[h * DH + lane];
            const float ninv = p.normalize ? 1.f / fmaxf(sqrtf(wave_sum(nkl * nkl)), 1e-12f) : 1.f;
            nkv = bf16_to_f32(f32_to_bf16(nkl * ninv * (p.normalize ? p.k_scale[lane] : 1.f)));
        }
        nk_row[lane] = nkv;
    }
    // ---- K^ -> LDS
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        float f[4][8];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (s_key >= nk) kr[c] = z;
            unpack8(kr[c], f[c]);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += f[c][j] * f[c][j];
        }
        if (p.normalize) {
            ss += __shfl_xor(ss, 1, 64);
            const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[c][j] = f[c][j] * inv * p.k_scale[s_half * 32 + c * 8 + j];
                kr[c] = pack8(f[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<uint4*>(Ks + sw_off(s_key, s_half * 4 + c)) = kr[c];
    }
    // The V image was copied by LDS-DMA: it is visible to other waves only after the ISSUING wave's vmcnt has drained and a barrier.  The
    // K loads above follow the DMA in program order and are consumed before this point, so the counter is already zero here, but the
    // ordering must not depend on that: drain explicitly, then barrier.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const float c1 = p.scale * 1.4426950408889634f;
    float nvv[4];                        // this lane's dims of the (bf16-rounded) null value
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) nvv[dt] = p.null_k ? bf16_to_f32(f32_to_bf16(p.null_v[h * DH + dt * 16 + fr])) : 0.f;
#pragma unroll 1
    for (int qb = 0; qb < 2; ++qb) {
        // ---- Q fragments of this wave's two 16-query blocks (B operand of S^T): query fr, d = ks*32 + 8*fg .. +7
        uint4 qf[2];
        float s_null;
        {
            const int qi = blockIdx.x * 256 + w * 32 + qb * 16 + fr;
            const bool q_ok = qi < p.nq;
            const bf16_t* qp = p.q + (size_t)b * p.q_sb + (size_t)h * p.q_sh + (size_t)(q_ok ? qi : 0) * p.q_sn;
            const uint4 z = make_uint4(0, 0, 0, 0);
            const uint4 l0 = *reinterpret_cast<const uint4*>(qp + 8 * fg);
            const uint4 l1 = *reinterpret_cast<const uint4*>(qp + 32 + 8 * fg);
            qf[0] = q_ok ? l0 : z;
            qf[1] = q_ok ? l1 : z;
            float qv0[8], qv1[8];
            unpack8(qf[0], qv0);
            unpack8(qf[1], qv1);
            if (p.normalize) {
                float ss = 0.f;
    #pragma unroll
                for (int j = 0; j < 8; ++j) ss += qv0[j] * qv0[j] + qv1[j] * qv1[j];
                ss += __shfl_xor(ss, 16, 64);
                ss += __shfl_xor(ss, 32, 64);
                const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);      // F.normalize eps (mmp.py:41-42)
    #pragma unroll
                for (int j = 0; j < 8; ++j) {
                    qv0[j] = qv0[j] * inv * qs_row[8 * fg + j];
                    qv1[j] = qv1[j] * inv * qs_row[32 + 8 * fg + j];
                }
                qf[0] = pack8(qv0);
                qf[1] = pack8(qv1);
                unpack8(qf[0], qv0);   // what the MFMA sees
                unpack8(qf[1], qv1);
            }
            s_null = NEG_BIG;
            if (p.null_k) {
                float part = 0.f;
    #pragma unroll
                for (int j = 0; j < 8; ++j) part += qv0[j] * nk_row[8 * fg + j] + qv1[j] * nk_row[32 + 8 * fg + j];
                part += __shfl_xor(part, 16, 64);
                part += __shfl_xor(part, 32, 64);
                s_null = part;      // raw units, like the scores
            }
        }
        // ---- S^T = K^ Q^T: acc_s[kb][r] -> key kb*16 + 4*fg + r, query fr.  The scores stay in the MFMA's raw units (q^ . k^):
        //      softmax(scale * s) = exp2((s - max s) * scale * log2 e) is one FMA + v_exp_f32 per score (scale > 0: mmp.py:98 has 8)
        f32x4_t acc_s[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            acc_s[kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 kf = *reinterpret_cast<const uint4*>(Ks + sw_off(kb * 16 + fr, ks * 4 + fg));
                acc_s[kb] = mfma16(kf, qf[ks], acc_s[kb]);
            }
        }
        float m = s_null;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {      // two v_max3_f32 per accumulator
            m = __builtin_fmaxf(__builtin_fmaxf(m, acc_s[kb][0]), acc_s[kb][1]);
            m = __builtin_fmaxf(__builtin_fmaxf(m, acc_s[kb][2]), acc_s[kb][3]);
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        // (the weights are rounded to bf16 for the PV MFMA, so the hardware exp2, ~1e-6 relative, is exact enough)
        const float mc = -m * c1;
        // packed fp32 (v_pk_fma_f32 / v_pk_add_f32): this kernel is bound by VALU issue, not by the MFMA pipe
        const f32x2_t c1v = {c1, c1}, mcv = {mc, mc};
        f32x2_t ps2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f32x2_t a2 = {acc_s[kb][2 * hh], acc_s[kb][2 * hh + 1]};
                const f32x2_t x2 = __builtin_elementwise_fma(a2, c1v, mcv);
                const f32x2_t e2 = {__builtin_amdgcn_exp2f(x2[0]), __builtin_amdgcn_exp2f(x2[1])};
                acc_s[kb][2 * hh] = e2[0];
                acc_s[kb][2 * hh + 1] = e2[1];
                ps2 += e2;
            }
        }
        float psum = ps2[0] + ps2[1];
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        const float p_null = p.null_k ? __builtin_amdgcn_exp2f(__builtin_fmaf(s_null, c1, mc)) : 0.f;
        const float linv = 1.f / (psum + p_null);
        // ---- O = p_null * v_null + P V: acc_o[dt][r] -> query 4*fg + r, d = dt*16 + fr
        f32x4_t acc_o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pn = __shfl(p_null, 4 * fg + r, 64);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                acc_o[dt][r] = pn * nvv[dt];
        }
        // transposing reads: a 16-key x 16-d block is 512 contiguous bytes of the image and lane l supplies byte 8*l of it (key
        // 4*fg + (fr >> 2), d 4*(fr & 3) .. +3) -- the one layout the LDS serves without bank conflicts for this instruction
#pragma unroll
        for (int ks = 0; ks < NKB / 2; ++ks) {
            {
                const float pa[4] = {acc_s[2 * ks][0], acc_s[2 * ks][1], acc_s[2 * ks][2], acc_s[2 * ks][3]};
                const float pb[4] = {acc_s[2 * ks + 1][0], acc_s[2 * ks + 1][1], acc_s[2 * ks + 1][2], acc_s[2 * ks + 1][3]};
                const uint4 pf = make_uint4(pack_bf16x2(pa[0], pa[1]), pack_bf16x2(pa[2], pa[3]), pack_bf16x2(pb[0], pb[1]), pack_bf16x2(pb[2], pb[3]));
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const uint2 lo = lds_read_tr16(Vs + dt * 8192 + ks * 1024 + lane * 8);
                    const uint2 hi = lds_read_tr16(Vs + dt * 8192 + ks * 1024 + 512 + lane * 8);
                    acc_o[dt] = mfma16(pf, make_uint4(lo.x, lo.y, hi.x, hi.y), acc_o[dt]);
                }
            }
        }
        // ---- O / l -> bf16, transposed through 512 B of LDS per wave, 4 queries (one accumulator register index r) per pass, so that
        //      a lane stores 16 B of one 128-byte output row instead of 16 scattered 2-byte values
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lr = __shfl(linv, 4 * fg + r, 64);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<bf16_t*>(Os + fg * 128 + (dt * 16 + fr) * 2) = f32_to_bf16(acc_o[dt][r] * lr);
            __builtin_amdgcn_wave_barrier();
            const int qo = blockIdx.x * 256 + w * 32 + qb * 16 + 4 * (lane >> 3) + r;      // lanes 0..31: row lane >> 3, chunk lane & 7
            if (lane < 32 && qo < p.nq) {
                const uint4 val = *reinterpret_cast<const uint4*>(Os + lane * 16);
                *reinterpret_cast<uint4*>(p.out + (size_t)b * p.o_sb + (size_t)h * p.o_sh + (size_t)qo * p.o_sn + (lane & 7) * 8) = val;
            }
        }
    }
}

}  // namespace

int k_attention(hipStream_t s, const AttnArgs& a_in) {
    AttnArgs a = a_in;
    a.debug = g_mm_debug;
    if (a.B <= 0 || a.H <= 0 || a.nq <= 0) return MM_OK;
    if (a.dh != 0 && a.dh != 64) {
        // dim_head 32 / 128 (muse_maskgit_pytorch.py:165-174 accepts any): the kernels of this file are built around 64-wide heads; the
        // fp32-MFMA kernel of attention_f32.hip is templated on the head width and reads / writes the bf16 operands directly
        AttnF32Args f;
        memset(&f, 0, sizeof(f));
        f.q = a.q; f.q_sb = a.q_sb; f.q_sh = a.q_sh; f.q_sn = a.q_sn;
        f.k = a.k; f.k_sb = a.k_sb; f.k_sh = a.k_sh; f.k_sn = a.k_sn;
        f.v = a.v; f.v_sb = a.v_sb; f.v_sh = a.v_sh; f.v_sn = a.v_sn;
        f.out = a.out; f.o_sb = a.o_sb; f.o_sh = a.o_sh; f.o_sn = a.o_sn;
        f.B = a.B; f.H = a.H; f.nq = a.nq; f.nk = a.nk; f.key_mask = a.key_mask; f.km_sb = a.km_sb; f.normalize = a.normalize;
        f.q_scale = a.q_scale; f.k_scale = a.k_scale; f.null_k = a.null_k; f.null_v = a.null_v; f.scale = a.scale;
        f.kv_batch_mod = a.kv_batch_mod; f.dh = a.dh; f.io_bf16 = 1;
        return k_attention_f32(s, f);
    }
    if (a.nk < 0) return mm_set_error(MM_ERR_SHAPE, "attention: nk < 0");
    if (a.nk == 0 && !a.null_k) return mm_set_error(MM_ERR_SHAPE, "attention: no keys at all");
    if ((a.q_sn % 8) || (a.k_sn % 8) || (a.v_sn % 8) || (a.q_sh % 8) || (a.k_sh % 8) || (a.v_sh % 8) ||
        (a.q_sb % 8) || (a.k_sb % 8) || (a.v_sb % 8) || (a.o_sn % 8) || (a.o_sh % 8) || (a.o_sb % 8))
        return mm_set_error(MM_ERR_ALIGN, "attention: q/k/v strides must be multiples of 8 elements");
    if (a.normalize && (!a.q_scale || !a.k_scale)) return mm_set_error(MM_ERR_SHAPE, "attention: normalize needs q_scale/k_scale");
    const int reps = 1 + ((g_mm_debug >> 16) & 0xFF);      // tools/attn_bench.py: back-to-back launches from C
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_full_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, FULL_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_full_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize, FULL_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_full_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, FULL_SMEM);
        if (e != hipSuccess) return mm_set_hip_error(e, "attention_full hipFuncSetAttribute");
        attr_set = true;
    }
    for (int r = 0; r < reps; ++r) {
        if (a.nq >= 128 && (a.nk == 128 || a.nk == 192 || a.nk == 256) && !a.key_mask && a.scale > 0.f && !(g_mm_debug & 32768)) {
            // all keys resident: one workgroup per (batch, head, 256 queries)
            dim3 grid((a.nq + 255) / 256, a.H, a.B);
            if (a.nk == 256) hipLaunchKernelGGL(attention_full_kernel<16>, grid, dim3(512), FULL_SMEM, s, a);
            else if (a.nk == 192) hipLaunchKernelGGL(attention_full_kernel<12>, grid, dim3(512), FULL_SMEM, s, a);
            else hipLaunchKernelGGL(attention_full_kernel<8>, grid, dim3(512), FULL_SMEM, s, a);
        } else if (a.nq >= 128 && !(g_mm_debug & 2048)) {      // (also for short contexts: 11.3 vs 13.7 us at 256 queries x 32 keys)
            dim3 grid((a.nq + 127) / 128, a.H, a.B);
            hipLaunchKernelGGL(attention_kernel<8>, grid, dim3(512), 0, s, a);
        } else {
            dim3 grid((a.nq + 63) / 64, a.H, a.B);
            hipLaunchKernelGGL(attention_kernel<4>, grid, dim3(256), 0, s, a);
        }
    }
    return mm_check_launch("attention_kernel");
}
