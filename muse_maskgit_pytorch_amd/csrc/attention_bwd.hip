// Backward of the Muse attention operator (muse_maskgit_pytorch.py:137-162 with attend.py:109-140 as the arithmetic
// definition; forward kernel: attention.hip) for gfx950.
//
// Forward, per (batch, head):  qn_i = l2norm(q_i) * q_scale,  kn_j = l2norm(k_j) * k_scale  (key 0 = the learned null key,
// normalised like the others; v_0 = null value),  s_ij = 8 * qn_i . kn_j (masked keys -> -inf),  p = softmax_j(s),  o_i = sum_j p_ij v_j.
// This kernel produces the gradients w.r.t. the NORMALISED operands and v:
//     dv_j  = sum_i p_ij do_i
//     ds_ij = p_ij (do_i . v_j - D_i),  D_i = do_i . o_i
//     dqn_i = 8 sum_j ds_ij kn_j,     dkn_j = 8 sum_i ds_ij qn_i
// (key 0's dkn / dv go to separate fp32 buffers: they are the null key/value gradients of this (batch, head)).  The l2norm /
// scale chain rule is a row kernel of its own (qk_norm_bwd_kernel below).
//
// One 256-thread workgroup per (batch, head); nq <= 256 queries (multiple of 64), keys in blocks of 64.
//   * every matrix product is v_mfma_f32_16x16x32_bf16.  The accumulator layout of a 16x16 block (lane: column = lane & 15,
//     rows 4*(lane >> 4) + r) IS an operand layout for a product that contracts over the block's row index: two blocks side by
//     side fill the 8 contraction slots of a lane (slot r -> row 4g + r of block 0, slot 4 + r -> the same row of block 1), the
//     other operand is read from a TRANSPOSED LDS copy with the same slot order.  So P and dS never leave registers.
//   * products that contract over queries (dV = P^T dO, dKn = dS^T Qn) need P / dS with the key on lane & 15, the one that
//     contracts over keys (dQn = dS Kn) needs the query on lane & 15: S and dP are therefore computed in both orientations
//     (7 block products instead of 5) rather than transposed through LDS.
//   * LDS (157 KiB): Qn, dO row-major (XOR-swizzled 128-byte rows) + their transposes (pitch +16 B: conflict-free ds_read_b64),
//     one 64-key block of Kn / V / Kn^T at a time, the per-query log-sum-exp and D.
//   * a first sweep over the key blocks computes the log-sum-exp per query (the forward does not save it).
// P and dS are rounded to bf16 as MFMA operands (like P in the forward); accumulation is fp32.
#include <math.h>

#include "common.h"
#include "muse_hip_internal.h"

namespace {

constexpr int QMAX = 256;
constexpr int TP_Q = QMAX + 8;        // transposed-copy pitch in elements (528 B)
constexpr int TP_K = 64 + 8;          // Kn^T block pitch (144 B)
constexpr int OFF_QN = 0;                                 // [256][64] bf16 swizzled
constexpr int OFF_DO = OFF_QN + QMAX * 128;               // [256][64]
constexpr int OFF_QNT = OFF_DO + QMAX * 128;              // [64][TP_Q]
constexpr int OFF_DOT = OFF_QNT + 64 * TP_Q * 2;          // [64][TP_Q]
constexpr int OFF_KN = OFF_DOT + 64 * TP_Q * 2;           // [64][64] swizzled
constexpr int OFF_V = OFF_KN + 64 * 128;                  // [64][64] swizzled
constexpr int OFF_KNT = OFF_V + 64 * 128;                 // [64][TP_K]
constexpr int OFF_LSE = OFF_KNT + 64 * TP_K * 2;          // float[256]
constexpr int OFF_DD = OFF_LSE + QMAX * 4;                // float[256]
constexpr int OFF_KB = OFF_DD + QMAX * 4;                 // float[64]: 0 for a live key, -inf for masked / out of range
constexpr int ABW_SMEM = OFF_KB + 64 * 4;

__device__ __forceinline__ int rm_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

struct AttnBwdArgs {
    const bf16_t* q; long q_sb, q_sh, q_sn;
    const bf16_t* k; long k_sb, k_sh, k_sn;
    const bf16_t* v; long v_sb, v_sh, v_sn;
    const bf16_t* o; long o_sb, o_sh, o_sn;
    const bf16_t* dout; long do_sb, do_sh, do_sn;
    bf16_t* dqn; long dq_sb, dq_sh, dq_sn;
    bf16_t* dkn; long dk_sb, dk_sh, dk_sn;
    bf16_t* dv; long dv_sb, dv_sh, dv_sn;
    float* dnk; float* dnv;                      // [B*H][64] gradients of the (normalised) null key / the null value
    int B, H, nq, nk;                            // nk real keys (null key excluded)
    const uint8_t* key_mask; long km_sb;
    const float* q_scale; const float* k_scale; const float* null_k; const float* null_v;
    float scale;
};

// fragment of a row-major swizzled [rows][64] array: rows r0..r0+15, contraction half ks (32 of the 64 dims)
__device__ __forceinline__ u32x4_t frag_rm(const unsigned char* base, int r0, int ks, int fr, int fg) {
    return *reinterpret_cast<const u32x4_t*>(base + rm_off(r0 + fr, ks * 4 + fg));
}
// fragment of a transposed [64][pitch] array for a 32-wide contraction span starting at column c0 in "two-block" slot order:
// slots 0..3 <- columns c0 + 4g .. +3, slots 4..7 <- columns c0 + 16 + 4g .. +3; rows d0 .. d0+15
__device__ __forceinline__ u32x4_t frag_tp(const unsigned char* base, int pitch_e, int d0, int c0, int fr, int fg) {
    const unsigned char* p = base + ((d0 + fr) * pitch_e + c0 + 4 * fg) * 2;
    const uint2 lo = *reinterpret_cast<const uint2*>(p);
    const uint2 hi = *reinterpret_cast<const uint2*>(p + 32);
    return u32x4_t{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ u32x4_t pack_two_blocks(const float (&a)[4], const float (&b)[4]) {
    return u32x4_t{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
}

template <int NQS>      // 16-query sub-blocks per wave: nq = 64 * NQS
__global__ __launch_bounds__(256) void attention_bwd_kernel(const AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lse_s = reinterpret_cast<float*>(smem + OFF_LSE);
    float* dd_s = reinterpret_cast<float*>(smem + OFF_DD);
    float* kbias = reinterpret_cast<float*>(smem + OFF_KB);
    bf16_t* qnt = reinterpret_cast<bf16_t*>(smem + OFF_QNT);
    bf16_t* dot = reinterpret_cast<bf16_t*>(smem + OFF_DOT);
    bf16_t* knt = reinterpret_cast<bf16_t*>(smem + OFF_KNT);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int nq = 64 * NQS;
    const int nkt = p.nk + 1;                       // keys including the null key at index 0
    const int nkb = (nkt + 63) >> 6;

    // ---- P0: this head's queries: normalise, stage Qn / dO in both orientations, D_i = do_i . o_i
    if (t < nq) {
        const bf16_t* qr = p.q + (long)b * p.q_sb + (long)h * p.q_sh + (long)t * p.q_sn;
        const bf16_t* dor = p.dout + (long)b * p.do_sb + (long)h * p.do_sh + (long)t * p.do_sn;
        const bf16_t* orow = p.o + (long)b * p.o_sb + (long)h * p.o_sh + (long)t * p.o_sn;
        float x[64];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float f[8];
            unpack8(*reinterpret_cast<const uint4*>(qr + c * 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { x[c * 8 + j] = f[j]; ss += f[j] * f[j]; }
        }
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);          // F.normalize eps (mmp.py:151)
        float dd = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = x[c * 8 + j] * inv * p.q_scale[c * 8 + j];
            const uint4 pk = pack8(f);
            *reinterpret_cast<uint4*>(smem + OFF_QN + rm_off(t, c)) = pk;
            const bf16_t* e = reinterpret_cast<const bf16_t*>(&pk);
#pragma unroll
            for (int j = 0; j < 8; ++j) qnt[(c * 8 + j) * TP_Q + t] = e[j];
            const uint4 dv = *reinterpret_cast<const uint4*>(dor + c * 8);
            *reinterpret_cast<uint4*>(smem + OFF_DO + rm_off(t, c)) = dv;
            const bf16_t* de = reinterpret_cast<const bf16_t*>(&dv);
            float df[8], of[8];
            unpack8(dv, df);
            unpack8(*reinterpret_cast<const uint4*>(orow + c * 8), of);
#pragma unroll
            for (int j = 0; j < 8; ++j) { dot[(c * 8 + j) * TP_Q + t] = de[j]; dd += df[j] * of[j]; }
        }
        dd_s[t] = dd;
    }

    // one 64-key block -> LDS (normalised keys row-major + transposed, values row-major, liveness bias); thread: row t >> 2, 16 dims
    auto load_kblock = [&](int kb) {
        const int r = t >> 2, part = t & 3;
        const int kidx = kb * 64 + r;
        const bool inr = kidx < nkt;
        float kx[16], vx[16];
        if (inr && kidx == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { kx[j] = p.null_k[h * 64 + part * 16 + j]; vx[j] = p.null_v[h * 64 + part * 16 + j]; }
        } else if (inr) {
            const bf16_t* kr = p.k + (long)b * p.k_sb + (long)h * p.k_sh + (long)(kidx - 1) * p.k_sn + part * 16;
            const bf16_t* vr = p.v + (long)b * p.v_sb + (long)h * p.v_sh + (long)(kidx - 1) * p.v_sn + part * 16;
            float f[8];
            unpack8(*reinterpret_cast<const uint4*>(kr), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) kx[j] = f[j];
            unpack8(*reinterpret_cast<const uint4*>(kr + 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) kx[8 + j] = f[j];
            unpack8(*reinterpret_cast<const uint4*>(vr), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) vx[j] = f[j];
            unpack8(*reinterpret_cast<const uint4*>(vr + 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) vx[8 + j] = f[j];
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) { kx[j] = 0.f; vx[j] = 0.f; }
        }
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) ss += kx[j] * kx[j];
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        float kn[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) kn[j] = inr ? kx[j] * inv * p.k_scale[part * 16 + j] : 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float f[8], g[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { f[j] = kn[c * 8 + j]; g[j] = vx[c * 8 + j]; }
            const uint4 pk = pack8(f);
            *reinterpret_cast<uint4*>(smem + OFF_KN + rm_off(r, part * 2 + c)) = pk;
            *reinterpret_cast<uint4*>(smem + OFF_V + rm_off(r, part * 2 + c)) = pack8(g);
            const bf16_t* e = reinterpret_cast<const bf16_t*>(&pk);
#pragma unroll
            for (int j = 0; j < 8; ++j) knt[(part * 16 + c * 8 + j) * TP_K + r] = e[j];
        }
        if (part == 0) {
            bool live = inr;
            if (live && kidx > 0 && p.key_mask) live = p.key_mask[(long)b * p.km_sb + (kidx - 1)] != 0;
            kbias[r] = live ? 0.f : -INFINITY;
        }
    };

    // ---- P1: log-sum-exp per query.  Wave w owns queries [16*NQS*w, +16*NQS) in the key-major orientation (lane: query = lane & 15).
    const int qw0 = w * 16 * NQS;
    float m_run[NQS], l_run[NQS];
#pragma unroll
    for (int i = 0; i < NQS; ++i) { m_run[i] = -INFINITY; l_run[i] = 0.f; }
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();
        load_kblock(kb);
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < NQS; ++qq) {
            const u32x4_t qf0 = frag_rm(smem + OFF_QN, qw0 + qq * 16, 0, fr, fg), qf1 = frag_rm(smem + OFF_QN, qw0 + qq * 16, 1, fr, fg);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                f32x4_t c = {0.f, 0.f, 0.f, 0.f};
                c = mfma16(frag_rm(smem + OFF_KN, kk * 16, 0, fr, fg), qf0, c);
                c = mfma16(frag_rm(smem + OFF_KN, kk * 16, 1, fr, fg), qf1, c);
                const float4 kb4 = *reinterpret_cast<const float4*>(kbias + kk * 16 + 4 * fg);
                const float s0 = p.scale * c[0] + kb4.x, s1 = p.scale * c[1] + kb4.y, s2 = p.scale * c[2] + kb4.z, s3 = p.scale * c[3] + kb4.w;
                const float bm = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
                if (bm > -INFINITY) {
                    const float nm = fmaxf(m_run[qq], bm);
                    l_run[qq] = l_run[qq] * __expf(m_run[qq] - nm) + __expf(s0 - nm) + __expf(s1 - nm) + __expf(s2 - nm) + __expf(s3 - nm);
                    m_run[qq] = nm;
                }
            }
        }
    }
#pragma unroll
    for (int qq = 0; qq < NQS; ++qq) {
        float m = m_run[qq], l = l_run[qq];
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float om = __shfl_xor(m, o, 64), ol = __shfl_xor(l, o, 64);
            const float nm = fmaxf(m, om);
            l = (nm == -INFINITY) ? 0.f : l * __expf(m - nm) + ol * __expf(om - nm);
            m = nm;
        }
        if (fg == 0) lse_s[qw0 + qq * 16 + fr] = m + __logf(l);       // the null key is always live: l > 0
    }

    // ---- P2: gradients, one key block at a time
    f32x4_t dq_acc[NQS][4];
#pragma unroll
    for (int i = 0; i < NQS; ++i)
#pragma unroll
        for (int d = 0; d < 4; ++d) dq_acc[i][d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const long bh = (long)b * p.H + h;
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();                 // (also orders the lse_s writes above before the first use below)
        load_kblock(kb);
        __syncthreads();
        // (a) query-major blocks (lane: key = lane & 15 of this wave's 16 keys): dV and dKn for keys kb*64 + 16w ..
        {
            f32x4_t dv_acc[4], dk_acc[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) { dv_acc[d] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dk_acc[d] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
            const u32x4_t kf0 = frag_rm(smem + OFF_KN, w * 16, 0, fr, fg), kf1 = frag_rm(smem + OFF_KN, w * 16, 1, fr, fg);
            const u32x4_t vf0 = frag_rm(smem + OFF_V, w * 16, 0, fr, fg), vf1 = frag_rm(smem + OFF_V, w * 16, 1, fr, fg);
            const float kbv = kbias[w * 16 + fr];
            for (int q0 = 0; q0 < nq; q0 += 32) {
                float pr[2][4], ds[2][4];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const int qb = q0 + blk * 16;
                    f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    s = mfma16(frag_rm(smem + OFF_QN, qb, 0, fr, fg), kf0, s);
                    s = mfma16(frag_rm(smem + OFF_QN, qb, 1, fr, fg), kf1, s);
                    dp = mfma16(frag_rm(smem + OFF_DO, qb, 0, fr, fg), vf0, dp);
                    dp = mfma16(frag_rm(smem + OFF_DO, qb, 1, fr, fg), vf1, dp);
                    const float4 l4 = *reinterpret_cast<const float4*>(lse_s + qb + 4 * fg);
                    const float4 d4 = *reinterpret_cast<const float4*>(dd_s + qb + 4 * fg);
                    const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __expf(p.scale * s[r] + kbv - ls[r]);      // masked key: exp(-inf) = 0
                        pr[blk][r] = pv;
                        ds[blk][r] = p.scale * pv * (dp[r] - dd[r]);
                    }
                }
                const u32x4_t p_op = pack_two_blocks(pr[0], pr[1]), ds_op = pack_two_blocks(ds[0], ds[1]);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    dv_acc[d] = mfma16(p_op, frag_tp(smem + OFF_DOT, TP_Q, d * 16, q0, fr, fg), dv_acc[d]);
                    dk_acc[d] = mfma16(ds_op, frag_tp(smem + OFF_QNT, TP_Q, d * 16, q0, fr, fg), dk_acc[d]);
                }
            }
            // lane: d = 16*dblk + (lane & 15), key = kb*64 + 16w + 4*(lane >> 4) + r
#pragma unroll
            for (int d = 0; d < 4; ++d) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kidx = kb * 64 + w * 16 + 4 * fg + r;
                    const int dc = d * 16 + fr;
                    if (kidx == 0) {
                        p.dnk[bh * 64 + dc] = dk_acc[d][r];
                        p.dnv[bh * 64 + dc] = dv_acc[d][r];
                    } else if (kidx < nkt) {
                        p.dkn[(long)b * p.dk_sb + (long)h * p.dk_sh + (long)(kidx - 1) * p.dk_sn + dc] = f32_to_bf16(dk_acc[d][r]);
                        p.dv[(long)b * p.dv_sb + (long)h * p.dv_sh + (long)(kidx - 1) * p.dv_sn + dc] = f32_to_bf16(dv_acc[d][r]);
                    }
                }
            }
        }
        // (b) key-major blocks (lane: query = lane & 15 of this wave's queries): dQn
#pragma unroll
        for (int qq = 0; qq < NQS; ++qq) {
            const int qb = qw0 + qq * 16;
            const u32x4_t qf0 = frag_rm(smem + OFF_QN, qb, 0, fr, fg), qf1 = frag_rm(smem + OFF_QN, qb, 1, fr, fg);
            const u32x4_t of0 = frag_rm(smem + OFF_DO, qb, 0, fr, fg), of1 = frag_rm(smem + OFF_DO, qb, 1, fr, fg);
            const float ls = lse_s[qb + fr], ddv = dd_s[qb + fr];
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                float ds[2][4];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const int k0 = kp * 32 + blk * 16;
                    f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    s = mfma16(frag_rm(smem + OFF_KN, k0, 0, fr, fg), qf0, s);
                    s = mfma16(frag_rm(smem + OFF_KN, k0, 1, fr, fg), qf1, s);
                    dp = mfma16(frag_rm(smem + OFF_V, k0, 0, fr, fg), of0, dp);
                    dp = mfma16(frag_rm(smem + OFF_V, k0, 1, fr, fg), of1, dp);
                    const float4 kb4 = *reinterpret_cast<const float4*>(kbias + k0 + 4 * fg);
                    const float kbv[4] = {kb4.x, kb4.y, kb4.z, kb4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __expf(p.scale * s[r] + kbv[r] - ls);
                        ds[blk][r] = p.scale * pv * (dp[r] - ddv);
                    }
                }
                const u32x4_t ds_op = pack_two_blocks(ds[0], ds[1]);
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    dq_acc[qq][d] = mfma16(ds_op, frag_tp(smem + OFF_KNT, TP_K, d * 16, kp * 32, fr, fg), dq_acc[qq][d]);
            }
        }
    }
    // lane: d = 16*dblk + (lane & 15), query = qw0 + 16*qq + 4*(lane >> 4) + r
#pragma unroll
    for (int qq = 0; qq < NQS; ++qq)
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = qw0 + qq * 16 + 4 * fg + r;
                p.dqn[(long)b * p.dq_sb + (long)h * p.dq_sh + (long)qi * p.dq_sn + d * 16 + fr] = f32_to_bf16(dq_acc[qq][d][r]);
            }
}

// ------------------------------------------------------------------------------------------------ l2norm * scale backward
// y = x / max(|x|, eps) * s   ->   dx = (g - xh (xh . g)) / max(|x|, eps) with g = dy * s, xh = x / max(|x|, eps);  ds += dy * xh.
// One row of 64 per 16 lanes (4 values each); per-workgroup partials of ds, reduced by colsum afterwards.  `x` rows are bf16
// (projection outputs) or, for the null key, fp32 parameters broadcast over the batch (x_f32 [H][64], row -> head = row % H).
__global__ __launch_bounds__(256) void qk_norm_bwd_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ x_f32, int H,
                                                          const bf16_t* __restrict__ dy, long lddy, const float* __restrict__ dy_f32,
                                                          const float* __restrict__ scale, long rows, int heads_per_row,
                                                          bf16_t* __restrict__ dx, long lddx, float* __restrict__ dx_f32,
                                                          float* __restrict__ dscale_part) {
    __shared__ float red[16][64];
    const int t = threadIdx.x, sub = t & 15, grp = t >> 4;      // 16 row-slots per workgroup, 16 lanes x 4 dims per row
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const long nvec = rows * heads_per_row;                     // one "vector" = one head's 64 dims of one row
    for (long vi = (long)blockIdx.x * 16 + grp; vi < nvec; vi += (long)gridDim.x * 16) {
        const long row = vi / heads_per_row;
        const int hh = (int)(vi % heads_per_row);
        float xv[4], gy[4];
        if (x_f32) {
            const float4 a = *reinterpret_cast<const float4*>(x_f32 + (long)(vi % H) * 64 + sub * 4);
            xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w;
        } else {
            const uint2 a = *reinterpret_cast<const uint2*>(x + row * ldx + hh * 64 + sub * 4);
            xv[0] = bf16lo(a.x); xv[1] = bf16hi(a.x); xv[2] = bf16lo(a.y); xv[3] = bf16hi(a.y);
        }
        if (dy_f32) {
            const float4 a = *reinterpret_cast<const float4*>(dy_f32 + vi * 64 + sub * 4);
            gy[0] = a.x; gy[1] = a.y; gy[2] = a.z; gy[3] = a.w;
        } else {
            const uint2 a = *reinterpret_cast<const uint2*>(dy + row * lddy + hh * 64 + sub * 4);
            gy[0] = bf16lo(a.x); gy[1] = bf16hi(a.x); gy[2] = bf16lo(a.y); gy[3] = bf16hi(a.y);
        }
        const float4 s4 = *reinterpret_cast<const float4*>(scale + sub * 4);
        const float sc[4] = {s4.x, s4.y, s4.z, s4.w};
        float ss = (xv[0] * xv[0] + xv[1] * xv[1]) + (xv[2] * xv[2] + xv[3] * xv[3]);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o, 64);
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        float xh[4], g[4], dotv = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { xh[j] = xv[j] * inv; g[j] = gy[j] * sc[j]; dotv += xh[j] * g[j]; acc[j] += gy[j] * xh[j]; }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) dotv += __shfl_xor(dotv, o, 64);
        float o4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o4[j] = (g[j] - xh[j] * dotv) * inv;
        if (dx_f32) *reinterpret_cast<float4*>(dx_f32 + vi * 64 + sub * 4) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        else *reinterpret_cast<uint2*>(dx + row * lddx + hh * 64 + sub * 4) = make_uint2(pack_bf16x2(o4[0], o4[1]), pack_bf16x2(o4[2], o4[3]));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[grp][sub * 4 + j] = acc[j];
    __syncthreads();
    if (t < 64) {
        float s = 0.f;
        for (int g2 = 0; g2 < 16; ++g2) s += red[g2][t];
        dscale_part[(long)blockIdx.x * 64 + t] = s;
    }
}

}  // namespace

int k_attention_bwd(hipStream_t s, const bf16_t* q, long q_sb, long q_sh, long q_sn, const bf16_t* k, long k_sb, long k_sh, long k_sn,
                    const bf16_t* v, long v_sb, long v_sh, long v_sn, const bf16_t* o, long o_sb, long o_sh, long o_sn,
                    const bf16_t* dout, long do_sb, long do_sh, long do_sn, bf16_t* dqn, long dq_sb, long dq_sh, long dq_sn,
                    bf16_t* dkn, long dk_sb, long dk_sh, long dk_sn, bf16_t* dv, long dv_sb, long dv_sh, long dv_sn, float* dnk, float* dnv,
                    int B, int H, int nq, int nk, const uint8_t* key_mask, long km_sb, const float* q_scale, const float* k_scale,
                    const float* null_k, const float* null_v, float scale) {
    if (B <= 0 || H <= 0) return MM_OK;
    if (nq != 64 && nq != 128 && nq != 256) return mm_set_error(MM_ERR_SHAPE, "attention_bwd: nq must be 64, 128 or 256 (longer sequences: later scope)");
    if (nk < 0 || !null_k || !null_v || !q_scale || !k_scale) return mm_set_error(MM_ERR_SHAPE, "attention_bwd: the Muse form (l2norm, scales, null kv) is required");
    AttnBwdArgs a;
    a.q = q; a.q_sb = q_sb; a.q_sh = q_sh; a.q_sn = q_sn;
    a.k = k; a.k_sb = k_sb; a.k_sh = k_sh; a.k_sn = k_sn;
    a.v = v; a.v_sb = v_sb; a.v_sh = v_sh; a.v_sn = v_sn;
    a.o = o; a.o_sb = o_sb; a.o_sh = o_sh; a.o_sn = o_sn;
    a.dout = dout; a.do_sb = do_sb; a.do_sh = do_sh; a.do_sn = do_sn;
    a.dqn = dqn; a.dq_sb = dq_sb; a.dq_sh = dq_sh; a.dq_sn = dq_sn;
    a.dkn = dkn; a.dk_sb = dk_sb; a.dk_sh = dk_sh; a.dk_sn = dk_sn;
    a.dv = dv; a.dv_sb = dv_sb; a.dv_sh = dv_sh; a.dv_sn = dv_sn;
    a.dnk = dnk; a.dnv = dnv; a.B = B; a.H = H; a.nq = nq; a.nk = nk; a.key_mask = key_mask; a.km_sb = km_sb;
    a.q_scale = q_scale; a.k_scale = k_scale; a.null_k = null_k; a.null_v = null_v; a.scale = scale;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, ABW_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, ABW_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, ABW_SMEM);
        if (e != hipSuccess) return mm_set_hip_error(e, "attention_bwd hipFuncSetAttribute");
        attr_set = true;
    }
    const dim3 grid(B * H), block(256);
    if (nq == 64) hipLaunchKernelGGL(attention_bwd_kernel<1>, grid, block, ABW_SMEM, s, a);
    else if (nq == 128) hipLaunchKernelGGL(attention_bwd_kernel<2>, grid, block, ABW_SMEM, s, a);
    else hipLaunchKernelGGL(attention_bwd_kernel<4>, grid, block, ABW_SMEM, s, a);
    return mm_check_launch("attention_bwd_kernel");
}

long k_qk_norm_bwd_blocks(long nvec) {
    long b = (nvec + 15) / 16;
    return b > 1024 ? 1024 : (b < 1 ? 1 : b);      // (round 4: was 128 -- half the CUs idle and 32 serial vectors per 16-lane group at the base size: 22.8 us per call)
}

int k_qk_norm_bwd(hipStream_t s, const bf16_t* x, long ldx, const float* x_f32, int H, const bf16_t* dy, long lddy, const float* dy_f32,
                  const float* scale, long rows, int heads_per_row, bf16_t* dx, long lddx, float* dx_f32, float* dscale_part) {
    if (rows <= 0) return MM_OK;
    const long nvec = rows * heads_per_row;
    hipLaunchKernelGGL(qk_norm_bwd_kernel, dim3((unsigned)k_qk_norm_bwd_blocks(nvec)), dim3(256), 0, s, x, ldx, x_f32, H, dy, lddy, dy_f32, scale,
                       rows, heads_per_row, dx, lddx, dx_f32, dscale_part);
    return mm_check_launch("qk_norm_bwd_kernel");
}
