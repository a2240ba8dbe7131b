// Cross-attention of the decode loop with the output projection folded into the values (round 4):
//     x += ( softmax(8 q^ . k^ + key mask) @ V ) W_o^T      (muse_maskgit_pytorch.py:139-162, context = the text encoding)
// The text context is the same at every decode step, so K, V and everything linear behind V are step-invariant:
//     (P_h V_h) W_o,h^T = P_h (V_h W_o,h^T) =: P_h VW_h          per head h, VW_h [keys][dim]
// Packed once per generate and layer (k_cross_fold_pack): K^ (l2-normalised, scaled, bf16, null key first) and VW (bf16, null value first) -- then a layer's
// cross-attention behind its q projection is ONE kernel instead of two (33-key attention + a 512 x 512 output projection with the fp32 residual, each of them
// pure launch latency at 8192 rows: 13 + 17 us): scores and softmax per head, then out[query][:] = P_flat[query][(head, key)] . VW_flat -- one MFMA contraction
// over the 8 x 36 = 288 (head, key) pairs -- the residual add, and the LayerNorm(dim)-fold producer outputs (bf16 row image + per-64-column statistics).
// Fewer flops too: 2 x 288 x 512 per query instead of 2 x 512 x 512 + the P V products.
//
// Shape class (the headline config): dim = inner = 512, 8 heads x 64, <= 35 context tokens (+ the null key = 36 keys per head).  Everything else takes
// the two-kernel path (model.hip cross_attn_block).
//
// One 512-thread workgroup = 32 queries of one sequence; grid = (ceil(nq / 32), sequences) = 256 workgroups at the base config.
//   phase 0  every wave requests ITS 64 output features of VW_flat^T as MFMA A fragments straight into registers (36 x 16 B per lane; the pack kernel stores
//            them fragment-major: one contiguous KiB per wave load).  They do not depend on the scores, so their L2 latency hides behind phases 1 and 2.
//   phase 1  wave h = head h: q^ fragments (l2norm * q_scale while loading, as attention.hip), S^T = K^ Q^T on the MFMA (3 key blocks x 2 query blocks),
//            mask, softmax over the head's <= 36 keys in registers (a lane owns 12 keys of ONE query), P -> bf16 -> LDS [32 queries][288] (592-byte rows:
//            conflict-free for the 8-byte writes and the 16-byte fragment reads).
//   phase 2  wave w = output features 64 w .. 64 w + 63: 9 k-blocks x (2 B-fragment reads + 8 MFMAs).
//   phase 3  accumulator fragment = 4 consecutive features of one query: residual add in place, bf16 image, and the wave's 64 columns ARE one statistics partial.
#include <string.h>

#include "common.h"
#include "muse_hip_internal.h"

namespace {

constexpr int XQ = 32;            // queries per workgroup
constexpr int XH = 8;             // heads (= waves)
constexpr int XKS = 36;           // keys per head in the flat (head, key) axis: null key + <= 35 context keys
constexpr int XKF = XH * XKS;     // 288 = 9 MFMA k-blocks of 32
constexpr int XKB = XKF / 32;
constexpr int XD = 512;           // model dim = 8 waves x 64 output features
constexpr int P_LD = 592;         // bytes per P row in LDS (288 bf16 + 16: rows start 20 banks apart -> 16 rows x 16 B cover the 64 banks exactly once)
constexpr float NEG_BIG = -3.0e38f;

__global__ __launch_bounds__(512, 2) void cross_fold_kernel(const CrossFoldArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char Ps[XQ * P_LD];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    // workgroup -> (sequence, query block): consecutive workgroup ids go round-robin to the 8 XCDs, each with its own L2.  All query blocks of one
    // sequence read the same 295 KB of VW fragments, so they are placed on ONE XCD (sequence b lives on XCD b % 8): the fragments come out of HBM / MALL
    // once per sequence instead of once per query block (round 4: 22.7 us -> see DESIGN.md, the kernel was bound by exactly that traffic)
    const int nqb = (p.nq + XQ - 1) / XQ;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = (slot / nqb) * 8 + xcd;
    if (b >= p.seqs) return;
    const int q0 = (slot % nqb) * XQ;
    const int kvb = p.kv_batch_mod > 0 ? b % p.kv_batch_mod : b;

    // key validity (the same for every query of the sequence): bit 0 = the null key (always attended, mmp.py:145-155), bit j = context token j - 1.  One byte per
    // lane and a ballot -> a wave-uniform 64-bit mask (twelve dependent byte loads per lane would each cost a memory round trip)
    unsigned long long valid64;
    {
        bool keep = lane < p.m;
        if (keep && p.key_mask) keep = p.key_mask[(size_t)b * p.km_sb + lane] != 0;
        valid64 = (__ballot(keep) << 1) | 1ull;
    }
    bool kvalid[3][4];
#pragma unroll
    for (int kb = 0; kb < 3; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) kvalid[kb][r] = (valid64 >> (kb * 16 + 4 * fg + r)) & 1ull;
    // ---- loads of phase 1 first (vector memory returns in order: waiting for them must not wait for the 36 fragment loads behind them).
    //      K^ fragments [kv sequence][head][3 key blocks][2 d-halves][64 lanes][8 bf16]; head = wave
    uint4 kf[3][2];
    {
        const uint4* kp = reinterpret_cast<const uint4*>(p.khat) + ((size_t)kvb * XH + w) * 6 * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < 3; ++kb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kf[kb][ks] = kp[(kb * 2 + ks) * 64];
    }
    float qs0[8], qs1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { qs0[j] = p.q_scale[8 * fg + j]; qs1[j] = p.q_scale[32 + 8 * fg + j]; }
    uint4 qraw[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + qb * 16 + fr;
        const bf16_t* qp = p.q + ((size_t)b * p.nq + (size_t)(qi < p.nq ? qi : 0)) * p.q_ld + w * 64;
        qraw[qb][0] = *reinterpret_cast<const uint4*>(qp + 8 * fg);
        qraw[qb][1] = *reinterpret_cast<const uint4*>(qp + 32 + 8 * fg);
    }
    // ---- phase 0: this wave's VW^T fragments (A operands of phase 2): [kv sequence][32 feature blocks][9 k-blocks][64 lanes][8 bf16]
    uint4 av[4][XKB];
    {
        const uint4* vp = reinterpret_cast<const uint4*>(p.vwt) + ((size_t)kvb * (XD / 16) + (size_t)w * 4) * XKB * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < XKB; ++kb)      // (requested in the order phase 2 consumes them)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) av[ob][kb] = vp[(ob * XKB + kb) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);      // (keep the request order: phase-1 operands, then the fragments)

    // ---- phase 1: head w
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float v0[8], v1[8];
        unpack8(qraw[qb][0], v0);
        unpack8(qraw[qb][1], v1);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += v0[j] * v0[j] + v1[j] * v1[j];
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);      // F.normalize eps (mmp.py:41-42)
#pragma unroll
        for (int j = 0; j < 8; ++j) { v0[j] = v0[j] * inv * qs0[j]; v1[j] = v1[j] * inv * qs1[j]; }
        const uint4 qf0 = pack8(v0), qf1 = pack8(v1);
        // S^T = K^ Q^T: a[r] = score(key 16 kb + 4 fg + r, query fr of this block)
        float s[3][4];
        float mx = NEG_BIG;
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) {
            f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
            a = mfma16(kf[kb][0], qf0, a);
            a = mfma16(kf[kb][1], qf1, a);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[kb][r] = kvalid[kb][r] ? a[r] * p.scale : NEG_BIG;
                mx = fmaxf(mx, s[kb][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 3; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[kb][r] = kvalid[kb][r] ? __expf(s[kb][r] - mx) : 0.f;      // (the null key is always valid: mx is a real score, sum >= 1)
                sum += s[kb][r];
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float linv = 1.f / sum;
        // P[query][36 w + key], keys 0 .. 35 of this head (the third key block only holds keys 32 .. 35: its lanes fg = 0)
        unsigned char* prow = Ps + (qb * 16 + fr) * P_LD + w * (XKS * 2);
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) {
            if (kb == 2 && fg != 0) continue;
            *reinterpret_cast<uint2*>(prow + (kb * 16 + 4 * fg) * 2) =
                make_uint2(pack_bf16x2(s[kb][0] * linv, s[kb][1] * linv), pack_bf16x2(s[kb][2] * linv, s[kb][3] * linv));
        }
    }
    // the residual rows of this wave's features (phase 3) are requested now: they arrive under phase 2
    float4 res[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + qb * 16 + fr;
        const float* xr = p.x + ((size_t)b * p.nq + (size_t)(qi < p.nq ? qi : 0)) * p.ldx + w * 64 + 4 * fg;
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) res[qb][ob] = *reinterpret_cast<const float4*>(xr + ob * 16);
    }
    __syncthreads();

    // ---- phase 2: out^T[feature][query] = VW^T . P^T over the 288 (head, key) pairs
    f32x4_t acc[4][2];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) acc[ob][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < XKB; ++kb) {
        uint4 pf[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) pf[qb] = *reinterpret_cast<const uint4*>(Ps + (qb * 16 + fr) * P_LD + kb * 64 + fg * 16);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) acc[ob][qb] = mfma16(av[ob][kb], pf[qb], acc[ob][qb]);
    }

    // ---- phase 3: x += out (fp32, in place), bf16 image + this wave's 64-column statistics partial of the new row (LayerNorm(dim) fold, producer side)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + qb * 16 + fr;
        const bool ok = qi < p.nq;
        const size_t row = (size_t)b * p.nq + (size_t)(ok ? qi : 0);
        // statistics in the canonical order of every fold producer (common.h row_stats16: a balanced tree over the 64 columns in natural order, so the
        // decode loop's null half -- whose rows get the constant null-pass row in the self-attention's output projection -- and the general path -- which
        // runs this kernel on the null pass too -- hand bit-identical (sum, sum of squares) to the next LayerNorm fold): columns 16 ob + 4 fg + r
        float q1[4], q2[4];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            float4 o;
            o.x = acc[ob][qb][0] + res[qb][ob].x; o.y = acc[ob][qb][1] + res[qb][ob].y;
            o.z = acc[ob][qb][2] + res[qb][ob].z; o.w = acc[ob][qb][3] + res[qb][ob].w;
            const int col = w * 64 + ob * 16 + 4 * fg;
            if (ok) {
                *reinterpret_cast<float4*>(p.x + row * p.ldx + col) = o;
                if (p.xb) *reinterpret_cast<uint2*>(p.xb + row * p.ldxb + col) = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
            } else {
                o = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            q1[ob] = (o.x + o.y) + (o.z + o.w);
            q2[ob] = (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
        }
        if (p.stp) {      // (wave-uniform)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {      // the four lane groups = the four 4-column shares of a 16-column block: (fg0 + fg1) + (fg2 + fg3)
                q1[ob] += __shfl_xor(q1[ob], 16, 64); q2[ob] += __shfl_xor(q2[ob], 16, 64);
                q1[ob] += __shfl_xor(q1[ob], 32, 64); q2[ob] += __shfl_xor(q2[ob], 32, 64);
            }
            const float s1 = (q1[0] + q1[1]) + (q1[2] + q1[3]), s2 = (q2[0] + q2[1]) + (q2[2] + q2[3]);
            if (ok && fg == 0) *reinterpret_cast<float2*>(p.stp + (row * p.st_np + w) * 2) = make_float2(s1, s2);
        }
    }
}

// ---- pack, once per generate and layer.  grid (kv sequences, heads, 2 halves of the features), 256 threads.
//   khat  [s][h][3 key blocks][2 d-halves][64 lanes][8]: K^ = bf16(k / max(|k|, eps) * k_scale) of key (16 kb + lane % 16), d = 32 ks + 8 (lane / 16) .. + 7; key 0 = the
//         null key (fp32 parameter, mmp.py:145-149), key j = context token j - 1 (the K half of ckv, bf16), keys > m zero
//   vwt   [s][32 feature blocks][9 k-blocks][64 lanes][8]: VW^T[feature 16 ob + lane % 16][flat k = 32 kb + 8 (lane / 16) .. + 7], flat k = 36 head + key,
//         VW[key][feature] = bf16( sum_d v[key][64 head + d] * W_o[feature][64 head + d] ), v[0] = bf16(null_v) like attention.hip, keys > m zero
__global__ __launch_bounds__(256) void cross_fold_pack_kernel(const bf16_t* __restrict__ ckv, int m, int I, const float* __restrict__ null_k, const float* __restrict__ null_v,
                                                               const float* __restrict__ k_scale, const bf16_t* __restrict__ w_out, int ldw, bf16_t* __restrict__ khat,
                                                               bf16_t* __restrict__ vwt) {
    __shared__ __attribute__((aligned(16))) float vs[XKS][64];
    const int s = blockIdx.x, h = blockIdx.y, half = blockIdx.z, t = threadIdx.x;
    // V_h (and, in the first half's workgroup, K^_h: threads 0 .. 47 take one key each)
    for (int i = t; i < XKS * 64; i += 256) {
        const int key = i >> 6, d = i & 63;
        float v = 0.f;
        if (key == 0) v = bf16_to_f32(f32_to_bf16(null_v[h * 64 + d]));
        else if (key <= m) v = bf16_to_f32(ckv[((size_t)s * m + key - 1) * 2 * I + I + h * 64 + d]);
        vs[key][d] = v;
    }
    if (half == 0 && t < 48) {
        const int key = t;
        float k[64];
        float ss = 0.f;
#pragma unroll
        for (int d = 0; d < 64; ++d) {
            float v = 0.f;
            if (key == 0) v = null_k[h * 64 + d];
            else if (key <= m) v = bf16_to_f32(ckv[((size_t)s * m + key - 1) * 2 * I + h * 64 + d]);
            k[d] = v;
            ss += v * v;
        }
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        bf16_t* kb_ = khat + ((size_t)s * XH + h) * 6 * 64 * 8;
#pragma unroll
        for (int d = 0; d < 64; ++d) {
            const int kb = key >> 4, frk = key & 15, ks = d >> 5, fgk = (d & 31) >> 3, j = d & 7;
            kb_[(((kb * 2 + ks) * 64) + fgk * 16 + frk) * 8 + j] = f32_to_bf16(k[d] * inv * k_scale[d]);
        }
    }
    __syncthreads();
    {
        const int o = half * 256 + t;      // one output feature per thread: its 64 weights of head h stay in registers, V_h is broadcast out of LDS 16 bytes at a time
        float wr[64];
        const bf16_t* wp = w_out + (size_t)o * ldw + h * 64;
#pragma unroll
        for (int d8 = 0; d8 < 8; ++d8) {
            float f[8];
            unpack8(*reinterpret_cast<const uint4*>(wp + d8 * 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) wr[d8 * 8 + j] = f[j];
        }
        const int ob = o >> 4, fro = o & 15;
        for (int key = 0; key < XKS; ++key) {
            float a = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 16; ++d4) {
                const float4 v = *reinterpret_cast<const float4*>(&vs[key][d4 * 4]);
                a = __builtin_fmaf(v.x, wr[d4 * 4], a); a = __builtin_fmaf(v.y, wr[d4 * 4 + 1], a);
                a = __builtin_fmaf(v.z, wr[d4 * 4 + 2], a); a = __builtin_fmaf(v.w, wr[d4 * 4 + 3], a);
            }
            const int kfl = h * XKS + key;
            const int kb = kfl >> 5, fgk = (kfl & 31) >> 3, j = kfl & 7;
            vwt[((((size_t)s * (XD / 16) + ob) * XKB + kb) * 64 + fgk * 16 + fro) * 8 + j] = f32_to_bf16(a);
        }
    }
}

// the null pass's cross-attention as a constant row: every text key masked -> P = 1 on each head's null key, so the kernel above adds sum over the heads (in
// head order: one non-zero product per MFMA k-block) of the bf16 VW entries of the null key.  The decode loop adds that row to the null half without running
// the kernel (model.hip); this reproduces the kernel's value bit for bit from the packed fragments of any sequence.
__global__ __launch_bounds__(256) void cross_fold_null_row_kernel(const bf16_t* __restrict__ vwt, float* __restrict__ out) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= XD) return;
    const int ob = o >> 4, fro = o & 15;
    float a = 0.f;
    for (int h = 0; h < XH; ++h) {
        const int kfl = h * XKS;
        const int kb = kfl >> 5, fgk = (kfl & 31) >> 3, j = kfl & 7;
        a += bf16_to_f32(vwt[((((size_t)ob) * XKB + kb) * 64 + fgk * 16 + fro) * 8 + j]);
    }
    out[o] = a;
}

}  // namespace

bool k_cross_fold_eligible(int D, int I, int H, int dh, int m) { return D == XD && I == XD && H == XH && dh == 64 && m >= 1 && m + 1 <= XKS; }
size_t k_cross_fold_khat_elems(int kv_seqs) { return (size_t)kv_seqs * XH * 6 * 64 * 8; }
size_t k_cross_fold_vwt_elems(int kv_seqs) { return (size_t)kv_seqs * XD * XKF; }

int k_cross_fold_pack(hipStream_t s, const bf16_t* ckv, int kv_seqs, int m, int I, const float* null_k, const float* null_v, const float* k_scale,
                      const bf16_t* w_out, int ldw, bf16_t* khat, bf16_t* vwt) {
    if (kv_seqs <= 0) return MM_OK;
    if (!null_k || !null_v || !k_scale) return mm_set_error(MM_ERR_SHAPE, "cross_fold_pack: null key / value and k_scale required");
    hipLaunchKernelGGL(cross_fold_pack_kernel, dim3(kv_seqs, XH, 2), dim3(256), 0, s, ckv, m, I, null_k, null_v, k_scale, w_out, ldw, khat, vwt);
    return mm_check_launch("cross_fold_pack_kernel");
}

int k_cross_fold_null_row(hipStream_t s, const bf16_t* vwt, float* out) {
    hipLaunchKernelGGL(cross_fold_null_row_kernel, dim3(XD / 256), dim3(256), 0, s, vwt, out);
    return mm_check_launch("cross_fold_null_row_kernel");
}

int k_cross_fold(hipStream_t s, const CrossFoldArgs& a) {
    if (a.seqs <= 0 || a.nq <= 0) return MM_OK;
    if (a.m < 1 || a.m + 1 > XKS) return mm_set_error(MM_ERR_SHAPE, "cross_fold: 1 <= context tokens <= 35");
    if ((a.ldx % 4) || (a.q_ld % 8) || (a.xb && (a.ldxb % 4))) return mm_set_error(MM_ERR_SHAPE, "cross_fold: unaligned rows");
    const int nqb = (a.nq + XQ - 1) / XQ;
    hipLaunchKernelGGL(cross_fold_kernel, dim3(8 * ((a.seqs + 7) / 8) * nqb), dim3(512), 0, s, a);
    return mm_check_launch("cross_fold_kernel");
}
