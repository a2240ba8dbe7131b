// Cross-attention block of the decode loop as ONE kernel (round 4):
//     x += ( softmax(8 q^ . k^ + key mask) @ V ) W_o^T,   q = LayerNorm(x) W_q^T      (muse_maskgit_pytorch.py:139-162, context = the text encoding)
// Two observations.  (1) The text context is the same at every decode step, so K, V and everything linear behind V are step-invariant:
//     (P_h V_h) W_o,h^T = P_h (V_h W_o,h^T) =: P_h VW_h          per head h, VW_h [keys][dim]
// K^ (l2-normalised, scaled, bf16, null key first) and VW (bf16, null value first) are packed once per generate and layer (k_cross_fold_pack), and the
// attention's P V product + the 512 x 512 output projection become one MFMA contraction over the 8 x 36 = 288 (head, key) pairs: fewer flops (2 x 288 x 512
// per query instead of 2 x 512 x 512 + P V) and no second kernel.  (2) At 8192 rows the q projection, the 33-key attention and the output projection were
// three launches of pure latency (17 + 13 + 17 us, matrix pipe 0.12 / 0.04); a workgroup that owns 32 complete rows can run all of it: q for its rows (the
// LayerNorm(dim) fold's consumer side: raw bf16 rows x gain-folded weight, rstd * (acc - mean * c1) + c2), scores, softmax, P . VW, the residual add, and the
// fold's producer outputs for the feed-forward behind it.
//
// Shape class (the headline config): dim = inner = 512, 8 heads x 64, LayerNorm(dim) fold on, and -- two instantiations, template parameter KS = key slots per head --
//   KS = 36: <= 35 context tokens (+ the null key); the wave's 36 VW^T fragments of phase D stay in registers (round 4);
//   KS = 80: <= 79 context tokens (round 5: the reference pads to the longest prompt, t5.py:78-79 -- 77 is a common text length): 20 MFMA k-blocks over the 640
//            (head, key) pairs, the VW^T fragments STREAMED through a ring of six k-blocks like the q weight in phase B (655 KB per sequence instead of 295).
// Beyond 79 tokens the fold stops paying: 2 x 8 KS x 512 flops and 8 KS x 512 x 2 bytes per sequence grow with KS, the unfolded P V + output projection do
// not (at KS = 80 the fold already costs 655 k vs 606 k flops per query and 655 KB of VW against 512 KB of W_o shared by ALL sequences); the super-resolution
// context (256 condition tokens + text) and text_len 256 take the three-kernel path (model.hip cross_attn_block), as does everything outside this class.
//
// One 512-thread workgroup = 32 queries of one sequence; all query blocks of a sequence sit on ONE XCD (they read the same 295 KB of VW fragments: out of HBM /
// MALL once per sequence instead of once per query block: 22.7 -> 17.8 us when this was measured on the two-kernel form).
//   phase A  the 32 raw rows (bf16 image of the residual stream) -> LDS (1056-byte rows: conflict-free fragment reads)
//   phase B  wave h = head h: q_h = rows . Wq_h^T, 16 k-blocks x (2 B-fragment reads + 8 MFMAs); the weight fragments stream from L2 straight into registers
//            (fragment-major pack: one contiguous KiB per wave load), a ring of five k-blocks in flight per wave
//   phase C  fold epilogue on the accumulators, l2norm * q_scale over the head's 64 features (16 per lane + two lane exchanges), q^ -> bf16.  An accumulator
//            fragment (4 consecutive features of one query per lane) IS the B operand of v_mfma_f32_16x16x16_bf16: S^T = K^ Q^T needs no transposition.
//            Mask, softmax over the head's <= 36 keys in registers, P -> bf16 -> LDS [32 queries][288] (608-byte rows).  The wave's 36 VW^T fragments
//            (phase D's A operands) are requested at the start of this phase.
//   phase D  wave w = output features 64 w .. 64 w + 63: 9 k-blocks x (2 B-fragment reads + 8 MFMAs)
//   phase E  accumulator fragment = 4 consecutive features of one query: residual add in place, bf16 image, and the wave's 64 columns ARE one statistics partial
#include <string.h>

#include "common.h"
#include "muse_hip_internal.h"

namespace {

constexpr int XH = 8;             // heads (= waves)
constexpr int XD = 512;           // model dim = 8 waves x 64 output features
constexpr int X_LD = 1056;        // bytes per raw row in LDS: 512 bf16 + 32.  A ds_read_b128 is served in four groups of 16 lanes that are NOT lane-contiguous ({0-3, 12-15, 20-27}, ..:
                                  // MI355X_MICROARCH.md, LDS table); with lane (fr, fg) reading row fr, chunk fg, a row stride of 64 k + 32 bytes puts every group on 16 distinct 16-byte
                                  // slots.  Round 4's 1040 / 592 (64 k + 16) were laid out for lane-contiguous groups and cost a 2-way conflict on every fragment read (SQ counters: 0.34)
constexpr int XKS_MAX = 80;       // the largest instantiation's key slots (workspace sizing)
template <int KS>                 // key slots per head in the flat (head, key) axis: the null key + <= KS - 1 context keys (KS % 4 == 0: 8 KS is a whole number of 32-deep k-blocks)
struct XGeo {
    static constexpr int XKF = XH * KS;          // 288 / 640 flat (head, key) pairs
    static constexpr int XKB = XKF / 32;         // 9 / 20 MFMA k-blocks
    static constexpr int NKB = (KS + 15) / 16;   // 3 / 5 key blocks of the 16 x 16 x 16 score MFMA
    static constexpr int P_LD = XKF * 2 + 32;    // bytes per P row in LDS (608 / 1312 = 64 k + 32: conflict-free fragment reads, see X_LD)
    static constexpr int RD = XKB <= 9 ? XKB : 6;   // VW^T k-blocks held (KS = 36: all of them) or in flight (ring) per wave in phase D
    static constexpr int SMEM = 32 * X_LD + 32 * P_LD;
    static_assert(XKF % 32 == 0, "whole k-blocks");
};
constexpr float NEG_BIG = -3.0e38f;

typedef __attribute__((ext_vector_type(4))) short bf16x4s_t;
__device__ __forceinline__ f32x4_t mfma16k16(const uint2& a, const uint2& b, f32x4_t c) {      // 16 x 16 x 16: 4 consecutive k per lane
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16x4s_t, a), __builtin_bit_cast(bf16x4s_t, b), c, 0, 0, 0);
}

constexpr int NQB = 2;            // 16-query blocks per workgroup (32 queries).  A 64-query form -- the packed operands streamed once for twice the rows, half as many
                                  // workgroups -- measured 31.5 vs 24.3 us per launch in round 4 (the operand feed is bound per CU: idling half of the CUs loses) and was removed
template <int KS>
__global__ __launch_bounds__(512, 2) void cross_fold_kernel(const CrossFoldArgs p) {
    using G = XGeo<KS>;
    constexpr int XKB = G::XKB, NKB = G::NKB, P_LD = G::P_LD, RD = G::RD;
    constexpr int XQ = 16 * NQB;
    constexpr int RING = 5;       // k-blocks of q-weight fragments in flight per wave (register budget: 256)
    extern __shared__ __attribute__((aligned(16))) unsigned char xf_smem[];
    unsigned char* Xs = xf_smem;                   // [XQ][X_LD]
    unsigned char* Ps = xf_smem + XQ * X_LD;       // [XQ][P_LD]
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    // workgroup -> (sequence, query block): consecutive workgroup ids go round-robin to the 8 XCDs, each with its own L2.  All query blocks of one
    // sequence read the same 295 KB of VW fragments, so they are placed on ONE XCD (sequence b lives on XCD b % 8)
    const int nqb = (p.nq + XQ - 1) / XQ;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = (slot / nqb) * 8 + xcd;
    if (b >= p.seqs) return;
    const int q0 = (slot % nqb) * XQ;
    const int kvb = p.kv_batch_mod > 0 ? b % p.kv_batch_mod : b;
    const size_t row0 = (size_t)b * p.nq;

    // ---- requests, oldest first (vector memory returns in order): the key-mask byte and the rows' statistics partials, the raw rows, then the first five
    //      k-blocks of this wave's q-weight fragments.  [head][4 feature blocks][16 k-blocks][64 lanes][8 bf16]: a wave load = one contiguous KiB
    int kmb = 1, kmb1 = 1;
    if (p.key_mask && lane < p.m) kmb = p.key_mask[(size_t)b * p.km_sb + lane];
    if (KS > 64 && p.key_mask && 64 + lane < p.m) kmb1 = p.key_mask[(size_t)b * p.km_sb + 64 + lane];
    float4 part[NQB][4];      // 8 partials (sum, sum of squares) per row: dim 512 = 8 x 64 columns
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const int qi = q0 + qb * 16 + fr;
        const float4* pp = reinterpret_cast<const float4*>(p.stp_in + (row0 + (size_t)(qi < p.nq ? qi : 0)) * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) part[qb][i] = pp[i];
    }
    // K^ fragments of head w for the 16 x 16 x 16 MFMA [kv sequence][head][3 key blocks][4 d-blocks][64 lanes][4 bf16], and the fold / scale constants of
    // this lane's 16 features (64 w + 16 ob + 4 fg + r)
    uint2 kf[NKB][4];
    auto load_kf = [&]() {
        const uint2* kp = reinterpret_cast<const uint2*>(p.khat) + ((size_t)kvb * XH + w) * (NKB * 4) * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) kf[kb][ob] = kp[(kb * 4 + ob) * 64];
    };
    load_kf();
    float4 c1[4], c2[4], qs[4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        c1[ob] = *reinterpret_cast<const float4*>(p.c1 + w * 64 + ob * 16 + 4 * fg);
        c2[ob] = p.c2 ? *reinterpret_cast<const float4*>(p.c2 + w * 64 + ob * 16 + 4 * fg) : make_float4(0.f, 0.f, 0.f, 0.f);      // (NULL: beta = 0)
        qs[ob] = *reinterpret_cast<const float4*>(p.q_scale + ob * 16 + 4 * fg);
    }
    uint4 xr[2 * NQB];
#pragma unroll
    for (int i = 0; i < 2 * NQB; ++i) {
        const int c = t + 512 * i, r = c >> 6, ch = c & 63;
        const int qi = q0 + r;
        xr[i] = *reinterpret_cast<const uint4*>(p.xb_in + (row0 + (size_t)(qi < p.nq ? qi : 0)) * p.ldxb_in + ch * 8);
    }
    const uint4* wp = reinterpret_cast<const uint4*>(p.wqf) + (size_t)w * 4 * 16 * 64 + lane;
    uint4 wa[RING][4];
#pragma unroll
    for (int kb = 0; kb < RING; ++kb)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) wa[kb][ob] = wp[(ob * 16 + kb) * 64];
    // ---- phase A: raw rows -> LDS (rows past nq: the sequence's first row, never stored)
#pragma unroll
    for (int i = 0; i < 2 * NQB; ++i) {
        const int c = t + 512 * i, r = c >> 6, ch = c & 63;
        *reinterpret_cast<uint4*>(Xs + r * X_LD + ch * 16) = xr[i];
    }
    // key validity (the same for every query of the sequence): bit 0 = the null key (always attended, mmp.py:145-155), bit j = context token j - 1: a
    // wave-uniform 64-bit mask from one byte per lane
    asm volatile("" : "+v"(kmb));      // (the byte is consumed HERE, behind the row loads it was requested in front of -- not right behind its own request)
    const unsigned long long tok_lo = __ballot(lane < p.m && kmb != 0);
    const unsigned long long valid64 = (tok_lo << 1) | 1ull;                                                         // keys 0 .. 63
    const unsigned long long valid_hi = KS > 64 ? ((tok_lo >> 63) | (__ballot(64 + lane < p.m && kmb1 != 0) << 1)) : 0ull;      // keys 64 .. 127
    // this lane's two queries: LayerNorm statistics of their raw rows (fold, consumer side; common.h ln_rstd_negmean on the partials read above)
    float2 lnst[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { s1 += part[qb][i].x; s2 += part[qb][i].y; s1 += part[qb][i].z; s2 += part[qb][i].w; }      // (partials in index order, as ln_rstd_negmean)
        const float mean = s1 * (1.f / (float)XD);
        const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * (1.f / (float)XD)), 0.f);
        lnst[qb] = make_float2(__builtin_amdgcn_rsqf(var + 1e-5f), -mean);
    }
    __syncthreads();      // the rows are in LDS
    // ---- phase B: q_h = rows . Wq_h^T (head = wave), the weight fragments RING k-blocks ahead
    f32x4_t accq[4][NQB];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) accq[ob][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        uint4 bx[NQB];
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) bx[qb] = *reinterpret_cast<const uint4*>(Xs + (qb * 16 + fr) * X_LD + kb * 64 + fg * 16);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) accq[ob][qb] = mfma16(wa[kb % RING][ob], bx[qb], accq[ob][qb]);
        if (kb + RING < 16) {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) wa[kb % RING][ob] = wp[(ob * 16 + kb + RING) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);      // (the refill is requested HERE: the scheduler otherwise sinks it to its use five k-blocks later and waits for it there)
    }
    // ---- phase C: fold epilogue + l2norm * q_scale: lane holds features 64 w + 16 ob + 4 fg + r of query fr (per query block)
    uint2 qh[4][NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const float rstd = lnst[qb].x, nmean = lnst[qb].y;
        float qv[4][4];
        float ss = 0.f;
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            qv[ob][0] = ln_fold_apply(accq[ob][qb][0], rstd, nmean, c1[ob].x, c2[ob].x);
            qv[ob][1] = ln_fold_apply(accq[ob][qb][1], rstd, nmean, c1[ob].y, c2[ob].y);
            qv[ob][2] = ln_fold_apply(accq[ob][qb][2], rstd, nmean, c1[ob].z, c2[ob].z);
            qv[ob][3] = ln_fold_apply(accq[ob][qb][3], rstd, nmean, c1[ob].w, c2[ob].w);
            ss += (qv[ob][0] * qv[ob][0] + qv[ob][1] * qv[ob][1]) + (qv[ob][2] * qv[ob][2] + qv[ob][3] * qv[ob][3]);
        }
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);      // F.normalize eps (mmp.py:41-42)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
            qh[ob][qb] = make_uint2(pack_bf16x2(qv[ob][0] * inv * qs[ob].x, qv[ob][1] * inv * qs[ob].y),
                                    pack_bf16x2(qv[ob][2] * inv * qs[ob].z, qv[ob][3] * inv * qs[ob].w));
    }
    __builtin_amdgcn_sched_barrier(0);
    // this wave's VW^T fragments (A operands of phase D) are requested now, behind the q accumulators' registers: [kv sequence][32 feature blocks][9 k-blocks]
    // [64 lanes][8 bf16]; they arrive under the scores and the softmax.  (64-query form: in two batches, one in front of each pair of query blocks -- all 36
    // fragments + the four blocks' q^ + K^ do not fit the register file at once -- NQB == 2: one batch)
    uint4 av[4][RD];
    const uint4* vp = reinterpret_cast<const uint4*>(p.vwt) + ((size_t)kvb * (XD / 16) + (size_t)w * 4) * XKB * 64 + lane;
#pragma unroll
    for (int qp = 0; qp < NQB / 2; ++qp) {
#pragma unroll
        for (int kb = 0; kb < RD; ++kb)      // (requested in the order phase D consumes them; KS = 80: the first six k-blocks, the rest follow through the ring)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) av[ob][kb] = vp[(ob * XKB + kb) * 64];
        __builtin_amdgcn_sched_barrier(0);
        {
            bool kvalid[NKB][4];
    #pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
    #pragma unroll
                for (int r = 0; r < 4; ++r) kvalid[kb][r] = kb < 4 ? ((valid64 >> (kb * 16 + 4 * fg + r)) & 1ull) : ((valid_hi >> ((kb - 4) * 16 + 4 * fg + r)) & 1ull);
    #pragma unroll
            for (int qb = 2 * qp; qb < 2 * qp + 2; ++qb) {
                // S^T = K^ Q^T: a[r] = score(key 16 kb + 4 fg + r, query fr of this block)
                float s[NKB][4];
                float mx = NEG_BIG;
    #pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                    for (int ob = 0; ob < 4; ++ob) a = mfma16k16(kf[kb][ob], qh[ob][qb], a);
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
                for (int kb = 0; kb < NKB; ++kb)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s[kb][r] = kvalid[kb][r] ? __expf(s[kb][r] - mx) : 0.f;      // (the null key is always valid: mx is a real score, sum >= 1)
                        sum += s[kb][r];
                    }
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const float linv = 1.f / sum;
                // P[query][KS w + key], keys 0 .. KS - 1 of this head (KS = 36: the third key block only holds keys 32 .. 35, its lanes fg = 0)
                unsigned char* prow = Ps + (qb * 16 + fr) * P_LD + w * (KS * 2);
    #pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    if (kb * 16 + 4 * fg >= KS) continue;
                    *reinterpret_cast<uint2*>(prow + (kb * 16 + 4 * fg) * 2) =
                        make_uint2(pack_bf16x2(s[kb][0] * linv, s[kb][1] * linv), pack_bf16x2(s[kb][2] * linv, s[kb][3] * linv));
                }
            }
        }
    }
    // ---- phases D / E, 32 queries per pass (the VW fragments stay in registers).  The residual rows of this wave's features are requested first: they arrive under
    //      phase D
    __syncthreads();      // P of every head is in LDS
#pragma unroll 1
    for (int pass = 0; pass < NQB / 2; ++pass) {
    float4 res[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + (2 * pass + qb) * 16 + fr;
        const float* xr = p.x + (row0 + (size_t)(qi < p.nq ? qi : 0)) * p.ldx + w * 64 + 4 * fg;
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) res[qb][ob] = *reinterpret_cast<const float4*>(xr + ob * 16);
    }

    // ---- phase D: out^T[feature][query] = VW^T . P^T over the 288 (head, key) pairs
    f32x4_t acc[4][2];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) acc[ob][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < XKB; ++kb) {
        uint4 pf[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) pf[qb] = *reinterpret_cast<const uint4*>(Ps + ((2 * pass + qb) * 16 + fr) * P_LD + kb * 64 + fg * 16);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) acc[ob][qb] = mfma16(av[ob][kb % RD], pf[qb], acc[ob][qb]);
        if (RD < XKB) {      // KS = 80: the slot just consumed takes k-block kb + RD (requested HERE: see phase B)
            if (kb + RD < XKB) {
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) av[ob][kb % RD] = vp[(ob * XKB + kb + RD) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- phase E: x += out (fp32, in place), bf16 image + this wave's 64-column statistics partial of the new row (LayerNorm(dim) fold, producer side)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + (2 * pass + qb) * 16 + fr;
        const bool ok = qi < p.nq;
        const size_t row = row0 + (size_t)(ok ? qi : 0);
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
}

// ---- pack, once per generate and layer.  grid (kv sequences, heads, 2 halves of the features), 256 threads.
//   khat  [s][h][3 key blocks][4 d-blocks][64 lanes][4]: K^ = bf16(k / max(|k|, eps) * k_scale) of key (16 kb + lane % 16), d = 16 ob + 4 (lane / 16) .. + 3 (the A operand
//         of the 16 x 16 x 16 MFMA); key 0 = the null key (fp32 parameter, mmp.py:145-149), key j = context token j - 1 (the K half of ckv, bf16), keys > m zero
//   vwt   [s][32 feature blocks][9 k-blocks][64 lanes][8]: VW^T[feature 16 ob + lane % 16][flat k = 32 kb + 8 (lane / 16) .. + 7], flat k = 36 head + key,
//         VW[key][feature] = bf16( sum_d v[key][64 head + d] * W_o[feature][64 head + d] ), v[0] = bf16(null_v) like attention.hip, keys > m zero
template <int KS>
__global__ __launch_bounds__(256) void cross_fold_pack_kernel(const bf16_t* __restrict__ ckv, int m, int I, const float* __restrict__ null_k, const float* __restrict__ null_v,
                                                               const float* __restrict__ k_scale, const bf16_t* __restrict__ w_out, int ldw, bf16_t* __restrict__ khat,
                                                               bf16_t* __restrict__ vwt) {
    constexpr int XKB = XGeo<KS>::XKB, NKB = XGeo<KS>::NKB;
    __shared__ __attribute__((aligned(16))) float vs[KS][64];
    const int s = blockIdx.x, h = blockIdx.y, half = blockIdx.z, t = threadIdx.x;
    // V_h (and, in the first half's workgroup, K^_h: threads 0 .. 16 NKB - 1 take one key slot each)
    for (int i = t; i < KS * 64; i += 256) {
        const int key = i >> 6, d = i & 63;
        float v = 0.f;
        if (key == 0) v = bf16_to_f32(f32_to_bf16(null_v[h * 64 + d]));
        else if (key <= m) v = bf16_to_f32(ckv[((size_t)s * m + key - 1) * 2 * I + I + h * 64 + d]);
        vs[key][d] = v;
    }
    if (half == 0 && t < NKB * 16) {
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
        bf16_t* kb_ = khat + ((size_t)s * XH + h) * (NKB * 4) * 64 * 4;
#pragma unroll
        for (int d = 0; d < 64; ++d) {
            const int kb = key >> 4, frk = key & 15, ob = d >> 4, fgk = (d & 15) >> 2, j = d & 3;
            kb_[(((kb * 4 + ob) * 64) + fgk * 16 + frk) * 4 + j] = f32_to_bf16(k[d] * inv * k_scale[d]);
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
        for (int key = 0; key < KS; ++key) {
            float a = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 16; ++d4) {
                const float4 v = *reinterpret_cast<const float4*>(&vs[key][d4 * 4]);
                a = __builtin_fmaf(v.x, wr[d4 * 4], a); a = __builtin_fmaf(v.y, wr[d4 * 4 + 1], a);
                a = __builtin_fmaf(v.z, wr[d4 * 4 + 2], a); a = __builtin_fmaf(v.w, wr[d4 * 4 + 3], a);
            }
            const int kfl = h * KS + key;
            const int kb = kfl >> 5, fgk = (kfl & 31) >> 3, j = kfl & 7;
            vwt[((((size_t)s * (XD / 16) + ob) * XKB + kb) * 64 + fgk * 16 + fro) * 8 + j] = f32_to_bf16(a);
        }
    }
}

// the null pass's cross-attention as a constant row: every text key masked -> P = 1 on each head's null key, so the kernel above adds sum over the heads (in
// head order: one non-zero product per MFMA k-block) of the bf16 VW entries of the null key.  The decode loop adds that row to the null half without running
// the kernel (model.hip); this reproduces the kernel's value bit for bit from the packed fragments of any sequence.
__global__ __launch_bounds__(256) void cross_fold_null_row_kernel(const bf16_t* __restrict__ vwt, int KS, float* __restrict__ out) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= XD) return;
    const int XKB = XH * KS / 32;
    const int ob = o >> 4, fro = o & 15;
    float a = 0.f;
    for (int h = 0; h < XH; ++h) {
        const int kfl = h * KS;
        const int kb = kfl >> 5, fgk = (kfl & 31) >> 3, j = kfl & 7;
        a += bf16_to_f32(vwt[((((size_t)ob) * XKB + kb) * 64 + fgk * 16 + fro) * 8 + j]);
    }
    out[o] = a;
}

// the gain-folded q weight [512 features][512] as MFMA A fragments: [head][4 feature blocks][16 k-blocks][64 lanes][8] (a wave load = one contiguous KiB).  Once per
// generate and layer (0.5 MB; the weight itself is step-invariant, the packed copy lives in the workspace)
__global__ __launch_bounds__(256) void cross_fold_wq_pack_kernel(const bf16_t* __restrict__ wq, int ldw, bf16_t* __restrict__ wqf) {
    const int c = blockIdx.x * 256 + threadIdx.x;      // one 16-byte chunk per thread: (feature o, k-chunk kc of 8)
    if (c >= XD * (XD / 8)) return;
    const int o = c >> 6, kc = c & 63;
    const int h = o >> 6, ob = (o & 63) >> 4, fro = o & 15, kb = kc >> 2, fgk = kc & 3;
    reinterpret_cast<uint4*>(wqf)[(((size_t)(h * 4 + ob) * 16 + kb) * 64) + fgk * 16 + fro] = *reinterpret_cast<const uint4*>(wq + (size_t)o * ldw + kc * 8);
}

}  // namespace

static int xf_ks(int m) { return m + 1 <= 36 ? 36 : 80; }      // key slots of the instantiation that takes m context tokens
bool k_cross_fold_eligible(int D, int I, int H, int dh, int m) { return D == XD && I == XD && H == XH && dh == 64 && m >= 1 && m + 1 <= XKS_MAX; }
// (sized for the largest instantiation whatever the context length: the strides between layers / the workspace must not depend on a debug switch or on m)
size_t k_cross_fold_khat_elems(int kv_seqs) { return (size_t)kv_seqs * XH * (XGeo<XKS_MAX>::NKB * 4) * 64 * 4; }
size_t k_cross_fold_wqf_elems() { return (size_t)XD * XD; }
size_t k_cross_fold_vwt_elems(int kv_seqs) { return (size_t)kv_seqs * XD * XGeo<XKS_MAX>::XKF; }

int k_cross_fold_pack(hipStream_t s, const bf16_t* ckv, int kv_seqs, int m, int I, const float* null_k, const float* null_v, const float* k_scale,
                      const bf16_t* w_out, int ldw, const bf16_t* w_q_ln, int ldwq, bf16_t* khat, bf16_t* vwt, bf16_t* wqf) {
    if (kv_seqs <= 0) return MM_OK;
    if (!null_k || !null_v || !k_scale || !w_q_ln) return mm_set_error(MM_ERR_SHAPE, "cross_fold_pack: null key / value, k_scale and the gain-folded q weight required");
    if (ldwq % 8) return mm_set_error(MM_ERR_SHAPE, "cross_fold_pack: unaligned q weight rows");
    if (m < 1 || m + 1 > XKS_MAX) return mm_set_error(MM_ERR_SHAPE, "cross_fold_pack: 1 <= context tokens <= 79");
    hipLaunchKernelGGL(cross_fold_wq_pack_kernel, dim3(XD * (XD / 8) / 256), dim3(256), 0, s, w_q_ln, ldwq, wqf);
    {
        const int rc = mm_check_launch("cross_fold_wq_pack_kernel");
        if (rc) return rc;
    }
    if (xf_ks(m) == 36) hipLaunchKernelGGL(cross_fold_pack_kernel<36>, dim3(kv_seqs, XH, 2), dim3(256), 0, s, ckv, m, I, null_k, null_v, k_scale, w_out, ldw, khat, vwt);
    else hipLaunchKernelGGL(cross_fold_pack_kernel<80>, dim3(kv_seqs, XH, 2), dim3(256), 0, s, ckv, m, I, null_k, null_v, k_scale, w_out, ldw, khat, vwt);
    return mm_check_launch("cross_fold_pack_kernel");
}

int k_cross_fold_null_row(hipStream_t s, const bf16_t* vwt, int m, float* out) {
    hipLaunchKernelGGL(cross_fold_null_row_kernel, dim3(XD / 256), dim3(256), 0, s, vwt, xf_ks(m), out);
    return mm_check_launch("cross_fold_null_row_kernel");
}

template <int KS>
static int launch_cross_fold(hipStream_t s, const CrossFoldArgs& a) {
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cross_fold_kernel<KS>), hipFuncAttributeMaxDynamicSharedMemorySize, XGeo<KS>::SMEM);
        if (e != hipSuccess) return mm_set_hip_error(e, "cross_fold hipFuncSetAttribute");
        attr_set = true;
    }
    const int nqb = (a.nq + 31) / 32;      // 32 queries per workgroup
    hipLaunchKernelGGL(cross_fold_kernel<KS>, dim3(8 * ((a.seqs + 7) / 8) * nqb), dim3(512), XGeo<KS>::SMEM, s, a);
    return mm_check_launch("cross_fold_kernel");
}

int k_cross_fold(hipStream_t s, const CrossFoldArgs& a) {
    if (a.seqs <= 0 || a.nq <= 0) return MM_OK;
    if (a.m < 1 || a.m + 1 > XKS_MAX) return mm_set_error(MM_ERR_SHAPE, "cross_fold: 1 <= context tokens <= 79");
    if ((a.ldx % 4) || (a.ldxb_in % 8) || (a.xb && (a.ldxb % 4))) return mm_set_error(MM_ERR_SHAPE, "cross_fold: unaligned rows");
    if (!a.xb_in || !a.stp_in || a.in_np != XD / 64 || !a.wqf || !a.c1 || !a.q_scale) return mm_set_error(MM_ERR_SHAPE, "cross_fold: fold inputs of the q projection required");
    return xf_ks(a.m) == 36 ? launch_cross_fold<36>(s, a) : launch_cross_fold<80>(s, a);
}
