// The 'f16x2' tier's cross-attention behind its q projection as ONE kernel (round 6):
//     x += ( softmax(8 q^ . k^ + key mask) @ V ) W_o^T        (muse_maskgit_pytorch.py:139-162, context = the text encoding)
// with the output projection folded into the step-invariant values, as cross_fold.hip does for the bf16 engine:
//     (P_h V_h) W_o,h^T = P_h (V_h W_o,h^T) =: P_h VW_h       per head h, VW_h [keys][dim] -- packed once per generate and layer from the fp32 values.
// Rounds 4-5 ran the tier's block as LayerNorm-split + q projection (term GEMM) + attention on the fp32 MFMA (attention_f32.hip: 2048 workgroups, each
// normalising the same 33 keys again, P . V at 1/16 of the fp16 rate) + output projection (term GEMM, fp32 residual in / out): 17 + 26 + 28.8 + 26 us per
// layer and step at the headline size against 25 us of the bf16 engine's single kernel.  Here all of it is one launch (template QP; QP = false takes q rows from a GEMM in
// front and runs phases C-E only):
//   phase A  wave w = rows 4 w .. 4 w + 3 of the workgroup's 32: LayerNorm with split.hip's arithmetic (two passes, the same lane <-> column assignment), the two fp16
//            terms of every value -> LDS [32][512] x 2 planes (1056-byte rows)
//   phase B  wave h = head h: q_h = rows . Wq_h^T as term products in the term GEMMs' order (per 32-deep k-block xh.wh, xl.wh, xh.wl) against the q weight's term planes
//            packed as fragments, streamed from L2 through a register ring -- the accumulator fragment (4 consecutive features of one query) is what phase C reads.
//            Same operations in the same order as layernorm_split + the 64-row term GEMM: the results agree BIT FOR BIT with the launches (tests).
//   phase C  wave h = head h, 32 queries: q^ = l2norm(q_h) * q_scale in fp32 registers, S^T = K^ Q^T on v_mfma_f32_16x16x4_f32 (exact fp32 products; K^ is the
//            fp32 pack, normalised ONCE per generate), mask, softmax in registers (expf), P x 2^10 -> its two fp16 terms -> LDS [32 queries][288 (head, key)] x 2 planes
//   phase D  wave w = output features 64 w .. + 63: out^T = VW^T . P^T over the 288 (head, key) pairs as THREE fp16 products per k-block
//            (VWh.Ph + VWh.Pl + VWl.Ph, fp32 accumulation; VW x 2^8 split into two fp16 terms at pack time: 22 significand bits each side), the VW^T fragments
//            streamed from L2 through a register ring
//   phase E  x += out * 2^-18 (fp32, in place)
// Precision: every product is exact in fp32 or carries 2^-22 relative operand error -- the level of the tier's term GEMMs; the association differs from the
// reference's (P V) W_o^T (tests: against the fp64 block, and the tier's goldens).
// Shape class: dim = inner = 512, 8 heads x 64, <= 35 context tokens (+ the null key): KS = 36 key slots per head.  Everything else keeps the four-launch path.
// Measured and NOT kept (round 6): a phase F taking the feed-forward's LayerNorm + term split for the workgroup's 32 complete rows (and, in the same workgroup, for
// the matching rows of the null half with their constant row added) -- it removes the 17 us LayerNorm-split launch and lengthens this kernel by as much: a single
// round of one-workgroup-per-CU latency chains pays every added barrier and 32-byte store segment in full, the stand-alone pass streams at 6 TB/s
// (97.4 / 97.6 ms per tier step with it, 97.5 / 97.0 without, same box).
// Measured (same process, general fp32 checkpoint, B = 32): four launches 106.3 / 106.6 ms per step, phases C-E behind LayerNorm-split + q GEMM 101.3 / 102.4, everything in
// here 99.3 / 99.5.
#include <float.h>
#include <string.h>

#include "common.h"
#include "muse_hip_internal.h"

namespace {

constexpr int VH = 8, VD = 512, KS = 36;
constexpr int XKF = VH * KS;            // 288 flat (head, key) pairs
constexpr int XKB = XKF / 32;           // 9 MFMA k-blocks
constexpr int NKB = 3;                  // key blocks of 16 per head
constexpr int P_LD = XKF * 2 + 32;      // bytes per P row in LDS (608 = 64 k + 32: conflict-free fragment reads, cross_fold.hip X_LD)
constexpr int VQ = 32;                  // queries per workgroup
constexpr int RD = 4;                   // VW^T k-blocks in flight per wave (two planes: 32 x 16 bytes per lane)
constexpr float P_SCALE = 1024.f, VW_SCALE = 256.f, OUT_SCALE = 1.f / (1024.f * 256.f);
constexpr float NEG_BIG = -3.0e38f;
constexpr int X_LD = 1056;              // bytes per normalised row (one term plane) in LDS: 512 fp16 + 32 (cross_fold.hip: conflict-free fragment reads)
constexpr int RQ = 4;                   // q-weight k-blocks in flight per wave and plane (QP form)
constexpr int SMEM_QP = 2 * VQ * X_LD + 2 * VQ * P_LD, SMEM_PLAIN = 2 * VQ * P_LD;

__device__ __forceinline__ f32x4_t mfma4(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// QP: the block's LayerNorm and q projection run in here as well (phases A / B) -- q never exists in HBM, the LayerNorm-split pass and the q GEMM disappear
template <bool QP>
__global__ __launch_bounds__(512, 1) void cross_vw_x2_kernel(const CrossVwArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vw_smem[];
    unsigned char* Ps = vw_smem;                        // [2 planes][VQ][P_LD]: plane 0 the leading terms of P, plane 1 the remainders
    unsigned char* Xs = vw_smem + 2 * VQ * P_LD;        // QP: [2 planes][VQ][X_LD] the normalised rows' terms
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    // workgroup -> (sequence, query block): all query blocks of a sequence on ONE XCD (they stream the same VW fragments), as cross_fold.hip
    const int nqb = (p.nq + VQ - 1) / VQ;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = (slot / nqb) * 8 + xcd;
    if (b >= p.seqs) return;
    const int q0 = (slot % nqb) * VQ;
    const int kvb = p.kv_batch_mod > 0 ? b % p.kv_batch_mod : b;
    const size_t row0 = (size_t)b * p.nq;

    int kmb = 1;
    if (p.key_mask && lane < p.m) kmb = p.key_mask[(size_t)b * p.km_sb + lane];
    // q rows of head w: lane (fr, fg) holds dims 16 j + 4 fg + i of query fr (per query block) in qf[qb][4 j + i]
    float qf[2][16];
    if constexpr (!QP) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qi = q0 + qb * 16 + fr;
            const float* qp = p.q + (row0 + (size_t)(qi < p.nq ? qi : 0)) * p.ldq + w * 64 + 4 * fg;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(qp + 16 * j);
                qf[qb][4 * j] = v.x; qf[qb][4 * j + 1] = v.y; qf[qb][4 * j + 2] = v.z; qf[qb][4 * j + 3] = v.w;
            }
        }
    } else {
        // ---- phase A: rows 4 w .. 4 w + 3 of the block: LayerNorm with split.hip's arithmetic (layernorm_split_kernel<2>: lane holds columns 4 (64 it + lane) .. + 3, two
        //      passes), the two fp16 terms of every value -> LDS.  The q weight's first k-blocks are requested in front: [plane][head][4 feature blocks][16 k-blocks][64 lanes][8]
        const bool t3 = p.wq_terms == 3;      // (uniform) three products: the weight has a remainder plane
        const uint4* wph = reinterpret_cast<const uint4*>(p.wqf) + (size_t)w * 4 * 16 * 64 + lane;
        const uint4* wpl = wph + (size_t)VH * 4 * 16 * 64;
        uint4 wah[RQ][4], wal[RQ][4];
#pragma unroll
        for (int kb = 0; kb < RQ; ++kb)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                wah[kb][ob] = wph[(ob * 16 + kb) * 64];
                wal[kb][ob] = t3 ? wpl[(ob * 16 + kb) * 64] : make_uint4(0u, 0u, 0u, 0u);
            }
        float4 xv[4][2];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int qi = q0 + 4 * w + rr;
            const float* xr = p.x + (row0 + (size_t)(qi < p.nq ? qi : 0)) * p.ldx;
#pragma unroll
            for (int it = 0; it < 2; ++it) xv[rr][it] = *reinterpret_cast<const float4*>(xr + (it * 64 + lane) * 4);
        }
        float4 lg[2], lb[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            lg[it] = *reinterpret_cast<const float4*>(p.ln_gamma + (it * 64 + lane) * 4);
            lb[it] = p.ln_beta ? *reinterpret_cast<const float4*>(p.ln_beta + (it * 64 + lane) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            float sum = 0.f;
#pragma unroll
            for (int it = 0; it < 2; ++it) sum += (xv[rr][it].x + xv[rr][it].y) + (xv[rr][it].z + xv[rr][it].w);
            const float mean = wave_sum(sum) / (float)VD;
            float sq = 0.f;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const float a = xv[rr][it].x - mean, b2 = xv[rr][it].y - mean, c2 = xv[rr][it].z - mean, d2 = xv[rr][it].w - mean;
                sq += (a * a + b2 * b2) + (c2 * c2 + d2 * d2);
            }
            const float rstd = 1.f / sqrtf(wave_sum(sq) / (float)VD + 1e-5f);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const float o[4] = {(xv[rr][it].x - mean) * rstd * lg[it].x + lb[it].x, (xv[rr][it].y - mean) * rstd * lg[it].y + lb[it].y,
                                    (xv[rr][it].z - mean) * rstd * lg[it].z + lb[it].z, (xv[rr][it].w - mean) * rstd * lg[it].w + lb[it].w};
                uint16_t h[4], l[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) split2_f16(o[r], h[r], l[r]);
                unsigned char* xd = Xs + (4 * w + rr) * X_LD + (it * 64 + lane) * 8;
                *reinterpret_cast<uint2*>(xd) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
                *reinterpret_cast<uint2*>(xd + VQ * X_LD) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
            }
        }
        __syncthreads();      // the 32 normalised rows are in LDS
        // ---- phase B: wave h = head h: q_h = rows . Wq_h^T as term products (xh.wh + xl.wh (+ xh.wl)), the weight fragments RQ k-blocks ahead
        f32x4_t accq[4][2];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) accq[ob][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            u32x4_t bh[2], bl[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                bh[qb] = *reinterpret_cast<const u32x4_t*>(Xs + (qb * 16 + fr) * X_LD + kb * 64 + fg * 16);
                bl[qb] = *reinterpret_cast<const u32x4_t*>(Xs + VQ * X_LD + (qb * 16 + fr) * X_LD + kb * 64 + fg * 16);
            }
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                const u32x4_t ah = __builtin_bit_cast(u32x4_t, wah[kb % RQ][ob]), al = __builtin_bit_cast(u32x4_t, wal[kb % RQ][ob]);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    accq[ob][qb] = mfma16t<true>(ah, bh[qb], accq[ob][qb]);
                    accq[ob][qb] = mfma16t<true>(ah, bl[qb], accq[ob][qb]);
                    if (t3) accq[ob][qb] = mfma16t<true>(al, bh[qb], accq[ob][qb]);
                }
            }
            if (kb + RQ < 16) {
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    wah[kb % RQ][ob] = wph[(ob * 16 + kb + RQ) * 64];
                    if (t3) wal[kb % RQ][ob] = wpl[(ob * 16 + kb + RQ) * 64];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // an accumulator fragment = 4 consecutive features (16 ob + 4 fg + r) of query fr: the layout phase C reads
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int r = 0; r < 4; ++r) qf[qb][4 * ob + r] = accq[ob][qb][r] * p.alpha;
    }
    // K^ of head w: [kv sequence][head][3 key blocks][4 dim blocks][64 lanes][4]: lane (fr, fg) gets key 16 kb + fr, dims 16 j + 4 fg .. + 3
    float4 kf[NKB][4];
    {
        const float4* kp = reinterpret_cast<const float4*>(p.khat) + ((size_t)kvb * VH + w) * (NKB * 4) * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int j = 0; j < 4; ++j) kf[kb][j] = kp[(kb * 4 + j) * 64];
    }
    float4 qs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) qs[j] = *reinterpret_cast<const float4*>(p.q_scale + 16 * j + 4 * fg);
    // this wave's first VW^T fragments (phase D) are requested now: they arrive under the scores and the softmax.  [kv sequence][plane][32 feature blocks][9 k-blocks][64 lanes][8]
    uint4 avh[4][RD], avl[4][RD];
    const uint4* vph = reinterpret_cast<const uint4*>(p.vwt) + (((size_t)kvb * 2) * (VD / 16) + (size_t)w * 4) * XKB * 64 + lane;
    const uint4* vpl = vph + (size_t)(VD / 16) * XKB * 64;
#pragma unroll
    for (int kb = 0; kb < RD; ++kb)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) { avh[ob][kb] = vph[(ob * XKB + kb) * 64]; avl[ob][kb] = vpl[(ob * XKB + kb) * 64]; }
    __builtin_amdgcn_sched_barrier(0);

    const unsigned long long tok_lo = __ballot(lane < p.m && kmb != 0);
    const unsigned long long valid64 = (tok_lo << 1) | 1ull;      // bit 0 = the null key (always attended, mmp.py:145-155), bit j = context token j - 1
    bool kvalid[NKB][4];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) kvalid[kb][r] = (valid64 >> (kb * 16 + 4 * fg + r)) & 1ull;

    // ---- phase C
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) ss += qf[qb][e] * qf[qb][e];
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float den = fmaxf(sqrtf(ss), 1e-12f);      // F.normalize eps (mmp.py:41-42)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qf[qb][4 * j] = qf[qb][4 * j] / den * qs[j].x; qf[qb][4 * j + 1] = qf[qb][4 * j + 1] / den * qs[j].y;
            qf[qb][4 * j + 2] = qf[qb][4 * j + 2] / den * qs[j].z; qf[qb][4 * j + 3] = qf[qb][4 * j + 3] / den * qs[j].w;
        }
        float s[NKB][4];
        float mx = NEG_BIG;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a = mfma4(kf[kb][j].x, qf[qb][4 * j], a);
                a = mfma4(kf[kb][j].y, qf[qb][4 * j + 1], a);
                a = mfma4(kf[kb][j].z, qf[qb][4 * j + 2], a);
                a = mfma4(kf[kb][j].w, qf[qb][4 * j + 3], a);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {      // a[r] = score(key 16 kb + 4 fg + r, query fr)
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
                s[kb][r] = kvalid[kb][r] ? expf(s[kb][r] - mx) : 0.f;      // (the null key is always valid: mx is a real score, sum >= 1)
                sum += s[kb][r];
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float linv = P_SCALE / sum;
        unsigned char* prow = Ps + (qb * 16 + fr) * P_LD + w * (KS * 2);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb * 16 + 4 * fg >= KS) continue;      // (the third key block only holds keys 32 .. 35: its lanes fg = 0)
            uint16_t h[4], l[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) split2_f16(s[kb][r] * linv, h[r], l[r]);
            *reinterpret_cast<uint2*>(prow + (kb * 16 + 4 * fg) * 2) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
            *reinterpret_cast<uint2*>(prow + VQ * P_LD + (kb * 16 + 4 * fg) * 2) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
        }
    }
    // the residual rows of this wave's features: requested before the barrier, consumed in phase E
    float4 res[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + qb * 16 + fr;
        const float* xr = p.x + (row0 + (size_t)(qi < p.nq ? qi : 0)) * p.ldx + w * 64 + 4 * fg;
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) res[qb][ob] = *reinterpret_cast<const float4*>(xr + ob * 16);
    }
    __syncthreads();      // P of every head is in LDS

    // ---- phase D: out^T[feature][query] = VW^T . P^T over the 288 (head, key) pairs, three term products per k-block
    f32x4_t acc[4][2];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) acc[ob][qb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < XKB; ++kb) {
        u32x4_t ph[2], pl[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            ph[qb] = *reinterpret_cast<const u32x4_t*>(Ps + (qb * 16 + fr) * P_LD + kb * 64 + fg * 16);
            pl[qb] = *reinterpret_cast<const u32x4_t*>(Ps + VQ * P_LD + (qb * 16 + fr) * P_LD + kb * 64 + fg * 16);
        }
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            const u32x4_t ah = __builtin_bit_cast(u32x4_t, avh[ob][kb % RD]), al = __builtin_bit_cast(u32x4_t, avl[ob][kb % RD]);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                acc[ob][qb] = mfma16t<true>(ah, ph[qb], acc[ob][qb]);
                acc[ob][qb] = mfma16t<true>(ah, pl[qb], acc[ob][qb]);
                acc[ob][qb] = mfma16t<true>(al, ph[qb], acc[ob][qb]);
            }
        }
        if (kb + RD < XKB) {      // the slot just consumed takes k-block kb + RD (requested HERE: the scheduler otherwise sinks the load to its use)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) { avh[ob][kb % RD] = vph[(ob * XKB + kb + RD) * 64]; avl[ob][kb % RD] = vpl[(ob * XKB + kb + RD) * 64]; }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- phase E: accumulator fragment = 4 consecutive features of one query: x += out (fp32, in place)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + qb * 16 + fr;
        if (qi >= p.nq) continue;
        float* xr = p.x + (row0 + (size_t)qi) * p.ldx + w * 64 + 4 * fg;
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            float4 o;
            o.x = acc[ob][qb][0] * OUT_SCALE + res[qb][ob].x; o.y = acc[ob][qb][1] * OUT_SCALE + res[qb][ob].y;
            o.z = acc[ob][qb][2] * OUT_SCALE + res[qb][ob].z; o.w = acc[ob][qb][3] * OUT_SCALE + res[qb][ob].w;
            *reinterpret_cast<float4*>(xr + ob * 16) = o;
        }
    }
}

// ---- pack, once per generate and layer.  grid (kv sequences, heads, 2 halves of the features), 256 threads.
//   khat  [s][h][3 key blocks][4 dim blocks][64 lanes][4] fp32: K^ = k / max(|k|, eps) * k_scale of key (16 kb + lane % 16), dims 16 j + 4 (lane / 16) .. + 3; key 0 = the
//         null key (mmp.py:145-149), key j = context token j - 1 (the K half of the fp32 ckv), keys > m zero
//   vwt   [s][2 planes][32 feature blocks][9 k-blocks][64 lanes][8] fp16: the two terms of 2^8 VW^T[feature 16 ob + lane % 16][flat k = 32 kb + 8 (lane / 16) .. + 7],
//         flat k = 36 head + key, VW[key][feature] = sum_d v[key][64 head + d] * W_o[feature][64 head + d] in fp32, W_o = (wh + wl) * alpha from the term pack
__global__ __launch_bounds__(256) void cross_vw_x2_pack_kernel(const float* __restrict__ ckv, int m, int I, const float* __restrict__ null_k, const float* __restrict__ null_v,
                                                                const float* __restrict__ k_scale, const uint16_t* __restrict__ w_out, int ldw, int terms, float alpha,
                                                                float* __restrict__ khat, uint16_t* __restrict__ vwt) {
    __shared__ __attribute__((aligned(16))) float vs[KS][64];
    const int s = blockIdx.x, h = blockIdx.y, half = blockIdx.z, t = threadIdx.x;
    for (int i = t; i < KS * 64; i += 256) {
        const int key = i >> 6, d = i & 63;
        float v = 0.f;
        if (key == 0) v = null_v[h * 64 + d];
        else if (key <= m) v = ckv[((size_t)s * m + key - 1) * 2 * I + I + h * 64 + d];
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
            else if (key <= m) v = ckv[((size_t)s * m + key - 1) * 2 * I + h * 64 + d];
            k[d] = v;
            ss += v * v;
        }
        const float den = fmaxf(sqrtf(ss), 1e-12f);
        float* kb_ = khat + ((size_t)s * VH + h) * (NKB * 4) * 64 * 4;
#pragma unroll
        for (int d = 0; d < 64; ++d) {
            const int kb = key >> 4, frk = key & 15, j = d >> 4, fgk = (d & 15) >> 2, i = d & 3;
            kb_[(((kb * 4 + j) * 64) + fgk * 16 + frk) * 4 + i] = k[d] / den * k_scale[d];      // (the operations of attention_f32.hip's staging, in its order)
        }
    }
    __syncthreads();
    {
        const int o = half * 256 + t;      // one output feature per thread: its 64 weights of head h in registers
        float wr[64];
        const uint16_t* wp = w_out + (size_t)o * ldw + h * 64;
#pragma unroll
        for (int d8 = 0; d8 < 8; ++d8) {      // 16-byte loads: 8 fp16 terms each (rows and segments are multiples of 64 elements)
            float fh[8], fl[8];
            unpack8s<true>(*reinterpret_cast<const uint4*>(wp + d8 * 8), fh);
            if (terms == 3) unpack8s<true>(*reinterpret_cast<const uint4*>(wp + 2 * I + d8 * 8), fl);      // segments [wh | wh | wl]
#pragma unroll
            for (int j = 0; j < 8; ++j) wr[d8 * 8 + j] = (terms == 3 ? fh[j] + fl[j] : fh[j]) * alpha;
        }
        const int ob = o >> 4, fro = o & 15;
        uint16_t* vh = vwt + ((size_t)s * 2) * VD * XKF;
        uint16_t* vl = vh + (size_t)VD * XKF;
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
            const size_t at = ((((size_t)ob) * XKB + kb) * 64 + fgk * 16 + fro) * 8 + j;
            uint16_t th, tl;
            split2_f16(a * VW_SCALE, th, tl);
            vh[at] = th;
            vl[at] = tl;
        }
    }
}

// the q weight's term planes [512 features][terms x 512] (segments [wh | wh | wl]) as MFMA A fragments: [plane][head][4 feature blocks][16 k-blocks][64 lanes][8] (a wave load =
// one contiguous KiB), once per generate and layer
__global__ __launch_bounds__(256) void cross_vw_x2_wq_pack_kernel(const uint16_t* __restrict__ wq, int ldw, int terms, uint16_t* __restrict__ wqf) {
    const int c = blockIdx.x * 256 + threadIdx.x;      // one 16-byte chunk per thread: (feature o, k-chunk kc of 8), both planes
    if (c >= VD * (VD / 8)) return;
    const int o = c >> 6, kc = c & 63;
    const int h = o >> 6, ob = (o & 63) >> 4, fro = o & 15, kb = kc >> 2, fgk = kc & 3;
    const size_t at = (((size_t)(h * 4 + ob) * 16 + kb) * 64) + fgk * 16 + fro;
    reinterpret_cast<uint4*>(wqf)[at] = *reinterpret_cast<const uint4*>(wq + (size_t)o * ldw + kc * 8);
    if (terms == 3) reinterpret_cast<uint4*>(wqf)[(size_t)VD * VD / 8 + at] = *reinterpret_cast<const uint4*>(wq + (size_t)o * ldw + 2 * VD + kc * 8);
}

}  // namespace

size_t k_cross_vw_x2_wqf_halves() { return (size_t)2 * VD * VD; }
int k_cross_vw_x2_wq_pack(hipStream_t s, const bf16_t* w_q, int ldw, int terms, bf16_t* wqf) {
    if (!w_q || !wqf || (terms != 2 && terms != 3) || ldw < terms * VD || (ldw % 8)) return mm_set_error(MM_ERR_SHAPE, "cross_vw_x2_wq_pack: the q weight's term pack (2 or 3 segments of 512)");
    hipLaunchKernelGGL(cross_vw_x2_wq_pack_kernel, dim3(VD * (VD / 8) / 256), dim3(256), 0, s, (const uint16_t*)w_q, ldw, terms, (uint16_t*)wqf);
    return mm_check_launch("cross_vw_x2_wq_pack_kernel");
}

bool k_cross_vw_x2_eligible(int D, int I, int H, int dh, int m) { return D == VD && I == VD && H == VH && dh == 64 && m >= 1 && m + 1 <= KS; }
size_t k_cross_vw_x2_khat_floats(int kv_seqs) { return (size_t)kv_seqs * VH * (NKB * 4) * 64 * 4; }
size_t k_cross_vw_x2_vwt_halves(int kv_seqs) { return (size_t)kv_seqs * 2 * VD * XKF; }

int k_cross_vw_x2_pack(hipStream_t s, const float* ckv, int kv_seqs, int m, int I, const float* null_k, const float* null_v, const float* k_scale,
                       const bf16_t* w_out, int ldw, int terms, float alpha, float* khat, bf16_t* vwt) {
    if (kv_seqs <= 0) return MM_OK;
    if (!null_k || !null_v || !k_scale || !w_out) return mm_set_error(MM_ERR_SHAPE, "cross_vw_x2_pack: null key / value, k_scale and the output projection's term pack required");
    if (m < 1 || m + 1 > KS || I != VD || (terms != 2 && terms != 3) || ldw < terms * I) return mm_set_error(MM_ERR_SHAPE, "cross_vw_x2_pack: 1 <= context tokens <= 35, inner 512, 2 or 3 term segments");
    hipLaunchKernelGGL(cross_vw_x2_pack_kernel, dim3(kv_seqs, VH, 2), dim3(256), 0, s, ckv, m, I, null_k, null_v, k_scale, (const uint16_t*)w_out, ldw, terms,
                       alpha == 0.f ? 1.f : alpha, khat, (uint16_t*)vwt);
    return mm_check_launch("cross_vw_x2_pack_kernel");
}

int k_cross_vw_x2(hipStream_t s, const CrossVwArgs& a) {
    if (a.seqs <= 0 || a.nq <= 0) return MM_OK;
    if (a.m < 1 || a.m + 1 > KS) return mm_set_error(MM_ERR_SHAPE, "cross_vw_x2: 1 <= context tokens <= 35");
    if ((a.ldx % 4) || !a.khat || !a.vwt || !a.q_scale || !a.x) return mm_set_error(MM_ERR_SHAPE, "cross_vw_x2: operands / alignment");
    if (a.wqf ? (!a.ln_gamma || (a.wq_terms != 2 && a.wq_terms != 3) || a.ldx != VD) : (!a.q || (a.ldq % 4)))
        return mm_set_error(MM_ERR_SHAPE, "cross_vw_x2: either q rows, or the q weight fragments with the LayerNorm's gain and dense 512-wide rows");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cross_vw_x2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_QP);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(cross_vw_x2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_PLAIN);
        if (e != hipSuccess) return mm_set_hip_error(e, "cross_vw_x2 hipFuncSetAttribute");
        attr_set = true;
    }
    const int nqb = (a.nq + VQ - 1) / VQ;
    const dim3 grid(8 * ((a.seqs + 7) / 8) * nqb);
    if (a.wqf) hipLaunchKernelGGL(cross_vw_x2_kernel<true>, grid, dim3(512), SMEM_QP, s, a);
    else hipLaunchKernelGGL(cross_vw_x2_kernel<false>, grid, dim3(512), SMEM_PLAIN, s, a);
    return mm_check_launch("cross_vw_x2_kernel");
}
