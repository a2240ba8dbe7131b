// The 'f16x2' tier's wide projections (self-attention q|k|v, FF w1) with every operand TERM staged ONCE.
//
// The tier multiplies fp32 values as sums of two fp16 terms: x = xh + xl, w = wh + wl, x . w ~ xh.wh + xl.wh + xh.wl (common.h split2_f16).  Rounds 4-5 ran
// that as ONE fp16 GEMM of depth 3 K over the segment packs X' = [xh | xl | xh], W' = [wh | wh | wl] on the unchanged kernels -- which stages xh and wh
// TWICE (as does the two-product form [xh | xl] . [wh | wh] of a bf16-representable checkpoint) through a k-loop that is bound by its L2 -> LDS operand
// feed, not by the matrix pipe (64 KiB per 3440-cycle step against 2048 cycles of MFMA issue).  The products SHARE operands, so this kernel
// reads the same packs as FOUR (three) term planes and runs all products of a 32-deep k-block from one staging of it:
//   NP = 3: a step stages [xh(32) | xl(32)] for 256 tokens and [wh(32) | wl(32)] for the weight tile (64 KiB, as a 64-deep step of gemm_wide.hip) and
//           issues 96 MFMAs per wave on it instead of 64: xh.wh, xl.wh, xh.wl per (weight fragment, token block), fragments reused from registers;
//   NP = 2: a step stages [xh(32) | xl(32)]; the weight tile [wh(64)] is staged every OTHER step into its own double buffer (48 KiB per 64 MFMAs).
// Everything else is gemm_wide.hip's structure: 256 tokens x 64 NFW weight rows per 512-thread workgroup (wave (wm, wn): 128 tokens x 16 NFW rows), 128-byte
// XOR-swizzled LDS rows filled by LDS-DMA (the swizzle and the plane offset of a lane's 16 bytes are part of its SOURCE address), one barrier per step,
// counted vmcnt, persistent over the tiles with the next tile's first step requested during the last step of the current one.
// Summation order per output element: k ascending in blocks of 32, per block hh, lh, hl -- NOT the concatenated order of the other kernels of the tier
// (all hh, then all lh, then all hl): same terms, fp32 accumulation, results equal to ~1e-7 relative, not bit for bit (tests/test_gpu_terms_gemm.py).
// Epilogues: EPI 0 -- fp32 rows, accumulators x alpha, written straight from the fragments (16 bytes per lane, 64 contiguous bytes per row and instruction);
//            EPI 1 -- FF w1 of the tier: GEGLU (exact erf, fp32) over the GEGLU-interleaved weight rows, the product split into its fp16 terms and written
//                     as the segment pack [hh | hl | hh][:NP] FF w2 multiplies, + the row's (sum, sum of squares) per 32 columns for the LayerNorm(inner)
//                     fold of w2 (mmp.py:85-88): the fp32 [rows][2 Fp] intermediate and the GEGLU / LayerNorm / split pass over it disappear.
#include "common.h"
#include "muse_hip_internal.h"

namespace {

constexpr int TM = 256, ROWB = 128;
constexpr int X_B = TM * ROWB;      // 32 KiB: 256 token rows x [xh(32) | xl(32)]

__device__ __forceinline__ int sw128(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

#define MT_VMCNT_IMM(n_) (0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14))
__device__ __forceinline__ void wait_vmcnt_t(int n) {      // wave-uniform n; above 31: wait for 31 (waiting for more is always safe)
    switch (n) {
#define MT_W(n_) case n_: __builtin_amdgcn_s_waitcnt(MT_VMCNT_IMM(n_)); break;
        MT_W(0) MT_W(1) MT_W(2) MT_W(3) MT_W(4) MT_W(5) MT_W(6) MT_W(7) MT_W(8) MT_W(9) MT_W(10) MT_W(11) MT_W(12) MT_W(13) MT_W(14) MT_W(15)
        MT_W(16) MT_W(17) MT_W(18) MT_W(19) MT_W(20) MT_W(21) MT_W(22) MT_W(23) MT_W(24) MT_W(25) MT_W(26) MT_W(27) MT_W(28) MT_W(29) MT_W(30)
#undef MT_W
        default: if (n < 0) __builtin_amdgcn_s_waitcnt(MT_VMCNT_IMM(0)); else __builtin_amdgcn_s_waitcnt(MT_VMCNT_IMM(31)); break;
    }
}

template <int NP, int NFW, int EPI, int DPOS = 1>      // DPOS: token block behind which the next step's requests go out (-1: right behind the barrier)
__global__ __launch_bounds__(512) void gemm_terms_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TN = 64 * NFW, W_B = TN * ROWB;
    // LDS: token stage s at s X_B; weight buffer s at 2 X_B + s W_B (NP 3: it follows the token stage; NP 2: it holds a PAIR of steps)
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int total = p.tiles_m * p.tiles_n, G = gridDim.x;
    const int KS = p.K / NP;            // the contraction length proper (one segment of the packs)
    const int KT = KS / 32;             // 32-deep steps (even: the launcher checks)
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    // a DMA instruction covers 8 rows of 128 bytes: lane l fills physical chunk l & 7 of row l >> 3 with LOGICAL chunk lc = (l & 7) ^ (row & 7); logical chunks
    // 0-3 are 32 k-values of the h plane, 4-7 the same 32 of the l plane (tokens: segment 1 of the row; weights, NP 3: segment 2); NP 2 weights: 64 k-values of wh
    const int lc = (lane & 7) ^ (lane >> 3);
    const int xoff = (lc & 3) * 16 + (lc >> 2) * KS * 2;
    const int woff = NP == 3 ? (lc & 3) * 16 + (lc >> 2) * KS * 4 : lc * 16;
    int voff_x[4], voff_w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        voff_x[i] = (32 * wid + 8 * i + (lane >> 3)) * p.ldx * 2 + xoff;
        voff_w[i] = (8 * NFW * wid + 8 * (i < NFW ? i : 0) + (lane >> 3)) * p.ldw * 2 + woff;
    }
    int vb = blockIdx.x;
    if (vb >= total) return;
    __amdgpu_buffer_rsrc_t rx, rw;
    int tile_m, tile_n;
#define TILE_SETUP(vb_)                                                                                                                \
    {                                                                                                                                  \
        xcd_grouped_tile(vb_, p.tiles_m, p.tiles_n, 8, tile_m, tile_n);                                                                \
        const int left_ = p.M - tile_m * TM;                                                                                           \
        rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X + (size_t)tile_m * TM * p.ldx), 0,                              \
                                               (unsigned)(left_ < TM ? left_ : TM) * (unsigned)p.ldx * 2u, 0x00020000);      /* rows beyond M read as zero */ \
        rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W + (size_t)tile_n * TN * p.ldw), 0, (unsigned)TN * (unsigned)p.ldw * 2u, 0x00020000); \
    }
#define ISSUE_X(kt_, st_)                                                                                                              \
    {                                                                                                                                  \
        unsigned char* xs_ = smem + (st_) * X_B + wid * 4096;                                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(xs_ + i * 1024), 16, voff_x[i], (kt_) * 64, 0, 0);               \
    }
#define ISSUE_W(byte_, buf_)      /* byte_: offset of the step (NP 3: 64 kt) or of the step pair (NP 2: 128 j) along the weight rows */ \
    {                                                                                                                                  \
        unsigned char* ws_ = smem + 2 * X_B + (buf_) * W_B + wid * (NFW * 1024);                                                       \
        _Pragma("unroll") for (int i = 0; i < NFW; ++i)                                                                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(ws_ + i * 1024), 16, voff_w[i], (byte_), 0, 0);                   \
    }
    // Issue order and waits.  NP 3: step kt requests step kt + 1 (tokens + weights) into the other stage; the wait at the top of a step is vmcnt(0).
    // NP 2: step kt requests the tokens of step kt + 1 and, at even kt, BEHIND them the weights of the next step pair (other weight buffer: last read
    // two steps ago) -- so the top of an odd step waits for everything but those NFW weight instructions, the top of an even step for everything.
    // Across tiles: the last step (odd) requests the next tile's step 0 (token stage 0, weight buffer 0: both idle since the step before -- NP 2 needs
    // KT % 4 == 0 for that); the epilogue's stores go out behind it, and the first wait of the next tile counts them (VMEM retires in order).
    TILE_SETUP(vb);
    ISSUE_X(0, 0);
    ISSUE_W(0, 0);
    int pending = 0;
    f32x4_t acc[NFW][8];             // [weight fragment a][token block b]: lane (fr, fg) holds weight rows 16 a + 4 fg .. + 3 for token 16 b + fr
    while (true) {
        const int m0 = tile_m * TM, n0 = tile_n * TN, cur_n = tile_n;
#pragma unroll
        for (int a = 0; a < NFW; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT; ++kt) {
            const int st = kt & 1;
#if defined(MM_EXP) && MM_EXP == 61      // tools A/B only (tools/build_exp.sh 61 gemm_terms): round 5's hand-counted wait across the tile boundary
            if (kt == 0) wait_vmcnt_t(pending);
#else
            // Round 6: the first wait of a tile takes EVERYTHING.  Round 5 waited vmcnt(number of epilogue stores issued behind this step's
            // LDS-DMA), which is only right if hipcc emits exactly the counted stores -- it may merge the GEGLU epilogue's 8-byte segment stores, and then the
            // wait would return before step 0's operands have landed: silent wrong results.  One drained wait per tile costs < 1 % (profiles/r06_*).
            if (kt == 0) { (void)pending; __builtin_amdgcn_s_waitcnt(MT_VMCNT_IMM(0)); }
#endif
            else if (NP == 2 && st == 1 && kt + 1 < KT) __builtin_amdgcn_s_waitcnt(MT_VMCNT_IMM(NFW));
            else __builtin_amdgcn_s_waitcnt(MT_VMCNT_IMM(0));
            __builtin_amdgcn_s_barrier();            // this step's operands have landed for everybody, and everybody is done reading what the next request overwrites
            const unsigned char* xs = smem + st * X_B + (wm * 128) * ROWB;
            const unsigned char* ws = smem + 2 * X_B + (NP == 3 ? st : ((kt >> 1) & 1)) * W_B + (wn * 16 * NFW) * ROWB;
            const int wc = NP == 3 ? 0 : st * 4;      // first chunk of this step's wh fragment in the weight row
            u32x4_t wh[NFW], wl[NFW], xh[2], xl[2];
#define TERMS_ISSUE_NEXT()                                                                                                             \
            {                                                                                                                          \
                if (kt + 1 < KT) {                                                                                                     \
                    ISSUE_X(kt + 1, st ^ 1);                                                                                           \
                    if constexpr (NP == 3) {                                                                                           \
                        ISSUE_W((kt + 1) * 64, st ^ 1);                                                                                \
                    } else {                                                                                                           \
                        if (st == 0 && kt + 2 < KT) ISSUE_W(((kt >> 1) + 1) * 128, ((kt >> 1) + 1) & 1);                               \
                    }                                                                                                                  \
                } else if (vb + G < total) {                                                                                           \
                    TILE_SETUP(vb + G);                                                                                                \
                    ISSUE_X(0, 0);                                                                                                     \
                    ISSUE_W(0, 0);                                                                                                     \
                }                                                                                                                      \
            }
            if constexpr (DPOS < 0) TERMS_ISSUE_NEXT();
#pragma unroll
            for (int a = 0; a < NFW; ++a) {
                wh[a] = *reinterpret_cast<const u32x4_t*>(ws + sw128(a * 16 + fr, wc + fg));
                if constexpr (NP == 3) wl[a] = *reinterpret_cast<const u32x4_t*>(ws + sw128(a * 16 + fr, 4 + fg));
            }
            xh[0] = *reinterpret_cast<const u32x4_t*>(xs + sw128(fr, fg));
            xl[0] = *reinterpret_cast<const u32x4_t*>(xs + sw128(fr, 4 + fg));
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b + 1 < 8) {      // the next block's token fragments are requested in front of this block's MFMAs
                    xh[(b + 1) & 1] = *reinterpret_cast<const u32x4_t*>(xs + sw128((b + 1) * 16 + fr, fg));
                    xl[(b + 1) & 1] = *reinterpret_cast<const u32x4_t*>(xs + sw128((b + 1) * 16 + fr, 4 + fg));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < NFW; ++a) acc[a][b] = mfma16t<true>(wh[a], xh[b & 1], acc[a][b]);
#pragma unroll
                for (int a = 0; a < NFW; ++a) acc[a][b] = mfma16t<true>(wh[a], xl[b & 1], acc[a][b]);
                if constexpr (NP == 3) {
#pragma unroll
                    for (int a = 0; a < NFW; ++a) acc[a][b] = mfma16t<true>(wl[a], xh[b & 1], acc[a][b]);
                }
                if (b == DPOS) TERMS_ISSUE_NEXT();      // the requests go out behind the first blocks (gemm_wide.hip: right behind the barrier they delay both waves' first MFMAs)
            }
        }
        const float al = p.alpha;
        const bool full = m0 + TM <= p.M;            // ragged tiles skip stores: their count is not wave-uniform, the next wait then takes everything
        int nstore = 0;
        if constexpr (EPI == 0) {
            float* outp = reinterpret_cast<float*>(p.out);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int m = m0 + wm * 128 + b * 16 + fr;
                float* orow = outp + (size_t)m * p.ldc + n0 + wn * 16 * NFW + 4 * fg;
#pragma unroll
                for (int a = 0; a < NFW; ++a) {
                    const f32x4_t v = acc[a][b] * al;
                    if (full || m < p.M) *reinterpret_cast<float4*>(orow + a * 16) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
            nstore = 8 * NFW;
        } else {
            // FF w1 of the tier.  Interleaved packing (ops.pack_w1_geglu): within a wave's 64 weight rows the first 32 are values, the next 32 their gates ->
            // 128 output columns per tile; lane (fr, fg) holds columns 32 wn + 16 a + 4 fg .. + 3 (a = 0, 1) of token 16 b + fr: 8 values per token block.
            static_assert(EPI == 0 || NFW == 4, "the GEGLU epilogue pairs weight fragments (a, a + 2)");
            bf16_t* outp = reinterpret_cast<bf16_t*>(p.out);
            const int Fp = p.N >> 1;
            const int code = MM_SPLIT_F16_BIT | NP | (p.terms_nodup ? MM_SPLIT_NODUP_BIT : 0);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int m = m0 + wm * 128 + b * 16 + fr;
                const bool ok = full || m < p.M;
                bf16_t* orow = outp + (size_t)m * p.ldc;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int col = cur_n * 128 + wn * 32 + a * 16 + 4 * fg;
                    const f32x4_t xv = acc[a][b] * al, gv = acc[a + 2][b] * al;
                    const mm_f32x2_t g01 = geglu_f2((mm_f32x2_t){xv[0], xv[1]}, (mm_f32x2_t){gv[0], gv[1]});
                    const mm_f32x2_t g23 = geglu_f2((mm_f32x2_t){xv[2], xv[3]}, (mm_f32x2_t){gv[2], gv[3]});
                    const float v[4] = {g01.x, g01.y, g23.x, g23.y};
                    if (ok) store_split4(orow, Fp, code, col, v);
                    if (p.ln_part) {
                        s1 += (v[0] + v[1]) + (v[2] + v[3]);
                        s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                    }
                }
                if (p.ln_part) {      // LayerNorm(inner) partials of the row's 32 columns in this wave: the four lane groups of the token, fixed order
                    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                    if (fg == 0 && ok) *reinterpret_cast<float2*>(p.ln_part + ((size_t)m * p.ln_np + cur_n * 4 + wn) * 2) = make_float2(s1, s2);
                }
            }
            nstore = 8 * 2 * (p.terms_nodup ? 2 : NP) + (p.ln_part ? 8 : 0);
        }
        pending = full ? nstore : 0;
        vb += G;
        if (vb >= total) break;
        // (no barrier here: the epilogue touches no LDS, and the next request into a stage is issued behind the next barrier)
    }
#undef TERMS_ISSUE_NEXT
#undef ISSUE_X
#undef ISSUE_W
#undef TILE_SETUP
}

template <int NP, int NFW, int EPI, int DPOS>
int launch_terms_at(GemmArgs a, hipStream_t stream) {
    constexpr int SM = 2 * X_B + 2 * 64 * NFW * ROWB;
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_terms_kernel<NP, NFW, EPI, DPOS>), hipFuncAttributeMaxDynamicSharedMemorySize, SM);
        if (e != hipSuccess) return mm_set_hip_error(e, "gemm_terms hipFuncSetAttribute");
        attr_set = true;
    }
    a.tiles_m = (a.M + TM - 1) / TM;
    a.tiles_n = a.N / (64 * NFW);
    const int total = a.tiles_m * a.tiles_n;
    hipLaunchKernelGGL((gemm_terms_kernel<NP, NFW, EPI, DPOS>), dim3(total < 256 ? total : 256), dim3(512), SM, stream, a);
    return mm_check_launch("gemm_terms_kernel");
}
template <int NP, int NFW, int EPI>
int launch_terms(GemmArgs a, hipStream_t stream) {
    return launch_terms_at<NP, NFW, EPI, 1>(a, stream);
}

// weight-tile height (in 64-row units) whose tile count fills the last round of CUs best: 4 (256 rows), or 3 (192 rows, plain epilogue only) when N is no
// multiple of 256 or the 192-row grid wastes less of its last round
int terms_nfw(const GemmArgs& a) {
    const long tm = (a.M + TM - 1) / TM;
    int best = 0;
    double best_eff = 0.;
    for (int nfw = 4; nfw >= 3; --nfw) {
        const int tn = 64 * nfw;
        if ((a.N % tn) || (nfw == 3 && a.epi != EPI_NONE)) continue;
        const long tiles = tm * (a.N / tn), rounds = (tiles + 255) / 256;
        if (g_mm_debug2 & 1) return nfw;      // tests: whatever the tile count
        if (tiles < 256) continue;
        const double eff = (double)tiles / (double)(rounds * 256);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = nfw; }
    }
    return best_eff >= 0.74 ? best : 0;
}

}  // namespace

// fp16 term-segment operands with a known product count (GemmArgs::terms = 2 / 3): plain fp32 output (q|k|v, FF w1 of the older form) or the tier's FF w1
// with GEGLU + term-split output; the segment length a multiple of 64 (NP 3) / 128 (NP 2), enough tiles to fill the chip
bool mm_gemm_terms_eligible(const GemmArgs& a) {
    if (g_mm_debug2 & 2) return false;
    if (!a.f16 || (a.terms != 2 && a.terms != 3) || a.mode != MODE_DENSE || a.bias || a.act != ACT_NONE || a.resid_bf16 || a.resid_f32 || a.fs_stats || a.m_dev ||
        a.splits > 1 || a.in_c1 || a.xb_out || a.ln_c1)
        return false;
    if (a.K % a.terms) return false;
    const int ks = a.K / a.terms;
    if ((ks % (a.terms == 3 ? 64 : 128)) || a.ldx < a.K || a.ldw < a.K || (a.ldx % 8) || (a.ldw % 8) || (((uintptr_t)a.out) & 15)) return false;
    if (a.epi == EPI_NONE) {
        if (a.out_kind != OUT_F32 || (a.ldc % 4) || a.ln_part) return false;
    } else if (a.epi == EPI_GEGLU) {
        if (a.out_kind != OUT_BF16 || (a.N % 256) || (a.ldc % 4) || a.ldc < (long)a.terms * (a.N / 2)) return false;
        if (a.ln_part && a.ln_np != a.N / 64) return false;
    } else {
        return false;
    }
    return terms_nfw(a) != 0;
}

int mm_gemm_terms_launch(GemmArgs a, hipStream_t stream) {
    if (a.alpha == 0.f) a.alpha = 1.f;
    const int nfw = terms_nfw(a);
    if (a.epi == EPI_GEGLU) return a.terms == 3 ? launch_terms<3, 4, 1>(a, stream) : launch_terms<2, 4, 1>(a, stream);
    if (a.terms == 3) return nfw == 4 ? launch_terms<3, 4, 0>(a, stream) : launch_terms<3, 3, 0>(a, stream);
    return nfw == 4 ? launch_terms<2, 4, 0>(a, stream) : launch_terms<2, 3, 0>(a, stream);
}
