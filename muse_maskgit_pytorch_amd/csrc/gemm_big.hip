// Large-tile variant of the bf16 MFMA GEMM (gemm.hip) for the token-parallel GEMMs of the transformer
// (M = tokens >= 256):  256 (m) x 128 (n) x 64 (k) per 512-thread workgroup = 8 waves as 4(m) x 2(n), each wave the same
// 64x64 = 4x4 fragment patch as the small kernel (so each output element sees the identical MFMA sequence: results are
// bit-identical between the two kernels).
//
// Why: at K = 512 the 128x128 kernel is latency-bound, not MFMA-bound (ablation in tools/gemm_bench.py: loop skeleton +
// exposed DMA latency ~ 2/3 of the time).  Here
//   * THREE 48 KiB stages with the LDS-DMA issued TWO tiles ahead and a COUNTED s_waitcnt vmcnt(6) (6 DMA instructions per
//     wave per tile), raw s_barrier -- a __syncthreads() would drain the queue with vmcnt(0);
//   * one workgroup per CU (144 KiB LDS), 2 waves per SIMD;
//   * 85 flop per LDS-DMA byte instead of 64, and half as many prologues / epilogues per output element.
// Same XOR-swizzled 128-byte rows, same LDS-staged row-contiguous epilogue as gemm.hip.
// MODE_CONV (implicit-GEMM NHWC convolution, vqgan_vae.py:224-232,255-261,271-277) is served when Cin % 64 == 0: a 64-wide
// k-tile then lies inside ONE filter tap, so the tap walk (ty, tx, channel offset) is wave-uniform running state and the
// per-lane im2col work per k-tile is a bounds test and one 64-bit multiply-add per staged row.
#include "common.h"
#include "muse_hip_internal.h"

namespace {

constexpr int BMB = 256;    // m rows per tile (tokens; CFG: 128 tokens x {cond, null})
constexpr int BNB = 128;    // n rows of W per tile
constexpr int BK = 64;
constexpr int W_BYTES = BNB * BK * 2;              // 16 KiB
constexpr int X_BYTES = BMB * BK * 2;              // 32 KiB
constexpr int STAGE_B = W_BYTES + X_BYTES;         // 48 KiB
constexpr int NSTAGE = 3;
constexpr int CT_LD = BNB + 4;                     // fp32 output tile row stride (floats)
constexpr int SMEM_B = NSTAGE * STAGE_B + BMB * 8;  // 144 KiB of stages (>= the 132 KiB fp32 output tile) + 2 KiB of per-row LayerNorm (mean, rstd)

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// source of the implicit-GEMM loader for taps that fall into the zero padding
__device__ __attribute__((aligned(16))) const unsigned int g_zero_page_big[64] = {0};

// NP (F16, MODE_DENSE only; round 5): 0 = segment packs as one contraction of depth K; 2 / 3 = term sharing as in gemm_terms.hip -- a 32-deep step stages the token
// rows as [xh(32) | xl(32)] and the weight rows as [wh(32) | wl(32)] (NP 3; NP 2: 64-byte rows [wh(32)], lane-linear -- conflict-free without a swizzle) and runs
// the products xh.wh, xl.wh (, xh.wl) of the k-block from that one staging: 48 (32) MFMAs per wave on the 48 (40) KiB that carried 32.
template <int MODE, bool F16 = false, int NP = 0>
__global__ __launch_bounds__(512) void gemm_big_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wave_m = wid >> 1, wave_n = wid & 1;
    int tile_m, tile_n;
    // (split-K: column tiles fastest inside an XCD's run -- the few column tiles of a row tile then stream the SAME long activation panel through one L2 together;
    //  with groups of 8 row tiles they sat on different XCDs and the head's dX read its 720 MB operand four times: 0.82 ms)
    xcd_grouped_tile(blockIdx.x, p.tiles_m, p.tiles_n, p.splits > 1 ? 1 : 8, tile_m, tile_n);
    const int n0 = tile_n * BNB;
    const int m0 = tile_m * (MODE == MODE_CFG ? 128 : BMB);

    // ---- per-lane DMA geometry: one instruction = 8 tile rows; lane l -> (row l>>3, physical chunk l&7) holding logical
    //      chunk (l&7) ^ (row&7).  Wave w stages W rows [16w, 16w+16) (2 instr) and X rows [32w, 32w+32) (4 instr).
    const int chunk = (lane & 7) ^ (lane >> 3);
    const int KS = NP ? p.K / NP : p.K;      // term sharing: the contraction length proper (one segment of the packs)
    // ... logical chunks 0-3 of a staged row are 32 k-values of the h plane, 4-7 the same 32 of the l plane (tokens: segment 1; weights, NP 3: segment 2).
    // Convolutions: the segments are per PIXEL (Cin = NP x channels) and per TAP of the weight row, so the plane stride is the channel count
    const int PS = (MODE == MODE_CONV && NP) ? p.Cin / NP : KS;
    const int xce = NP ? (chunk & 3) * 8 + (chunk >> 2) * PS : chunk * 8;
    const int wce = NP == 3 ? (chunk & 3) * 8 + (chunk >> 2) * 2 * PS : chunk * 8;
    const bf16_t* wptr[2];
    const bf16_t* xptr[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if constexpr (NP == 2) {      // one instruction = 16 rows of 64 bytes: lane l -> row l >> 2, chunk l & 3
            const int n = n0 + 16 * wid + (lane >> 2);
            wptr[i] = p.W + (size_t)(n < p.N ? n : 0) * p.ldw + (lane & 3) * 8;
        } else {
            const int n = n0 + 16 * wid + 8 * i + (lane >> 3);
            wptr[i] = p.W + (size_t)(n < p.N ? n : 0) * p.ldw + wce;      // clamped rows feed only unstored outputs
        }
    }
    int ciy[4], cix[4], cpix[4];      // conv: input y / x of tap (0,0) and the flat pixel index of that position, per staged row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 32 * wid + 8 * i + (lane >> 3);
        if constexpr (MODE == MODE_CFG) {
            const int wm = r >> 6, jj = r & 63;
            const int tok = m0 + wm * 32 + (jj & 31);
            xptr[i] = ((jj >> 5) ? p.X2 : p.X) + (size_t)(tok < p.M ? tok : 0) * p.ldx + chunk * 8;
        } else if constexpr (MODE == MODE_CONV) {
            const int m = m0 + r;
            const int mm = m < p.M ? m : 0;
            const int hw = p.Hv * p.Wv;
            const int cb = mm / hw;
            const int rem = mm - cb * hw;
            const int cy = rem / p.Wv, cx = rem - cy * p.Wv;
            ciy[i] = cy * p.stride + p.off_y;
            cix[i] = cx * p.stride + p.off_x;
            cpix[i] = (cb * p.Hin + ciy[i]) * p.Win + cix[i];
            xptr[i] = p.X + xce;
        } else {
            const int m = m0 + r;
            xptr[i] = p.X + (size_t)(m < p.M ? m : 0) * p.ldx + xce;
        }
    }
    int tap_y = 0, tap_x = 0, tap_c = 0;      // conv: filter tap and channel offset of the NEXT k-tile to issue (wave-uniform)
    int tap_k = 0;                            // conv with term sharing: first weight column of that tap (taps x Cin: the walk skips the l / repeated segments)

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define ISSUE_TILE(kt_, stage_)                                                                              \
    {                                                                                                        \
        const int k0_ = (MODE == MODE_CONV && NP) ? tap_k + tap_c : (kt_) * (NP ? 32 : BK);                  \
        unsigned char* ws_ = smem + (stage_) * STAGE_B + wid * (NP == 2 ? 1024 : 2048);                      \
        unsigned char* xs_ = smem + (stage_) * STAGE_B + W_BYTES + wid * 4096;                               \
        _Pragma("unroll") for (int i = 0; i < (NP == 2 ? 1 : 2); ++i)                                        \
            __builtin_amdgcn_global_load_lds(wptr[i] + k0_, (lds_ptr_t)(ws_ + i * 1024), 16, 0, 0);          \
        if constexpr (MODE != MODE_CONV) {                                                                   \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
                __builtin_amdgcn_global_load_lds(xptr[i] + k0_, (lds_ptr_t)(xs_ + i * 1024), 16, 0, 0);      \
        } else {                                                                                             \
            const int dpix_ = tap_y * p.Win + tap_x;                                                         \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
                const bool ok_ = (unsigned)(ciy[i] + tap_y) < (unsigned)p.Hin && (unsigned)(cix[i] + tap_x) < (unsigned)p.Win; \
                const bf16_t* src_ = xptr[i] + ((size_t)(cpix[i] + dpix_) * p.Cin + tap_c);                  \
                if (!ok_) src_ = reinterpret_cast<const bf16_t*>(g_zero_page_big);                           \
                __builtin_amdgcn_global_load_lds(src_, (lds_ptr_t)(xs_ + i * 1024), 16, 0, 0);               \
            }                                                                                                \
            tap_c += NP ? 32 : BK;                                                                           \
            if (tap_c >= (NP ? PS : p.Cin)) { tap_c = 0; tap_k += p.Cin; if (++tap_x == p.TW) { tap_x = 0; ++tap_y; } } \
        }                                                                                                    \
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // split-K (round 6; plain dense fp32-output products with a LONG contraction and few tiles -- the training head's dX = dlogits . W over the vocabulary):
    // blockIdx.y contracts its K / splits share into its own fp32 slab, the caller sums the slabs in a fixed order (k_colsum)
    const int nsplit = (MODE == MODE_DENSE && !NP && !F16 && p.splits > 1) ? p.splits : 1;
    if constexpr (MODE == MODE_DENSE && !NP && !F16) {
        if (nsplit > 1) {
            const int kbase = (int)blockIdx.y * (p.K / nsplit);
#pragma unroll
            for (int i = 0; i < 2; ++i) wptr[i] += kbase;
#pragma unroll
            for (int i = 0; i < 4; ++i) xptr[i] += kbase;
        }
    }
    const int fr = lane & 15, fg = lane >> 4;
    const int KT = NP ? KS / 32 : p.K / nsplit / BK;
    // fp32 residual (out = x + ...: attention out-projection, FF w2): this lane's 16 x 16 B of the tile's 128 KiB are fetched FIRST, ahead
    // of the DMA (VMEM returns in order, so the counted waits below see them retire before k-tile 0).  Read in the epilogue, the
    // residual made the write-out a read-modify-write latency tail of every workgroup at once (gemm.hip: 16 k vs 2.3 k cycles).
    constexpr bool RESID_PF = MODE == MODE_DENSE;
    float4 rres[RESID_PF ? BMB / 16 : 1];
    const bool resid_pf = RESID_PF && p.resid_f32 && p.out_kind == OUT_F32 && p.epi != EPI_GEGLU && (p.N % 4) == 0;
    if constexpr (RESID_PF) {
        if (resid_pf) {
            const int n_ = n0 + (t & 31) * 4;
#pragma unroll
            for (int pass = 0; pass < BMB / 16; ++pass) {
                const int m_ = m0 + pass * 16 + (t >> 5);
                rres[pass] = (m_ < p.M && n_ < p.N) ? *reinterpret_cast<const float4*>(p.resid_f32 + (size_t)m_ * p.ldr + n_) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    // LayerNorm(inner) folded into this GEMM (GemmArgs::ln_c1): per-row mean / rstd from the w1 kernel's partial sums, parked in LDS
    float2* ln_stat = reinterpret_cast<float2*>(smem + NSTAGE * STAGE_B);
    if constexpr (RESID_PF) {
        if (p.ln_c1) {      // 512 threads = 2 per tile row
            const int r_ = t >> 1, m_ = m0 + r_;
            const float2 st = ln_stats_from_partials(p.ln_part, p.ln_np, m_ < p.M ? m_ : 0, p.ln_F, t & 1, m_ < p.M);
            if (!(t & 1)) ln_stat[r_] = st;
        }
    }
    ISSUE_TILE(0, 0);
    if (KT > 1) ISSUE_TILE(1, 1);

    for (int kt = 0; kt < KT; ++kt) {
        // tile kt has landed once at most the NEXT tile's 6 DMA instructions of this wave are still outstanding
        if (kt + 1 < KT && !(p.debug & 2)) {
            if constexpr (NP == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // (one weight instruction per wave and step)
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my ds_reads of the previous tile are complete
        __builtin_amdgcn_s_barrier();                          // everybody's DMA of tile kt landed; stage (kt+2)%3 is free
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < KT && !(p.debug & 2)) ISSUE_TILE(kt + 2, (kt + 2) % NSTAGE);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* ws = smem + (kt % NSTAGE) * STAGE_B;
        const unsigned char* xs = ws + W_BYTES;
        if constexpr (NP != 0) {      // term sharing: every product of this 32-deep k-block from one staging of its term planes
            u32x4_t wh[4], wl[4], xh[4], xl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (NP == 3) {
                    wh[i] = *reinterpret_cast<const u32x4_t*>(ws + lds_off(wave_n * 64 + i * 16 + fr, fg));
                    wl[i] = *reinterpret_cast<const u32x4_t*>(ws + lds_off(wave_n * 64 + i * 16 + fr, 4 + fg));
                } else {
                    wh[i] = *reinterpret_cast<const u32x4_t*>(ws + (wave_n * 64 + i * 16 + fr) * 64 + fg * 16);
                }
                xh[i] = *reinterpret_cast<const u32x4_t*>(xs + lds_off(wave_m * 64 + i * 16 + fr, fg));
                xl[i] = *reinterpret_cast<const u32x4_t*>(xs + lds_off(wave_m * 64 + i * 16 + fr, 4 + fg));
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = mfma16t<F16>(wh[a], xh[b], acc[a][b]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = mfma16t<F16>(wh[a], xl[b], acc[a][b]);
            if constexpr (NP == 3) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = mfma16t<F16>(wl[a], xh[b], acc[a][b]);
            }
            continue;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4_t af[4], bfm[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const u32x4_t*>(ws + lds_off(wave_n * 64 + i * 16 + fr, ks * 4 + fg));
                bfm[i] = *reinterpret_cast<const u32x4_t*>(xs + lds_off(wave_m * 64 + i * 16 + fr, ks * 4 + fg));
            }
            if (!(p.debug & 4)) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = mfma16t<F16>(af[a], bfm[b], acc[a][b]);
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][0][0] += __uint_as_float(af[a][0] ^ bfm[a][1]);
            }
        }
    }
    if ((p.debug & 1) && acc[0][0][0] != 12345.678f) return;
    if constexpr (F16) {      // fp16 term products: undo the power-of-two scale of the packed weight terms (exact)
        const float al = p.alpha;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] *= al;
    }

    // ---- epilogue: fp32 tile through LDS, row-contiguous 16-byte write-out (see gemm.hip)
    constexpr int MT = (MODE == MODE_CFG) ? 2 : 4;
    constexpr int TROWS = (MODE == MODE_CFG) ? 128 : BMB;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* ct = reinterpret_cast<float*>(smem);
    const bool geglu = (MODE == MODE_DENSE) && p.epi == EPI_GEGLU;
#pragma unroll
    for (int b = 0; b < MT; ++b) {
        const int ml = (MODE == MODE_CFG) ? (wave_m * 32 + b * 16 + fr) : (wave_m * 64 + b * 16 + fr);
        if (geglu) {      // see gemm.hip: fragments 0,1 = gelu half, 2,3 = gate half of the same 32 output columns
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int nl = wave_n * 32 + a * 16 + fg * 4;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = geglu_f(acc[a][b][r], acc[a + 2][b][r]);
                *reinterpret_cast<float4*>(ct + ml * CT_LD + nl) = make_float4(v[0], v[1], v[2], v[3]);
            }
            continue;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int nl = wave_n * 64 + a * 16 + fg * 4;
            float v[4];
            if constexpr (MODE == MODE_CFG) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float c = acc[a][b][r], nlv = acc[a][(b + 2) & 3][r];
                    v[r] = nlv + (c - nlv) * p.cfg_scale;      // muse_maskgit_pytorch.py:254
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[a][b][r];
            }
            if (p.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (n0 + nl + r < p.N) ? p.bias[n0 + nl + r] : 0.f;
            }
            if (p.act == ACT_LEAKY) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.1f * v[r];   // vqgan_vae.py:103-104
            }
            *reinterpret_cast<float4*>(ct + ml * CT_LD + nl) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    __syncthreads();
    auto out_row = [&](int m) -> size_t {
        if constexpr (MODE == MODE_CONV) {
            const int hw = p.Hv * p.Wv;
            const int ob = m / hw;
            const int rem = m - ob * hw;
            const int oy = (rem / p.Wv) * p.os + p.py, ox = (rem % p.Wv) * p.os + p.px;
            return ((size_t)ob * p.Hout + oy) * p.Wout + ox;
        } else {
            return (size_t)m;
        }
    };

    if (geglu) {
        const int c8 = (t & 7) * 8;
        const int no = tile_n * 64 + c8;
#pragma unroll 4
        for (int pass = 0; pass < BMB / 64; ++pass) {
            const int ml = pass * 64 + (t >> 3);
            const int m = m0 + ml;
            const float4 lo = *reinterpret_cast<const float4*>(ct + ml * CT_LD + c8);
            const float4 hi = *reinterpret_cast<const float4*>(ct + ml * CT_LD + c8 + 4);
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const uint4 pk = pack8(v);
            if (p.ln_part) {      // LayerNorm(inner) partial sums of this row's 64 columns (common.h); all 8 lanes of a row take part
                const float2 st = ln_partial_row64(pk);
                if ((t & 7) == 0 && m < p.M) *reinterpret_cast<float2*>(p.ln_part + ((size_t)m * p.ln_np + tile_n) * 2) = st;
            }
            if (m >= p.M || no >= p.N / 2) continue;
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldc + no) = pk;
        }
        return;
    }

    if (p.out_kind == OUT_F32) {
        const int c4 = (t & 31) * 4;
        const int n = n0 + c4;
        if constexpr (RESID_PF) {
            if (resid_pf) {      // the residual is already in registers: a pure store phase
                float4 lc1 = make_float4(0.f, 0.f, 0.f, 0.f), lc2 = lc1;
                if (p.ln_c1 && n < p.N) { lc1 = *reinterpret_cast<const float4*>(p.ln_c1 + n); lc2 = *reinterpret_cast<const float4*>(p.ln_c2 + n); }
                float4 arow = make_float4(0.f, 0.f, 0.f, 0.f);      // LayerNorm(dim) fold, producer side (GemmArgs::xb_out): see below
                if (p.add_row && n < p.N) arow = *reinterpret_cast<const float4*>(p.add_row + n);
                const bool fold_out = p.xb_out != nullptr;         // (wave-uniform)
#pragma unroll
                for (int pass = 0; pass < BMB / 16; ++pass) {
                    const int ml = pass * 16 + (t >> 5);
                    const int m = m0 + ml;
                    const bool ok_ = m < p.M && n < p.N;
                    if (!fold_out && !ok_) continue;
                    if (fold_out && __ballot(ok_) == 0ull) continue;
                    // statistics first, retired, then the tile row: see the same loop in gemm.hip (the 128x128 kernel's round-1 failure)
                    float2 st = make_float2(0.f, 1.f);
                    if (p.ln_c1) { st = ln_stat[ml]; __builtin_amdgcn_s_waitcnt(0xC07F); }
                    float4 cv = *reinterpret_cast<const float4*>(ct + ml * CT_LD + c4);
                    if (p.ln_c1) {      // z = (a - mean) * rstd * gamma + beta contracted with W2:  rstd * (a . W2g) - rstd * mean * c1 + c2
                        const float rs = st.y, rm = st.x * st.y;
                        cv.x = rs * cv.x - rm * lc1.x + lc2.x; cv.y = rs * cv.y - rm * lc1.y + lc2.y;
                        cv.z = rs * cv.z - rm * lc1.z + lc2.z; cv.w = rs * cv.w - rm * lc1.w + lc2.w;
                    }
                    float4 o = make_float4(cv.x + rres[pass].x, cv.y + rres[pass].y, cv.z + rres[pass].z, cv.w + rres[pass].w);
                    if (p.add_row && m >= p.add_row_from) { o.x += arow.x; o.y += arow.y; o.z += arow.z; o.w += arow.w; }      // (acc + resid) + add_row
                    if (ok_) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n) = o;
                    if (fold_out) {
                        // the new residual row also leaves as bf16 (the operand of the GEMM behind the next LayerNorm, which then needs no pass of its own)
                        // together with this tile's two 64-column shares of the row's (sum, sum of squares), taken from the fp32 values
                        if (!ok_) o = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (ok_) *reinterpret_cast<uint2*>(p.xb_out + (size_t)m * p.ldxb + n) = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
                        float2 st2 = row_stats16((o.x + o.y) + (o.z + o.w), (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w));
                        if (p.st_gran == 64) {      // two partials per 128-column tile (N <= 512: no cross-row exchange)
                            const int pi_ = tile_n * 2 + ((t >> 4) & 1);      // (a width of an odd number of 64-column blocks has no second half in its last tile)
                            if ((t & 15) == 0 && m < p.M && pi_ < p.st_np) *reinterpret_cast<float2*>(p.st_part + ((size_t)m * p.st_np + pi_) * 2) = st2;
                        } else {                    // one per tile: wide rows keep the consumers' partial count small (it rides in their LDS)
                            st2.x += __shfl_xor(st2.x, 16, 64);
                            st2.y += __shfl_xor(st2.y, 16, 64);
                            if ((t & 31) == 0 && m < p.M) *reinterpret_cast<float2*>(p.st_part + ((size_t)m * p.st_np + tile_n) * 2) = st2;
                        }
                    }
                }
                return;
            }
        }
#pragma unroll 4
        for (int pass = 0; pass < TROWS / 16; ++pass) {
            const int ml = pass * 16 + (t >> 5);
            const int m = m0 + ml;
            if (m >= p.M || n >= p.N) continue;
            const float4 cv = *reinterpret_cast<const float4*>(ct + ml * CT_LD + c4);
            float v[4] = {cv.x, cv.y, cv.z, cv.w};
            const bool full = n + 3 < p.N;
            const size_t orow = out_row(m);
            if (p.resid_f32) {
                const float* rp = p.resid_f32 + orow * p.ldr + n;
                if (full) {
                    const float4 r0 = *reinterpret_cast<const float4*>(rp);
                    v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += rp[r];
                }
            }
            float* op = reinterpret_cast<float*>(p.out) + (nsplit > 1 ? (size_t)blockIdx.y * p.split_stride : (size_t)0) + orow * p.ldc + n;
            if (full) *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < p.N) op[r] = v[r];
            }
        }
    } else {
        const int c8 = (t & 15) * 8;
        const int n = n0 + c8;
#pragma unroll 4
        for (int pass = 0; pass < TROWS / 32; ++pass) {
            const int ml = pass * 32 + (t >> 4);
            const int m = m0 + ml;
            if (m >= p.M || n >= p.N) continue;
            const float4 lo = *reinterpret_cast<const float4*>(ct + ml * CT_LD + c8);
            const float4 hi = *reinterpret_cast<const float4*>(ct + ml * CT_LD + c8 + 4);
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const size_t orow = out_row(m);
            if (p.resid_bf16) {
                const bf16_t* rp = p.resid_bf16 + orow * p.ldr + n;
                if (n + 7 < p.N) {
                    float rv[8];
                    if (F16 && p.half_io) unpack8s<true>(*reinterpret_cast<const uint4*>(rp), rv); else unpack8(*reinterpret_cast<const uint4*>(rp), rv);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += rv[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) if (n + r < p.N) v[r] += (F16 && p.half_io) ? ld16s<true>(rp[r]) : bf16_to_f32(rp[r]);
                }
            }
            bf16_t* op = reinterpret_cast<bf16_t*>(p.out) + orow * p.ldc + n;
            if (n + 7 < p.N) *reinterpret_cast<uint4*>(op) = (F16 && p.half_io) ? pack8s<true>(v) : pack8(v);
            else {
#pragma unroll
                for (int r = 0; r < 8; ++r) if (n + r < p.N) op[r] = (F16 && p.half_io) ? st16s<true>(v[r]) : f32_to_bf16(v[r]);
            }
        }
    }
}

template <int MODE, bool F16 = false, int NP = 0>
int launch_big(const GemmArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>((gemm_big_kernel<MODE, F16, NP>)),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
        if (e != hipSuccess) return mm_set_hip_error(e, "gemm_big hipFuncSetAttribute");
        attr_set = true;
    }
    const int blocks = a.tiles_m * a.tiles_n;
    hipLaunchKernelGGL((gemm_big_kernel<MODE, F16, NP>), dim3(blocks, (MODE == MODE_DENSE && !NP && !F16 && a.splits > 1) ? a.splits : 1), dim3(512), SMEM_B, stream, a);
    return mm_check_launch("gemm_big_kernel");
}

}  // namespace

// GEMMs large enough to fill 256-row tiles; convolutions only when a k-tile stays inside one filter tap
bool mm_gemm_big_eligible(const GemmArgs& a) {
    if (a.out_kind == OUT_NCHW_F32) return false;
    if (a.mode == MODE_CONV && ((a.Cin % BK) != 0 || a.K != a.Ktrue)) return false;
    const int tok = a.mode == MODE_CFG ? 128 : BMB;
    const long tiles = (long)((a.M + tok - 1) / tok) * ((a.N + BNB - 1) / BNB);
    return a.M >= 2 * tok && a.N >= BNB && tiles >= 256;      // one workgroup per CU: fewer tiles than CUs idles the chip
}

// split-K on the 256 x 128 tile: a long contraction whose 128 x 128 tiling is bound by operand traffic (64 flop per staged byte) -- each split keeps >= 4096 of K
bool mm_gemm_big_split_eligible(const GemmArgs& a) {
    return a.mode == MODE_DENSE && !a.f16 && a.splits > 1 && a.out_kind == OUT_F32 && !a.resid_f32 && !a.bias && a.epi == EPI_NONE && !a.m_dev && !a.ln_c1 && !a.xb_out &&
           a.M >= 2 * BMB && a.N >= BNB && (a.N % 4) == 0 && ((a.K / BK) % a.splits) == 0 && a.K / a.splits >= 4096;
}

int mm_gemm_big_launch(GemmArgs a, hipStream_t stream) {
    if (a.splits > 1 && !mm_gemm_big_split_eligible(a)) return mm_set_error(MM_ERR_UNSUPPORTED, "gemm_big: split-K needs a plain dense bf16 product with >= 4096 of K per split");
    a.tiles_n = (a.N + BNB - 1) / BNB;
    const int tm = a.mode == MODE_CFG ? 128 : BMB;
    a.tiles_m = (a.M + tm - 1) / tm;
    if (a.f16 && a.mode == MODE_DENSE && (a.terms == 2 || a.terms == 3) && !(g_mm_debug2 & 2) && !(a.debug & (1 | 2 | 4)) && (a.K % a.terms) == 0 &&
        ((a.K / a.terms) % 32) == 0 && a.ldx >= a.K && a.ldw >= a.K)      // term sharing (round 5): equal-length term segments, every term plane staged once
        return a.terms == 3 ? launch_big<MODE_DENSE, true, 3>(a, stream) : launch_big<MODE_DENSE, true, 2>(a, stream);
    // ... convolutions: segments per pixel / per tap, a 32-channel step inside one tap.  Three products only: with two the step count and the MFMAs per step equal the
    // concatenated form's and the loader's per-instruction address arithmetic doubles per flop -- measured 0.45 ms per generate SLOWER (VAE part 15.99 vs 15.54 ms, same box)
    if (a.f16 && a.mode == MODE_CONV && a.terms == 3 && !(g_mm_debug2 & 2) && !(a.debug & (1 | 2 | 4)) && (a.Cin % 3) == 0 && ((a.Cin / 3) % 32) == 0 && a.K == a.Ktrue)
        return launch_big<MODE_CONV, true, 3>(a, stream);
    if (a.f16) return a.mode == MODE_CONV ? launch_big<MODE_CONV, true>(a, stream) : launch_big<MODE_DENSE, true>(a, stream);      // (mm_gemm_launch admits dense / conv only)
    if (a.mode == MODE_CFG) return launch_big<MODE_CFG>(a, stream);
    if (a.mode == MODE_CONV) return launch_big<MODE_CONV>(a, stream);
    return launch_big<MODE_DENSE>(a, stream);
}
