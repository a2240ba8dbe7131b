// bf16 MFMA GEMM family for gfx950:  out[m][n] = sum_k X[m][k] * W[n][k]   (nn.Linear layout, K contiguous
// in both operands), fp32 accumulate.  One main loop serves
//   MODE_DENSE : Linear layers of the transformer            (muse_maskgit_pytorch.py:85,88,118-124,225,233)
//   MODE_CFG   : to_logits on the cond AND null rows of one token at once, epilogue writes
//                null + (cond - null) * cond_scale            (muse_maskgit_pytorch.py:250-254)
//   MODE_CONV  : implicit-GEMM NHWC convolution / one parity class of ConvTranspose2d(4,2,1)
//                                                            (vqgan_vae.py:224-232, 255-261, 271-277)
//
// Tiling (CDNA4): 128 (n) x 128 (m) x 64 (k) per 256-thread workgroup, 4 waves as 2x2, each wave a
// 64x64 patch = 4x4 v_mfma_f32_16x16x32_bf16 fragments.  The WEIGHT tile is the MFMA A operand and the
// activation tile the B operand, so one accumulator fragment holds 4 CONSECUTIVE output features of one
// token: the epilogue stores 16 B (fp32) / 8 B (bf16) per lane instead of 4 scalars.
// LDS: two 32 KiB stages (W tile + X tile, row = 128 B = 8 chunks of 16 B, chunk index XOR (row & 7) so
// every ds_read_b128 lane group hits 16 distinct slots).  Staging is LDS-DMA (global_load_lds_dwordx4, 1 KiB =
// 8 tile rows per wave instruction): no VGPR round trip and -- what matters on CDNA4 -- no ds_write_b128, which at
// ~13 LDS cycles each made the register-staged first version of this kernel LDS-bound.  The DMA writes lane-linear,
// so the XOR swizzle is applied to each lane's SOURCE address.  The next tile's DMA is issued before the current
// tile's MFMAs; one __syncthreads (vmcnt(0) + barrier) per K-tile.
// Grid: 1-D, XCD-aware grouped tile order (common.h).
#include "common.h"
#include "muse_hip_internal.h"

#ifdef MM_GEMM_TIMING      // tools/small_gemm_timing.py only: shader-clock stamps of workgroup 0 / wave 0
__device__ unsigned long long g_gemm_stamps[64];
#define TSTAMP(i_) if (blockIdx.x == 0 && threadIdx.x < 64) { g_gemm_stamps[i_] = __builtin_readcyclecounter(); }
#else
#define TSTAMP(i_)
#endif

namespace {

constexpr int BT = 128;   // tile edge (both n and m)
constexpr int BK = 64;    // k per stage
constexpr int STAGE_BYTES = 2 * BT * BK * 2;   // W tile + X tile
constexpr int SMEM_BYTES = BT * (BT + 4) * 4 + BT * 8 + BT * 8;  // max(2 stages = 64 KiB, fp32 output tile with padded rows = 66 KiB) + 1 KiB of per-row LayerNorm (mean, rstd) + 1 KiB of the LayerNorm(dim) fold's c1 | c2 tile entries

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// source of the implicit-GEMM loader for taps that fall into the zero padding
__device__ __attribute__((aligned(16))) const unsigned int g_zero_page[64] = {0};

// BM: token rows per tile.  128 everywhere, except dense launches whose 128 x 128 tiles would leave a CU with fewer than two workgroups (the cross-attention's
// projections: 8192 rows x 512 columns = 256 tiles, ONE four-wave workgroup per CU, i.e. one wave per SIMD with every latency exposed -- 16.9 us for 4.3 GFLOP):
// there BM = 64 gives each CU two or three independent workgroups whose barrier phases drift apart and cover one another (round 4).  Same MFMA order per
// output element: bit-identical results.
// NP (F16, MODE_DENSE; round 5): 0 = segment packs as one contraction of depth K; 2 / 3 = term sharing as in gemm_terms.hip / gemm_big.hip -- 32-deep steps on
// [xh(32) | xl(32)] token rows and [wh(32) | wl(32)] weight rows (NP 2: 64-byte rows [wh(32)], lane-linear), every product of the k-block from one staging.
template <int MODE, bool F16 = false, int BM = 128, int NP = 0>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs p) {
    static_assert(BM == 128 || (BM == 64 && MODE == MODE_DENSE), "64-token tiles: dense mode only");
    static_assert(NP == 0 || (F16 && MODE == MODE_DENSE), "term sharing: dense fp16 term operands only");
    constexpr int MB = BM / 32;      // token fragments per wave (a wave: 64 weight rows x BM / 2 tokens)
    // Term sharing on 64-token tiles (the tier's cross-attention projections): a 32-deep step carries only 24 MFMAs per wave, far less than one L2 round trip, so
    // the two-stage loop below (request step kt + 1, compute step kt, wait) exposed the DMA latency at every step (28 us for 19 GFLOP, matrix pipe 0.25).  These
    // instantiations run THREE stages of 24 KiB (weights 16 + tokens 8) with the requests two steps ahead and a counted vmcnt, as gemm_big.hip; still two workgroups per CU.
    constexpr bool P3 = NP != 0 && BM == 64;
    constexpr int STG = P3 ? BT * BK * 2 + BM * BK * 2 : STAGE_BYTES;
    constexpr int TAIL = (P3 && 3 * STG > BT * (BT + 4) * 4) ? 3 * STG : BT * (BT + 4) * 4;      // LayerNorm statistics / c1 | c2 entries live behind the stages and the output tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TSTAMP(0)
    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int wave_n = wid >> 1, wave_m = wid & 1;
    int tile_m, tile_n;
    xcd_grouped_tile(blockIdx.x, p.tiles_m, p.tiles_n, 16, tile_m, tile_n);
    const int n0 = tile_n * BT;
    const int m0 = tile_m * (MODE == MODE_CFG ? 64 : BM);   // CFG: 64 tokens x {cond, null} per tile
    // device-side row count (the per-row fallback of the fused sampler: the number of rows to redo is only known on the device): tiles
    // beyond it leave at once, so a launch sized for the capacity costs a launch and nothing else when no row failed
    if (p.m_dev && m0 >= *p.m_dev) return;

    // ---- per-lane DMA geometry.  Wave w stages tile rows [32w, 32w+32) with 4 instructions of 8 rows each; within an
    //      instruction lane l lands on (row l>>3, physical chunk l&7), which must hold logical chunk (l&7) ^ (row & 7).
    const int chunk = (lane & 7) ^ (lane >> 3);
    const int KS = NP ? p.K / NP : p.K;      // term sharing: the contraction length proper; logical chunks 0-3 = 32 k-values of the h plane, 4-7 = of the l plane
    const int xce = NP ? (chunk & 3) * 8 + (chunk >> 2) * KS : chunk * 8;
    const int wce = NP == 3 ? (chunk & 3) * 8 + (chunk >> 2) * 2 * KS : chunk * 8;
    const int row0 = 32 * wid + (lane >> 3);          // rows row0 + 8*i
    constexpr int XI = BM / 32;                       // X tile: this wave's BM / 4 rows = XI instructions (rows xrow0 + 8*i)
    const int xrow0 = (BM / 4) * wid + (lane >> 3);
    const bf16_t* wptr[4];
    const bf16_t* xptr[4];
    bool xok[4];
    int cb[4], cy[4], cx[4];   // conv: batch / output y / output x of the staged rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 8 * i;
        const int n = n0 + r;
        // out-of-range rows are clamped to row 0: they only feed outputs the epilogue never stores
        if constexpr (NP == 2) {      // one instruction = 16 rows of 64 bytes: lane l -> row l >> 2, chunk l & 3 (two instructions per wave)
            const int n2 = n0 + 32 * wid + 16 * (i & 1) + (lane >> 2);
            wptr[i] = p.W + (size_t)(n2 < p.N ? n2 : 0) * p.ldw + (lane & 3) * 8;
        } else {
            wptr[i] = p.W + (size_t)(n < p.N ? n : 0) * p.ldw + wce;
        }
        if constexpr (MODE == MODE_DENSE) {
            const int m = m0 + xrow0 + 8 * (i < XI ? i : 0);
            xok[i] = m < p.M;
            xptr[i] = p.X + (size_t)(xok[i] ? m : 0) * p.ldx + xce;
        } else if constexpr (MODE == MODE_CFG) {
            const int wm = r >> 6, jj = r & 63;
            const int tok = m0 + wm * 32 + (jj & 31);
            xok[i] = tok < p.M;
            xptr[i] = ((jj >> 5) ? p.X2 : p.X) + (size_t)(xok[i] ? tok : 0) * p.ldx + chunk * 8;
        } else {
            const int m = m0 + r;
            xok[i] = m < p.M;
            const int mm = xok[i] ? m : 0;
            const int hw = p.Hv * p.Wv;
            cb[i] = mm / hw;
            const int rem = mm - cb[i] * hw;
            cy[i] = rem / p.Wv;
            cx[i] = rem - cy[i] * p.Wv;
            xptr[i] = p.X;
        }
    }

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define ISSUE_TILE(kt_, stage_)                                                                                    \
    {                                                                                                              \
        const int k0_ = (kt_) * (NP ? 32 : BK);                                                                    \
        unsigned char* ws_ = smem + (stage_) * STG + wid * (NP == 2 ? 2048 : 4096);   /* this wave's 32 rows of the W tile */ \
        unsigned char* xs_ = smem + (stage_) * STG + BT * BK * 2 + wid * (BM * 32);   /* ... its BM / 4 rows of the X tile */ \
        _Pragma("unroll") for (int i = 0; i < (NP == 2 ? 2 : 4); ++i)                                              \
            __builtin_amdgcn_global_load_lds(wptr[i] + k0_, (lds_ptr_t)(ws_ + i * 1024), 16, 0, 0);                \
        if constexpr (MODE != MODE_CONV) {                                                                         \
            _Pragma("unroll") for (int i = 0; i < XI; ++i)                                                         \
                __builtin_amdgcn_global_load_lds(xptr[i] + k0_, (lds_ptr_t)(xs_ + i * 1024), 16, 0, 0);            \
        } else {                                                                                                   \
            /* im2col on the fly: this lane's 8 channels of K-index k belong to tap k / Cin; taps that fall in the */ \
            /* zero padding (or k >= Ktrue) are fetched from a zero page instead */                                \
            const int k_ = k0_ + chunk * 8;                                                                        \
            const int tap_ = k_ / p.Cin;                                                                           \
            const int c_ = k_ - tap_ * p.Cin;                                                                      \
            const int ty_ = tap_ / p.TW, tx_ = tap_ - ty_ * p.TW;                                                  \
            const bool kok_ = k_ < p.Ktrue;                                                                        \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                        \
                const int iy_ = cy[i] * p.stride + ty_ + p.off_y;                                                  \
                const int ix_ = cx[i] * p.stride + tx_ + p.off_x;                                                  \
                const bool ok_ = kok_ && iy_ >= 0 && iy_ < p.Hin && ix_ >= 0 && ix_ < p.Win;                       \
                const size_t off_ = (((size_t)cb[i] * p.Hin + iy_) * p.Win + ix_) * p.Cin + c_;                    \
                const bf16_t* src_ = ok_ ? p.X + off_ : reinterpret_cast<const bf16_t*>(g_zero_page);              \
                __builtin_amdgcn_global_load_lds(src_, (lds_ptr_t)(xs_ + i * 1024), 16, 0, 0);                     \
            }                                                                                                      \
        }                                                                                                          \
    }

    f32x4_t acc[4][MB];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // split-K: this workgroup covers k-tiles [kt0, kt0 + KT) and writes its partial sums to slab blockIdx.y of `out`
    const int nsplit = p.splits > 1 ? p.splits : 1;
    const int KT = NP ? KS / 32 : p.K / BK / nsplit;
    const int kt0 = (int)blockIdx.y * KT;
    void* const outp = p.splits > 1 ? static_cast<void*>(reinterpret_cast<float*>(p.out) + (size_t)blockIdx.y * p.split_stride) : p.out;
    TSTAMP(1)
    ISSUE_TILE(kt0, 0);
    // fp32 residual (attention / FF output projections: out = x + ...): this lane's 16 x 16 B of the 64 KiB tile are fetched NOW, behind
    // the first k-tile's DMA -- in the epilogue the read-modify-write of all workgroups at once was a pure latency / bandwidth tail
    // (cycle stamps, 8192 x 512 x 512: write-out 16.1 k cycles with the residual read there, 2.3 k without)
    constexpr bool RESID_PF = (MODE == MODE_DENSE);
    float4 rres[RESID_PF ? BM / 8 : 1];
    const bool resid_pf = RESID_PF && p.resid_f32 && p.out_kind == OUT_F32 && p.splits <= 1 && (p.N % 4) == 0;
    if constexpr (RESID_PF) {
        if (resid_pf) {
            const int n_ = n0 + (t & 31) * 4;
#pragma unroll
            for (int pass = 0; pass < BM / 8; ++pass) {
                const int m_ = m0 + pass * 8 + (t >> 5);
                rres[pass] = (m_ < p.M && n_ < p.N) ? *reinterpret_cast<const float4*>(p.resid_f32 + (size_t)m_ * p.ldr + n_) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    // LayerNorm(inner) folded into this GEMM (GemmArgs::ln_c1): per-row mean / rstd from the w1 kernel's partial sums, parked behind the tile
    float2* ln_stat = reinterpret_cast<float2*>(smem + TAIL);
    if constexpr (RESID_PF) {
        if (resid_pf && p.ln_c1) {      // 256 threads = 2 per tile row
            const int r_ = t >> 1, m_ = m0 + r_;
            const float2 st = ln_stats_from_partials(p.ln_part, p.ln_np, m_ < p.M ? m_ : 0, p.ln_F, t & 1, m_ < p.M && r_ < BM);
            if (!(t & 1)) ln_stat[r_] = st;
        } else if (p.in_c1) {           // LayerNorm(dim) fold, consumer side (GemmArgs::in_c1): the operand rows are raw residual rows, their statistics come with them
            if (t >= 128) {             // one thread per tile row: (rstd, -mean) by the shared routine (common.h ln_rstd_negmean)
                const int r_ = t - 128, m_ = m0 + r_;
                const bool ok_ = m_ < p.M && r_ < BM;
                ln_stat[r_] = ln_rstd_negmean(reinterpret_cast<const float2*>(p.in_part) + (size_t)(ok_ ? m_ : 0) * p.in_np, ok_ ? p.in_np : 0, 1.f / (float)p.in_F);
            }
            if (t < 64) {               // this tile's 128 entries of c1 | c2 go to LDS now: a global load in the epilogue would be pure exposed latency
                const int n_ = n0 + (t & 31) * 4;
                const float* src_ = (t < 32) ? p.in_c1 : p.in_c2;
                float4 v_ = make_float4(0.f, 0.f, 0.f, 0.f);
                if (src_ && n_ + 3 < p.N) v_ = *reinterpret_cast<const float4*>(src_ + n_);
                reinterpret_cast<float4*>(smem + TAIL + BT * 8)[t] = v_;
            }
        }
    }
    if constexpr (P3) {      // three stages: the second step goes out now (behind the residual / statistics loads: VMEM retires in order); step 0 has landed once at most
                             // that step's requests are outstanding
        constexpr int NREQ = (NP == 2 ? 2 : 4) + BM / 32;      // DMA instructions per wave and step
        if (KT > 1) {
            ISSUE_TILE(kt0 + 1, 1);
            if constexpr (NREQ == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's LDS-DMA of the first k-tile has landed (explicit: a barrier alone does not drain VMEM)
    __syncthreads();
    }
    TSTAMP(2)

    const int fr = lane & 15, fg = lane >> 4;
#define COMPUTE_TILE(stage_)                                                                                       \
    {                                                                                                              \
        const unsigned char* ws_ = smem + (stage_) * STG;                                                          \
        const unsigned char* xs_ = ws_ + BT * BK * 2;                                                              \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                         \
            u32x4_t af[4], bfm[MB];                                                                                \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
                af[i] = *reinterpret_cast<const u32x4_t*>(ws_ + lds_off(wave_n * 64 + i * 16 + fr, ks * 4 + fg)); \
            _Pragma("unroll") for (int i = 0; i < MB; ++i)                                                         \
                bfm[i] = *reinterpret_cast<const u32x4_t*>(xs_ + lds_off(wave_m * (BM / 2) + i * 16 + fr, ks * 4 + fg)); \
            if (!(p.debug & 4)) {                                                                                  \
            _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                          \
                _Pragma("unroll") for (int b = 0; b < MB; ++b) acc[a][b] = mfma16t<F16>(af[a], bfm[b], acc[a][b]);  \
            } else { _Pragma("unroll") for (int a = 0; a < 4; ++a) { acc[a][0][0] += __uint_as_float(af[a][0] ^ bfm[a][1]); } } \
        }                                                                                                          \
    }
#define COMPUTE_TERMS(stage_)      /* term sharing: xh.wh, xl.wh (, xh.wl) of this 32-deep k-block from one staging of its term planes */ \
    {                                                                                                              \
        const unsigned char* ws_ = smem + (stage_) * STG;                                                          \
        const unsigned char* xs_ = ws_ + BT * BK * 2;                                                              \
        u32x4_t wh[4], wl[4], xh[MB], xl[MB];                                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                            \
            if constexpr (NP == 3) {                                                                               \
                wh[i] = *reinterpret_cast<const u32x4_t*>(ws_ + lds_off(wave_n * 64 + i * 16 + fr, fg));           \
                wl[i] = *reinterpret_cast<const u32x4_t*>(ws_ + lds_off(wave_n * 64 + i * 16 + fr, 4 + fg));       \
            } else {                                                                                               \
                wh[i] = *reinterpret_cast<const u32x4_t*>(ws_ + (wave_n * 64 + i * 16 + fr) * 64 + fg * 16);       \
            }                                                                                                      \
        }                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < MB; ++i) {                                                           \
            xh[i] = *reinterpret_cast<const u32x4_t*>(xs_ + lds_off(wave_m * (BM / 2) + i * 16 + fr, fg));         \
            xl[i] = *reinterpret_cast<const u32x4_t*>(xs_ + lds_off(wave_m * (BM / 2) + i * 16 + fr, 4 + fg));     \
        }                                                                                                          \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                              \
            _Pragma("unroll") for (int b = 0; b < MB; ++b) acc[a][b] = mfma16t<F16>(wh[a], xh[b], acc[a][b]);      \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                              \
            _Pragma("unroll") for (int b = 0; b < MB; ++b) acc[a][b] = mfma16t<F16>(wh[a], xl[b], acc[a][b]);      \
        if constexpr (NP == 3) {                                                                                   \
            _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                          \
                _Pragma("unroll") for (int b = 0; b < MB; ++b) acc[a][b] = mfma16t<F16>(wl[a], xh[b], acc[a][b]);  \
        }                                                                                                          \
    }
#define COMPUTE_STEP(stage_) { if constexpr (NP != 0) COMPUTE_TERMS(stage_) else COMPUTE_TILE(stage_) }
    if constexpr (P3) {
        constexpr int NREQ = (NP == 2 ? 2 : 4) + BM / 32;
        for (int kt = 0; kt < KT; ++kt) {
            if (kt > 0) {
                // step kt has landed once at most step kt + 1's requests of this wave are outstanding; the barrier publishes it and frees stage (kt + 2) % 3
                if (kt + 1 < KT) { if constexpr (NREQ == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 2 < KT) ISSUE_TILE(kt0 + kt + 2, (kt + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);
            COMPUTE_TERMS(kt % 3);
        }
    } else {
    // steady state: the next tile's DMA is in flight while this tile's MFMAs run; one barrier per tile
    for (int kt = 0; kt < KT - 1; ++kt) {
        if (!(p.debug & 2)) ISSUE_TILE(kt0 + kt + 1, (kt + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);   // keep the DMA issue ahead of the MFMA block
        COMPUTE_STEP(kt & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile's DMA (issued above by this wave) has landed before the barrier publishes it
        __syncthreads();
        TSTAMP(3 + kt)
    }
    COMPUTE_STEP((KT - 1) & 1);
    }
    TSTAMP(40)
    if constexpr (F16) {      // fp16 term products: undo the power-of-two scale of the packed weight terms (exact)
        const float al = p.alpha;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < MB; ++b) acc[a][b] *= al;
    }

    if constexpr (MODE == MODE_DENSE) {
        if (p.in_c1) {
            // LayerNorm(dim) fold, consumer side: X held the raw residual rows, W the gains -> acc = rstd * (acc - mean * c1[n]) + c2[n], applied on the
            // accumulators (before GEGLU when there is one), the same expression in the same order as gemm_wide.hip's
            const float4* lc = reinterpret_cast<const float4*>(smem + TAIL + BT * 8);      // [32] c1 then [32] c2 of this tile's 128 columns
            float4 c1v[4], c2v[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int q4 = wave_n * 16 + a * 4 + fg;
                c1v[a] = lc[q4];
                c2v[a] = lc[32 + q4];
            }
#pragma unroll
            for (int b = 0; b < MB; ++b) {
                const float2 stv = ln_stat[wave_m * (BM / 2) + b * 16 + fr];
                const float rs = stv.x, nm = stv.y;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    acc[a][b][0] = ln_fold_apply(acc[a][b][0], rs, nm, c1v[a].x, c2v[a].x);
                    acc[a][b][1] = ln_fold_apply(acc[a][b][1], rs, nm, c1v[a].y, c2v[a].y);
                    acc[a][b][2] = ln_fold_apply(acc[a][b][2], rs, nm, c1v[a].z, c2v[a].z);
                    acc[a][b][3] = ln_fold_apply(acc[a][b][3], rs, nm, c1v[a].w, c2v[a].w);
                }
            }
        }
    }
    // ---- epilogue.  A lane holds out[m][n..n+3] per fragment (16 rows x 64 B per store instruction): storing that
    //      directly touches half cache lines and was measured at 35-45 % of the kernel.  Instead the tile goes
    //      through LDS (fp32, row stride 132 floats: conflict-free ds_write_b128) and is written out row-contiguously,
    //      16 B per lane, 512 B (fp32) / 256 B (bf16) of one row per quarter wave; the residual is read the same way.
    if ((p.debug & 1) && acc[0][0][0] != 12345.678f) return;
    constexpr int MT = (MODE == MODE_CFG) ? 2 : MB;
    constexpr int TROWS = (MODE == MODE_CFG) ? 64 : BM;

    if (p.out_kind == OUT_NCHW_F32) {
        // narrow conv head (Cout = image channels): direct scalar stores, out[b][n][y][x]
        float* op = reinterpret_cast<float*>(outp);
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            const int m = m0 + wave_m * 64 + b * 16 + fr;
            if (m >= p.M) continue;
            const int hw = p.Hv * p.Wv;
            const int ob = m / hw;
            const int rem = m - ob * hw;
            const int oy = (rem / p.Wv) * p.os + p.py, ox = (rem % p.Wv) * p.os + p.px;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int n = n0 + wave_n * 64 + a * 16 + fg * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) {
                        float v = acc[a][b][r] + (p.bias ? p.bias[n + r] : 0.f);
                        if (p.act == ACT_LEAKY) v = v > 0.f ? v : 0.1f * v;
                        op[(((size_t)ob * p.N + n + r) * p.Hout + oy) * p.Wout + ox] = v;
                    }
            }
        }
        return;
    }

    __syncthreads();                       // every wave is done reading the last stage
    TSTAMP(41)
    // Round 6 -- an unsynchronised overlay found by inspection while looking for the root cause of the round-1 LDS hazard: the fp32 staging
    // tile below OVERLAYS the k-loop's stages (rows 64 .. 127 of `ct` = bytes 33.8 .. 67.6 KiB = all of stage 1), and rounds 1-5 had NO workgroup barrier between
    // a wave's last fragment reads (its final COMPUTE_STEP, which follows the loop's last barrier) and another wave's first staging write.  The four waves leave
    // that barrier together and run the same ~600 cycles of work, so the window is a wave lagging by more than half a step (LDS arbitration against the co-resident
    // workgroup) -- rare, timing-dependent, and exactly the kind of failure the determinism stress can only screen for.  gemm_big.hip / gemm_wide.hip / gemm_fp8.hip
    // always had this barrier; this kernel now has it too: every wave's fragment reads have returned (lgkmcnt) before anybody overwrites a stage.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* ct = reinterpret_cast<float*>(smem);
    constexpr int CT_LD = BT + 4;          // floats per tile row (528 B)
    const bool geglu = (MODE == MODE_DENSE) && p.epi == EPI_GEGLU;
#pragma unroll
    for (int b = 0; b < MT; ++b) {
        const int ml = (MODE == MODE_CFG) ? (wave_m * 32 + b * 16 + fr) : (wave_m * (BM / 2) + b * 16 + fr);
        if (geglu) {
            // W rows are interleaved per wave: fragments a = 0,1 hold the gelu half, a = 2,3 the gate half of the SAME
            // 32 output columns -> gate * gelu(x) is lane-local; the tile emits 64 columns
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
    TSTAMP(42)

    if (geglu) {
        // 64 output columns per tile row (128 B bf16): 8 lanes x 8 columns per row, 32 rows per pass
        const int c8 = (t & 7) * 8;
        const int no = tile_n * 64 + c8;           // output column; the output has N/2 columns
#pragma unroll 4
        for (int pass = 0; pass < BM / 32; ++pass) {
            const int ml = pass * 32 + (t >> 3);
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
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(outp) + (size_t)m * p.ldc + no) = pack8(v);
        }
        return;
    }

    // row-contiguous write-out, 16 B per lane, consecutive lanes on consecutive addresses
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
    if (p.out_kind == OUT_F32) {
        // 4 columns per lane: 32 lanes cover one 512-byte tile row, 8 rows per pass
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
                for (int pass = 0; pass < BM / 8; ++pass) {
                    const int ml = pass * 8 + (t >> 5);
                    const int m = m0 + ml;
                    const bool ok_ = m < p.M && n < p.N;
                    if (!fold_out && !ok_) continue;
                    if (fold_out && __ballot(ok_) == 0ull) continue;
                    // The row's statistics are read -- and RETIRED -- before the tile row is requested.  Round-1 order (tile row, then the
                    // statistics right behind it, both LDS reads in flight) produced, in ~4 % of full-size launches, rstd * acc.x == 0 in the
                    // last quarter-wave (lanes 48-63) of one pass of one workgroup: one output row off by its whole feed-forward term in 16
                    // columns (DESIGN.md "Round-1 nondeterminism"; tools/determinism_stress.py reproduces and screens it).
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
                    if (ok_) *reinterpret_cast<float4*>(reinterpret_cast<float*>(outp) + (size_t)m * p.ldc + n) = o;
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
                TSTAMP(43)
                return;
            }
        }
#pragma unroll 4
        for (int pass = 0; pass < TROWS / 8; ++pass) {
            const int ml = pass * 8 + (t >> 5);
            const int m = m0 + ml;
            if (m >= p.M || n >= p.N) continue;
            const size_t orow = out_row(m);
            const float4 cv = *reinterpret_cast<const float4*>(ct + ml * CT_LD + c4);
            float v[4] = {cv.x, cv.y, cv.z, cv.w};
            const bool full = n + 3 < p.N;
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
            float* op = reinterpret_cast<float*>(outp) + orow * p.ldc + n;
            if (full) *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < p.N) op[r] = v[r];
            }
        }
    } else {
        // bf16: 8 columns per lane: 16 lanes cover one 256-byte tile row, 16 rows per pass
        const int c8 = (t & 15) * 8;
        const int n = n0 + c8;
#pragma unroll 4
        for (int pass = 0; pass < TROWS / 16; ++pass) {
            const int ml = pass * 16 + (t >> 4);
            const int m = m0 + ml;
            if (m >= p.M || n >= p.N) continue;
            const size_t orow = out_row(m);
            const float4 lo = *reinterpret_cast<const float4*>(ct + ml * CT_LD + c8);
            const float4 hi = *reinterpret_cast<const float4*>(ct + ml * CT_LD + c8 + 4);
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const bool full = n + 7 < p.N;
            if (p.resid_bf16) {
                const bf16_t* rp = p.resid_bf16 + orow * p.ldr + n;
                if (full) {
                    float rv[8];
                    if (F16 && p.half_io) unpack8s<true>(*reinterpret_cast<const uint4*>(rp), rv); else unpack8(*reinterpret_cast<const uint4*>(rp), rv);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += rv[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) if (n + r < p.N) v[r] += (F16 && p.half_io) ? ld16s<true>(rp[r]) : bf16_to_f32(rp[r]);
                }
            }
            bf16_t* op = reinterpret_cast<bf16_t*>(outp) + orow * p.ldc + n;
            if (full) *reinterpret_cast<uint4*>(op) = (F16 && p.half_io) ? pack8s<true>(v) : pack8(v);      // (round 6: fp16 storage of the single-term fp16 VAE decode)
            else {
#pragma unroll
                for (int r = 0; r < 8; ++r) if (n + r < p.N) op[r] = (F16 && p.half_io) ? st16s<true>(v[r]) : f32_to_bf16(v[r]);
            }
        }
    }
    TSTAMP(43)
}

template <int MODE, bool F16 = false, int BM = 128, int NP = 0>
int launch(const GemmArgs& a, hipStream_t stream) {
    // (three-stage term-sharing instantiations: 3 x 24 KiB of stages + the 2 KiB tail; gemm_kernel's P3)
    constexpr int SM = (NP != 0 && BM == 64) ? 3 * (BT * BK * 2 + BM * BK * 2) + BT * 8 + BT * 8 : SMEM_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>((gemm_kernel<MODE, F16, BM, NP>)),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SM);
        if (e != hipSuccess) return mm_set_hip_error(e, "gemm hipFuncSetAttribute");
        attr_set = true;
    }
    const int blocks = a.tiles_m * a.tiles_n;
    if (blocks <= 0) return MM_OK;
    hipLaunchKernelGGL((gemm_kernel<MODE, F16, BM, NP>), dim3(blocks, a.splits > 1 ? a.splits : 1), dim3(256), SM, stream, a);
    return mm_check_launch("gemm_kernel");
}

}  // namespace

int g_mm_debug = 0;
int g_mm_debug2 = 0;

// dense launch on the 128 x 128 kernel: 64-token tiles when the 128-token grid would give a CU fewer than two workgroups (see gemm_kernel)
static int launch_dense_small(GemmArgs a, hipStream_t stream) {
    a.tiles_n = (a.N + BT - 1) / BT;
    a.tiles_m = (a.M + BT - 1) / BT;
    if ((long)a.tiles_m * a.tiles_n < 512 && a.M > 64 && a.splits <= 1 && !a.m_dev) {
        a.tiles_m = (a.M + 63) / 64;
        return launch<MODE_DENSE, false, 64>(a, stream);
    }
    return launch<MODE_DENSE>(a, stream);
}

int mm_gemm_launch(GemmArgs a, hipStream_t stream) {
    a.debug = g_mm_debug;
    if (a.K <= 0 || (a.K % BK) != 0) return mm_set_error(MM_ERR_SHAPE, "gemm: K must be a positive multiple of 64 (pad at pack time)");
    if ((a.ldw % 8) != 0 || (a.mode != MODE_CONV && (a.ldx % 8) != 0)) return mm_set_error(MM_ERR_ALIGN, "gemm: row strides must be multiples of 8 elements (16 B)");
    if (a.mode == MODE_CONV && (a.Cin % 8) != 0) return mm_set_error(MM_ERR_SHAPE, "conv: Cin must be a multiple of 8");
    if (a.out_kind != OUT_NCHW_F32 && a.N >= 8 && ((a.out_kind == OUT_F32 && (a.ldc % 4)) || (a.out_kind == OUT_BF16 && (a.ldc % 8))))
        return mm_set_error(MM_ERR_ALIGN, "gemm: ldc must be a multiple of 4 (fp32 out) / 8 (bf16 out) elements");
    if ((a.resid_f32 && (a.ldr % 4)) || (a.resid_bf16 && (a.ldr % 8))) return mm_set_error(MM_ERR_ALIGN, "gemm: residual stride alignment");
    if ((a.resid_f32 && a.out_kind != OUT_F32) || (a.resid_bf16 && a.out_kind != OUT_BF16))
        return mm_set_error(MM_ERR_DTYPE, "gemm: the residual must have the output's dtype");
    if (a.epi == EPI_GEGLU && (a.mode != MODE_DENSE || (a.N % 128) || a.out_kind != OUT_BF16 || a.bias || a.resid_f32 || a.resid_bf16))
        return mm_set_error(MM_ERR_SHAPE, "gemm: GEGLU epilogue needs a dense bf16 GEMM with N % 128 == 0");
    if (a.splits > 1) {
        if (a.mode != MODE_DENSE || a.out_kind != OUT_F32 || a.resid_f32 || a.bias || a.epi != EPI_NONE || (a.K / BK) % a.splits)
            return mm_set_error(MM_ERR_SHAPE, "gemm: split-K needs a plain dense fp32-output GEMM with K/64 divisible by the split count");
        if (mm_gemm_big_split_eligible(a)) return mm_gemm_big_launch(a, stream);      // (round 6: long contractions, see gemm_big.hip)
        a.tiles_n = (a.N + BT - 1) / BT;
        a.tiles_m = (a.M + BT - 1) / BT;
        return launch<MODE_DENSE>(a, stream);
    }
    if (a.ln_part && !a.ln_c1) {
        // FF w1 that also emits the LayerNorm(inner) partial sums (every GEGLU epilogue of the family does, identically)
        if (a.epi != EPI_GEGLU) return mm_set_error(MM_ERR_UNSUPPORTED, "gemm: LayerNorm partial sums ride on the GEGLU epilogue");
        a.ln_np = a.f16 ? a.N / 64 : a.N / 128;      // (the tier's w1 on gemm_terms.hip: one partial per 32 output columns)
    }
    if (a.ln_c1) {
        // FF w2 with the LayerNorm(inner) folded in: the fp32-residual epilogue of the 128x128 / 256x128 kernels
        if (a.mode != MODE_DENSE || !a.resid_f32 || a.out_kind != OUT_F32 || (a.N % 4) || !a.ln_part || !a.ln_c2 || a.ln_np <= 0 || a.ln_F <= 0 ||
            a.splits > 1)
            return mm_set_error(MM_ERR_UNSUPPORTED, "gemm: folded LayerNorm needs a dense fp32-residual GEMM");
    }
    if (a.fs_stats && a.f16) {      // fp16 term operands: the emission exists on the 256 x 256 persistent kernel only (model.hip checks eligibility first)
        if (!mm_gemm_wide_fused_eligible(a)) return mm_set_error(MM_ERR_UNSUPPORTED, "gemm: fused sampling on fp16 term operands needs the 256 x 256 logits kernel (>= 1024 rows)");
        return mm_gemm_wide_fused_launch(a, stream);
    }
    if (a.fs_stats) {      // fused sampling: only the 256-column guidance kernel implements the emission (model.hip checks eligibility first)
        if ((a.mode != MODE_CFG && !(a.mode == MODE_DENSE && a.wide_tok)) || !mm_gemm_cfg2_eligible(a))
            return mm_set_error(MM_ERR_UNSUPPORTED, "gemm: fused sampling needs the 256-column guidance-logits kernel");
        return mm_gemm_cfg2_launch(a, stream);
    }
    if (a.f16) {      // fp16 term operands ('f16x2' tier): fp32 output, dense / convolution; the 256x128 kernel when it fills the chip, else the 128x128 one
        if (a.alpha == 0.f) a.alpha = 1.f;
        if (a.epi == EPI_GEGLU) {      // the tier's FF w1 with GEGLU + term-split output: gemm_terms.hip only (model.hip checks eligibility first)
            if (!mm_gemm_terms_eligible(a)) return mm_set_error(MM_ERR_UNSUPPORTED, "gemm: the GEGLU epilogue on fp16 term operands needs the term-sharing kernel's shape class");
            return mm_gemm_terms_launch(a, stream);
        }
        if (((a.out_kind == OUT_BF16 || a.resid_bf16) && !a.half_io) || a.mode == MODE_CFG || a.epi != EPI_NONE || (a.ln_part && !a.ln_c1) || a.splits > 1)
            return mm_set_error(MM_ERR_UNSUPPORTED, "gemm: fp16 term operands take the fp32-output dense / convolution forms only (16-bit fp16 outputs: half_io)");
        if (a.half_io && (a.terms || a.xb_out || a.in_c1 || a.ln_c1)) return mm_set_error(MM_ERR_UNSUPPORTED, "gemm: half_io is the plain single-term fp16 form");
        if (a.half_io && mm_gemm_wide_conv_eligible(a)) return mm_gemm_wide_conv_launch(a, stream);      // round 6: convolutions on the persistent 256 x 256 x 64 tile
        if (a.head_w || a.par_w[1]) return mm_set_error(MM_ERR_UNSUPPORTED, "gemm: the fused head / the parity-batched form need the 256 x 256 convolution kernel's shape class");
        if (!(a.debug & (8 | (1 << 30))) && mm_gemm_terms_eligible(a)) return mm_gemm_terms_launch(a, stream);
        if (!a.m_dev && !(a.debug & 8) && mm_gemm_big_eligible(a)) return mm_gemm_big_launch(a, stream);
        a.tiles_n = (a.N + BT - 1) / BT;
        a.tiles_m = (a.M + BT - 1) / BT;
        // term sharing (round 5; equal-length term segments stated by the caller): every term plane staged once, the products of a k-block from that staging
        const int np = (a.mode == MODE_DENSE && (a.terms == 2 || a.terms == 3) && !(g_mm_debug2 & 2) && !(a.debug & (1 | 2 | 4)) && (a.K % a.terms) == 0 &&
                        ((a.K / a.terms) % 32) == 0 && a.ldx >= a.K && a.ldw >= a.K) ? a.terms : 0;
        if (a.mode == MODE_DENSE && (long)a.tiles_m * a.tiles_n < 512 && a.M > 64 && !a.m_dev) {      // 64-token tiles, as launch_dense_small (the tier's cross-attention projections)
            a.tiles_m = (a.M + 63) / 64;
            return np == 3 ? launch<MODE_DENSE, true, 64, 3>(a, stream) : np == 2 ? launch<MODE_DENSE, true, 64, 2>(a, stream) : launch<MODE_DENSE, true, 64>(a, stream);
        }
        if (np == 3) return launch<MODE_DENSE, true, 128, 3>(a, stream);
        if (np == 2) return launch<MODE_DENSE, true, 128, 2>(a, stream);
        return a.mode == MODE_CONV ? launch<MODE_CONV, true>(a, stream) : launch<MODE_DENSE, true>(a, stream);
    }
    if (a.in_c1) {      // LayerNorm(dim) fold, consumer side: the wide kernels or the 128x128 kernel (bf16 output, dense)
        if (a.mode != MODE_DENSE || a.out_kind != OUT_BF16 || !a.in_part || a.in_np <= 0 || a.in_F <= 0 || a.resid_bf16 || a.bias || a.splits > 1 || a.m_dev || (a.N % 4))
            return mm_set_error(MM_ERR_UNSUPPORTED, "gemm: the LayerNorm(dim) fold needs a plain dense bf16-output GEMM");
        if (!(a.debug & (8 | 4096 | 8192 | (1 << 30))) && mm_gemm_wide_eligible(a)) return mm_gemm_wide_launch(a, stream);
        return launch_dense_small(a, stream);
    }
    if (a.xb_out) {     // ... producer side: the fp32-residual epilogue of the 256x128 / 128x128 kernels
        if (a.mode != MODE_DENSE || a.out_kind != OUT_F32 || !a.resid_f32 || (a.N % 4) || !a.st_part || a.splits > 1 || a.m_dev || a.epi != EPI_NONE)
            return mm_set_error(MM_ERR_UNSUPPORTED, "gemm: the LayerNorm(dim) fold is produced by the dense fp32-residual epilogue");
        a.st_gran = a.N <= 512 ? 64 : 128;      // (model.hip ln_fold_np computes the same count for the consumers)
        a.st_np = (a.N + a.st_gran - 1) / a.st_gran;
        if (!(a.debug & 8) && mm_gemm_big_eligible(a)) return mm_gemm_big_launch(a, stream);
        return launch_dense_small(a, stream);
    }
    if (a.mode == MODE_CONV && mm_gemm_wide_conv_eligible(a)) return mm_gemm_wide_conv_launch(a, stream);      // round 6: convolutions on the persistent 256 x 256 x 64 tile
    if (a.head_w || a.par_w[1]) return mm_set_error(MM_ERR_UNSUPPORTED, "gemm: the fused head / the parity-batched form need the 256 x 256 convolution kernel's shape class");
    if (a.m_dev) a.debug |= 8;      // only the 128x128 kernel reads the device-side row count
    if (!(a.debug & (8 | 4096 | 8192 | (1 << 30))) && mm_gemm_wide_eligible(a)) return mm_gemm_wide_launch(a, stream);      // (bit 1 << 30: A/B against the older kernels)
    if (!(a.debug & (8 | 4096 | 8192)) && mm_gemm_cfg2_eligible(a)) return mm_gemm_cfg2_launch(a, stream);
    if (!(a.debug & (8 | 4096)) && mm_gemm_pers_eligible(a)) return mm_gemm_pers_launch(a, stream);
    if (!(a.debug & 8) && mm_gemm_big_eligible(a)) return mm_gemm_big_launch(a, stream);
    a.tiles_n = (a.N + BT - 1) / BT;
    const int tm = a.mode == MODE_CFG ? 64 : BT;
    a.tiles_m = (a.M + tm - 1) / tm;
    switch (a.mode) {
        case MODE_DENSE: return launch_dense_small(a, stream);
        case MODE_CFG: return launch<MODE_CFG>(a, stream);
        case MODE_CONV: return launch<MODE_CONV>(a, stream);
    }
    return mm_set_error(MM_ERR_SHAPE, "gemm: bad mode");
}

#ifdef MM_GEMM_TIMING
extern "C" int mm_debug_gemm_stamps(unsigned long long* host_dst, int n) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_gemm_stamps), sizeof(unsigned long long) * n);
}
#endif
