// bf16 GEMM on a 256-token x 256-weight-row tile with a 64-deep k-step ("more flops per barrier"), for FF w1 + GEGLU (+ the LayerNorm(inner) partial
// sums of the folded feed-forward) and for plain bf16-output projections whose tile count fills the CUs.
//
// Why another kernel: every k-loop of this family loses a roughly constant ~0.3-0.6 us per barrier interval whatever the tile and the look-ahead
// (256 x 256 x 32 with three stages and a hand-placed software pipeline: 1970 cycles per 1031-cycle step; the fp8 kernel's plain
// two-stage loop with 64 KiB per step: 0.33 us lost per 0.71 us of MFMAs).  So this kernel simply maximises the work between two barriers that fits
// the LDS: two stages of (256 + 256) rows x 128 bytes = 128 KiB, 64 MFMAs per wave and step, ONE barrier per step, the next step's LDS-DMA issued
// behind the first MFMA sub-step, fragments read and MFMAs issued in program order for the compiler to interleave -- the structure of gemm_fp8.hip with
// the bf16 instruction.  Persistent when the k-step count is even: the next tile's first step is requested during the last step of the current one into
// the stage that step does not read; the epilogue goes through the other stage so that every global store is 16 bytes of one output row per lane.
// Accumulation order per output element: k ascending in chunks of 32, one MFMA each -- the order of every GEMM kernel of the family: bit-identical
// results (and bit-identical LayerNorm partial sums: common.h ln_partial_row64 over the same bf16 values).
#include "common.h"
#include "muse_hip_internal.h"

#ifdef MM_GEMM_TIMING      // tools (gemm_harness `stamps`): s_memtime stamps of workgroup 0, waves 0 and 4 of gemm_wide_fused_kernel along their tiles
__device__ unsigned long long g_wide_stamps[2][2048];
#define WD_STAMP() { if (ts_on && ts_i < 2048) g_wide_stamps[ts_g][ts_i++] = __builtin_readcyclecounter(); }
#else
#define WD_STAMP()
#endif

namespace {

constexpr int TM = 256, TN = 256, BKB = 128;      // tile (TN: the 256-row weight tile; the dense kernel also has a 192-row form); bytes per k-step row (64 bf16)
constexpr int X_B = TM * BKB, W_B = TN * BKB, STG = X_B + W_B, SMEM = 2 * STG;
constexpr int WIDE_MAX_NP = 8;      // LayerNorm(dim) fold: up to 8 partials per row (two per 128 columns: dim <= 512) fit behind the stages (160 KiB of LDS)

__device__ __forceinline__ int sw128(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

#define MM_VMCNT_IMM(n_) (0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14))
__device__ __forceinline__ void wait_vmcnt_w(int n) {      // wave-uniform n; above 31: wait for 31 (waiting for more is always safe)
    switch (n) {
#define MM_W(n_) case n_: __builtin_amdgcn_s_waitcnt(MM_VMCNT_IMM(n_)); break;
        MM_W(0) MM_W(1) MM_W(2) MM_W(3) MM_W(4) MM_W(5) MM_W(6) MM_W(7) MM_W(8) MM_W(9) MM_W(10) MM_W(11) MM_W(12) MM_W(13) MM_W(14) MM_W(15)
        MM_W(16) MM_W(17) MM_W(18) MM_W(19) MM_W(20) MM_W(21) MM_W(22) MM_W(23) MM_W(24) MM_W(25) MM_W(26) MM_W(27) MM_W(28) MM_W(29) MM_W(30)
#undef MM_W
        default: if (n < 0) __builtin_amdgcn_s_waitcnt(MM_VMCNT_IMM(0)); else __builtin_amdgcn_s_waitcnt(MM_VMCNT_IMM(31)); break;
    }
}
template <bool GEGLU, int NFW>      // NFW: weight fragments per wave -- 4: 256-row weight tile, 3: 192-row tile (N a multiple of 192 but not of 256: q|k|v of the base config)
__global__ __launch_bounds__(512) void gemm_wide_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TN = 64 * NFW, W_B = TN * BKB, STG = X_B + W_B;
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int total = p.tiles_m * p.tiles_n, G = gridDim.x;
    const int KT = p.K / 64;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    // a DMA instruction covers 8 rows: lane l fetches row l >> 3, logical chunk (l & 7) ^ (row & 7) into physical chunk l & 7; this wave stages token rows
    // 32 wid .. + 31 and weight rows 8 NFW wid .. (4 + NFW instructions per step)
    const int dchunk = ((lane & 7) ^ (lane >> 3)) * 16;
    int voff_x[4], voff_w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        voff_x[i] = (32 * wid + 8 * i + (lane >> 3)) * p.ldx * 2 + dchunk;
        voff_w[i] = (8 * NFW * wid + 8 * (i < NFW ? i : 0) + (lane >> 3)) * p.ldw * 2 + dchunk;
    }
    int vb = blockIdx.x;
    if (vb >= total) return;
    __amdgpu_buffer_rsrc_t rx, rw;
    int tile_m, tile_n;
#define TILE_SETUP(vb_)                                                                                                                \
    {                                                                                                                                  \
        xcd_grouped_tile(vb_, p.tiles_m, p.tiles_n, p.group_m, tile_m, tile_n);                                                                \
        const int left_ = p.M - tile_m * TM;                                                                                           \
        rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X + (size_t)tile_m * TM * p.ldx), 0,                              \
                                               (unsigned)(left_ < TM ? left_ : TM) * (unsigned)p.ldx * 2u, 0x00020000);      /* rows beyond M read as zero */ \
        rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W + (size_t)tile_n * TN * p.ldw), 0, (unsigned)TN * (unsigned)p.ldw * 2u, 0x00020000); \
    }
#define ISSUE(kt_, st_)                                                                                                                \
    {                                                                                                                                  \
        unsigned char* xs_ = smem + (st_) * STG + wid * 4096;                                                                          \
        unsigned char* ws_ = smem + (st_) * STG + X_B + wid * (NFW * 1024);                                                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(xs_ + i * 1024), 16, voff_x[i], (kt_) * BKB, 0, 0);              \
        _Pragma("unroll") for (int i = 0; i < NFW; ++i)                                                                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(ws_ + i * 1024), 16, voff_w[i], (kt_) * BKB, 0, 0);              \
    }
    // Persistent when the step count is even (the launcher then starts one workgroup per CU): the NEXT tile's first k-step is requested during the last step of
    // the current one, into stage 0, which that step (odd index: stage 1) does not read; the epilogue stages its output through stage 1 and its global stores
    // are still retiring while the next tile's first steps run (the wait of that tile's step 0 counts them: VMEM retires in order).
    TILE_SETUP(vb);
    ISSUE(0, 0);
    int pending = 0;                 // VMEM stores this wave issued BEHIND the DMA of the coming tile's first step
    unsigned char* stg = smem + STG; // output staging: stage 1
    f32x4_t acc[NFW][8];             // [weight fragment a][token fragment b]: lane (fr, fg) holds weight rows 16 a + 4 fg .. + 3 for token 16 b + fr
    while (true) {
        const int m0 = tile_m * TM, n0 = tile_n * TN, cur_n = tile_n;
#pragma unroll
        for (int a = 0; a < NFW; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT; ++kt) {
            const int st = kt & 1;
            if (kt == 0) wait_vmcnt_w(pending); else __builtin_amdgcn_s_waitcnt(0x0F70);      // this step's DMA has landed (only younger stores may still be in flight)
            __builtin_amdgcn_s_barrier();            // ... for everybody, and everybody is done reading the other stage
            if (kt == 1 && p.in_c1 && wid < 4) {
                // LayerNorm(dim) fold: the partials of this tile's 256 rows landed with step 1's wait; ONE thread per row turns them into (rstd, -mean) in
                // the shadow of this step's MFMAs (read again only in the epilogue, many barriers from here)
                const int np_ = p.in_np;
                reinterpret_cast<float2*>(smem + 2 * STG + TM * np_ * 8 + 2048)[t] =
                    ln_rstd_negmean(reinterpret_cast<const float2*>(smem + 2 * STG) + t * np_, np_, 1.f / (float)p.in_F);
            }
            const unsigned char* xs = smem + st * STG + (wm * 128) * BKB;
            const unsigned char* ws = smem + st * STG + X_B + (wn * 16 * NFW) * BKB;
            // Software pipeline of the fragment reads (round 4): hipcc schedules "read the token fragment, wait for it, four MFMAs" strictly in that order, so
            // every 64 cycles of MFMA work waited for a full LDS round trip (ISA: ds_read_b128, s_waitcnt lgkmcnt(0), 4 x v_mfma, repeated).  Here the token
            // fragment of block b + 1 is requested BEFORE the MFMAs of block b, and the second sub-step's weight fragments before the last block of the first
            // (same MFMA order per accumulator: bit-identical results).
            u32x4_t wf[NFW], wf2[NFW], xr[3];
#pragma unroll
            for (int a = 0; a < NFW; ++a) wf[a] = *reinterpret_cast<const u32x4_t*>(ws + sw128(a * 16 + fr, fg));
            xr[0] = *reinterpret_cast<const u32x4_t*>(xs + sw128(fr, fg));
            xr[1] = *reinterpret_cast<const u32x4_t*>(xs + sw128(16 + fr, fg));
#pragma unroll
            for (int it = 0; it < 16; ++it) {        // 16 blocks of 16 tokens x 32 k: sub-step ks = it >> 3 (k ascending), token block b = it & 7
                const int b = it & 7;
                if (it + 2 < 16) xr[(it + 2) % 3] = *reinterpret_cast<const u32x4_t*>(xs + sw128(((it + 2) & 7) * 16 + fr, ((it + 2) >> 3) * 4 + fg));
                if (it == 6) {
#pragma unroll
                    for (int a = 0; a < NFW; ++a) wf2[a] = *reinterpret_cast<const u32x4_t*>(ws + sw128(a * 16 + fr, 4 + fg));
                }
                __builtin_amdgcn_sched_barrier(0);      // the prefetches stay in FRONT of this block's MFMAs (the scheduler otherwise sinks them behind)
#pragma unroll
                for (int a = 0; a < NFW; ++a) acc[a][b] = mfma16(it < 8 ? wf[a] : wf2[a], xr[it % 3], acc[a][b]);
                // The next step's DMA goes out BEHIND the first sub-step: right behind the barrier it delays the first MFMAs of both waves of a SIMD, behind
                // the second sub-step it lands too late (FF w1 of the base config, tools/gemm_harness: 62.0 / 58.6 / 67.7 us for the three placements)
                if (it == 7) {
                    if (kt + 1 < KT) {
                        ISSUE(kt + 1, st ^ 1);
                    } else if (vb + G < total) {     // (only with an even KT: see the launcher)
                        TILE_SETUP(vb + G);
                        ISSUE(0, 0);
                    }
                    if (kt == 0 && p.in_c1) {
                        // LayerNorm(dim) fold (GemmArgs::in_c1): this tile's row statistics partials (256 rows x np x 8 bytes) and its 64 NFW entries of c1 / c2
                        // arrive by LDS-DMA too, behind the stages (retired by the vmcnt(0) of step 1: K >= 128)
                        const int np_ = p.in_np;
                        const int left_ = p.M - m0;
                        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in_part + (size_t)m0 * np_ * 2), 0,
                                                                                             (unsigned)(left_ < TM ? left_ : TM) * (unsigned)np_ * 8u, 0x00020000);
                        for (int pc = wid; pc < 2 * np_; pc += 8)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(smem + 2 * STG + pc * 1024), 16, lane * 16, pc * 1024, 0, 0);
                        if (wid == 7 || (wid == 6 && p.in_c2)) {
                            const float* src_ = (wid == 7 ? p.in_c1 : p.in_c2) + n0;
                            const __amdgpu_buffer_rsrc_t rc_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src_), 0, (unsigned)TN * 4u, 0x00020000);
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rc_, (lds_ptr_t)(smem + 2 * STG + TM * np_ * 8 + (wid == 7 ? 0 : 1024)), 16, lane * 16, 0, 0, 0);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();                // everybody is done with stage 1: it becomes the output staging tile
        const bool full = m0 + TM <= p.M;            // ragged tiles skip stores: their count is not wave-uniform, the next wait then takes everything
        int nstore = 0;
        if (p.in_c1) {
            // LayerNorm(dim) fold, consumer side: X held the raw residual rows, W the gains -> out = rstd * (acc - mean * c1[n]) + c2[n] (before GEGLU);
            // one canonical evaluation with the 128 x 128 kernel (ln_rstd_negmean, ln_fold_apply)
            const int np_ = p.in_np;
            const float2* rsm = reinterpret_cast<const float2*>(smem + 2 * STG + TM * np_ * 8 + 2048);      // per row (rstd, -mean), written at k-step 1
            const float* lc1 = reinterpret_cast<const float*>(smem + 2 * STG + TM * np_ * 8);
            const bool has_c2 = p.in_c2 != nullptr;
            float4 c1v[NFW], c2v[NFW];
#pragma unroll
            for (int a = 0; a < NFW; ++a) {
                const int ci = wn * 16 * NFW + a * 16 + 4 * fg;
                c1v[a] = *reinterpret_cast<const float4*>(lc1 + ci);
                c2v[a] = has_c2 ? *reinterpret_cast<const float4*>(lc1 + 256 + ci) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const float2 stv = rsm[wm * 128 + b * 16 + fr];
                const float rs = stv.x, nm = stv.y;      // rstd, -mean
#pragma unroll
                for (int a = 0; a < NFW; ++a) {          // ln_fold_apply (common.h): two fused multiply-adds per value, one definition for both consumer kernels
                    acc[a][b][0] = ln_fold_apply(acc[a][b][0], rs, nm, c1v[a].x, c2v[a].x);
                    acc[a][b][1] = ln_fold_apply(acc[a][b][1], rs, nm, c1v[a].y, c2v[a].y);
                    acc[a][b][2] = ln_fold_apply(acc[a][b][2], rs, nm, c1v[a].z, c2v[a].z);
                    acc[a][b][3] = ln_fold_apply(acc[a][b][3], rs, nm, c1v[a].w, c2v[a].w);
                }
            }
        }
        if constexpr (GEGLU && NFW == 4) {
            // interleaved w1 packing: within a wave's 64 weight rows the first 32 are values, the next 32 their gates -> 128 output columns per tile
            constexpr int ROWB = 256;                // staging row bytes (128 bf16): 256 rows = 64 KiB = stage 1
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int row = wm * 128 + b * 16 + fr;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int col = wn * 32 + a * 16 + 4 * fg;
                    const mm_f32x2_t g01 = geglu_f2((mm_f32x2_t){acc[a][b][0], acc[a][b][1]}, (mm_f32x2_t){acc[a + 2][b][0], acc[a + 2][b][1]});      // (packed fp32: same values as geglu_f)
                    const mm_f32x2_t g23 = geglu_f2((mm_f32x2_t){acc[a][b][2], acc[a][b][3]}, (mm_f32x2_t){acc[a + 2][b][2], acc[a + 2][b][3]});
                    *reinterpret_cast<uint2*>(stg + row * ROWB + (((col >> 3) ^ (row & 7)) << 4) + (col & 4) * 2) =
                        make_uint2(pack_bf16x2(g01.x, g01.y), pack_bf16x2(g23.x, g23.y));
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
#pragma unroll 2
            for (int i = t; i < TM * 16; i += 512) {              // 16 chunks of 8 columns per row; 8 adjacent lanes = 64 columns = one LayerNorm part
                const int row = i >> 4, c = i & 15;
                const int m = m0 + row;
                const uint4 v = *reinterpret_cast<const uint4*>(stg + row * ROWB + ((c ^ (row & 7)) << 4));
                if (p.ln_part) {                                  // LayerNorm(inner) partial sums of this row's 64 columns (common.h): all 8 lanes take part
                    const float2 stv = ln_partial_row64(v);
                    if ((c & 7) == 0 && m < p.M) *reinterpret_cast<float2*>(p.ln_part + ((size_t)m * p.ln_np + cur_n * 2 + (c >> 3)) * 2) = stv;
                }
                if (m < p.M) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldc + cur_n * 128 + c * 8) = v;
            }
            nstore = (TM * 16 / 512) * (p.ln_part ? 2 : 1);
        } else {
            constexpr int ROWB = TN * 2, CPR = TN / 8;      // staging row bytes, 16-byte chunks per row; one token half (128 rows <= 64 KiB) at a time
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (wm == half) {
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const int row = b * 16 + fr;
#pragma unroll
                        for (int a = 0; a < NFW; ++a) {
                            const int col = wn * 16 * NFW + a * 16 + 4 * fg;
                            *reinterpret_cast<uint2*>(stg + row * ROWB + (((col >> 3) ^ (row & 7)) << 4) + (col & 4) * 2) =
                                make_uint2(pack_bf16x2(acc[a][b][0], acc[a][b][1]), pack_bf16x2(acc[a][b][2], acc[a][b][3]));
                        }
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_s_barrier();
#pragma unroll 2
                for (int i = t; i < 128 * CPR; i += 512) {
                    const int row = i / CPR, c = i - row * CPR;
                    const int m = m0 + half * 128 + row;
                    if (m < p.M) {
                        const uint4 v = *reinterpret_cast<const uint4*>(stg + row * ROWB + ((c ^ (row & 7)) << 4));
                        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldc + n0 + c * 8) = v;
                    }
                }
                if (half == 0) {
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_s_barrier();      // the first half is read out before the second one overwrites it
                }
            }
            nstore = 2 * (128 * CPR / 512);
        }
        pending = full ? nstore : 0;
        vb += G;
        if (vb >= total) break;
        // (no barrier here: a wave reaches the next tile's first barrier only behind its own staging reads, and the DMA into stage 1 is issued behind that barrier)
    }
#undef ISSUE
#undef TILE_SETUP
}

// ---- the logits GEMM of the decode loop on the same k-loop: to_logits of the guidance-mixed embeddings with the fused-sampling emission straight from the
// accumulators (the tile end of gemm_cfg2_kernel<WIDE_MIX2>, same canonical statistics and candidate format: common.h tile_softmax_stats / fs_slot_index).
// Persistent (one workgroup per CU walks its tiles): the first k-step of the NEXT tile is requested during the last step of the current one, into the stage
// that step does not read, so the emission (whose exchange arrays live in the other stage) covers its latency.  K % 128 == 0 (an even number of steps: the
// stage parity is the same for every tile).
__device__ __forceinline__ float max2_w(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max4_w(float a, float b, float c, float d) {      // (MFMA results: no canonicalising v_max x, x needed)
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(r), "v"(d));
    return r;
}

constexpr int SMEM_F = SMEM + 1024;      // + the tile's per-token bounds

// F16: the operands are fp16 terms ('f16x2' tier) -> the fp16 MFMA, accumulators scaled by p.alpha before the emission.  NP (F16 only; round 5): 0 = the segment
// packs as one contraction of depth K (rounds 4-5); 2 / 3 = term sharing as in gemm_terms.hip -- 32-deep steps on [xh | xl] token rows and [wh | wl] weight rows
// (NP 3; NP 2: [wh(64)] weight rows staged every other step), every product of a k-block from one staging of its term planes: 96 (64) MFMAs per wave and step on
// the 64 (48) KiB that carried 64 before.  Same stage layout (stage s = token tile s | weight tile s), so the emission is untouched.
template <bool F16, int NP = 0>
__global__ __launch_bounds__(512) void gemm_wide_fused_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int total = p.tiles_m * p.tiles_n, G = gridDim.x;
    const int KS = NP ? p.K / NP : p.K;      // term sharing: the contraction length proper (one segment of the packs)
    const int KT = NP ? KS / 32 : p.K / 64;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    unsigned char* xch = smem + STG;                                           // exchange arrays of the emission: stage 1 (free at a tile's end)
    float* lthr = reinterpret_cast<float*>(smem + SMEM);                       // this tile's per-token bounds (LDS-DMA at its first k-step)
    const __amdgpu_buffer_rsrc_t thr_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.fs_thr), 0, (unsigned)p.M * 4u, 0x00020000);
    const int lchunk = (lane & 7) ^ (lane >> 3);
    // term sharing: logical chunks 0-3 of a staged row are 32 k-values of the h plane, 4-7 the same 32 of the l plane (tokens: segment 1; weights, NP 3: segment 2)
    const int dchunk = NP ? (lchunk & 3) * 16 + (lchunk >> 2) * KS * 2 : lchunk * 16;
    const int dchunk_w = NP == 3 ? (lchunk & 3) * 16 + (lchunk >> 2) * KS * 4 : lchunk * 16;
    int voff_x[4], voff_w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        voff_x[i] = (32 * wid + 8 * i + (lane >> 3)) * p.ldx * 2 + dchunk;
        voff_w[i] = (32 * wid + 8 * i + (lane >> 3)) * p.ldw * 2 + dchunk_w;
    }
    int vb = blockIdx.x;
    if (vb >= total) return;
#ifdef MM_GEMM_TIMING
    const bool ts_on = blockIdx.x == 0 && (t == 0 || t == 256);
    const int ts_g = t >> 8;
    int ts_i = 0;
#endif
    __amdgpu_buffer_rsrc_t rx, rw;
    int tile_m, tile_n;
#define TILE_SETUP(vb_)                                                                                                                \
    {                                                                                                                                  \
        xcd_grouped_tile(vb_, p.tiles_m, p.tiles_n, 8, tile_m, tile_n);                                                                \
        const int left_ = p.M - tile_m * TM;                                                                                           \
        rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X + (size_t)tile_m * TM * p.ldx), 0,                              \
                                               (unsigned)(left_ < TM ? left_ : TM) * (unsigned)p.ldx * 2u, 0x00020000);                \
        rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W + (size_t)tile_n * TN * p.ldw), 0, (unsigned)TN * (unsigned)p.ldw * 2u, 0x00020000); \
    }
#define ISSUE(kt_, st_)                                                                                                                \
    {                                                                                                                                  \
        unsigned char* xs_ = smem + (st_) * STG + wid * 4096;                                                                          \
        unsigned char* ws_ = smem + (st_) * STG + X_B + wid * 4096;                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(xs_ + i * 1024), 16, voff_x[i], (kt_) * BKB, 0, 0);              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(ws_ + i * 1024), 16, voff_w[i], (kt_) * BKB, 0, 0);              \
    }
#define ISSUE_TX(kt_, st_)      /* term sharing: the token rows of 32-deep step kt_ */                                               \
    {                                                                                                                                  \
        unsigned char* xs_ = smem + (st_) * STG + wid * 4096;                                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(xs_ + i * 1024), 16, voff_x[i], (kt_) * 64, 0, 0);               \
    }
#define ISSUE_TW(byte_, st_)    /* ... the weight rows: byte_ = 64 kt (NP 3, per step) or 128 j (NP 2, per step pair j) */             \
    {                                                                                                                                  \
        unsigned char* ws_ = smem + (st_) * STG + X_B + wid * 4096;                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(ws_ + i * 1024), 16, voff_w[i], (byte_), 0, 0);                   \
    }
    TILE_SETUP(vb);
    if constexpr (NP != 0) {
        ISSUE_TX(0, 0);
        ISSUE_TW(0, 0);
    } else {
        ISSUE(0, 0);
    }
    int pending = 0;                 // VMEM stores this wave issued BEHIND the DMA of the coming tile's first step (the previous tile's emission)
    f32x4_t acc[4][8];
    while (true) {
        const int cur_m = tile_m, cur_n = tile_n;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        WD_STAMP();      // tile start
        if constexpr (NP != 0) {
        // term sharing (gemm_terms.hip's loop; requests right behind the barrier as in this kernel's plain loop).  NP 2: the weights of step pair j + 1 go out at
        // the even step 2 j BEHIND the tokens of step 2 j + 1, so the top of an odd step waits for everything but those 4 instructions.
        for (int kt = 0; kt < KT; ++kt) {
            const int st = kt & 1;
            if (kt == 0) wait_vmcnt_w(pending);
            else if (NP == 2 && st == 1 && kt + 1 < KT) __builtin_amdgcn_s_waitcnt(MM_VMCNT_IMM(4));
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < KT) {
                ISSUE_TX(kt + 1, st ^ 1);
                if constexpr (NP == 3) {
                    ISSUE_TW((kt + 1) * 64, st ^ 1);
                } else {
                    if (st == 0 && kt + 2 < KT) ISSUE_TW(((kt >> 1) + 1) * 128, ((kt >> 1) + 1) & 1);
                }
            } else if (vb + G < total) {      // last step (odd): the next tile's first step goes into stage 0 (NP 2: K / NP % 128 == 0, so the last pair sits in stage 1)
                TILE_SETUP(vb + G);
                ISSUE_TX(0, 0);
                ISSUE_TW(0, 0);
            }
            if (kt == 0 && wid == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(thr_rs, (lds_ptr_t)(lthr), 16, lane * 16, cur_m * TM * 4, 0, 0);
            const unsigned char* xs = smem + st * STG + (wm * 128) * BKB;
            const unsigned char* ws = smem + (NP == 3 ? st : ((kt >> 1) & 1)) * STG + X_B + (wn * 64) * BKB;
            const int wc = NP == 3 ? 0 : st * 4;
            u32x4_t wh[4], wl[4], xh[2], xl[2];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                wh[a] = *reinterpret_cast<const u32x4_t*>(ws + sw128(a * 16 + fr, wc + fg));
                if constexpr (NP == 3) wl[a] = *reinterpret_cast<const u32x4_t*>(ws + sw128(a * 16 + fr, 4 + fg));
            }
            xh[0] = *reinterpret_cast<const u32x4_t*>(xs + sw128(fr, fg));
            xl[0] = *reinterpret_cast<const u32x4_t*>(xs + sw128(fr, 4 + fg));
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b + 1 < 8) {
                    xh[(b + 1) & 1] = *reinterpret_cast<const u32x4_t*>(xs + sw128((b + 1) * 16 + fr, fg));
                    xl[(b + 1) & 1] = *reinterpret_cast<const u32x4_t*>(xs + sw128((b + 1) * 16 + fr, 4 + fg));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][b] = mfma16t<F16>(wh[a], xh[b & 1], acc[a][b]);
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][b] = mfma16t<F16>(wh[a], xl[b & 1], acc[a][b]);
                if constexpr (NP == 3) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[a][b] = mfma16t<F16>(wl[a], xh[b & 1], acc[a][b]);
                }
            }
        }
        } else {
        for (int kt = 0; kt < KT; ++kt) {
            const int st = kt & 1;
#ifdef MM_GEMM_TIMING
            if (kt > 0 && vb == (int)blockIdx.x + 3 * G) WD_STAMP();      // the fourth tile: one stamp per k-step
#endif
            if (kt == 0) wait_vmcnt_w(pending); else __builtin_amdgcn_s_waitcnt(0x0F70);      // this step's DMA has landed (in-order retirement: only younger stores may be in flight)
            __builtin_amdgcn_s_barrier();
            // (the DMA issue stays right behind the barrier; behind the first sub-step -- what the non-persistent kernel above did -- measured 2 %
            //  slower in this kernel, same box, 583-586 vs 594-597 us per launch at R = 5140)
            if (kt + 1 < KT) {
                ISSUE(kt + 1, st ^ 1);
            } else if (vb + G < total) {      // last step (stage 1, KT even): the next tile's first step goes into stage 0
                TILE_SETUP(vb + G);
                ISSUE(0, 0);
            }
            if (kt == 0 && wid == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(thr_rs, (lds_ptr_t)(lthr), 16, lane * 16, cur_m * TM * 4, 0, 0);
            const unsigned char* xs = smem + st * STG + (wm * 128) * BKB;
            const unsigned char* ws = smem + st * STG + X_B + (wn * 64) * BKB;
            // (the two-block-ahead prefetch of gemm_wide_kernel was measured here as well: 546 vs 568 us stand-alone, but 565 vs 535 us inside the decode loop --
            //  this kernel keeps hipcc's own order of the fragment reads)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4_t wf[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) wf[a] = *reinterpret_cast<const u32x4_t*>(ws + sw128(a * 16 + fr, ks * 4 + fg));
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const u32x4_t xf = *reinterpret_cast<const u32x4_t*>(xs + sw128(b * 16 + fr, ks * 4 + fg));
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[a][b] = mfma16t<F16>(wf[a], xf, acc[a][b]);
                }
            }
        }
        }
        if constexpr (F16) {      // undo the power-of-two scale of the packed weight terms (exact) before statistics and candidates
            const float al = p.alpha;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[a][b] *= al;
        }
        WD_STAMP();      // k-loop end
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();            // everybody is done with stage 1: it becomes the exchange area
        // ---- emission from the accumulators (gemm_cfg.hip, tile end of WIDE_MIX2; token of fragment block b: 128 wm + 16 b + fr)
        int le_ = lane;
        asm volatile("" : "+v"(le_));            // opaque copy: keeps the emission's address arithmetic out of the k-loop's live ranges
        const int FR_ = le_ & 15, FG_ = le_ >> 4;
        const int m0t = cur_m * TM;
        uint32_t* xmask = reinterpret_cast<uint32_t*>(xch);                          // [4 quarters][tokens]: the quarter's 32 keep bits of a token (bit 8 a + 2 f + h)
        float2* xml = reinterpret_cast<float2*>(xch + TM * 16);                      // [16 lane groups][tokens] (ml, pl)
        int nstore = 0;
        uint32_t mq[8];                          // the quarter's 32 keep bits of this lane's token of block b (bit 8 a + 2 f + h), after the exchange
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int tokl = wm * 128 + b * 16 + FR_;
            const bool valid = m0t + tokl < p.M;
            const float thr = lthr[tokl];
            // this lane IS lane group (wn, FG_) of the token: its 8 granules (a, h), no lane exchange anywhere
            float g2[4][2], gm[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                g2[a][0] = max2_w(acc[a][b][0], acc[a][b][1]);
                g2[a][1] = max2_w(acc[a][b][2], acc[a][b][3]);
                gm[a] = max2_w(g2[a][0], g2[a][1]);
            }
            const float ml = max4_w(gm[0], gm[1], gm[2], gm[3]);
            float gs[4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
                gs[a] = fs_exp_sum4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3], ml);
            const float pl = (gs[0] + gs[1]) + (gs[2] + gs[3]);
            // keep bits: this lane's 8 granules sit at bits 8 a + 2 f + h of the quarter's word; the token's four lane groups (lanes FR_, 16 + FR_, 32 + FR_,
            // 48 + FR_) OR their shares together with two lane exchanges
            uint32_t m32 = 0;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const uint32_t k2 = ((valid && g2[a][0] >= thr) ? 1u : 0u) | ((valid && g2[a][1] >= thr) ? 2u : 0u);
                m32 |= k2 << (8 * a);
            }
            m32 <<= 2 * FG_;
            m32 |= (uint32_t)__shfl_xor((int)m32, 16, 64);
            m32 |= (uint32_t)__shfl_xor((int)m32, 32, 64);
            mq[b] = m32;
            if (FG_ == 0) xmask[wn * TM + tokl] = m32;
            xml[(wn * 4 + FG_) * TM + tokl] = make_float2(ml, pl);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();      // all 16 lane groups of every token have published their masks and (ml, pl)
        WD_STAMP();      // statistics + masks published, exchange barrier passed
        // ---- the kept granules go out compacted in column order: a quarter starts behind the kept granules of the quarters in front of it
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint32_t mine = (mq[b] >> (2 * FG_)) & 0x03030303u;      // bit 8 a + h: this lane's granule (a, h) of block b
            if (__ballot(mine != 0u) == 0ull) continue;                   // wave-uniform
            const int tokl = wm * 128 + b * 16 + FR_;
            int base = 0;                                                 // kept granules of the quarters in front of this one
            if (wn > 0) base += __popc(xmask[tokl]);
            if (wn > 1) base += __popc(xmask[TM + tokl]);
            if (wn > 2) base += __popc(xmask[2 * TM + tokl]);
            float2* slot = reinterpret_cast<float2*>(p.fs_cand + ((size_t)(m0t + tokl) * p.tiles_n + cur_n) * FS_SLOT) + base;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                // one rank per granule PAIR (round 5): granule (a, 1) sits right behind (a, 0) when that one is kept
                const int pa = __popc(mq[b] & ((1u << (8 * a + 2 * FG_)) - 1u));
                const bool kp0 = (mine >> (8 * a)) & 1u, kp1 = (mine >> (8 * a + 1)) & 1u;
                if (__ballot(kp0) != 0ull) {                            // wave-uniform: the store below is ISSUED (exact VMEM count for the waits)
                    if (kp0) slot[pa] = make_float2(acc[a][b][0], acc[a][b][1]);
                    ++nstore;
                }
                if (__ballot(kp1) != 0ull) {
                    if (kp1) slot[pa + (kp0 ? 1 : 0)] = make_float2(acc[a][b][2], acc[a][b][3]);
                    ++nstore;
                }
            }
        }
        // ---- one record per (token, piece): this wave combines the tokens of its half's blocks wn and wn + 4 (common.h tile_combine16).  Round 5: ALL four lane
        // groups of a token take part -- lane group FG_ owns quarter FG_ of the 16 (ml, pl) pairs: its max, its w(q) = (t0 + t1) + (t2 + t3) with four exponentials,
        // then the maximum and the sum travel across the four lane groups by two lane exchanges each, in the canonical order (w0 + w1) + (w2 + w3) (a + b == b + a
        // exactly, so the pairing is what matters).  Rounds 3-4: lane group 0 alone, 16 exponentials in a row, three quarters of the lanes idle.
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const int tokl = wm * 128 + (hb * 4 + wn) * 16 + FR_;
            const int tok = m0t + tokl;
            const float2* gq = xml + (4 * FG_) * TM + tokl;      // this lane group's quarter: pairs 4 FG_ .. 4 FG_ + 3 of the token
            const float2 u0 = gq[0], u1 = gq[TM], u2 = gq[2 * TM], u3 = gq[3 * TM];
            float M_ = fmaxf(fmaxf(u0.x, u1.x), fmaxf(u2.x, u3.x));
            M_ = fmaxf(M_, __shfl_xor(M_, 16, 64));
            M_ = fmaxf(M_, __shfl_xor(M_, 32, 64));
            float wq = (u0.y * __expf(u0.x - M_) + u1.y * __expf(u1.x - M_)) + (u2.y * __expf(u2.x - M_) + u3.y * __expf(u3.x - M_));
            wq += __shfl_xor(wq, 16, 64);      // w(0) + w(1) | w(2) + w(3)
            wq += __shfl_xor(wq, 32, 64);      // (w(0) + w(1)) + (w(2) + w(3))
            const bool w_ = FG_ == 0 && tok < p.M;
            if (__ballot(w_) != 0ull) {
                if (w_) {
                    float4* rec = p.fs_stats + ((size_t)tok * p.tiles_n + cur_n) * FS_REC;
                    rec[0] = make_float4(M_, wq, 0.f, 0.f);
                    rec[1] = make_float4(__uint_as_float(xmask[tokl]), __uint_as_float(xmask[TM + tokl]), __uint_as_float(xmask[2 * TM + tokl]), __uint_as_float(xmask[3 * TM + tokl]));
                }
                nstore += 2;
            }
        }
        WD_STAMP();      // candidate stores + records issued
        pending = nstore;
        vb += G;
        if (vb >= total) break;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();      // the exchange area (stage 1) is read out before the next tile's second step lands in it
    }
#undef ISSUE_TX
#undef ISSUE_TW
#undef ISSUE
#undef TILE_SETUP
}

template <bool GEGLU, int NFW>
int launch_wide(GemmArgs a, hipStream_t stream) {
    constexpr int SM = 2 * (TM + 64 * NFW) * BKB + TM * WIDE_MAX_NP * 8 + 2048 + TM * 8;      // two stages + the LayerNorm-fold data (row statistics partials, c1, c2, per-row rstd / mean * rstd)
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<GEGLU, NFW>), hipFuncAttributeMaxDynamicSharedMemorySize, SM);
        if (e != hipSuccess) return mm_set_hip_error(e, "gemm_wide hipFuncSetAttribute");
        attr_set = true;
    }
    a.tiles_m = (a.M + TM - 1) / TM;
    a.tiles_n = a.N / (64 * NFW);
    a.group_m = 8;      // row-tiles per group of the XCD-grouped tile order (round 6 sweep 1 .. 12 on w1 / q|k|v: 59.0-60.4 / 34.2-34.7 us -- the order does not matter)
    const int total = a.tiles_m * a.tiles_n;
    // persistent (one workgroup per CU) when the k-step count is even -- the prefetch across tiles relies on the stage parity --, one tile per workgroup otherwise
    const int grid = ((a.K / 64) % 2 == 0 && !(g_mm_debug & (1 << 22))) ? (total < 256 ? total : 256) : total;
    hipLaunchKernelGGL((gemm_wide_kernel<GEGLU, NFW>), dim3(grid), dim3(512), SM, stream, a);
    return mm_check_launch("gemm_wide_kernel");
}

// weight-tile height (in 64-row units) whose tile count fills the last round of CUs to >= 90 %: 4 (256 rows) first, then 3 (192 rows, dense only); 0: none
int wide_nfw(const GemmArgs& a) {
    const long tm = (a.M + TM - 1) / TM;
    for (int nfw = 4; nfw >= 3; --nfw) {
        const int tn = 64 * nfw;
        if ((a.N % tn) || (nfw == 3 && a.epi != EPI_NONE)) continue;
        const long tiles = tm * (a.N / tn), rounds = (tiles + 255) / 256;
        if (tiles >= 256 && tiles * 10 >= rounds * 256 * 9) return nfw;
    }
    return 0;
}

}  // namespace

// dense bf16-output GEMMs without bias / activation / residual (plain or GEGLU with optional LayerNorm partial sums), K % 64 == 0, N % 256 == 0 (or % 192, plain
// only), 16-byte aligned rows, and a tile count that fills the last round of CUs to >= 90 % (a coarse tile loses the remainder)
bool mm_gemm_wide_eligible(const GemmArgs& a) {
    if (a.f16) return false;
    if (a.mode != MODE_DENSE || a.bias || a.act != ACT_NONE || a.resid_bf16 || a.resid_f32 || a.out_kind != OUT_BF16 || a.fs_stats || a.m_dev) return false;
    if (a.epi != EPI_NONE && a.epi != EPI_GEGLU) return false;
    if ((a.K % 64) || a.K < 128 || (a.ldx % 8) || (a.ldw % 8) || (a.ldc % 8) || (((uintptr_t)a.out) & 15)) return false;
    if (a.ln_part && a.epi != EPI_GEGLU) return false;
    if (a.in_c1 && (a.in_np < 1 || a.in_np > WIDE_MAX_NP || !a.in_part || (a.N % 4))) return false;
    return wide_nfw(a) != 0;
}

// the single-pass logits GEMM with the fused-sampling emission (x = the guidance-mixed embeddings): K % 128 == 0, N % 256 == 0
bool mm_gemm_wide_fused_eligible(const GemmArgs& a) {
    return a.mode == MODE_DENSE && a.wide_tok && a.fs_stats && a.fs_cand && a.fs_thr && !a.bias && a.act == ACT_NONE && !a.resid_f32 && !a.resid_bf16 &&
           a.epi == EPI_NONE && (a.K % 128) == 0 && a.K >= 128 && (a.N % TN) == 0 && (a.ldx % 8) == 0 && (a.ldw % 8) == 0 && a.M >= 1024;
}

int mm_gemm_wide_fused_launch(GemmArgs a, hipStream_t stream) {
#ifdef MM_TOOLS_PP      // tools/build_timing.sh only: the rejected out-of-lock-step forms (tools/experiments/gemm_pp.hip); never compiled into libmuse_hip.so
    if (mm_gemm_pp_fused_selected(a)) return mm_gemm_pp_fused_launch(a, stream);
#endif
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_F);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_F);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_fused_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_F);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_fused_kernel<true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_F);
        if (e != hipSuccess) return mm_set_hip_error(e, "gemm_wide_fused hipFuncSetAttribute");
        attr_set = true;
    }
    a.tiles_m = (a.M + TM - 1) / TM;
    a.tiles_n = a.N / TN;
    const int total = a.tiles_m * a.tiles_n;
    if (a.f16) {
        if (a.alpha == 0.f) a.alpha = 1.f;
        // term sharing (round 5) when the caller states the term count and the segment length fits the step structure (mm_debug_set2 bit 2: off, A/B)
        const int ks = (a.terms == 2 || a.terms == 3) && (a.K % a.terms) == 0 ? a.K / a.terms : 0;
        const bool share = ks && !(g_mm_debug2 & 2) && (ks % (a.terms == 3 ? 64 : 128)) == 0 && a.ldx >= a.K && a.ldw >= a.K;
        if (share && a.terms == 3) hipLaunchKernelGGL((gemm_wide_fused_kernel<true, 3>), dim3(total < 256 ? total : 256), dim3(512), SMEM_F, stream, a);
        else if (share) hipLaunchKernelGGL((gemm_wide_fused_kernel<true, 2>), dim3(total < 256 ? total : 256), dim3(512), SMEM_F, stream, a);
        else hipLaunchKernelGGL(gemm_wide_fused_kernel<true>, dim3(total < 256 ? total : 256), dim3(512), SMEM_F, stream, a);
    } else {
        hipLaunchKernelGGL(gemm_wide_fused_kernel<false>, dim3(total < 256 ? total : 256), dim3(512), SMEM_F, stream, a);
    }
    return mm_check_launch("gemm_wide_fused_kernel");
}

int mm_gemm_wide_launch(GemmArgs a, hipStream_t stream) {
    if (a.epi == EPI_GEGLU) return launch_wide<true, 4>(a, stream);
    return wide_nfw(a) == 4 ? launch_wide<false, 4>(a, stream) : launch_wide<false, 3>(a, stream);
}

#ifdef MM_GEMM_TIMING
extern "C" int mm_debug_wide_stamps(unsigned long long* host_dst, int n) {      // [2][2048]
    (void)n;
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_wide_stamps), sizeof(unsigned long long) * 2 * 2048);
}
#endif
