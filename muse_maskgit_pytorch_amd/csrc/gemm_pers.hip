// Persistent variant of the 256x128 bf16 MFMA GEMM (gemm_big.hip) for the GEMM classes whose epilogue is a pure store:
//   MODE_DENSE, bf16 output (q|k|v projection, FF w1 with the fused GEGLU epilogue)
//   MODE_CFG,   fp32 logits  (to_logits of both guidance passes + the combine; only when gemm_cfg.hip's 256-column tile does not apply)
// Measured on the non-persistent kernels (tools/gemm_bench.py ablation): the output stores are 25-40 % of the kernel and
// overlap with nothing -- every workgroup on the chip reaches its store phase at about the same time (HBM idle during the
// MFMA phase, MFMA idle during the store phase) -- and every tile pays a DMA-latency bubble at its start.  Storing straight
// from the MFMA fragment layout inside the next tile's loop (first attempt) was slower still: 16 rows x 64 B per
// instruction wastes the write path.  So here one workgroup per CU walks its tiles as ONE software pipeline:
//   * two 48 KiB DMA stages + a 64 KiB output tile `ct` in LDS (160 KiB = the whole CU);
//   * at the end of a tile the accumulators are combined (CFG / GEGLU) and transposed into ct (XOR-swizzled rows);
//   * ct is written out row-contiguously, 16 B per lane, ONE 8 KiB piece per k-iteration of the NEXT tile, i.e. inside its
//     MFMA stream (the piece is read from ct right after the barrier, with a raw ds_read, and stored after the MFMAs);
//   * a DMA cursor of its own runs two k-steps ahead of the MFMAs, across tile boundaries;
//   * the fragments are double-buffered over the two 32-wide halves of a stage: the ds_reads of the next half (after the
//     barrier: of the next stage) are in flight while the MFMAs of the current half run;
//   * counted waits as __builtin_amdgcn_s_waitcnt (the compiler's waitcnt pass must see them): behind the DMA we wait for
//     there is at most the one store of the previous iteration.
// Same tile shape and MFMA order as gemm.hip / gemm_big.hip -> bit-identical results.
#include "common.h"
#include "muse_hip_internal.h"

// Ablation hooks (skip stores / DMA / MFMA / the ct transpose via mm_debug_set bits 1 / 2 / 4 / 16, tools/gemm_bench.py) are compiled
// in only with -DMM_GEMM_ABLATE: as run-time branches they split the k-loop into a dozen basic blocks, which stops the
// scheduler from overlapping ds_reads, DMA issue and MFMAs.
#ifdef MM_GEMM_ABLATE
#define ABL(p_, bit_) ((p_).debug & (bit_))
#else
#define ABL(p_, bit_) 0
#endif

// (non-temporal output stores were tried here as in gemm_cfg.hip: no effect -- these outputs are 50 MB and re-read by the next kernel)
#define ST16(ptr_, v_) (*reinterpret_cast<uint4*>(ptr_) = (v_))

namespace {

constexpr int BMB = 256, BNB = 128, BK = 64;
constexpr int W_BYTES = BNB * BK * 2, X_BYTES = BMB * BK * 2, STAGE_B = W_BYTES + X_BYTES;
constexpr int CT_OFF = 2 * STAGE_B;                 // 96 KiB
constexpr int SMEM_B = CT_OFF + 65536;              // + 64 KiB output tile = 160 KiB

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// The DMA uses BUFFER loads (buffer_load_dwordx4 ... lds), see gemm_cfg.hip: a tile is two / three wave-uniform resource descriptors
// (first weight row, first activation row of the cond / null pass), the per-lane byte offsets never change, and rows beyond M are
// zero-filled by the descriptor's num_records -- the tile switch is a handful of scalar instructions.
struct TilePtrs {
    __amdgpu_buffer_rsrc_t w;
    __amdgpu_buffer_rsrc_t x[2];      // CFG: cond / null rows; dense: x[1] = x[0]
};

template <int MODE>
__device__ __forceinline__ void tile_setup(const GemmArgs& p, int vb, TilePtrs& tp) {
    int tile_m, tile_n;
    xcd_grouped_tile(vb, p.tiles_m, p.tiles_n, 8, tile_m, tile_n);
    constexpr int rows = (MODE == MODE_CFG) ? 128 : BMB;
    const int n0 = tile_n * BNB, m0 = tile_m * rows;
    const int left = p.M - m0;
    const unsigned xbytes = (unsigned)(left < rows ? left : rows) * (unsigned)p.ldx * 2u;
    tp.w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W + (size_t)n0 * p.ldw), 0, (unsigned)BNB * (unsigned)p.ldw * 2u, 0x00020000);
    tp.x[0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X + (size_t)m0 * p.ldx), 0, xbytes, 0x00020000);
    tp.x[1] = (MODE == MODE_CFG) ? __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X2 + (size_t)m0 * p.ldx), 0, xbytes, 0x00020000) : tp.x[0];
}

// accumulators -> ct (LDS).  Layouts (all rows XOR-swizzled at 16-byte chunk granularity):
//   CFG   : 128 rows x 128 fp32 (512 B rows, 32 chunks, swizzle row & 7)
//   dense : 256 rows x 128 bf16 (256 B rows, 16 chunks, swizzle row & 15)
//   GEGLU : 256 rows x  64 bf16 (128 B rows,  8 chunks, swizzle row & 7)
template <int MODE>
__device__ __forceinline__ void acc_to_ct(const GemmArgs& p, const f32x4_t (&acc)[4][4], unsigned char* ct, int wave_m, int wave_n,
                                          int fr, int fg) {
    if constexpr (MODE == MODE_CFG) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int row = wave_m * 32 + b * 16 + fr;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int c = (wave_n * 64 + a * 16 + fg * 4) >> 2;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float cv = acc[a][b][r], nl = acc[a][b + 2][r];
                    v[r] = nl + (cv - nl) * p.cfg_scale;      // muse_maskgit_pytorch.py:254
                }
                *reinterpret_cast<float4*>(ct + row * 512 + ((c ^ (row & 7)) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    } else if (p.epi == EPI_GEGLU) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int row = wave_m * 64 + b * 16 + fr;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int col = wave_n * 32 + a * 16 + fg * 4;
                *reinterpret_cast<uint2*>(ct + row * 128 + (((col >> 3) ^ (row & 7)) << 4) + (fg & 1) * 8) =
                    make_uint2(pack_bf16x2(geglu_f(acc[a][b][0], acc[a + 2][b][0]), geglu_f(acc[a][b][1], acc[a + 2][b][1])),
                               pack_bf16x2(geglu_f(acc[a][b][2], acc[a + 2][b][2]), geglu_f(acc[a][b][3], acc[a + 2][b][3])));
            }
        }
    } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int row = wave_m * 64 + b * 16 + fr;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int col = wave_n * 64 + a * 16 + fg * 4;
                *reinterpret_cast<uint2*>(ct + row * 256 + (((col >> 3) ^ (row & 15)) << 4) + (fg & 1) * 8) =
                    make_uint2(pack_bf16x2(acc[a][b][0], acc[a][b][1]), pack_bf16x2(acc[a][b][2], acc[a][b][3]));
            }
        }
    }
}

// one 8 KiB piece of ct, 16 B per lane, consecutive lanes on consecutive addresses of one output row: the LDS read ...
// (inline asm on purpose: a ds_read the compiler can see gets an s_waitcnt vmcnt(0) in front of it -- its waitcnt pass
//  cannot tell ct from the DMA stages and assumes the LDS-DMA in flight may write what is being read.  The caller waits
//  lgkmcnt(0) itself before using the value.)
__device__ __forceinline__ uint4 lds_read_b128_raw(const unsigned char* ptr) {
    typedef __attribute__((address_space(3))) const unsigned char* lds_cptr_t;
    const unsigned addr = (unsigned)(size_t)(lds_cptr_t)ptr;
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return make_uint4(v[0], v[1], v[2], v[3]);
}

template <int MODE>
__device__ __forceinline__ uint4 read_piece(const GemmArgs& p, const unsigned char* ct, int piece, int t) {
    if constexpr (MODE == MODE_CFG) {
        const int row = piece * 16 + (t >> 5), c = t & 31;
        return lds_read_b128_raw(ct + row * 512 + ((c ^ (row & 7)) << 4));
    } else if (p.epi == EPI_GEGLU) {
        const int row = piece * 64 + (t >> 3), c = t & 7;
        return lds_read_b128_raw(ct + row * 128 + ((c ^ (row & 7)) << 4));
    } else {
        const int row = piece * 32 + (t >> 4), c = t & 15;
        return lds_read_b128_raw(ct + row * 256 + ((c ^ (row & 15)) << 4));
    }
}

// ... and the global store of what read_piece returned
template <int MODE>
__device__ __forceinline__ void write_piece(const GemmArgs& p, const uint4 v, int piece, int t, int m0, int n0, int tile_n) {
    if constexpr (MODE == MODE_CFG) {
        const int row = piece * 16 + (t >> 5), c = t & 31;
        const int m = m0 + row;
        if (m < p.M) *reinterpret_cast<uint4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n0 + c * 4) = v;
    } else if (p.epi == EPI_GEGLU) {
        const int row = piece * 64 + (t >> 3), c = t & 7;
        const int m = m0 + row;
        if (p.ln_part) {      // LayerNorm(inner) partial sums of this row's 64 columns (common.h), inside the next tile's k-loop
            const float2 st = ln_partial_row64(v);
            if (c == 0 && m < p.M) *reinterpret_cast<float2*>(p.ln_part + ((size_t)m * p.ln_np + tile_n) * 2) = st;
        }
        if (m < p.M) ST16(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldc + tile_n * 64 + c * 8, v);
    } else {
        const int row = piece * 32 + (t >> 4), c = t & 15;
        const int m = m0 + row;
        if (m < p.M) ST16(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldc + n0 + c * 8, v);
    }
}

template <int MODE>
__device__ __forceinline__ void store_piece(const GemmArgs& p, const unsigned char* ct, int piece, int t, int m0, int n0, int tile_n) {
    if constexpr (MODE == MODE_CFG) {
        const int row = piece * 16 + (t >> 5), c = t & 31;
        const uint4 v = *reinterpret_cast<const uint4*>(ct + row * 512 + ((c ^ (row & 7)) << 4));
        const int m = m0 + row;
        if (m < p.M) *reinterpret_cast<uint4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n0 + c * 4) = v;
    } else if (p.epi == EPI_GEGLU) {
        const int row = piece * 64 + (t >> 3), c = t & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(ct + row * 128 + ((c ^ (row & 7)) << 4));
        const int m = m0 + row;
        if (p.ln_part) {
            const float2 st = ln_partial_row64(v);
            if (c == 0 && m < p.M) *reinterpret_cast<float2*>(p.ln_part + ((size_t)m * p.ln_np + tile_n) * 2) = st;
        }
        if (m < p.M) ST16(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldc + tile_n * 64 + c * 8, v);
    } else {
        const int row = piece * 32 + (t >> 4), c = t & 15;
        const uint4 v = *reinterpret_cast<const uint4*>(ct + row * 256 + ((c ^ (row & 15)) << 4));
        const int m = m0 + row;
        if (m < p.M) ST16(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldc + n0 + c * 8, v);
    }
}

template <int MODE>
__global__ __launch_bounds__(512) void gemm_pers_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ct = smem + CT_OFF;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wave_m = wid >> 1, wave_n = wid & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int total = p.tiles_m * p.tiles_n;
    const int G = gridDim.x;
    const int KT = p.K / BK;
    const int tile_rows = (MODE == MODE_CFG) ? 128 : BMB;
    const int npieces = (MODE == MODE_DENSE && p.epi == EPI_GEGLU) ? 4 : 8;

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    // per-lane byte offsets inside a tile (fixed for the whole kernel): a DMA instruction covers 8 rows, lane l fetches row l >> 3,
    // logical chunk (l & 7) ^ (row & 7) into physical chunk l & 7; this wave stages weight rows 16*wid + 8*i and activation rows
    // 32*wid + 8*i (CFG: row r = wave row r >> 6, pass (r >> 5) & 1 -- wave-uniform --, token (r >> 6)*32 + (r & 31))
    int voff_w[2], voff_x[4];
    {
        const int chunk = (lane & 7) ^ (lane >> 3);
#pragma unroll
        for (int i = 0; i < 2; ++i) voff_w[i] = (16 * wid + 8 * i + (lane >> 3)) * p.ldw * 2 + chunk * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 32 * wid + 8 * i + (lane >> 3);
            const int xrow = (MODE == MODE_CFG) ? (r >> 6) * 32 + (r & 31) : r;
            voff_x[i] = xrow * p.ldx * 2 + chunk * 16;
        }
    }
    const bool x_null = (MODE == MODE_CFG) && (wid & 1);
#define ISSUE_TILE(tp_, kt_, st_)                                                                             \
    {                                                                                                         \
        const int k0_ = (kt_) * (BK * 2);                                                                     \
        unsigned char* ws_ = smem + (st_) * STAGE_B + wid * 2048;                                             \
        unsigned char* xs_ = smem + (st_) * STAGE_B + W_BYTES + wid * 4096;                                   \
        const __amdgpu_buffer_rsrc_t rx_ = x_null ? (tp_).x[1] : (tp_).x[0];                                  \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                         \
            __builtin_amdgcn_raw_ptr_buffer_load_lds((tp_).w, (lds_ptr_t)(ws_ + i * 1024), 16, voff_w[i], k0_, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, (lds_ptr_t)(xs_ + i * 1024), 16, voff_x[i], k0_, 0, 0);     \
    }

    int vb = blockIdx.x;
    if (vb >= total) return;
    const int steps_total = ((total - 1 - vb) / G + 1) * KT;      // k-steps this workgroup runs over all its tiles

    // ---- load cursor: the DMA runs ahead of the MFMAs across tile boundaries (step s lands in stage s & 1)
    TilePtrs lc;
    int l_vb = vb, l_k = 0;
    bool l_live = true;
    tile_setup<MODE>(p, l_vb, lc);
#define LOAD_NEXT(st_)                                                                                        \
    if (l_live) {                                                                                             \
        if (!ABL(p, 2)) ISSUE_TILE(lc, l_k, st_);                                                         \
        if (++l_k == KT) {                                                                                    \
            l_k = 0;                                                                                          \
            l_vb += G;                                                                                        \
            l_live = l_vb < total;                                                                            \
            if (l_live) tile_setup<MODE>(p, l_vb, lc);                                                        \
        }                                                                                                     \
    }
    LOAD_NEXT(0);
    LOAD_NEXT(1);

    // fragment registers, double-buffered over the two 32-wide k-halves of a stage: while the MFMAs of one half run, the
    // ds_reads of the next half (after the barrier: of the NEXT stage) are in flight -- the LDS pipe and the matrix pipe
    // overlap instead of alternating (the single-buffered loop idled the MFMAs ~45 % of the time even without any DMA)
    u32x4_t a0[4], b0[4], a1[4], b1[4];
#define READ_FRAGS(af_, bf_, st_, ks_)                                                                        \
    {                                                                                                         \
        const unsigned char* ws_ = smem + (st_) * STAGE_B;                                                    \
        const unsigned char* xs_ = ws_ + W_BYTES;                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
            af_[i] = *reinterpret_cast<const u32x4_t*>(ws_ + lds_off(wave_n * 64 + i * 16 + fr, (ks_) * 4 + fg)); \
            bf_[i] = *reinterpret_cast<const u32x4_t*>(xs_ + lds_off(wave_m * 64 + i * 16 + fr, (ks_) * 4 + fg)); \
        }                                                                                                     \
    }
#define MFMA_HALF(af_, bf_)                                                                                   \
    if (!ABL(p, 4)) {                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                         \
            _Pragma("unroll") for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(af_[a], bf_[b], acc[a][b]);      \
        __builtin_amdgcn_s_setprio(0);                                                                        \
    } else { _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(af_[i]), "v"(bf_[i])); }
    // first half-step of a tile: C = 0 as an inline constant instead of 64 accumulator clears per tile
#define MFMA_HALF0(af_, bf_)                                                                                  \
    {                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                         \
            _Pragma("unroll") for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(af_[a], bf_[b], f32x4_t{0.f, 0.f, 0.f, 0.f}); \
        __builtin_amdgcn_s_setprio(0);                                                                        \
    }

    // step 0 has landed once only step 1's six DMA instructions are still in flight
    if (steps_total > 1 && !ABL(p, 2)) __builtin_amdgcn_s_waitcnt(0x0F76);      // vmcnt(6)
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    READ_FRAGS(a0, b0, 0, 0);

    f32x4_t acc[4][4];
    int prv_m0 = 0, prv_n0 = 0, prv_tile_n = 0;
    bool have_prev = false;
    int g = 0;                  // global k-step counter: the stage of step g is g & 1
    int st_prev = 0;            // VMEM stores this wave issued in the previous step (they sit behind the DMA we wait for)

    // Output pieces of the PREVIOUS tile ride on this tile's k-steps: piece q is read from ct right after the barrier of step q
    // (q <= KT - 2), so every read of the old ct is complete before anybody passes the barrier of step KT - 1 -- and the tile end can
    // overwrite ct without a barrier of its own; the new ct is first read after the barrier of the next tile's step 0, which every
    // wave reaches with its own ct writes retired (lgkmcnt(0)).  When a tile has exactly as many k-steps as pieces, step KT - 2
    // reads two pieces and the second one is stored in step KT - 1.
    const bool two_in_one = KT == npieces;
    const bool stat_store = (MODE == MODE_DENSE) && p.epi == EPI_GEGLU && p.ln_part != nullptr;
    uint4 pv2 = make_uint4(0, 0, 0, 0);
#define STEP_BODY(FIRST_)                                                                                     \
    {                                                                                                         \
        const int st = g & 1;                                                                                 \
        /* a0/b0 were requested 16 MFMAs ago: retiring them HERE (a wait the compiler's scoreboard sees) keeps it from placing */ \
        /* an lgkmcnt(0) behind the a1/b1 reads below, which would serialise those reads with the first MFMA half */ \
        __builtin_amdgcn_s_waitcnt(0xC07F);                     /* lgkmcnt(0) */                               \
        READ_FRAGS(a1, b1, st, 1);                                                                            \
        if (FIRST_ && !ABL(p, 4)) MFMA_HALF0(a0, b0) else MFMA_HALF(a0, b0)                                   \
        /* step g+1 (issued after the previous barrier) has landed once at most the store of the previous step is in flight. */ \
        /* A ragged tile may skip a whole store instruction (all lanes predicated off): then st_prev = 0 -- under-counting */ \
        /* only makes the wait conservative, over-counting would let the DMA we need slip. */                 \
        /* (waits as builtins, not inline asm: the compiler's own waitcnt pass must SEE them, or it assumes the LDS-DMA may still */ \
        /*  be in flight and drains vmcnt(0) in front of later ds_reads) */                                   \
        if (st_prev == 2) __builtin_amdgcn_s_waitcnt(0x0F72); else if (st_prev) __builtin_amdgcn_s_waitcnt(0x0F71); else __builtin_amdgcn_s_waitcnt(0x0F70);      /* vmcnt(2 / 1 / 0) */ \
        __builtin_amdgcn_s_waitcnt(0xC07F);                     /* lgkmcnt(0): my reads of stage st (and my ct writes) are complete */ \
        __builtin_amdgcn_s_barrier();                           /* everybody's are: stage st is free, step g+1 is visible */ \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        /* read the piece(s) BEFORE the DMA issue (nothing but the previous store is in flight here), store after the MFMAs */ \
        const bool piece = have_prev && kt < npieces && kt <= KT - 2 && !ABL(p, 1);                           \
        const bool piece_late = have_prev && two_in_one && kt == KT - 1 && !ABL(p, 1);                        \
        uint4 pv = make_uint4(0, 0, 0, 0);                                                                    \
        if (piece) pv = read_piece<MODE>(p, ct, kt, t);                                                       \
        if (piece && two_in_one && kt == KT - 2) pv2 = read_piece<MODE>(p, ct, KT - 1, t);                    \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        LOAD_NEXT(st);                                          /* step g+2 */                                \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (g + 1 < steps_total) READ_FRAGS(a0, b0, st ^ 1, 0);                                               \
        MFMA_HALF(a1, b1);                                                                                    \
        st_prev = 0;                                                                                          \
        if (piece || piece_late) {                                                                            \
            __builtin_amdgcn_s_waitcnt(0xC07F);                 /* the raw ds_read of pv (and, long done, a0/b0) */ \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            write_piece<MODE>(p, piece ? pv : pv2, kt, t, prv_m0, prv_n0, prv_tile_n);                        \
            st_prev = (prv_m0 + tile_rows <= p.M) ? (stat_store ? 2 : 1) : 0;      /* (a folded FF's statistics store is a second one) */ \
        }                                                                                                     \
        ++g;                                                                                                  \
    }

    while (true) {
        {
            const int kt = 0;
            STEP_BODY(true)
        }
        for (int kt = 1; kt < KT; ++kt) STEP_BODY(false)
        // tile boundary: nobody reads the previous ct any more (see above) -> overwrite it with this tile's output
        int tile_m, tile_n;
        xcd_grouped_tile(vb, p.tiles_m, p.tiles_n, 8, tile_m, tile_n);
        if (!ABL(p, 16)) acc_to_ct<MODE>(p, acc, ct, wave_m, wave_n, fr, fg);
        prv_m0 = tile_m * tile_rows; prv_n0 = tile_n * BNB; prv_tile_n = tile_n; have_prev = true;
        vb += G;
        if (vb >= total) break;
    }
    // drain the last tile
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    if (!ABL(p, 1))
        for (int q = 0; q < npieces; ++q) store_piece<MODE>(p, ct, q, t, prv_m0, prv_n0, prv_tile_n);
}

template <int MODE>
int launch_pers(const GemmArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pers_kernel<MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
        if (e != hipSuccess) return mm_set_hip_error(e, "gemm_pers hipFuncSetAttribute");
        attr_set = true;
    }
    const int total = a.tiles_m * a.tiles_n;
    const int grid = total < 256 ? total : 256;         // one workgroup per CU (it owns all 160 KiB of LDS)
    hipLaunchKernelGGL(gemm_pers_kernel<MODE>, dim3(grid), dim3(512), SMEM_B, stream, a);
    return mm_check_launch("gemm_pers_kernel");
}

}  // namespace

// pure-store epilogues only (no bias / activation / residual), K >= 512 so the 8 store pieces of a tile fit in the next tile's
// k-loop, N a multiple of the tile width (M may be ragged), 16-byte aligned output rows
bool mm_gemm_pers_eligible(const GemmArgs& a) {
    if (a.mode == MODE_CONV || a.bias || a.act != ACT_NONE || a.resid_bf16 || a.resid_f32) return false;
    if (a.mode == MODE_DENSE && a.out_kind != OUT_BF16) return false;
    if (a.mode == MODE_CFG && a.out_kind != OUT_F32) return false;
    const int tok = a.mode == MODE_CFG ? 128 : BMB;
    if (a.K < 8 * BK || (a.N % BNB) != 0) return false;
    if ((a.out_kind == OUT_BF16 && (a.ldc % 8)) || (a.out_kind == OUT_F32 && (a.ldc % 4)) || (((uintptr_t)a.out) & 15)) return false;
    const long tiles = (long)((a.M + tok - 1) / tok) * (a.N / BNB);
    return tiles >= 256;
}

int mm_gemm_pers_launch(GemmArgs a, hipStream_t stream) {
    a.tiles_n = a.N / BNB;
    const int tok = a.mode == MODE_CFG ? 128 : BMB;
    a.tiles_m = (a.M + tok - 1) / tok;
    if (a.mode == MODE_CFG) return launch_pers<MODE_CFG>(a, stream);
    return launch_pers<MODE_DENSE>(a, stream);
}
