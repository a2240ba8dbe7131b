// to_logits + classifier-free guidance, second generation (muse_maskgit_pytorch.py:250-254, 332): persistent MFMA GEMM with a
// 128-token x 256-vocabulary tile.
//
// Why another kernel: the 256x128 kernels of this family all plateau where their LDS-DMA operand stream reaches ~45 GB/s per
// CU (tools/gemm_bench.py ablations: removing the MFMAs changes nothing, removing the DMA gives +40..50 %; 48 KiB per
// 4.2 MFLOP k-step / 45 GB/s = 1.03 PFLOP/s chip-wide, which is what they measure).  Doubling the vocabulary width of the tile
// cuts the operand bytes per flop by a third (32 KiB per 4.2 MFLOP) and the ds_read traffic per MFMA by a quarter.
//
// Shape of the pipeline (one 512-thread workgroup per CU, 8 waves = 2 token halves x 4 vocabulary quarters, each wave
// 64 tokens x {cond, null} x 64 columns = 4 x 8 accumulator fragments):
//   * K is walked in steps of 32 (one v_mfma_f32_16x16x32_bf16 per fragment pair); three 32 KiB LDS stages, the LDS-DMA runs
//     TWO k-steps ahead of the MFMAs across tile boundaries (its own tile cursor), one barrier per k-step;
//   * LDS rows are 64 B; a 16-row block is stored row-major with the 16-byte chunk index XORed by (-(row >> 2)) & 3, which
//     makes every ds_read_b128 lane group hit 16 distinct slots (MI355X_MICROARCH.md, LDS table) while 4 consecutive DMA
//     lanes still fetch one contiguous 64-byte row segment; the swizzle is applied to the DMA's SOURCE address;
//   * the fp32 output tile (128 KiB) does not fit beside the stages, so it leaves in two halves through a 64 KiB staging
//     tile `ct`: at the end of a tile every wave combines cond / null, writes the first 32 of its 64 tokens to ct and keeps the
//     other 32 (32 VGPRs) until the middle of the NEXT tile; ct drains row-contiguously (1 KiB per wave instruction), one
//     8 KiB piece per k-step, inside the next tile's MFMA stream;
//   * a piece is read from ct right after a step's barrier (raw ds_read: see lds_read_b128_raw) and stored at the end of the
//     step, after the DMA issue; a piece is one token row per wave, so whether a wave issues the store is wave-uniform and the
//     counted s_waitcnt vmcnt(N) at the top of a step knows exactly what may still be in flight behind the stage it is about
//     to read (the stores of the last two steps + the DMA one step ahead; VMEM retires in order and vmcnt counts stores).
// All waits are builtins, not inline asm, so that the compiler's own waitcnt pass sees them -- otherwise it assumes the LDS-DMA
// may still be writing and drains vmcnt(0) in front of ds_reads.
// The accumulation order per output element (k ascending in chunks of 32, one MFMA each) equals gemm.hip / gemm_big.hip /
// gemm_pers.hip: results are bit-identical to those kernels.
// Second instantiation (WIDE_GEGLU): FF w1 with the GEGLU epilogue on a 256-row x 256-weight-row tile for long K (see
// mm_gemm_cfg2_eligible); same pipeline, the bf16 output tile (64 KiB) goes through ct in one piece schedule.
// Third instantiation (WIDE_MIX, round 3): the guidance logits as ONE pass.  null + (cond - null) * s is linear in the embedding
// (mmp.py:254 with logits = embed @ W^T, :332), so the decode loop mixes the two passes' final embeddings first (mm_cfg_mix) and multiplies
// once: 128 mixed token rows x 256 columns per tile, half the MFMAs and 3/4 of the operand stream of the two-pass tile for the same
// 128 x 256 logits, the SAME output path (pieces through ct, logits store or fused-sampling emission) -- the formats downstream do not change.
#include "common.h"
#include "muse_hip_internal.h"

#ifdef MM_GEMM_ABLATE
#define ABL(p_, bit_) ((p_).debug & (bit_))
#else
#define ABL(p_, bit_) 0
#endif

#ifdef MM_GEMM_TIMING      // tools/cfg2_timing.py only: s_memtime stamps of workgroup 0 / wave 0 along its tiles
__device__ unsigned long long g_cfg2_stamps[1024];
#define TSTAMP() if (ts_on && ts_i < 1024) { g_cfg2_stamps[ts_i++] = __builtin_readcyclecounter(); }
#else
#define TSTAMP()
#endif

namespace {

constexpr int TOK = 128, BN = 256, BK = 32, NST = 3;
constexpr int XT_B = 2 * TOK * BK * 2;      // 16 KiB: 256 activation rows (128 tokens x {cond, null}) x 64 B
constexpr int WT_B = BN * BK * 2;           // 16 KiB
constexpr int STG_B = XT_B + WT_B;          // 32 KiB
constexpr int CT_OFF = NST * STG_B;         // 96 KiB
constexpr int SMEM_B = CT_OFF + 65536;      // 160 KiB (every instantiation is launched with all of it: Geo<>)

#define WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

// vmcnt is a 6-bit field split over the s_waitcnt immediate: bits 3:0 and 15:14 (lgkmcnt 0xF and expcnt 7 = no wait on those)
#define MM_VMCNT_IMM(n_) (0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14))
__device__ __forceinline__ void wait_vmcnt(int n) {      // n is wave-uniform; values above 31 wait for 31 (waiting for more is always safe)
    switch (n) {
#define MM_W(n_) case n_: __builtin_amdgcn_s_waitcnt(MM_VMCNT_IMM(n_)); break;
        MM_W(0) MM_W(1) MM_W(2) MM_W(3) MM_W(4) MM_W(5) MM_W(6) MM_W(7) MM_W(8) MM_W(9) MM_W(10) MM_W(11) MM_W(12) MM_W(13) MM_W(14) MM_W(15)
        MM_W(16) MM_W(17) MM_W(18) MM_W(19) MM_W(20) MM_W(21) MM_W(22) MM_W(23) MM_W(24) MM_W(25) MM_W(26) MM_W(27) MM_W(28) MM_W(29) MM_W(30)
#undef MM_W
        default: if (n < 0) __builtin_amdgcn_s_waitcnt(MM_VMCNT_IMM(0)); else __builtin_amdgcn_s_waitcnt(MM_VMCNT_IMM(31)); break;
    }
}

// raw LDS read the compiler's waitcnt pass cannot see (it would put an s_waitcnt vmcnt(0) in front: it cannot tell ct from the
// DMA stages); the caller waits lgkmcnt(0) before using the value
__device__ __forceinline__ uint4 lds_read_b128_raw(const unsigned char* ptr) {
    typedef __attribute__((address_space(3))) const unsigned char* lds_cptr_t;
    const unsigned addr = (unsigned)(size_t)(lds_cptr_t)ptr;
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return make_uint4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ float max2_asm(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// max of 4 as two instructions (fmaxf would add a canonicalising v_max x, x per operand: these are MFMA results, never signalling NaNs)
__device__ __forceinline__ float max4_asm(float a, float b, float c, float d) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(r), "v"(d));
    return r;
}

// the logits are written once and read once by the sampler, 1.35 GB per launch against 4 MiB of L2 per XCD: a non-temporal store
// keeps them from evicting the weight / activation tiles the other CUs of the XCD are about to re-read
__device__ __forceinline__ void store_stream(float* ptr, const uint4 v) {
    __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_t*>(ptr));
}

// The DMA uses BUFFER loads (buffer_load_dwordx4 ... lds): a tile is addressed by wave-uniform resource descriptors (its first
// activation / weight row) plus ONE 32-bit per-lane offset that never changes (row-in-tile * pitch + swizzled chunk), and the k
// position is a scalar offset.  Against global_load_lds with per-lane 64-bit addresses this halves the address data each DMA
// instruction moves, removes the per-tile 64-bit address arithmetic, and lets the descriptor's num_records zero-fill the rows
// beyond M instead of clamping them.
struct LoadCur {
    __amdgpu_buffer_rsrc_t x[2];      // CFG: cond rows / null rows of the tile; GEGLU: both the tile's activation rows
    __amdgpu_buffer_rsrc_t w;
};

constexpr int WIDE_CFG = 0, WIDE_GEGLU = 1, WIDE_MIX = 2, WIDE_MIXF = 3, WIDE_MIX2 = 4;
// WIDE_MIX2 = the fused-only single pass on a 256-token x 256-column tile: the two-pass tile's geometry (each wave 2 x 64 tokens x 64 columns, 128
// accumulator VGPRs) with the "null" half of the activation stage holding the NEXT 128 tokens instead of the second pass.  Measured with in-kernel
// stamps (tools/mixf_timing.py): the k-loop of this family streams its operands at ~16 B per clock and CU through the LDS-DMA path whatever the
// look-ahead, so the lever is flops per operand byte -- 128 flop/B here against 85 for 128 x 256 -- and the accumulator emission needs no output tile.
// WIDE_MIXF = WIDE_MIX with the fused-sampling emission ONLY (mm_generate's default path).  That epilogue needs no staging tile, and the k-loop of
// this family is bound by the LATENCY of the LDS-DMA, not its rate: a k-step costs ~0.9-1.0 us whether it carries 32 or 16 MFMAs per wave,
// i.e. DMA latency / look-ahead (two steps with three stages).  The 64 KiB of the staging tile become stages: FIVE stages of 24 KiB (8 KiB of
// activation rows + 16 KiB of weight rows), the DMA four k-steps ahead of the MFMAs.
template <int WMODE> struct Geo {
    static constexpr bool mix = WMODE == WIDE_MIX || WMODE == WIDE_MIXF;      // 128-token single pass: 4 activation blocks per wave
    static constexpr int nst = WMODE == WIDE_MIXF ? 5 : 3;                    // LDS stages
    static constexpr int xtb = mix ? 8192 : 16384;                            // activation bytes of a stage
    static constexpr int stg = (WMODE == WIDE_MIXF) ? 24576 : 32768;          // stage stride
    static constexpr int woff = (WMODE == WIDE_MIXF) ? 8192 : 16384;          // weight rows inside a stage
    static constexpr int ct_off = nst * stg;                                  // staging tile / exchange area behind the stages
};

template <int WMODE>
__device__ __forceinline__ void cfg_tile_setup(const GemmArgs& p, int vb, LoadCur& lc) {
    int tile_m, tile_n;
    xcd_grouped_tile(vb, p.tiles_m, p.tiles_n, 8, tile_m, tile_n);
    constexpr int TROWS = (WMODE == WIDE_GEGLU || WMODE == WIDE_MIX2) ? 2 * TOK : TOK;      // activation rows of a tile
    const int m0 = tile_m * TROWS, n0 = tile_n * BN;
    const int rows_left = p.M - m0;                                                     // > 0
    if constexpr (WMODE == WIDE_MIX2) {      // descriptor 0: tokens [m0, m0 + 128), descriptor 1: tokens [m0 + 128, m0 + 256)
        const int r0 = rows_left < TOK ? rows_left : TOK, r1 = rows_left - TOK < 0 ? 0 : (rows_left - TOK < TOK ? rows_left - TOK : TOK);
        lc.x[0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X + (size_t)m0 * p.ldx), 0, (unsigned)r0 * (unsigned)p.ldx * 2u, 0x00020000);
        lc.x[1] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X + (size_t)(m0 + (r1 ? TOK : 0)) * p.ldx), 0, (unsigned)r1 * (unsigned)p.ldx * 2u, 0x00020000);
        lc.w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W + (size_t)n0 * p.ldw), 0, (unsigned)BN * (unsigned)p.ldw * 2u, 0x00020000);
        return;
    }
    const int xrows = rows_left < TROWS ? rows_left : TROWS;
    const unsigned xbytes = (unsigned)xrows * (unsigned)p.ldx * 2u;                     // reads past the last real row return 0
    lc.x[0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X + (size_t)m0 * p.ldx), 0, xbytes, 0x00020000);
    lc.x[1] = (WMODE == WIDE_CFG) ? __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X2 + (size_t)m0 * p.ldx), 0, xbytes, 0x00020000) : lc.x[0];
    lc.w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W + (size_t)n0 * p.ldw), 0, (unsigned)BN * (unsigned)p.ldw * 2u, 0x00020000);
}

template <int WMODE>
__global__ __launch_bounds__(512) void gemm_cfg2_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef Geo<WMODE> G_;
    constexpr int NST = G_::nst, STG_B = G_::stg, XT_B = G_::woff;      // (shadow the file-level three-stage constants)
    constexpr bool MIXK = G_::mix;                                      // one mixed pass: 4 activation blocks per wave, 3 DMA instructions per step
    constexpr bool MIX2 = WMODE == WIDE_MIX2;                           // 256-token single pass on the two-pass tile's geometry
    constexpr bool TWOX = WMODE == WIDE_CFG || MIX2;                    // two activation descriptors per stage (cond / null passes, or tokens 0..127 / 128..255)
    constexpr int TTOK = MIX2 ? 2 * TOK : TOK;                          // tokens of a tile (TOKT modes)
    unsigned char* ct = smem + G_::ct_off;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int total = p.tiles_m * p.tiles_n;
    const int G = gridDim.x;
    const int KT = p.K / BK;
    const int KH = KT >> 1;
    const int rd = fr * 64 + ((fg ^ ((-(fr >> 2)) & 3)) << 4);      // this lane's fragment offset inside a 16-row block
    constexpr bool TOKT = WMODE != WIDE_GEGLU;      // token-row tiles with an fp32 output (two-pass guidance / single mixed pass)
    constexpr int NDMA = MIXK ? 3 : 4;      // LDS-DMA instructions per wave and k-step

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    int vb = blockIdx.x;
    if (vb >= total) return;
    const int steps_total = ((total - 1 - vb) / G + 1) * KT;

    // ---- load cursor: runs two k-steps ahead of the compute cursor, across tile boundaries; step s lands in stage s % 3
    LoadCur lc;
    int l_vb = vb, l_k = 0;
    bool l_live = true;
    cfg_tile_setup<WMODE>(p, l_vb, lc);
    // per-lane byte offsets inside a tile, fixed for the whole kernel.  A DMA instruction covers one 16-row block: lane l fetches
    // row l >> 2, logical 16-byte chunk (l & 3) ^ swizzle into physical chunk l & 3.  This wave stages blocks 2*wid, 2*wid + 1 of
    // both operands; activation block xb: CFG wave row xb >> 3, pass (xb >> 2) & 1 (0 cond, 1 null), token block xb & 3.
    int voff_x[2], voff_w[2];
    {
        const int c = (lane & 3) ^ ((-(lane >> 4)) & 3);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int xb = 2 * wid + i;
            // WIDE_MIX: 8 activation blocks per stage, this wave stages block wid (voff_x[1] unused)
            const int xrow = TWOX ? (xb >> 3) * 64 + (xb & 3) * 16 + (lane >> 2) : (MIXK ? wid * 16 + (lane >> 2) : xb * 16 + (lane >> 2));
            voff_x[i] = xrow * p.ldx * 2 + c * 16;
            voff_w[i] = (xb * 16 + (lane >> 2)) * p.ldw * 2 + c * 16;
        }
    }
    const bool x_null0 = TWOX && (((2 * wid) >> 2) & 1);      // wave-uniform: blocks 2*wid and 2*wid + 1 share the pass
#define LOAD_NEXT(st_)                                                                                         \
    if (l_live) {                                                                                              \
        if (!ABL(p, 2)) {                                                                                      \
            unsigned char* xs_ = smem + (st_) * STG_B + wid * 2048;                                            \
            const int k0_ = l_k * (BK * 2);                                                                    \
            const __amdgpu_buffer_rsrc_t rx_ = x_null0 ? lc.x[1] : lc.x[0];                                    \
            if constexpr (MIXK) {                                                                              \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, (lds_ptr_t)(smem + (st_) * STG_B + wid * 1024), 16, voff_x[0], k0_, 0, 0); \
            } else {                                                                                           \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, (lds_ptr_t)(xs_), 16, voff_x[0], k0_, 0, 0);     \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, (lds_ptr_t)(xs_ + 1024), 16, voff_x[1], k0_, 0, 0); \
            }                                                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(lc.w, (lds_ptr_t)(xs_ + XT_B), 16, voff_w[0], k0_, 0, 0); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(lc.w, (lds_ptr_t)(xs_ + XT_B + 1024), 16, voff_w[1], k0_, 0, 0); \
        }                                                                                                      \
        if (++l_k == KT) {                                                                                     \
            l_k = 0;                                                                                           \
            l_vb += G;                                                                                         \
            l_live = l_vb < total;                                                                             \
            if (l_live) cfg_tile_setup<WMODE>(p, l_vb, lc);                                                    \
        }                                                                                                      \
    }
#pragma unroll
    for (int st0 = 0; st0 < NST; ++st0) { LOAD_NEXT(st0); }

    // Fused sampling (round 3: from the ACCUMULATORS): no logits leave the kernel and nothing goes through the staging tile.  At the end of a tile
    // every wave holds 64 tokens x 64 columns as fragments -- a lane owns four GRANULES (4 consecutive columns) of one token per token block, which
    // is exactly the candidate unit of sampling_fused.hip -- so a lane's share of the statistics is a purely in-lane reduction, the 16 lane groups
    // of a token (4 vocabulary quarters x 4) meet through 18 KiB of LDS (keep nibbles + (max, sum exp) pairs: no lane exchange at all), and each
    // kept granule is stored straight to its slot (rank = kept granules of the token's 256-column piece in front of it).  The k-loop of a fused
    // launch carries no output work.
    // The counted waits only need a LOWER bound of what was issued behind a DMA (under-counting waits for more, never for less).
    const bool fused = TOKT && (WMODE == WIDE_MIXF || MIX2 || p.fs_stats != nullptr);
    unsigned char* xch = ct;                       // fused: [tokens][4 quarters] keep-nibble words (16 B per token) | [tokens][16 lane groups] (group max, group sum exp) (float2)
    float* lthr = reinterpret_cast<float*>(ct + TTOK * (16 + 128));      // fused: the tile's per-token bounds (1 KiB, written by LDS-DMA at the tile's first k-step)
    const __amdgpu_buffer_rsrc_t thr_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fused ? p.fs_thr : nullptr), 0, fused ? (unsigned)p.M * 4u : 0u, 0x00020000);
    f32x4_t acc[4][8];
    f32x4_t held[2][4];             // second half of the previous tile's output (tokens 32..63 of this wave), combined
    bool have_prev = false;
    int pm0 = 0, pn0 = 0;
    int g = 0;                      // global k-step counter of the compute cursor
    int st1 = 0, st2 = 0, st3 = 0, st4 = 0;      // VMEM stores this wave issued in the previous .. fourth-previous k-step (wave-uniform; NST - 1 of them matter)

#define HELD_TO_CT()                                                                                           \
    {                                                                                                          \
        _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                        \
            const int hrow_ = wm * 32 + b * 16 + fr;                                                           \
            _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                                    \
                const int c_ = wn * 16 + a * 4 + fg;                                                           \
                *reinterpret_cast<f32x4_t*>(ct + hrow_ * 1024 + ((c_ ^ (hrow_ & 7)) << 4)) = held[b][a];       \
            }                                                                                                  \
        }                                                                                                      \
    }
    // one 8 KiB piece of ct = 8 token rows x 1 KiB: this wave handles row q*8 + wid, lane = 16-byte chunk
#define PIECE_ROW(q_) ((q_) * 8 + wid)
#define PIECE_TOKEN(hrow_, half_) (pm0 + ((hrow_) >> 5) * 64 + (half_) * 32 + ((hrow_) & 31))

    // Software pipeline of one k-step (fragments: afA/afB = the 4 weight fragments of the current / next step, p0 / p1 = two
    // 2-fragment activation buffers).  Entering step g, afA and p0 (token pair 0) of step g are in registers:
    //     read p1 <- pair 1 | MFMA pair 0 | read p0 <- pair 2 | MFMA pair 1 | read p1 <- pair 3
    //     wait (DMA of step g+1 landed; my reads of stage g done) - BARRIER - stage g is free: issue the DMA of step g+3
    //     MFMA pair 2 | read afB, p0 <- step g+1 (weights, pair 0) | MFMA pair 3 | store the piece read after the barrier
    // so every MFMA group runs with the reads of a later group in flight, also across the barrier.
#define X_FRAG(st_, j_) (*reinterpret_cast<const u32x4_t*>(smem + (st_) * STG_B + wm * (MIXK ? 4096 : 8192) + rd + (j_) * 1024))
#define W_FRAG(st_, i_) (*reinterpret_cast<const u32x4_t*>(smem + (st_) * STG_B + XT_B + wn * 4096 + rd + (i_) * 1024))
#define MFMA_PAIR(af_, src_, j_)                                                                               \
    if (!ABL(p, 4)) {                                                                                          \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                          \
            _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                      \
                acc[a][2 * (j_) + h] = mfma16(af_[a], src_[h], acc[a][2 * (j_) + h]);                          \
    } else { asm volatile("" ::"v"(af_[0]), "v"(af_[1]), "v"(af_[2]), "v"(af_[3]), "v"(src_[0]), "v"(src_[1])); }
#define STEP(AF_, AFN_)                                                                                        \
    {                                                                                                          \
        const int st_ = g % NST, stn_ = (g + 1) % NST;                                                         \
        p1[0] = X_FRAG(st_, 2); p1[1] = X_FRAG(st_, 3);                                                        \
        MFMA_PAIR(AF_, p0, 0)                                                                                  \
        if constexpr (!MIXK) {                  /* one mixed pass: 4 activation blocks per wave, pairs 0 and 1 only */ \
            p0[0] = X_FRAG(st_, 4); p0[1] = X_FRAG(st_, 5);                                                    \
            MFMA_PAIR(AF_, p1, 1)                                                                              \
            p1[0] = X_FRAG(st_, 6); p1[1] = X_FRAG(st_, 7);                                                    \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        /* step g+1 has landed once only what this wave issued after ITS DMA can be in flight: the store of step g-2, the */ \
        /* 4 DMA instructions of step g+2, the store of step g-1 */                                            \
        /* (general form: the DMAs of steps g+2 .. g+NST-1 and the stores of the last NST-1 steps were issued behind the DMA of step g+1) */ \
        {                                                                                                      \
            int ahead_ = steps_total - (g + 2);                                                                \
            ahead_ = ahead_ < 0 ? 0 : (ahead_ > NST - 2 ? NST - 2 : ahead_);                                   \
            wait_vmcnt((ABL(p, 2) ? 0 : ahead_ * NDMA) + st1 + st2 + (NST > 3 ? st3 + st4 : 0));               \
        }                                                                                                      \
        WAIT_LGKM0();                                                                                          \
        __builtin_amdgcn_s_barrier();                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (TOKT && kt == KH && have_prev) {                                                                   \
            /* every wave has read the last piece of the first half (step KH-1 at the latest): second half -> ct */ \
            HELD_TO_CT();                                                                                      \
            WAIT_LGKM0();                                                                                      \
            __builtin_amdgcn_s_barrier();                                                                      \
        }                                                                                                      \
        /* this step's piece of the previous tile.  CFG: first half during steps 0..7, second half during steps KH..KH+7, one */ \
        /* token row (1 KiB fp32) per wave.  GEGLU: the whole 256 x 128 bf16 tile during steps 0..7, 4 rows (256 B each) per wave */ \
        const int half_ = (TOKT && kt >= KH) ? 1 : 0;                                                          \
        const int q_ = kt - (half_ ? KH : 0);                                                                  \
        int prow_, ptok_;                                                                                      \
        const unsigned char* psrc_;                                                                            \
        unsigned char* pv_ptr_;                                                                                \
        if constexpr (TOKT) {                                                                                  \
            prow_ = PIECE_ROW(q_);                                                                             \
            ptok_ = PIECE_TOKEN(prow_, half_);                                                                 \
            psrc_ = ct + prow_ * 1024 + ((lane ^ (prow_ & 7)) << 4);                                           \
            pv_ptr_ = reinterpret_cast<unsigned char*>(reinterpret_cast<float*>(p.out) + (size_t)ptok_ * p.ldc + pn0 + lane * 4); \
        } else {                                                                                               \
            prow_ = q_ * 32 + 4 * wid + (lane >> 4);                                                           \
            ptok_ = pm0 + q_ * 32 + 4 * wid;                   /* first of this wave's 4 rows: decides (wave-uniformly) the issue */ \
            psrc_ = ct + prow_ * 256 + (((lane & 15) ^ (prow_ & 15)) << 4);                                    \
            pv_ptr_ = reinterpret_cast<unsigned char*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)(pm0 + prow_) * p.ldc + (pn0 >> 1) + (lane & 15) * 8); \
        }                                                                                                      \
        const bool piece_ = have_prev && q_ < 8 && ptok_ < p.M && !ABL(p, 1);      /* wave-uniform */          \
        uint4 pv_ = make_uint4(0, 0, 0, 0);                                                                    \
        if (piece_) pv_ = lds_read_b128_raw(psrc_);                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        LOAD_NEXT(st_);                                     /* step g+3 into the stage just consumed */        \
        if (TOKT && fused && kt == 0 && wid == 0) {         /* this tile's per-token bounds -> LDS (one more DMA of wave 0: its counted waits */ \
            int tm_, tn_;                                   /* under-count by one for two steps, which only waits for more) */ \
            xcd_grouped_tile(vb, p.tiles_m, p.tiles_n, 8, tm_, tn_);                                           \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(thr_rs, (lds_ptr_t)(lthr), 16, lane * 16, tm_ * TTOK * 4, 0, 0); \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if constexpr (!MIXK) { MFMA_PAIR(AF_, p0, 2) }                                                         \
        if (g + 1 < steps_total) {                                                                             \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) AFN_[i] = W_FRAG(stn_, i);                           \
            p0[0] = X_FRAG(stn_, 0); p0[1] = X_FRAG(stn_, 1);                                                  \
        }                                                                                                      \
        if constexpr (!MIXK) { MFMA_PAIR(AF_, p1, 3) } else { MFMA_PAIR(AF_, p1, 1) }                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        st4 = st3; st3 = st2;                                                                                  \
        st2 = st1;                                                                                             \
        st1 = 0;                                                                                               \
        if (piece_) {                                                                                          \
            WAIT_LGKM0();                                       /* the raw read of pv_ (the fragments are long there) */ \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
            if constexpr (TOKT) {                                                                              \
                store_stream(reinterpret_cast<float*>(pv_ptr_), pv_);                                          \
                st1 = 1;                                                                                       \
            }                                                                                                  \
            else {                                                                                             \
                if (p.ln_part) {      /* LayerNorm(inner) partial sums of the row's two 64-column groups (common.h) */ \
                    const float2 lst_ = ln_partial_row64(pv_);                                                 \
                    if ((lane & 7) == 0 && pm0 + prow_ < p.M)                                                  \
                        *reinterpret_cast<float2*>(p.ln_part + ((size_t)(pm0 + prow_) * p.ln_np + (pn0 >> 7) + ((lane >> 3) & 1)) * 2) = lst_; \
                }                                                                                              \
                if (pm0 + prow_ < p.M) *reinterpret_cast<uint4*>(pv_ptr_) = pv_;      /* re-read by the next kernel: a plain store */ \
                st1 = p.ln_part ? 2 : 1;                        /* (the wave's first row is valid -- piece_ --, so both stores were issued) */ \
            }                                                                                                  \
        }                                                                                                      \
        ++g;                                                                                                   \
        ++kt;                                                                                                  \
    }

#ifdef MM_GEMM_TIMING
    const bool ts_on = blockIdx.x == 0 && wid == 0;
    int ts_i = 0;
#endif
    TSTAMP()
    // prologue: step 0 has landed once only the DMA of steps 1 and 2 (8 instructions) is in flight
    wait_vmcnt((steps_total > NST - 1 ? NST - 1 : steps_total - 1) * NDMA);
    __builtin_amdgcn_s_barrier();
    u32x4_t afA[4], afB[4], p0[2], p1[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) afA[i] = W_FRAG(0, i);
    p0[0] = X_FRAG(0, 0); p0[1] = X_FRAG(0, 1);

    while (true) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT;) {      // KT is even: two steps per trip, the weight fragments ping-pong between afA and afB
            TSTAMP()
            STEP(afA, afB)
            TSTAMP()
            STEP(afB, afA)
        }
        TSTAMP()
        // ---- tile end: all pieces of the previous tile have been read; combine this tile and park it
        int tile_m, tile_n;
        xcd_grouped_tile(vb, p.tiles_m, p.tiles_n, 8, tile_m, tile_n);
        WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        TSTAMP()
        bool emitted = false;
        if constexpr (TOKT) {
            if (fused) {
                // ---- fused-sampling emission straight from the accumulator fragments (see the comment at `fused`; canonical form: common.h)
                emitted = true;
                int le_ = lane;                                     // opaque copy: keeps the emission's per-lane address arithmetic out of the k-loop's
                asm volatile("" : "+v"(le_));                       // live ranges (hoisted, it spilled loop-invariant values to scratch = VMEM traffic in the loop)
                const int FR_ = le_ & 15, FG_ = le_ >> 4;
                const int m0t = tile_m * TTOK;
                constexpr int NBLK = MIX2 ? 8 : 4;                 // token blocks of 16 per wave
                const int tilec = tile_n;                          // the piece index of this tile's 256 columns in a row of V / 256 pieces
                uint32_t* xmask = reinterpret_cast<uint32_t*>(xch);                          // [4 quarters][tokens]: the quarter's 32 keep bits of a token (bit 8 a + 2 f + h)
                float2* xml = reinterpret_cast<float2*>(xch + TTOK * 16);                    // [16 lane groups][tokens] (ml, pl)
                int nstore = 0;
                uint32_t mq[8];                          // the quarter's 32 keep bits of this lane's token of block b (bit 8 a + 2 f + h), after the exchange
#pragma unroll
                for (int b = 0; b < NBLK; ++b) {
                    const int tokl = (b >> 2) * TOK + wm * 64 + (b & 3) * 16 + FR_;
                    const bool valid = m0t + tokl < p.M;
                    const float thr = lthr[tokl];
                    if constexpr (WMODE == WIDE_CFG) {      // combine the two passes in place: the null fragments are dead afterwards (mmp.py:254)
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float cv = acc[a][b][r], nl = acc[a][b + 4][r];
                                acc[a][b][r] = nl + (cv - nl) * p.cfg_scale;
                            }
                    }
                    // this lane IS lane group (wn, FG_) of the token: its 8 granules (a, h), no lane exchange anywhere
                    float g2[4][2], gm[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        g2[a][0] = max2_asm(acc[a][b][0], acc[a][b][1]);
                        g2[a][1] = max2_asm(acc[a][b][2], acc[a][b][3]);
                        gm[a] = max2_asm(g2[a][0], g2[a][1]);
                    }
                    const float ml = max4_asm(gm[0], gm[1], gm[2], gm[3]);
                    float gs[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        gs[a] = (fs_exp(acc[a][b][0], ml) + fs_exp(acc[a][b][1], ml)) + (fs_exp(acc[a][b][2], ml) + fs_exp(acc[a][b][3], ml));
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
                    if (FG_ == 0) xmask[wn * TTOK + tokl] = m32;
                    xml[(wn * 4 + FG_) * TTOK + tokl] = make_float2(ml, pl);
                }
                TSTAMP()
                WAIT_LGKM0();
                __builtin_amdgcn_s_barrier();      // all 16 lane groups of every token have published their masks and (ml, pl)
                TSTAMP()
                // ---- the kept granules go out compacted in column order: a quarter starts behind the kept granules of the quarters in front of it
#pragma unroll
                for (int b = 0; b < NBLK; ++b) {
                    const uint32_t mine = (mq[b] >> (2 * FG_)) & 0x03030303u;      // bit 8 a + h: this lane's granule (a, h) of block b
                    if (__ballot(mine != 0u) == 0ull) continue;                   // wave-uniform
                    const int tokl = (b >> 2) * TOK + wm * 64 + (b & 3) * 16 + FR_;
                    int base = 0;                                                 // kept granules of the quarters in front of this one
                    if (wn > 0) base += __popc(xmask[tokl]);
                    if (wn > 1) base += __popc(xmask[TTOK + tokl]);
                    if (wn > 2) base += __popc(xmask[2 * TTOK + tokl]);
                    float2* slot = reinterpret_cast<float2*>(p.fs_cand + ((size_t)(m0t + tokl) * p.tiles_n + tilec) * FS_SLOT) + base;
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const bool kp = (mine >> (8 * a + h)) & 1u;
                            if (__ballot(kp) != 0ull) {                         // wave-uniform: the store below is ISSUED (exact VMEM count for the waits)
                                if (kp) slot[__popc(mq[b] & ((1u << (8 * a + 2 * FG_ + h)) - 1u))] = make_float2(acc[a][b][2 * h], acc[a][b][2 * h + 1]);
                                ++nstore;
                            }
                        }
                }
                // one record per (token, piece): this wave combines the tokens of the token blocks b with (b & 3) == wn (its lanes FG_ == 0)
#pragma unroll
                for (int hb = 0; hb < NBLK / 4; ++hb) {
                    const int tokl = hb * TOK + wm * 64 + wn * 16 + FR_;
                    const int tok = m0t + tokl;
                    const bool w_ = FG_ == 0 && tok < p.M;
                    if (__ballot(w_) != 0ull) {
                        if (w_) {
                            // tile_combine16 (common.h) streamed from LDS in two sweeps (max, then the weighted sums in the canonical order)
                            const float2* gq = xml + tokl;                          // lane group g of this token: gq[g * TTOK]
                            float M_ = -INFINITY;
#pragma unroll
                            for (int i = 0; i < 16; ++i) M_ = fmaxf(M_, gq[i * TTOK].x);
                            float wq[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float2 u0 = gq[(4 * q) * TTOK], u1 = gq[(4 * q + 1) * TTOK], u2 = gq[(4 * q + 2) * TTOK], u3 = gq[(4 * q + 3) * TTOK];
                                wq[q] = (u0.y * __expf(u0.x - M_) + u1.y * __expf(u1.x - M_)) + (u2.y * __expf(u2.x - M_) + u3.y * __expf(u3.x - M_));
                            }
                            const float E_ = (wq[0] + wq[1]) + (wq[2] + wq[3]);
                            float4* rec = p.fs_stats + ((size_t)tok * p.tiles_n + tilec) * FS_REC;
                            rec[0] = make_float4(M_, E_, 0.f, 0.f);
                            rec[1] = make_float4(__uint_as_float(xmask[tokl]), __uint_as_float(xmask[TTOK + tokl]), __uint_as_float(xmask[2 * TTOK + tokl]),
                                                 __uint_as_float(xmask[3 * TTOK + tokl]));
                        }
                        nstore += 2;
                    }
                }
                st1 += nstore;      // issued after this wave's last DMA: the next NST - 1 steps' counted waits allow for them
                TSTAMP()
            }
        }
        if constexpr (TOKT) {
            if (!emitted) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    f32x4_t v;
                    if constexpr (MIXK) {
                        v = acc[a][b];      // the passes were combined in the embedding (mm_cfg_mix): these ARE the guidance logits
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float cv = acc[a][b][r], nl = acc[a][b + 4][r];
                            v[r] = nl + (cv - nl) * p.cfg_scale;      // muse_maskgit_pytorch.py:254
                        }
                    }
                    if (b < 2) {
                        if (!ABL(p, 16)) {
                            const int hrow = wm * 32 + b * 16 + fr;
                            const int c = wn * 16 + a * 4 + fg;
                            *reinterpret_cast<f32x4_t*>(ct + hrow * 1024 + ((c ^ (hrow & 7)) << 4)) = v;
                        }
                    } else {
                        held[b - 2][a] = v;
                    }
                }
            }
            }
        } else {
            // GEGLU (mmp.py:72-77): the weight rows are interleaved so that fragments a = 0,1 hold the gelu half and a = 2,3 the gate
            // half of the SAME 32 output columns of this wave; the tile emits 256 rows x 128 columns of bf16 = the whole 64 KiB ct
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int row = wm * 128 + b * 16 + fr;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int col = wn * 32 + a * 16 + fg * 4;
                    *reinterpret_cast<uint2*>(ct + row * 256 + (((col >> 3) ^ (row & 15)) << 4) + (fg & 1) * 8) =
                        make_uint2(pack_bf16x2(geglu_f(acc[a][b][0], acc[a + 2][b][0]), geglu_f(acc[a][b][1], acc[a + 2][b][1])),
                                   pack_bf16x2(geglu_f(acc[a][b][2], acc[a + 2][b][2]), geglu_f(acc[a][b][3], acc[a + 2][b][3])));
                }
            }
        }
        pm0 = tile_m * (TOKT ? TOK : 2 * TOK); pn0 = tile_n * BN;
        have_prev = !emitted;      // (a fused tile has left already: nothing rides on the next tile's k-loop)
        vb += G;
        if (vb >= total) break;
    }
    // ---- drain the last tile
    WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    if constexpr (TOKT) {
        if (fused) return;
        for (int half = 0; half < 2; ++half) {
            if (half) {
                HELD_TO_CT();
                WAIT_LGKM0();
                __builtin_amdgcn_s_barrier();
            }
            for (int q = 0; q < 8; ++q) {
                const int hrow = PIECE_ROW(q);
                const int ptok = PIECE_TOKEN(hrow, half);
                if (ptok < p.M && !ABL(p, 1)) {
                    const uint4 pv = *reinterpret_cast<const uint4*>(ct + hrow * 1024 + ((lane ^ (hrow & 7)) << 4));
                    store_stream(reinterpret_cast<float*>(p.out) + (size_t)ptok * p.ldc + pn0 + lane * 4, pv);
                }
            }
            WAIT_LGKM0();
            __builtin_amdgcn_s_barrier();
        }
    } else {
        for (int q = 0; q < 8; ++q) {
            const int row = q * 32 + 4 * wid + (lane >> 4);
            const uint4 pv = *reinterpret_cast<const uint4*>(ct + row * 256 + (((lane & 15) ^ (row & 15)) << 4));
            if (p.ln_part) {
                const float2 lst = ln_partial_row64(pv);
                if ((lane & 7) == 0 && pm0 + row < p.M)
                    *reinterpret_cast<float2*>(p.ln_part + ((size_t)(pm0 + row) * p.ln_np + (pn0 >> 7) + ((lane >> 3) & 1)) * 2) = lst;
            }
            if (pm0 + row < p.M && !ABL(p, 1))
                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)(pm0 + row) * p.ldc + (pn0 >> 1) + (lane & 15) * 8) = pv;
        }
    }
}

}  // namespace

// K a multiple of 64 with at least 16 k-steps of 32 (the store pieces of a tile ride on the next tile's k-loop), N a multiple of
// the 256-row weight tile, 16-byte aligned output rows, enough tiles for one workgroup per CU.
// CFG: fp32 logits of both guidance passes + the combine.  GEGLU: FF w1 with the fused gate * gelu(x) epilogue (bf16, N/2 columns).
bool mm_gemm_cfg2_eligible(const GemmArgs& a) {
    if (a.bias || a.act != ACT_NONE || a.resid_bf16 || a.resid_f32) return false;
    if ((a.K % (2 * BK)) != 0 || a.K < 16 * BK || (a.N % BN) != 0 || (((uintptr_t)a.out) & 15)) return false;
    if (a.fs_stats && a.mode != MODE_CFG && !(a.mode == MODE_DENSE && a.wide_tok)) return false;
    if (a.mode == MODE_CFG || (a.mode == MODE_DENSE && a.wide_tok && a.epi == EPI_NONE)) {      // fp32 logits of 128-token tiles: two-pass guidance / one mixed pass
        if (a.out_kind != OUT_F32 || (a.ldc % 4)) return false;
        return (long)((a.M + TOK - 1) / TOK) * (a.N / BN) >= 256;
    }
    if (a.mode == MODE_DENSE && a.epi == EPI_GEGLU) {
        if (a.out_kind != OUT_BF16 || (a.ldc % 8)) return false;
        const long tiles = (long)((a.M + 2 * TOK - 1) / (2 * TOK)) * (a.N / BN);
        // measured (tools/gemm_harness, MI355X): per 256 x 256 of output the k-loop is ~7 % faster than gemm_pers', the tile end costs about the
        // same, and the coarser tile loses more to the last partial round -- a win when the tile count fills the last round of CUs well
        // (FF w1 of the base config, 16384 x 2816 x 512 = 704 tiles = 2.75 rounds: 66.3 vs 71.3 us; 16384 x 4096 x 2048: 253 vs 267 us)
        const long rounds = (tiles + 255) / 256;
        return tiles >= 256 && (tiles * 10 >= rounds * 256 * 9 || tiles >= 4096);
    }
    return false;
}

int mm_gemm_cfg2_launch(GemmArgs a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_cfg2_kernel<WIDE_CFG>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_cfg2_kernel<WIDE_GEGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_cfg2_kernel<WIDE_MIX>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_cfg2_kernel<WIDE_MIXF>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_cfg2_kernel<WIDE_MIX2>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
        if (e != hipSuccess) return mm_set_hip_error(e, "gemm_cfg2 hipFuncSetAttribute");
        attr_set = true;
    }
    const bool cfg = a.mode == MODE_CFG;
    const bool mix = a.mode == MODE_DENSE && a.wide_tok && a.epi == EPI_NONE;
    // fused single pass: 256-token tiles when the launch has enough rows to fill them (fewer operand bytes per flop), 128-token tiles with five stages
    // otherwise (debug bits, A/B only: 1 << 26 three-stage 128-token kernel, 1 << 28 no 256-token tiles; the values are the same)
    if (mix && !(g_mm_debug & ((1 << 26) | (1 << 28) | (1 << 30))) && mm_gemm_wide_fused_eligible(a)) return mm_gemm_wide_fused_launch(a, stream);      // (bit 1 << 30: A/B)
    const bool mix2 = mix && a.fs_stats && a.M >= 1024 && !(g_mm_debug & ((1 << 26) | (1 << 28)));
    a.tiles_n = a.N / BN;
    a.tiles_m = ((cfg || mix) && !mix2) ? (a.M + TOK - 1) / TOK : (a.M + 2 * TOK - 1) / (2 * TOK);
    const int total = a.tiles_m * a.tiles_n;
    const int grid = total < 256 ? total : 256;
    if (cfg) hipLaunchKernelGGL(gemm_cfg2_kernel<WIDE_CFG>, dim3(grid), dim3(512), SMEM_B, stream, a);
    else if (mix2) hipLaunchKernelGGL(gemm_cfg2_kernel<WIDE_MIX2>, dim3(grid), dim3(512), SMEM_B, stream, a);
    else if (mix && a.fs_stats && !(g_mm_debug & (1 << 26))) hipLaunchKernelGGL(gemm_cfg2_kernel<WIDE_MIXF>, dim3(grid), dim3(512), SMEM_B, stream, a);      // (bit 1 << 26: A/B against three stages)
    else if (mix) hipLaunchKernelGGL(gemm_cfg2_kernel<WIDE_MIX>, dim3(grid), dim3(512), SMEM_B, stream, a);
    else hipLaunchKernelGGL(gemm_cfg2_kernel<WIDE_GEGLU>, dim3(grid), dim3(512), SMEM_B, stream, a);
    return mm_check_launch("gemm_cfg2_kernel");
}

#ifdef MM_GEMM_TIMING
extern "C" int mm_debug_cfg2_stamps(unsigned long long* host_dst, int n) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_cfg2_stamps), sizeof(unsigned long long) * n);
}
#endif
