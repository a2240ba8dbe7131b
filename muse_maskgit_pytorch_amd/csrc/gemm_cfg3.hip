// to_logits + classifier-free guidance, third generation (muse_maskgit_pytorch.py:250-254, 332): the 128-token x {cond, null} x 256-column
// persistent tile of gemm_cfg.hip with the k-step split into a COMPUTE phase and a LOAD phase, and the two wave groups of the workgroup
// (waves 0-3 = token half 0, waves 4-7 = token half 1: one of each per SIMD) running HALF A STEP OUT OF PHASE.
//
// Why: cycle stamps of gemm_cfg2 showed a k-step of ~2050 cycles for 1088 cycles of MFMA work per SIMD.  Both waves of a SIMD sat in the
// same phase: after the step's barrier both issued their LDS-DMA (4 instructions of 100-185 cycles each when the partner is in the same
// issue, MI355X_MICROARCH.md) and their fragment reads while the matrix pipe idled, then both queued MFMAs.  Here, in every interval
// between two barriers one wave of each SIMD issues nothing but its 32 MFMAs (+ 6 fragment reads) and the other does everything else:
// LDS-DMA for a later step, the first fragments of its next step, the output pieces of the previous tile (LDS staging tile -> fused
// sampling emission or logits store) and, once per tile, the guidance combine.  Roles swap at every barrier.
//
//   interval        2g          2g+1         2g+2         2g+3
//   group A      C(g)         L(g)         C(g+1)       L(g+1)         C = compute k-step, L = load phase
//   group B      L(g-1)       C(g)         L(g)         C(g+1)
//
// Three 32 KiB stages (step s in stage s % 3), hazards:
//   * L_A(g) refills the stage of step g-1 (last read by B in interval 2g-1) with step g+2; L_B(g) refills the stage of step g (last read
//     by B itself in 2g+1) with step g+3.  So A's DMA runs two steps ahead, B's three.
//   * A's part of step g+1 (issued in L_A(g-1)) is waited for at the END of C_A(g) (counted vmcnt), B's part (issued in L_B(g-2)) at the END
//     of L_B(g-1): both before barrier 2g+1, after which A (in L_A(g)) and B (in L_B(g), one interval later) read the first fragments of
//     step g+1.  Every DMA has more than a full interval to land.
// The fp32 output tile leaves as in gemm_cfg2 through the 64 KiB staging tile `ct` in two halves (the second half waits in 32 VGPRs), but
// one token row per wave and load phase (each group drains the rows its own waves staged); the row's statistics / stores are the first thing
// of the following load phase.  No VMEM or VALU-heavy work sits in a compute phase: whatever is added there stalls the one wave that feeds
// the SIMD's matrix pipe.  Same accumulation order per output element as every other kernel of the family -> bit-identical values.
#include "common.h"
#include "muse_hip_internal.h"

#ifdef MM_GEMM_TIMING      // tools/cfg3_timing.py only: s_memtime stamps of workgroup 0, wave 0 (group A) and wave 4 (group B)
__device__ unsigned long long g_cfg3_stamps[2][2048];
#define TSTAMP() if (ts_on && ts_i < 2048) { g_cfg3_stamps[wm][ts_i++] = __builtin_readcyclecounter(); }
#else
#define TSTAMP()
#endif
#ifndef MM_EXP
#define MM_EXP 0
#endif

namespace {

constexpr int TOK = 128, BN = 256, BK = 32, NST = 3;
constexpr int XT_B = 2 * TOK * BK * 2;      // 16 KiB: 256 activation rows (128 tokens x {cond, null}) x 64 B
constexpr int WT_B = BN * BK * 2;           // 16 KiB
constexpr int STG_B = XT_B + WT_B;          // 32 KiB
constexpr int CT_OFF = NST * STG_B;         // 96 KiB
constexpr int SMEM_B = CT_OFF + 65536;      // 160 KiB

#define WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

__device__ __forceinline__ void wait_vmcnt(int n) {      // n is wave-uniform; waits are builtins so that hipcc's waitcnt pass sees them
    switch (n) {
        case 0: __builtin_amdgcn_s_waitcnt(0x0F70); break;
        case 2: __builtin_amdgcn_s_waitcnt(0x0F72); break;
        case 4: __builtin_amdgcn_s_waitcnt(0x0F74); break;
        case 6: __builtin_amdgcn_s_waitcnt(0x0F76); break;
        case 1: __builtin_amdgcn_s_waitcnt(0x0F71); break;
        case 3: __builtin_amdgcn_s_waitcnt(0x0F73); break;
        case 5: __builtin_amdgcn_s_waitcnt(0x0F75); break;
        case 7: __builtin_amdgcn_s_waitcnt(0x0F77); break;
        case 8: __builtin_amdgcn_s_waitcnt(0x0F78); break;
        default: __builtin_amdgcn_s_waitcnt(0x0F70); break;
    }
}

// raw LDS read of the staging tile (the compiler cannot tell ct from the DMA stages and would drain vmcnt(0) in front of a visible read);
// the caller waits lgkmcnt(0) before using the value
__device__ __forceinline__ uint4 lds_read_b128_raw(const unsigned char* ptr) {
    typedef __attribute__((address_space(3))) const unsigned char* lds_cptr_t;
    const unsigned addr = (unsigned)(size_t)(lds_cptr_t)ptr;
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// scalar load at a wave-uniform address (valid after the next lgkmcnt(0)); inline asm: an ordinary load becomes a global_load and hipcc then
// drains vmcnt(0) -- the whole DMA pipeline -- in front of its first use
__device__ __forceinline__ float sload_f32(const float* ptr) {
    float v;
    asm volatile("s_load_dword %0, %1, 0x0" : "=s"(v) : "s"(ptr) : "memory");
    return v;
}

__device__ __forceinline__ void store_stream(float* ptr, const uint4 v) {
    __builtin_nontemporal_store(u32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_t*>(ptr));
}

struct LoadCur {
    __amdgpu_buffer_rsrc_t x[2];      // cond rows / null rows of the tile
    __amdgpu_buffer_rsrc_t w;
};

__device__ __forceinline__ void tile_setup(const GemmArgs& p, int vb, LoadCur& lc) {
    int tile_m, tile_n;
    xcd_grouped_tile(vb, p.tiles_m, p.tiles_n, 8, tile_m, tile_n);
    const int m0 = tile_m * TOK, n0 = tile_n * BN;
    const int rows_left = p.M - m0;
    const int xrows = rows_left < TOK ? rows_left : TOK;
    const unsigned xbytes = (unsigned)xrows * (unsigned)p.ldx * 2u;                     // reads past the last real row return 0
    lc.x[0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X + (size_t)m0 * p.ldx), 0, xbytes, 0x00020000);
    lc.x[1] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X2 + (size_t)m0 * p.ldx), 0, xbytes, 0x00020000);
    lc.w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W + (size_t)n0 * p.ldw), 0, (unsigned)BN * (unsigned)p.ldw * 2u, 0x00020000);
}

// the whole kernel body for one wave group (WM = 0: group A, 1: group B): the group-specific waits and look-ahead are compile-time
template <int WM, bool FUSED>
__device__ __forceinline__ void cfg3_body(const GemmArgs& p, unsigned char* smem) {
    unsigned char* ct = smem + CT_OFF;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int wm = WM;
    const int wn = wid & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int total = p.tiles_m * p.tiles_n;
    const int G = gridDim.x;
    const int KT = p.K / BK;
    const int rd = fr * 64 + ((fg ^ ((-(fr >> 2)) & 3)) << 4);      // this lane's fragment offset inside a 16-row block

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    int vb = blockIdx.x;
    const int S = ((total - 1 - vb) / G + 1) * KT;      // k-steps of this workgroup over all its tiles

    // ---- load cursor of this wave: blocks 2*wid, 2*wid + 1 of both operands (group A: its own token rows + weight rows 0..127, group B: its
    // token rows + weight rows 128..255).  Activation block xb: wave row xb >> 3, pass (xb >> 2) & 1 (0 cond, 1 null), token block xb & 3.
    LoadCur lc;
    int l_vb = vb, l_k = 0;
    bool l_live = true;
    tile_setup(p, l_vb, lc);
    int voff_x[2], voff_w[2];
    {
        const int c = (lane & 3) ^ ((-(lane >> 4)) & 3);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int xb = 2 * wid + i;
            const int xrow = (xb >> 3) * 64 + (xb & 3) * 16 + (lane >> 2);
            voff_x[i] = xrow * p.ldx * 2 + c * 16;
            voff_w[i] = (xb * 16 + (lane >> 2)) * p.ldw * 2 + c * 16;
        }
    }
    const bool x_null0 = ((2 * wid) >> 2) & 1;      // wave-uniform: blocks 2*wid and 2*wid + 1 share the pass
    int l_st = 0;                                   // stage the next DMA goes to
#define LOAD_NEXT()                                                                                            \
    if (l_live) {                                                                                              \
        unsigned char* xs_ = smem + l_st * STG_B + wid * 2048;                                                 \
        const int k0_ = l_k * (BK * 2);                                                                        \
        const __amdgpu_buffer_rsrc_t rx_ = x_null0 ? lc.x[1] : lc.x[0];                                        \
        if (!(MM_EXP == 3 || MM_EXP == 8)) {                                                                   \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, (lds_ptr_t)(xs_), 16, voff_x[0], k0_, 0, 0);             \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, (lds_ptr_t)(xs_ + 1024), 16, voff_x[1], k0_, 0, 0);      \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(lc.w, (lds_ptr_t)(xs_ + XT_B), 16, voff_w[0], k0_, 0, 0);     \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(lc.w, (lds_ptr_t)(xs_ + XT_B + 1024), 16, voff_w[1], k0_, 0, 0); \
        }                                                                                                      \
        l_st = l_st == NST - 1 ? 0 : l_st + 1;                                                                 \
        if (++l_k == KT) {                                                                                     \
            l_k = 0;                                                                                           \
            l_vb += G;                                                                                         \
            l_live = l_vb < total;                                                                             \
            if (l_live) tile_setup(p, l_vb, lc);                                                               \
        }                                                                                                      \
    }
    LOAD_NEXT();
    LOAD_NEXT();
    if (wm) { LOAD_NEXT(); }

    constexpr bool fused = FUSED;
    f32x4_t acc[4][8];
    f32x4_t held[2][4];             // second half of the previous tile's output (tokens 32..63 of this wave), combined
    bool have_prev = false;
    int pm0 = 0, pn0 = 0;
    int g = 0;                      // global k-step counter
    int st = 0;                     // stage of step g

#define X_FRAG(st_, j_) (*reinterpret_cast<const u32x4_t*>(smem + (st_) * STG_B + wm * 8192 + rd + (j_) * 1024))
#define W_FRAG(st_, i_) (*reinterpret_cast<const u32x4_t*>(smem + (st_) * STG_B + XT_B + wn * 4096 + rd + (i_) * 1024))
#define MFMA_PAIR(src_, j_)                                                                                    \
    if (MM_EXP != 4) {                                                                                         \
    _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                              \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                          \
            acc[a][2 * (j_) + h] = mfma16(af[a], src_[h], acc[a][2 * (j_) + h]);                               \
    } else { asm volatile("" ::"v"(af[0]), "v"(af[1]), "v"(af[2]), "v"(af[3]), "v"(src_[0]), "v"(src_[1])); }
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
#define PIECE_TOKEN(hrow_, half_) (pm0 + ((hrow_) >> 5) * 64 + (half_) * 32 + ((hrow_) & 31))
// Fused sampling emission of a staged row, in two parts (common.h fused_emit_piece is the one-piece form of the same arithmetic): the
// statistics are branch-free VALU work that rides between the MFMAs of a compute phase; the two stores go out in the following load phase
// (a VMEM instruction inside a compute phase would stall the only wave that is feeding the matrix pipe of its SIMD).
#define STATS_ROW(i_)                                                                                          \
    if (FUSED && MM_EXP != 11 && pok[i_]) {                                                                                               \
        const float x0_ = __uint_as_float(pv[i_].x), x1_ = __uint_as_float(pv[i_].y), x2_ = __uint_as_float(pv[i_].z), x3_ = __uint_as_float(pv[i_].w); \
        const float m4_ = fmaxf(fmaxf(x0_, x1_), fmaxf(x2_, x3_));                                             \
        const float m_ = wave_max_dpp(m4_);                                                                    \
        se[i_] = wave_sum_dpp((__expf(x0_ - m_) + __expf(x1_ - m_)) + (__expf(x2_ - m_) + __expf(x3_ - m_)));  \
        sm[i_] = m_;                                                                                           \
        skp[i_] = m4_ >= fthr[i_];                                                                             \
        sbal[i_] = __ballot(skp[i_]);                                                                          \
        srank[i_] = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(sbal[i_] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sbal[i_], 0u)); \
    }
#define STORE_ROW(i_)                                                                                          \
    if (pok[i_] && MM_EXP != 9) {                                                                                             \
        if (FUSED) {                                                                                           \
            const size_t slot_ = (size_t)ptok[i_] * p.tiles_n + (pn0 >> 8);                                    \
            if (sbal[i_] != 0ull) {      /* wave-uniform: the store below is issued exactly when a lane keeps (the counted vmcnt relies on it) */ \
                if (skp[i_]) p.fs_cand[slot_ * FS_SLOT + srank[i_]] = make_float4(__uint_as_float(pv[i_].x), __uint_as_float(pv[i_].y), __uint_as_float(pv[i_].z), __uint_as_float(pv[i_].w)); \
                st_now += 1;                                                                                   \
            }                                                                                                  \
            if (lane == 0) p.fs_stats[slot_] = make_float4(sm[i_], se[i_], __uint_as_float((uint32_t)sbal[i_]), __uint_as_float((uint32_t)(sbal[i_] >> 32))); \
        } else {                                                                                               \
            store_stream(reinterpret_cast<float*>(p.out) + (size_t)ptok[i_] * p.ldc + pn0 + lane * 4, pv[i_]); \
        }                                                                                                      \
        st_now += 1;                                                                                           \
    }

#ifdef MM_GEMM_TIMING
    const bool ts_on = blockIdx.x == 0 && wn == 0;
    int ts_i = 0;
#endif
    // ---- prologue: step 0 has landed once only the DMA of the later steps is in flight
    wait_vmcnt(wm ? 8 : 4);
    __builtin_amdgcn_s_barrier();                                       // barrier 0
    u32x4_t af[4], p0[2], p1[2], p2[2], p3[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = W_FRAG(0, i);
    p0[0] = X_FRAG(0, 0); p0[1] = X_FRAG(0, 1);
    if (wm) {                                                           // group B starts half a step late; its part of step 1 must be there at barrier 1
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt(4);
        WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    // the two staged output rows this wave took out of ct in its last load phase: emitted inside the next compute phase, between the MFMAs
    uint4 pv[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    float fthr[2] = {0.f, 0.f};
    int ptok[2] = {0, 0};
    bool pok[2] = {false, false};
    float sm[2] = {0.f, 0.f}, se[2] = {0.f, 0.f};
    bool skp[2] = {false, false};
    unsigned long long sbal[2] = {0ull, 0ull};
    int srank[2] = {0, 0};

    while (true) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT; ++kt) {
            const int stn = st == NST - 1 ? 0 : st + 1;
            // ================================================================ compute phase: 32 MFMAs, 6 fragment reads, the emission of the rows
            // staged in the last load phase's statistics (branch-free VALU between the MFMAs; no VMEM instruction in this phase)
            TSTAMP()
            // all six remaining activation fragments are requested up front: the first MFMA group covers their latency and no later group waits
            p1[0] = X_FRAG(st, 2); p1[1] = X_FRAG(st, 3);
            p2[0] = X_FRAG(st, 4); p2[1] = X_FRAG(st, 5);
            p3[0] = X_FRAG(st, 6); p3[1] = X_FRAG(st, 7);
            MFMA_PAIR(p0, 0)
            MFMA_PAIR(p1, 1)
            MFMA_PAIR(p2, 2)
            MFMA_PAIR(p3, 3)
            __builtin_amdgcn_sched_barrier(0);
            TSTAMP()
            if (!wm && MM_EXP != 7) wait_vmcnt(0);                                     // group A: its DMA of step g+1 (issued one load phase ago) has landed
            WAIT_LGKM0();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ================================================================ load phase
            TSTAMP()
            int st_now = 0;                                             // VMEM instructions this phase issues in front of its DMA (exact)
            STATS_ROW(0)
            STORE_ROW(0)
            STATS_ROW(1)
            STORE_ROW(1)
            __builtin_amdgcn_sched_barrier(0);
            const bool issued = l_live;
            LOAD_NEXT();
            __builtin_amdgcn_sched_barrier(0);
            TSTAMP()
            // Staged output of the previous tile: a group only ever reads the ct rows its own waves wrote (rows wm*32 .. wm*32+31), so the two
            // groups' half step of skew creates no hazard on ct.  One row per wave and load phase, requested here, turned into statistics +
            // stores at the start of the NEXT load phase: first half in L(0..7); second half -> ct in L(8); its rows in L(9), L(10) (two each)
            // and L(11..14); the next combine writes ct in L(KT-1) (KT >= 16).
            int nrow = 0, r0 = 0, half = 0;
            if (have_prev && !(MM_EXP == 5 || MM_EXP == 8)) {
                if (kt < 8) { nrow = 1; r0 = kt; }
                else if (kt == 9 || kt == 10) { nrow = 2; r0 = (kt - 9) * 2; half = 1; }
                else if (kt >= 11 && kt <= 14) { nrow = 1; r0 = kt - 7; half = 1; }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int prow = wm * 32 + (r0 + i) * 4 + wn;
                ptok[i] = PIECE_TOKEN(prow, half);
                pok[i] = i < nrow && ptok[i] < p.M;                     // wave-uniform
                if (pok[i] && MM_EXP != 10) {
                    pv[i] = lds_read_b128_raw(ct + prow * 1024 + ((lane ^ (prow & 7)) << 4));
                    if (fused) fthr[i] = sload_f32(p.fs_thr + __builtin_amdgcn_readfirstlane(ptok[i]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < S) {                                            // first fragments of this wave's next step
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = W_FRAG(stn, i);
                p0[0] = X_FRAG(stn, 0); p0[1] = X_FRAG(stn, 1);
            }
            if (kt == 8 && have_prev) {                                 // this group's waves have read its first-half rows (L(7))
                HELD_TO_CT();
            }
            if (kt == KT - 1) {
                // ---- tile end: combine cond / null (muse_maskgit_pytorch.py:254), first token half -> ct, second half stays in registers
                int tile_m, tile_n;
                xcd_grouped_tile(vb, p.tiles_m, p.tiles_n, 8, tile_m, tile_n);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        f32x4_t v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float cv = acc[a][b][r], nl = acc[a][b + 4][r];
                            v[r] = nl + (cv - nl) * p.cfg_scale;
                        }
                        if (b < 2) {
                            const int hrow = wm * 32 + b * 16 + fr;
                            const int c = wn * 16 + a * 4 + fg;
                            *reinterpret_cast<f32x4_t*>(ct + hrow * 1024 + ((c ^ (hrow & 7)) << 4)) = v;
                        } else {
                            held[b - 2][a] = v;
                        }
                    }
                }
                pm0 = tile_m * TOK; pn0 = tile_n * BN;
            }
            __builtin_amdgcn_sched_barrier(0);
            TSTAMP()
            if (wm && MM_EXP != 7) wait_vmcnt((issued ? 4 : 0) + st_now);                // group B: its DMA of step g+2 (issued in its previous load phase) has landed
            WAIT_LGKM0();
            TSTAMP()
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            st = stn;
            ++g;
        }
        have_prev = true;
        vb += G;
        if (vb >= total) break;
    }
    if (!wm) __builtin_amdgcn_s_barrier();                              // group A's half step of skew
    // ---- drain the last tile (every wave is past its last load phase)
    for (int half = 0; half < 2; ++half) {
        if (half) {
            HELD_TO_CT();
            WAIT_LGKM0();
            __builtin_amdgcn_s_barrier();
        }
        for (int q = 0; q < 8; ++q) {
            const int hrow = q * 8 + wid;
            const int ptk = PIECE_TOKEN(hrow, half);
            if (ptk < p.M) {
                const uint4 v = *reinterpret_cast<const uint4*>(ct + hrow * 1024 + ((lane ^ (hrow & 7)) << 4));
                if (fused) fused_emit_piece(make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)), ptk, pn0 >> 8,
                                            p.tiles_n, lane, p.fs_thr[ptk], p.fs_stats, p.fs_cand);
                else store_stream(reinterpret_cast<float*>(p.out) + (size_t)ptk * p.ldc + pn0 + lane * 4, v);
            }
        }
        WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
    }
}

__global__ __launch_bounds__(512) void gemm_cfg3_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x >= p.tiles_m * p.tiles_n) return;
    const bool grp_b = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) != 0;
    if (p.fs_stats) { if (grp_b) cfg3_body<1, true>(p, smem); else cfg3_body<0, true>(p, smem); }
    else { if (grp_b) cfg3_body<1, false>(p, smem); else cfg3_body<0, false>(p, smem); }
}

}  // namespace

int mm_gemm_cfg3_launch(GemmArgs a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_cfg3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
        if (e != hipSuccess) return mm_set_hip_error(e, "gemm_cfg3 hipFuncSetAttribute");
        attr_set = true;
    }
    a.tiles_n = a.N / BN;
    a.tiles_m = (a.M + TOK - 1) / TOK;
    const int total = a.tiles_m * a.tiles_n;
    const int grid = total < 256 ? total : 256;
    hipLaunchKernelGGL(gemm_cfg3_kernel, dim3(grid), dim3(512), SMEM_B, stream, a);
    return mm_check_launch("gemm_cfg3_kernel");
}

#ifdef MM_GEMM_TIMING
extern "C" int mm_debug_cfg3_stamps(unsigned long long* host_dst, int n) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_cfg3_stamps), sizeof(unsigned long long) * n);
}
#endif
