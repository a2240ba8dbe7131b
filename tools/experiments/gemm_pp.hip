// The wide bf16 GEMMs with their wave groups OUT of lock-step (round 5): the logits GEMM of the decode loop with the fused-sampling emission.
//
// Why: every kernel of this family so far (gemm_wide.hip, gemm_cfg.hip, gemm_big.hip) runs all eight waves of a workgroup through the same phase at the same
// time -- barrier, LDS-DMA issue, fragment reads, MFMAs -- so both waves of a SIMD wait on the same `vmcnt` / `lgkmcnt` / barrier together and the matrix pipe
// idles meanwhile (SQ counters of round 4: pipe busy 0.31, waves parked 0.37, issue-stalled 0.33; ~1000 cycles lost per barrier interval whatever the tile).
// Two forms of breaking that, one kernel template:
//
//   GROUPS == 2 ("ping-pong"): a 512-thread workgroup = two groups of four waves (group g owns tokens 128 g .. 128 g + 127 of the 256-token tile; wave w of a group
//     and wave w of the other share a SIMD).  The k-loop is cut into 32-deep PHASES, each a LOAD segment (12 ds_read_b128 of this phase's fragments + the
//     LDS-DMA issue of a phase further ahead + the counted vmcnt for the next phase) and a COMPUTE segment (32 MFMAs on registers only), separated by
//     workgroup barriers; group 1 executes ONE extra barrier before its first phase, so at every moment one wave of each SIMD computes while the other loads
//     (the 8-phase schedule of the CDNA4 guide, section 5: T3 + T4 + T5).
//   GROUPS == 1: a 256-thread workgroup owns a 128-token x 256-row tile by itself, TWO workgroups per CU with their own LDS rings; the second half of the
//     grid starts half a tile late, so one workgroup's emission (VALU only) runs under the other's k-loop.
//
// Operand staging: ring of NS slots, one 32-deep phase each (token rows and weight rows as 64-byte LDS rows, 16-byte chunks XOR-swizzled by (-(row >> 2)) & 3:
// conflict-free ds_read_b128, the swizzle applied to the LDS-DMA's source address -- the layout of gemm_cfg.hip); the DMA runs NS - 1 phases ahead across tile
// boundaries (persistent workgroups).  Accumulation order per output element: k ascending in chunks of 32, one MFMA each -- the order of every GEMM kernel of
// the family: bit-identical results.  The emission is gemm_wide_fused_kernel's (common.h tile_softmax_stats / fs_slot_index formats), per wave group.
#include "common.h"
#include "muse_hip_internal.h"

#ifdef MM_GEMM_TIMING      // tools/pp_timing (gemm_harness `stamps`): s_memtime stamps of workgroup 0, waves 0 and 4, along their tiles
__device__ unsigned long long g_pp_stamps[2][2048];
__device__ unsigned long long g_pp_tile_stamps[512][128];      // every workgroup's wave 0: {hw id, start, then per tile: k-loop end, emission end}
#define PP_STAMP() { if (ts_on && ts_i < 2048) g_pp_stamps[ts_g][ts_i++] = __builtin_readcyclecounter(); }
#define PP_TSTAMP() { if (t == 0 && tl_i < 128 && blockIdx.x < 512) g_pp_tile_stamps[blockIdx.x][tl_i++] = __builtin_readcyclecounter(); }
#else
#define PP_STAMP()
#define PP_TSTAMP()
#endif

namespace {

#define PP_VMCNT_IMM(n_) (0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14))
__device__ __forceinline__ void pp_wait_vmcnt(int n) {      // wave-uniform n, rounded DOWN to a handful of immediates (waiting for more is always safe)
    if (n >= 48) __builtin_amdgcn_s_waitcnt(PP_VMCNT_IMM(48));
    else if (n >= 32) __builtin_amdgcn_s_waitcnt(PP_VMCNT_IMM(32));
    else if (n >= 24) __builtin_amdgcn_s_waitcnt(PP_VMCNT_IMM(24));
    else if (n >= 16) __builtin_amdgcn_s_waitcnt(PP_VMCNT_IMM(16));
    else if (n >= 12) __builtin_amdgcn_s_waitcnt(PP_VMCNT_IMM(12));
    else if (n >= 8) __builtin_amdgcn_s_waitcnt(PP_VMCNT_IMM(8));
    else if (n >= 6) __builtin_amdgcn_s_waitcnt(PP_VMCNT_IMM(6));
    else if (n >= 4) __builtin_amdgcn_s_waitcnt(PP_VMCNT_IMM(4));
    else __builtin_amdgcn_s_waitcnt(PP_VMCNT_IMM(0));
}
#define PP_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

__device__ __forceinline__ float pp_max2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float pp_max4(float a, float b, float c, float d) {      // (MFMA results: no canonicalising v_max x, x needed)
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(r), "v"(d));
    return r;
}

constexpr int PP_XCH = 128 * 4 * 4 + 16 * 128 * 8;      // exchange arrays of one wave group's emission: keep masks [4 quarters][128 tokens] + (ml, pl) [16 lane groups][128 tokens] = 18 KiB

template <int GROUPS, int NS>
struct PPGeo {
    static constexpr int NW = 4 * GROUPS, TM = 128 * GROUPS, XS_B = TM * 64, WS_B = 256 * 64, SLOT = XS_B + WS_B;
    static constexpr int NXI = TM / 16, NI = NXI + 16, NPW = NI / NW, NXW = NXI / NW;      // LDS-DMA instructions per phase: token blocks, all, per wave, token blocks per wave
    static constexpr int DIST = NS - 1;
    static constexpr int THR_OFF = NS * SLOT;                  // the tile's per-token bounds (1 KiB)
    static constexpr int XCH_OFF = THR_OFF + 1024;             // GROUPS == 2: the groups' exchange areas behind the ring (group 0 of a 4-slot ring: the free ring slot)
    static constexpr int SMEM = GROUPS == 1 ? THR_OFF + 1024 : (NS == 4 ? XCH_OFF + PP_XCH : XCH_OFF + 2 * PP_XCH);
    static_assert(NXI % NW == 0 && NI % NW == 0, "whole LDS-DMA instructions per wave");
    static_assert(SMEM <= (GROUPS == 1 ? 80 : 160) * 1024, "LDS budget");
    static_assert(GROUPS == 2 || PP_XCH <= SLOT, "one ring slot holds the exchange arrays");
};

template <int GROUPS, int NS, bool F16>
__global__ __launch_bounds__(256 * GROUPS, 2) void gemm_pp_fused_kernel(const GemmArgs p, const int prio, const int delay_cycles, const int abl) {
    // abl (tools only, MM_PP_ABL): 1 = no emission, 2 = no LDS-DMA after the prologue, 4 = no MFMAs, 8 = no fragment reads, 16 = no workgroup barriers inside the k-loop
    // (round 6: only meaningful with 2 | 8 -- nothing is shared then), 32 = no counted waits inside the k-loop (with 2: nothing is in flight)
    using G = PPGeo<GROUPS, NS>;
    constexpr int NW = G::NW, TM = G::TM, XS_B = G::XS_B, SLOT = G::SLOT, NPW = G::NPW, NXW = G::NXW, DIST = G::DIST;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wid >> 2, wn = wid & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int total = p.tiles_m * p.tiles_n, GR = gridDim.x;
    const int NP = p.K / 32;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    float* lthr = reinterpret_cast<float*>(smem + G::THR_OFF);
    const __amdgpu_buffer_rsrc_t thr_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.fs_thr), 0, (unsigned)p.M * 4u, 0x00020000);
    // LDS-DMA: one instruction = 16 rows x 64 bytes; lane l fetches row l >> 2, logical chunk (l & 3) ^ swizzle(row) into physical chunk l & 3.  Wave w issues token
    // blocks w, w + NW, .. and weight blocks w, w + NW, .. of every phase (NPW instructions); the row of a block goes into the per-lane offset (range-checked
    // against the descriptor: rows beyond M read as zero), the k position into the scalar offset
    const int lrow = lane >> 2;
    const int lchunk = ((lane & 3) ^ ((-(lrow >> 2)) & 3)) * 16;
    const int voff_x = (16 * wid + lrow) * p.ldx * 2 + lchunk;
    const int voff_w = (16 * wid + lrow) * p.ldw * 2 + lchunk;
    const int xstep = 16 * NW * p.ldx * 2, wstep = 16 * NW * p.ldw * 2;
    // fragment reads: lane (fr, fg) reads row fr of a 16-row block, logical chunk fg
    const int lane_off = fr * 64 + ((fg ^ ((-(fr >> 2)) & 3)) << 4);
    int vb = blockIdx.x;
    if (vb >= total) return;
#ifdef MM_GEMM_TIMING
    const bool ts_on = blockIdx.x == 0 && (t == 0 || t == 256);
    const int ts_g = t >> 8;
    int ts_i = 0, tl_i = 2;
    if (t == 0 && blockIdx.x < 512)      // HW_ID (wave slot, SIMD, CU, SH, SE) | XCC_ID << 32
        g_pp_tile_stamps[blockIdx.x][0] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32);
#endif
    __amdgpu_buffer_rsrc_t rx, rw;
    int tile_m, tile_n;
#define PP_TILE_SETUP(vb_)                                                                                                             \
    {                                                                                                                                  \
        xcd_grouped_tile(vb_, p.tiles_m, p.tiles_n, 8 * (3 - GROUPS), tile_m, tile_n);                                                 \
        const int left_ = p.M - tile_m * TM;                                                                                           \
        rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X + (size_t)tile_m * TM * p.ldx), 0,                              \
                                               (unsigned)(left_ < TM ? left_ : TM) * (unsigned)p.ldx * 2u, 0x00020000);                \
        rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W + (size_t)tile_n * 256 * p.ldw), 0, 256u * (unsigned)p.ldw * 2u, 0x00020000); \
    }
#define PP_ISSUE(q_, slot_)                                                                                                            \
    {                                                                                                                                  \
        unsigned char* sb_ = smem + (slot_) * SLOT + wid * 1024;                                                                       \
        _Pragma("unroll") for (int i = 0; i < NXW; ++i)                                                                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(sb_ + i * (NW * 1024)), 16, voff_x + i * xstep, (q_) * 64, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < NPW - NXW; ++i)                                                                          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(sb_ + XS_B + i * (NW * 1024)), 16, voff_w + i * wstep, (q_) * 64, 0, 0); \
    }
    if (GROUPS == 1 && delay_cycles > 0) {
        // One of the two workgroups of a CU starts late, so that its emissions (VALU + stores) fall under the other one's k-loops (matrix pipe).  Which one: the
        // workgroup whose waves sit in an ODD wave slot of their SIMD (HW_ID.wave_id: a CU's first workgroup gets slot 0, the second slot 1) -- observed placement,
        // used for speed only; prio >= 4 selects the older guess (second half of the grid) for A/B
        const bool late = prio >= 4 ? (int)blockIdx.x >= (GR >> 1) : (__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 1) != 0;
        if (late) {
            const long long t0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - t0 < delay_cycles) __builtin_amdgcn_s_sleep(32);
        }
    }
#ifdef MM_GEMM_TIMING
    if (t == 0 && blockIdx.x < 512) g_pp_tile_stamps[blockIdx.x][1] = __builtin_readcyclecounter();
#endif
    PP_TILE_SETUP(vb);
#pragma unroll
    for (int q = 0; q < DIST; ++q) PP_ISSUE(q, q);
    pp_wait_vmcnt((DIST - 1) * NPW);
    __builtin_amdgcn_s_barrier();                    // phase 0 of the first tile has landed for everybody
    if (GROUPS == 2 && grp == 1) __builtin_amdgcn_s_barrier();      // the skew: group 1 runs one barrier interval behind group 0 from here on
    if (prio == 2 && GROUPS == 2 && grp == 1) __builtin_amdgcn_s_setprio(1);      // static priority for the later-dispatched half (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    int rs = 0, ws = DIST % NS;                      // ring slot the coming phase reads / the next DMA writes
    int pending = 0;                                 // VMEM stores of the previous tile's emission (issued between this tile's prefetched phases and its phase DIST)
    f32x4_t acc[4][8];                               // [weight fragment a][token fragment b]: lane (fr, fg) holds weight rows 16 a + 4 fg .. + 3 for token 16 b + fr
    while (true) {
        const int cur_m = tile_m, cur_n = tile_n;
        const bool has_next = vb + GR < total;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        PP_STAMP();      // tile start
        if (wid == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(thr_rs, (lds_ptr_t)(smem + G::THR_OFF), 16, lane * 16, cur_m * TM * 4, 0, 0);
        for (int ph = 0; ph < NP; ++ph) {
            // ---------------- LOAD segment of phase ph: its fragments -> registers; the DMA of phase ph + DIST; phase ph + 1 landed (this wave's share)
            const unsigned char* xs = smem + rs * SLOT + grp * 8192 + lane_off;
            const unsigned char* wsr = smem + rs * SLOT + XS_B + wn * 4096 + lane_off;
            u32x4_t wf[4], xf[8];
            if (!(abl & 8)) {
#pragma unroll
                for (int a = 0; a < 4; ++a) wf[a] = *reinterpret_cast<const u32x4_t*>(wsr + a * 1024);
#pragma unroll
                for (int b = 0; b < 8; ++b) xf[b] = *reinterpret_cast<const u32x4_t*>(xs + b * 1024);
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a) wf[a] = u32x4_t{(unsigned)lane, 0x3c003c00u, (unsigned)ph, 0u};
#pragma unroll
                for (int b = 0; b < 8; ++b) xf[b] = u32x4_t{0x3c003c00u, (unsigned)lane, 0u, (unsigned)ph};
            }
            if (ph + DIST < NP) {
                if (!(abl & 2)) PP_ISSUE(ph + DIST, ws);
            } else if (has_next) {
                if (ph + DIST == NP) PP_TILE_SETUP(vb + GR);
                if (!(abl & 2)) PP_ISSUE(ph + DIST - NP, ws);
            }
            ws = ws + 1 == NS ? 0 : ws + 1;
            // "this wave's share of phase ph + 1 has landed": in flight behind it are the phases ph + 2 .. ph + DIST that were really issued (none beyond the last
            // tile's last phase) and -- in a tile's first DIST - 1 phases -- the previous tile's emission stores (VMEM retires in order; counting too few is safe)
            const int ahead = has_next ? DIST - 1 : min(DIST - 1, max(NP - 2 - ph, 0));
            const bool plain_wait = ahead == DIST - 1 && (ph >= DIST - 1 || pending == 0);
#define PP_WAIT_NEXT() { if (plain_wait) __builtin_amdgcn_s_waitcnt(PP_VMCNT_IMM((DIST - 1) * NPW)); else pp_wait_vmcnt(ahead * NPW + (ph < DIST - 1 ? pending : 0)); }
            if constexpr (GROUPS == 2) {
                // group 1's share is read by group 0 in the very next interval: it waits here; group 0's own next LOAD comes behind its COMPUTE segment: it waits there
                if (grp == 1 && !(abl & 32)) PP_WAIT_NEXT();
                if (!(abl & 32)) PP_LGKM0();             // this group is done with the slot once the barrier is passed
                if (!(abl & 16)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (prio == 1) __builtin_amdgcn_s_setprio(1);
            }
            // ---------------- COMPUTE segment: 32 MFMAs (ping-pong form: on registers only; single group: the compiler interleaves them with the fragment reads)
            if (!(abl & 4)) {
#pragma unroll
                for (int b = 0; b < 8; ++b)
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[a][b] = mfma16t<F16>(wf[a], xf[b], acc[a][b]);
            } else {
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[b & 3][b][0] += __uint_as_float(wf[b & 3][0] ^ xf[b][1]);      // (keeps the fragment reads alive)
            }
            if constexpr (GROUPS == 2) {
                if (prio == 1) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (grp == 0 && !(abl & 32)) PP_WAIT_NEXT();
            } else {
                if (!(abl & 32)) { PP_WAIT_NEXT(); PP_LGKM0(); }
            }
#undef PP_WAIT_NEXT
            if (!(abl & 16)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            rs = rs + 1 == NS ? 0 : rs + 1;
#ifdef MM_GEMM_TIMING
            if (vb == (int)blockIdx.x + 3 * GR) PP_STAMP();      // the fourth tile: one stamp per phase
#endif
        }
        PP_STAMP();      // k-loop end
        PP_TSTAMP();
        if constexpr (F16) {      // undo the power-of-two scale of the packed weight terms (exact) before statistics and candidates
            const float al = p.alpha;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[a][b] *= al;
        }
        // ---------------- emission from the accumulators (gemm_wide_fused_kernel's, per wave group: token of fragment block b = 16 b + fr of the group's 128)
        // exchange area: behind the ring -- except a single-group workgroup (the ring slot of the phase just finished: free until the next tile's first DMA
        // issue, which the barrier at the end of the emission guards) and group 0 of a 4-slot ring (same slot, same argument: group 0 itself issues that DMA)
        unsigned char* xch;
        {
            const int fs_ = rs == 0 ? NS - 1 : rs - 1;      // slot of the phase just finished
            if (GROUPS == 1) xch = smem + fs_ * SLOT;
            else if (NS == 4) xch = grp == 0 ? smem + fs_ * SLOT : smem + G::XCH_OFF;
            else xch = smem + G::XCH_OFF + grp * PP_XCH;
        }
        int le_ = lane;
        asm volatile("" : "+v"(le_));            // opaque copy: keeps the emission's address arithmetic out of the k-loop's live ranges
        const int FR_ = le_ & 15, FG_ = le_ >> 4;
        const int m0t = cur_m * TM + grp * 128;
        const float* gthr = lthr + grp * 128;
        uint32_t* xmask = reinterpret_cast<uint32_t*>(xch);                          // [4 quarters][128 tokens]: the quarter's 32 keep bits of a token (bit 8 a + 2 f + h)
        float2* xml = reinterpret_cast<float2*>(xch + 128 * 16);                     // [16 lane groups][128 tokens] (ml, pl)
        int nstore = 0;
        uint32_t mq[8];
        if (abl & 1) {      // (ablation: keep the accumulators alive, emit nothing)
            float sink = 0.f;
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int a = 0; a < 4; ++a) sink += (acc[a][b][0] + acc[a][b][1]) + (acc[a][b][2] + acc[a][b][3]);
            if (sink == 1.2345e-30f) xmask[t & 127] = 1u;
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (abl & 1) { mq[b] = 0u; continue; }
            const int tokl = b * 16 + FR_;
            const bool valid = m0t + tokl < p.M;
            const float thr = gthr[tokl];
            float g2[4][2], gm[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                g2[a][0] = pp_max2(acc[a][b][0], acc[a][b][1]);
                g2[a][1] = pp_max2(acc[a][b][2], acc[a][b][3]);
                gm[a] = pp_max2(g2[a][0], g2[a][1]);
            }
            const float ml = pp_max4(gm[0], gm[1], gm[2], gm[3]);
            float gs[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) gs[a] = fs_exp_sum4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3], ml);
            const float pl = (gs[0] + gs[1]) + (gs[2] + gs[3]);
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
            if (FG_ == 0) xmask[wn * 128 + tokl] = m32;
            xml[(wn * 4 + FG_) * 128 + tokl] = make_float2(ml, pl);
        }
        PP_LGKM0();
        __builtin_amdgcn_s_barrier();      // all 16 lane groups of every token of this wave group have published their masks and (ml, pl)
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP();      // statistics + masks published, exchange barrier passed
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint32_t mine = (mq[b] >> (2 * FG_)) & 0x03030303u;      // bit 8 a + h: this lane's granule (a, h) of block b
            if (__ballot(mine != 0u) == 0ull) continue;                   // wave-uniform
            const int tokl = b * 16 + FR_;
            int base = 0;                                                 // kept granules of the quarters in front of this one
            if (wn > 0) base += __popc(xmask[tokl]);
            if (wn > 1) base += __popc(xmask[128 + tokl]);
            if (wn > 2) base += __popc(xmask[2 * 128 + tokl]);
            float2* slot = reinterpret_cast<float2*>(p.fs_cand + ((size_t)(m0t + tokl) * p.tiles_n + cur_n) * FS_SLOT) + base;
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
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {   // one record per (token, piece): this wave combines the tokens of its group's blocks wn and wn + 4
            const int tokl = (hb * 4 + wn) * 16 + FR_;
            const int tok = m0t + tokl;
            const bool w_ = FG_ == 0 && tok < p.M && !(abl & 1);
            if (__ballot(w_) != 0ull) {
                if (w_) {
                    const float2* gq = xml + tokl;              // tile_combine16 (common.h), streamed from LDS in two sweeps
                    float M_ = -INFINITY;
#pragma unroll
                    for (int i = 0; i < 16; ++i) M_ = fmaxf(M_, gq[i * 128].x);
                    float wq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float2 u0 = gq[(4 * q) * 128], u1 = gq[(4 * q + 1) * 128], u2 = gq[(4 * q + 2) * 128], u3 = gq[(4 * q + 3) * 128];
                        wq[q] = (u0.y * __expf(u0.x - M_) + u1.y * __expf(u1.x - M_)) + (u2.y * __expf(u2.x - M_) + u3.y * __expf(u3.x - M_));
                    }
                    const float E_ = (wq[0] + wq[1]) + (wq[2] + wq[3]);
                    float4* rec = p.fs_stats + ((size_t)tok * p.tiles_n + cur_n) * FS_REC;
                    rec[0] = make_float4(M_, E_, 0.f, 0.f);
                    rec[1] = make_float4(__uint_as_float(xmask[tokl]), __uint_as_float(xmask[128 + tokl]), __uint_as_float(xmask[2 * 128 + tokl]), __uint_as_float(xmask[3 * 128 + tokl]));
                }
                nstore += 2;
            }
        }
        PP_STAMP();      // candidate stores + records issued
        PP_TSTAMP();
        pending = __builtin_amdgcn_readfirstlane(nstore);      // (wave-uniform by construction: every increment sits under a ballot)
        vb += GR;
        PP_LGKM0();
        __builtin_amdgcn_s_barrier();      // the exchange area is read out (single group / 4-slot ring: before the next tile's first DMA issue lands in it)
        __builtin_amdgcn_sched_barrier(0);
        if (vb >= total) break;
    }
    if (GROUPS == 2 && grp == 0) __builtin_amdgcn_s_barrier();      // balances group 1's extra barrier
#undef PP_ISSUE
#undef PP_TILE_SETUP
}

int pp_variant() {      // tools / A-B only: MM_PP = 0 (off) | GROUPS * 100 + NS * 10 + prio, e.g. 230 = ping-pong, 3-slot ring; 131 = two workgroups per CU ...
    static int v = -1;
    if (v < 0) { const char* e = getenv("MM_PP"); v = e ? atoi(e) : 0; }
    return v;
}
int pp_abl() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MM_PP_ABL"); v = e ? atoi(e) : 0; }
    return v;
}
int pp_delay() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MM_PP_DELAY"); v = e ? atoi(e) : 12000; }
    return v;
}

template <int GROUPS, int NS, bool F16>
int launch_pp_fused(GemmArgs a, int prio, hipStream_t stream) {
    using G = PPGeo<GROUPS, NS>;
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pp_fused_kernel<GROUPS, NS, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
        if (e != hipSuccess) return mm_set_hip_error(e, "gemm_pp_fused hipFuncSetAttribute");
        attr_set = true;
    }
    a.tiles_m = (a.M + G::TM - 1) / G::TM;
    a.tiles_n = a.N / 256;
    const int total = a.tiles_m * a.tiles_n, cap = 256 * (3 - GROUPS);
    hipLaunchKernelGGL((gemm_pp_fused_kernel<GROUPS, NS, F16>), dim3(total < cap ? total : cap), dim3(256 * GROUPS), G::SMEM, stream, a, prio, pp_delay(), pp_abl());
    return mm_check_launch("gemm_pp_fused_kernel");
}

}  // namespace

// same eligibility as the 256 x 256 lock-step kernel (gemm_wide.hip mm_gemm_wide_fused_eligible) plus K % 32 == 0 and at least NS phases
bool mm_gemm_pp_fused_selected(const GemmArgs& a) { return pp_variant() != 0 && !a.f16 && (a.K % 32) == 0 && a.K >= 128; }

int mm_gemm_pp_fused_launch(GemmArgs a, hipStream_t stream) {
    const int v = pp_variant(), prio = v % 10;
    switch (v / 10) {
        case 23: return launch_pp_fused<2, 3, false>(a, prio, stream);
        case 24: return launch_pp_fused<2, 4, false>(a, prio, stream);
        case 13: return launch_pp_fused<1, 3, false>(a, prio, stream);
        default: return mm_set_error(MM_ERR_UNSUPPORTED, "gemm_pp: unknown MM_PP variant");
    }
}

#ifdef MM_GEMM_TIMING
extern "C" int mm_debug_pp_tile_stamps(unsigned long long* host_dst) {      // [512][128]
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_pp_tile_stamps), sizeof(unsigned long long) * 512 * 128);
}
extern "C" int mm_debug_pp_stamps(unsigned long long* host_dst, int n) {      // [2][n <= 2048]
    hipError_t e = hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_pp_stamps), sizeof(unsigned long long) * 2 * 2048);
    (void)n;
    return (int)e;
}
#endif
