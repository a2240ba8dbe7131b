// Sampling without the logits round trip (muse_maskgit_pytorch.py:576-609; SURVEY.md 8d "fused floor = 0 bytes of logits to HBM").
//
// The guidance-logits GEMM (gemm_cfg.hip) does not write its [R][V] fp32 output any more.  While a 256-column tile of a token row is
// on its way out of the accumulators (4 consecutive values per lane) the epilogue emits, per (row, tile):
//     stats  {tile max, sum exp(x - tile max), -, -} + the 128-bit mask of the kept granules                32 B
//     cand   the 2 values of every 2-column granule whose larger value reaches thr_lo[row], compacted        8 B per kept granule (~23 % of them)
// thr_lo[row] is a LOWER bound estimate of the row's k-th largest logit, known BEFORE the GEMM runs: a row's logits over the vocabulary
// are <e, w_v> for the row's (guidance-combined) embedding e, so their mean and variance are <e, mean_w> and e' Cov_w e with the weight
// statistics packed once per model (k_fused_threshold: one small MFMA GEMM against Cov_w); thr_lo = mean + (z_k - margin) sigma.
// A finishing kernel (sample_fused_kernel) then works on the ~9 k values >= thr_lo of a row instead of 65536 logits: exact k-th largest
// (value-linear histogram in LDS + rank counting inside one bin), Gumbel argmax over the entries >= it, confidence
// 1 - exp(x_pred - max) / sum from the tile statistics combined in a fixed order (deterministic).
// The candidate set is a superset of the kept set iff at least k values passed thr_lo; the finishing kernel CHECKS that (and its list
// capacity) per row and raises a device flag otherwise -- the caller then repeats the generate on the logits path (sampling.hip), so
// the result never depends on the estimate, only the speed does.  Same ids as sample_kernel on the same logits (tests).
#include <math.h>
#include <string.h>

#include "common.h"
#include "muse_hip_internal.h"

#ifndef MM_EXP
#define MM_EXP 0
#endif
namespace {

__device__ __forceinline__ uint32_t fkey(float f) {      // order-preserving float -> uint32 (as in sampling.hip)
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Philox2x32-10, philox_uniform2, gumbel_of / gumbel_of_unit: common.h (one definition for both sampling paths)
__device__ __forceinline__ float noise_gumbel(const FusedSampleArgs& p, long pos_flat, int idx) {
    if (p.noise_kind == MM_NOISE_GUMBEL) return p.noise[(size_t)pos_flat * p.noise_ld + idx];
    if (p.noise_kind == MM_NOISE_UNIFORM) return gumbel_of(p.noise[(size_t)pos_flat * p.noise_ld + idx]);
    if (p.noise_kind == MM_NOISE_PHILOX) {
        float u[2];
        philox_uniform2(p.seed, p.row_offset + (uint64_t)pos_flat, p.step, (uint32_t)(idx >> 1), u);
        return gumbel_of_unit((idx & 1) ? u[1] : u[0]);
    }
    return 0.f;
}

// ---- thr_lo[r] = mean_r + z * sigma_r with mean_r = <e_r, wmean>, sigma_r^2 = e_r' Cov e_r, e_r = null_r + (cond_r - null_r) * s (what the GEMM
//      multiplies, mmp.py:254, by linearity).  Three launches: e (bf16) and mean per row, T = E Cov on the bf16 MFMA GEMM, sigma from <e, T>.
__global__ __launch_bounds__(256) void fused_combine_kernel(const bf16_t* __restrict__ ec, const bf16_t* __restrict__ en, long ld, int R, int D, float s,
                                                            const float* __restrict__ wmean, bf16_t* __restrict__ ebf, float* __restrict__ mu) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    float m = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float a = bf16_to_f32(ec[(size_t)row * ld + c]), b = bf16_to_f32(en[(size_t)row * ld + c]);
        const bf16_t eb = f32_to_bf16(b + (a - b) * s);
        ebf[(size_t)row * D + c] = eb;
        m += bf16_to_f32(eb) * wmean[c];
    }
    m = wave_sum(m);
    if (lane == 0) mu[row] = m;
}
__global__ __launch_bounds__(256) void fused_sigma_kernel(const bf16_t* __restrict__ ebf, const float* __restrict__ tq, const float* __restrict__ mu, int R, int D,
                                                          float z, float* __restrict__ thr) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    float q = 0.f;
    for (int c = lane; c < D; c += 64) q += bf16_to_f32(ebf[(size_t)row * D + c]) * tq[(size_t)row * D + c];
    q = wave_sum(q);
    if (lane == 0) thr[row] = mu[row] + z * sqrtf(fmaxf(q, 0.f));
}

// ---- the GEMM epilogue's emission, as a stand-alone kernel over materialised logits (tests; shapes the 256-column GEMM does not take)
__global__ __launch_bounds__(256) void fused_emit_kernel(const float* __restrict__ logits, long ld, int R, int V, const float* __restrict__ thr,
                                                         float4* __restrict__ stats, float4* __restrict__ cand) {
    const int NT = V / 256;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long piece = (long)blockIdx.x * 4 + w;              // (row, tile)
    if (piece >= (long)R * NT) return;
    const int row = (int)(piece / NT), tile = (int)(piece - (long)row * NT);
    const float4 x4 = *reinterpret_cast<const float4*>(logits + (size_t)row * ld + tile * 256 + lane * 4);
    fused_emit_piece(x4, row, tile, NT, lane, thr[row], stats, cand);
}

// ---- finishing kernel: 512-thread workgroups, two per CU (78 KiB of LDS each), one row at a time
constexpr int FT = 512, FW = FT / 64;
constexpr int WSL = 1408;                       // per-wave slice of the candidate list; a wave gathers 1/8 of the tiles: ~1150 +- 35 values at V = 65536
constexpr int LIST_CAP = FW * WSL;              // 11264 candidates of one row held in LDS as (fp32 value, u16 index): 66 KiB
constexpr int NBF = 1024;                       // value-linear histogram bins over [thr_lo, row max]
constexpr int GU = 16;                          // tiles per wave whose candidate loads are in flight together (2 rounds per row at V = 65536; 8: same time)
constexpr int CANDF = 1024;                     // exact-select capacity (members of the bin that holds the k-th largest)

struct FusedShared {
    float xs[LIST_CAP];
    uint16_t cols[LIST_CAP];
    uint32_t hist[NBF];                         // TRANSPOSED (hslotf): lane l's 16 consecutive bins form a conflict-free column
    uint32_t cand[CANDF];                       // members of the bin that holds the k-th largest value: order-preserving keys ...
    uint16_t candc[CANDF];                      // ... and their columns (the members >= the k-th largest belong to the kept set)
    float redf[FW], redx[FW];
    int redi[FW];
    int wcnt[FW];
    int ncand;
    uint32_t thr_key;
};
__device__ __forceinline__ int hslotf(int b) { return ((b & 15) << 6) | (b >> 4); }

// both values of one list entry per lane (columns col, col + 1): those >= lo are appended to the wave's slice in lane order (ballot prefix, no atomics; a
// dead lane carries NaN, which passes no comparison).  A slice that would overflow is not written at all: its count still grows, so the row fails the
// `fits` check below and goes to the fallback.
__device__ __forceinline__ void fs_append2(const float2 x, int col, float lo, float* myx, uint16_t* myc, int& wcount) {
    const bool k0 = x.x >= lo, k1 = x.y >= lo;
    const unsigned long long b0 = __ballot(k0), b1 = __ballot(k1);
    const int nw = __popcll(b0) + __popcll(b1);
    if (wcount + nw <= WSL) {                                     // wave-uniform
        int idx = wcount + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)) +
                  (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
        if (k0) { myx[idx] = x.x; myc[idx] = (uint16_t)col; }
        idx += k0 ? 1 : 0;
        if (k1) { myx[idx] = x.y; myc[idx] = (uint16_t)(col + 1); }
    }
    wcount += nw;
}
// per-lane select by a wave-uniform 64-bit lane mask held in scalar registers: bit `lane` of mask ? if_set : if_clear (one v_cndmask)
__device__ __forceinline__ int fs_sel_mask(int if_clear, int if_set, unsigned long long mask) {
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}

__global__ __launch_bounds__(FT, 4) void sample_fused_kernel(const FusedSampleArgs p_in) {
    FusedSampleArgs p = p_in;
    if (p.seed_dev) { p.seed = p.seed_dev[0]; p.row_offset = p.seed_dev[1] * (uint64_t)p.row_mul; }      // (wave-uniform scalar loads: the keys of a replayed graph)
    extern __shared__ __attribute__((aligned(16))) unsigned char fs_raw[];
    FusedShared& S = *reinterpret_cast<FusedShared*>(fs_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the gather's piece walk, its addresses and its branches are wave-uniform
    const int NT = p.V / 256;
    // lane u < 32 of wave w holds the keep mask of piece w + 8 u (the wave's u-th piece) of the current row: read once per row straight from the
    // records, one row ahead, and handed to the gather through v_readlane (scalar registers, no LDS round trip)
    auto load_masks = [&](int r) -> uint4 {
        uint4 m = make_uint4(0u, 0u, 0u, 0u);
        const int tp = wid + lane * FW;
        if (r < p.R && lane < 32 && tp < NT) {
            const float4 mk = p.stats[((size_t)r * NT + tp) * FS_REC + 1];
            m = make_uint4(__float_as_uint(mk.x), __float_as_uint(mk.y), __float_as_uint(mk.z), __float_as_uint(mk.w));
        }
        return m;
    };
    uint4 mkv = load_masks(blockIdx.x);
    // thread t < V / 256 holds piece t's softmax statistics {max, sum exp(x - max)} of the current row, read one row ahead as well
    auto load_stats = [&](int r) -> float2 {
        float2 st = make_float2(-INFINITY, 0.f);
        if (r < p.R && tid < NT) { const float4 q = p.stats[((size_t)r * NT + tid) * FS_REC]; st = make_float2(q.x, q.y); }
        return st;
    };
    float2 stv = load_stats(blockIdx.x);
    for (int row = blockIdx.x; row < p.R; row += gridDim.x) {
        const long pos_flat = p.rows ? (long)p.rows[row] : (long)row;
        // ---- tile statistics -> row max, softmax denominator
        const float tmax = stv.x, tsum = stv.y;
        const float wm = wave_max(tmax);
        if (lane == 0) S.redf[wid] = wm;
        if (tid == 0) S.ncand = 0;
        for (int i = tid; i < NBF; i += FT) S.hist[i] = 0;
        __syncthreads();
        float M = S.redf[0];
#pragma unroll
        for (int w2 = 1; w2 < FW; ++w2) M = fmaxf(M, S.redf[w2]);
        // softmax denominator: sum_t tsum_t * exp(tmax_t - M), combined in a fixed order (deterministic)
        const float term = wave_sum((tid < NT) ? tsum * expf(tmax - M) : 0.f);
        if (lane == 0) S.redx[wid] = term;
        // ---- gather: wave w takes tiles w, w + 8, ...: the kept lanes' float4s (coalesced reads, issued GU tiles at a time), every value >= the
        //      bound is appended to THIS WAVE's slice of the LDS list (running count + ballot prefix: no atomics) and binned
        const float lo = p.thr[row];
        const float span = M - lo;
        const bool fast = (span >= 1e-30f) && (span < 3.0e38f);
        const float inv_w = fast ? (float)NBF / span : 0.f;
        float* myx = S.xs + wid * WSL;
        uint16_t* myc = S.cols + wid * WSL;
        int wcount = 0;                                           // wave-uniform
        // Dense gather: a piece's slot IS a compacted list (cnt = popcount of its keep mask granules, in column order), so lane e reads granule e -- one
        // coalesced 8-byte load per lane, no per-lane position arithmetic -- and only the granule NUMBER of entry e (the position of the e-th set bit of
        // the 128-bit mask) has to be found.  The mask is wave-uniform (scalar registers): the lane that owns bit J knows its rank (v_mbcnt), and one
        // ds_permute per mask half sends J to the lane of that rank (the lanes that own a cleared bit are routed behind the list, so each permute is a
        // bijection: no destination is written twice).  ~45 VALU instructions per piece against ~100 of the per-column walk it replaces (round 4: this
        // loop is VALU-issue-bound).
        for (int t0 = wid, pl0 = 0; t0 < NT; t0 += FW * GU, pl0 += GU) {      // pl0 + u: the lane of mkv that holds piece t0 + u FW
            float2 v[GU];
            uint32_t gr[GU / 2];                                  // 16 bits per piece: granule number of list entry `lane` | (pass-2 sender) << 8 for the tail round
            uint32_t big = 0;                                     // wave-uniform: pieces with more than 64 kept granules
            const float dead = __uint_as_float(0x7FC00000u);
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int t_ = t0 + u * FW;
                v[u] = make_float2(dead, dead);
                if ((u & 1) == 0) gr[u / 2] = 0;
                if (t_ < NT) {                                    // wave-uniform
                    const uint32_t m0 = __builtin_amdgcn_readlane(mkv.x, pl0 + u), m1 = __builtin_amdgcn_readlane(mkv.y, pl0 + u);
                    const uint32_t m2 = __builtin_amdgcn_readlane(mkv.z, pl0 + u), m3 = __builtin_amdgcn_readlane(mkv.w, pl0 + u);
                    const int c01 = __popc(m0) + __popc(m1), cnt = c01 + __popc(m2) + __popc(m3);
                    if (lane < cnt) v[u] = reinterpret_cast<const float2*>(p.cand + ((size_t)row * NT + t_) * FS_SLOT)[lane];
                    // pass 1: lane l owns granule l (bit l of m1:m0), list position = its rank; pass 2: granule 64 + l (bit l of m3:m2), list position c01 + rank
                    // taken mod 64 -- a position >= 64 (more than 64 kept granules in the piece) wraps to lane position - 64 < c01, where the tail round
                    // of the second loop picks it up.  Owners of a cleared bit are routed behind the owners: each permute is a bijection of the lanes.
                    const int ra = (int)__builtin_amdgcn_mbcnt_hi(m1, __builtin_amdgcn_mbcnt_lo(m0, 0u));
                    const int rb = (int)__builtin_amdgcn_mbcnt_hi(m3, __builtin_amdgcn_mbcnt_lo(m2, 0u));
                    const int da = fs_sel_mask(c01 + lane - ra, ra, ((unsigned long long)m1 << 32) | m0);
                    const int db = fs_sel_mask(cnt + lane - rb, c01 + rb, ((unsigned long long)m3 << 32) | m2) & 63;
                    const int ja = __builtin_amdgcn_ds_permute(da << 2, lane), jb = __builtin_amdgcn_ds_permute(db << 2, lane);
                    gr[u / 2] |= (uint32_t)((lane < c01 ? ja : 64 + jb) | (jb << 8)) << (16 * (u & 1));
                    big |= (cnt > 64 ? 1u : 0u) << u;
                }
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int t_ = t0 + u * FW;
                if (t_ < NT) {                                    // wave-uniform
                    const uint32_t g16 = (gr[u / 2] >> (16 * (u & 1))) & 0xFFFFu;
                    fs_append2(v[u], t_ * 256 + 2 * (int)(g16 & 255u), lo, myx, myc, wcount);
                    if ((big >> u) & 1u) {                        // wave-uniform, rare (more than half of the piece kept): entries 64 .. cnt - 1
                        const int cnt = __popc(__builtin_amdgcn_readlane(mkv.x, pl0 + u)) + __popc(__builtin_amdgcn_readlane(mkv.y, pl0 + u)) +
                                        __popc(__builtin_amdgcn_readlane(mkv.z, pl0 + u)) + __popc(__builtin_amdgcn_readlane(mkv.w, pl0 + u));
                        float2 w = make_float2(dead, dead);
                        if (64 + lane < cnt) w = reinterpret_cast<const float2*>(p.cand + ((size_t)row * NT + t_) * FS_SLOT)[64 + lane];
                        fs_append2(w, t_ * 256 + 2 * (int)(64u + (g16 >> 8)), lo, myx, myc, wcount);
                    }
                }
            }
        }
        mkv = load_masks(row + (int)gridDim.x);                  // the next row's masks and statistics arrive under the rest of this row
        stv = load_stats(row + (int)gridDim.x);
        // the histogram over THIS wave's slice, in a dense sweep (all 64 lanes busy: ~18 LDS atomic instructions per wave instead of the 128 sparse ones
        // an atomic per appended value inside the gather costs -- the LDS instruction slots, not the bytes, bound this kernel)
        if (fast) {
            const int cw = min(wcount, WSL);
            for (int i = lane; i < cw; i += 64) atomicAdd(&S.hist[hslotf(min(NBF - 1, max(0, (int)((myx[i] - lo) * inv_w))))], 1u);
        }
        if (MM_EXP == 1) { __syncthreads(); continue; }      /* tools: gather only */
        if (lane == 0) S.wcnt[wid] = wcount;
        __syncthreads();
        float sumexp = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < FW; ++w2) sumexp += S.redx[w2];
        int total = 0;
        bool fits = true;
#pragma unroll
        for (int w2 = 0; w2 < FW; ++w2) { total += S.wcnt[w2]; fits = fits && S.wcnt[w2] <= WSL; }
        // the candidate set holds the kept set iff >= k values passed the lower bound and every wave's slice held its share
        // (debug bit 1 << 27, tests only: every 97th row is declared unverifiable, so the on-device fallback runs on Gaussian logits too)
        const bool ok = total >= p.k_keep && fits && !((p.debug & (1 << 27)) && row % 97 == 5);
        if (!ok) {
            if (tid == 0) {
                // the bound could not be verified for this row: hand it to the on-device fallback (its logits are recomputed and sampled by
                // sample_kernel inside the same generate, model.hip); only a full list -- or a caller without one -- raises the flag
                int slot = -1;
                if (p.fail_rows) { slot = atomicAdd(p.fail_count, 1); if (slot < p.fail_cap) p.fail_rows[slot] = row; }
                if (!p.fail_rows || slot >= p.fail_cap) { atomicExch(p.fail_flag, 1); if (p.ids) p.ids[pos_flat] = 0; if (p.scores) p.scores[pos_flat] = 0.f; }
            }
            __syncthreads();
            continue;
        }
        // ---- bin that holds the k-th largest value (every wave scans redundantly), its members -> exact select by rank counting
        const int need = p.k_keep;
        int tbin = 0, above = 0, bcnt = 0;
        if (fast) {
            uint32_t mine = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) mine += S.hist[j * 64 + lane];          // bins 16*lane .. 16*lane + 15
            uint32_t suf = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t_ = (uint32_t)__shfl_down((int)suf, o, 64); if (lane + o < 64) suf += t_; }
            const uint32_t ab = suf - mine;
            const unsigned long long own = __ballot(ab < (uint32_t)need && (uint32_t)need <= suf);
            const int Lq = __ffsll((long long)own) - 1;                           // total >= need, so some lane owns it
            uint32_t abq = (uint32_t)__shfl((int)ab, Lq, 64);
#pragma unroll 1
            for (int j = 15; j >= 0; --j) {
                const uint32_t h = S.hist[j * 64 + Lq];
                if (abq < (uint32_t)need && (uint32_t)need <= abq + h) { tbin = Lq * 16 + j; bcnt = (int)h; break; }
                abq += h;
            }
            above = (int)abq;
        }
        const bool slow = !fast || bcnt > CANDF;
        uint32_t thr;
        int kept_w = 0;                                           // wave-uniform: this wave's kept entries, squeezed to the front of its slice
        int n_side = 0;                                           // fast path: members of the threshold bin parked in S.cand / S.candc
        const int cw = min(wcount, WSL);                          // (== wcount: the row passed `fits`)
        if (!slow) {
            // ONE sweep over the wave's slice: an entry above the threshold bin is kept whatever the exact threshold turns out to be -> squeezed to the front of
            // the slice, in place (the write position never passes the read position; same wave: LDS accesses complete in program order); a member of the
            // threshold bin is parked (key + column) for the exact select, and joins the Gumbel loop below if it reaches the k-th largest value
            int wpos = 0;
            for (int b0 = 0; b0 < cw; b0 += 64) {
                const int i = b0 + lane;
                const bool live = i < cw;
                const float x = live ? myx[i] : 0.f;
                const uint16_t c = live ? myc[i] : (uint16_t)0;
                const int bn = min(NBF - 1, max(0, (int)((x - lo) * inv_w)));
                const bool kp = live && bn > tbin;
                const unsigned long long bal = __ballot(kp);
                const int idx = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, (uint32_t)wpos));
                if (kp) { myx[idx] = x; myc[idx] = c; }
                wpos += __popcll(bal);
                if (live && bn == tbin) { const int sl = atomicAdd(&S.ncand, 1); if (sl < CANDF) { S.cand[sl] = fkey(x); S.candc[sl] = c; } }
            }
            kept_w = wpos;
            __syncthreads();
            const int need_in = need - above;
            n_side = min(S.ncand, CANDF);                         // == bcnt <= CANDF
            for (int i = tid; i < n_side; i += FT) {
                const uint32_t ki = S.cand[i];
                int gt = 0, ge = 0;
#pragma unroll 4
                for (int j = 0; j < n_side; ++j) { const uint32_t kj = S.cand[j]; gt += kj > ki; ge += kj >= ki; }
                if (gt < need_in && need_in <= ge) S.thr_key = ki;      // every thread that satisfies this holds the same key
            }
            __syncthreads();
            thr = S.thr_key;
        } else {
            // massive ties / degenerate span: bisection over the 32 key bits on the LDS list, then the slice is squeezed down to the entries >= the k-th largest
            uint32_t prefix = 0;
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t trial = prefix | (1u << bit);
                int c = 0;
                for (int i = lane; i < cw; i += 64) c += fkey(myx[i]) >= trial;
                c = wave_sum_i(c);
                __syncthreads();
                if (lane == 0) S.redi[wid] = c;
                __syncthreads();
                int tot = 0;
                for (int w2 = 0; w2 < FW; ++w2) tot += S.redi[w2];
                if (tot >= need) prefix = trial;
            }
            __syncthreads();
            thr = prefix;
            int wpos = 0;
            for (int b0 = 0; b0 < cw; b0 += 64) {
                const int i = b0 + lane;
                const float x = i < cw ? myx[i] : 0.f;
                const uint16_t c = i < cw ? myc[i] : (uint16_t)0;
                const bool kp = i < cw && fkey(x) >= thr;
                const unsigned long long bal = __ballot(kp);
                const int idx = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, (uint32_t)wpos));
                if (kp) { myx[idx] = x; myc[idx] = c; }
                wpos += __popcll(bal);
            }
            kept_w = wpos;
        }
        if (MM_EXP == 2) { __syncthreads(); continue; }      /* tools: up to the kept list */
        // ---- Gumbel argmax over the kept entries (mmp.py:410-411); ties -> lower index.  Every wave scans its own dense slice (its own LDS writes: no barrier
        //      needed), then the parked members of the threshold bin, 512 per round over the workgroup, those below the k-th largest masked out.  The result does
        //      not depend on which lane sees which entry (ties go by column index).
        const float T = p.temperature;
        float best = -INFINITY, best_x = 0.f;
        int best_i = 0x7FFFFFFF;
        {
            const int nit = (kept_w + 63) >> 6;
            const int nsd = (n_side > wid * 64 ? 1 : 0) + (n_side > FT + wid * 64 ? 1 : 0);      // rounds of parked members this wave takes part in (CANDF = 2 FT)
            for (int it = 0; it < nit + nsd; ++it) {
                const bool side = it >= nit;
                const int i = side ? (it - nit) * FT + tid : it * 64 + lane;
                bool live = side ? i < n_side : i < kept_w;
                float x = 0.f;
                int idx = 0;
                if (live) {
                    if (side) {
                        const uint32_t k = S.cand[i];
                        x = __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);      // inverse of fkey
                        idx = (int)S.candc[i];
                        live = k >= thr;
                    } else {
                        x = myx[i];
                        idx = (int)myc[i];
                    }
                }
                if (live) {
                    const float y = x / T + noise_gumbel(p, pos_flat, idx);
                    if (y > best || (y == best && idx < best_i)) { best = y; best_i = idx; best_x = x; }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(best_i, o, 64);
            const float ox = __shfl_xor(best_x, o, 64);
            if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; best_x = ox; }
        }
        if (lane == 0) { S.redf[wid] = best; S.redi[wid] = best_i; S.redx[wid] = best_x; }
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < FW; ++i)
                if (S.redf[i] > best || (S.redf[i] == best && S.redi[i] < best_i)) { best = S.redf[i]; best_i = S.redi[i]; best_x = S.redx[i]; }
            const float score = 1.f - expf(best_x - M) / sumexp;      // mmp.py:603-606 on the unfiltered logits
            if (p.ids) p.ids[pos_flat] = (int64_t)best_i;
            if (p.scores) p.scores[pos_flat] = score;
            if (p.pred_out) p.pred_out[row] = (int64_t)best_i;
            if (p.score_out) p.score_out[row] = score;
        }
        __syncthreads();
    }
}

double norm_quantile(double pq) {      // Acklam's rational approximation (placement of the lower bound only, never a result)
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    if (pq <= 0.0) return -1e9;
    if (pq >= 1.0) return 1e9;
    if (pq < 0.02425) { const double q = sqrt(-2 * log(pq)); return (((((c[0]*q+c[1])*q+c[2])*q+c[3])*q+c[4])*q+c[5]) / ((((d[0]*q+d[1])*q+d[2])*q+d[3])*q+1); }
    if (pq > 1 - 0.02425) { const double q = sqrt(-2 * log(1 - pq)); return -(((((c[0]*q+c[1])*q+c[2])*q+c[3])*q+c[4])*q+c[5]) / ((((d[0]*q+d[1])*q+d[2])*q+d[3])*q+1); }
    const double q = pq - 0.5, r = q * q;
    return (((((a[0]*r+a[1])*r+a[2])*r+a[3])*r+a[4])*r+a[5])*q / (((((b[0]*r+b[1])*r+b[2])*r+b[3])*r+b[4])*r+1);
}

}  // namespace

float k_fused_z(int k_keep, int V, float margin) { return (float)(norm_quantile(1.0 - (double)k_keep / (double)V) - (double)margin); }

// ---- distribution-free bound (round 5): thr[r] = the rank-th largest of S sampled logits of row r (sub [R][S]: the row's embedding times S vocabulary rows of
// to_logits drawn once per model).  The number of sampled columns above the row's true k-th largest logit is hypergeometric with mean S k / V, whatever the
// logits' distribution: a rank 4.5 standard deviations beyond that mean leaves the true top-k inside the candidates except for ~3e-6 of the rows (which the
// finisher detects and the on-device fallback finishes).  Exact select by radix descent on the order-preserving keys: 32 counting steps, one barrier each.
template <int PER>
__global__ __launch_bounds__(256) void quantile_rows_kernel(const float* __restrict__ sub, long ld, int R, int S, int rank, float* __restrict__ thr) {
    __shared__ int cnt[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int row = blockIdx.x; row < R; row += gridDim.x) {
        uint32_t key[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = i * 256 + tid;
            key[i] = c < S ? fkey(sub[(size_t)row * ld + c]) : 0u;      // (padding: below every real key)
        }
        uint32_t prefix = 0;
#pragma unroll 1
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = prefix | (1u << bit);
            int c = 0;
#pragma unroll
            for (int i = 0; i < PER; ++i) c += __popcll(__ballot(key[i] >= trial));
            if (lane == 0) cnt[bit & 1][wid] = c;
            __syncthreads();
            const int tot = (cnt[bit & 1][0] + cnt[bit & 1][1]) + (cnt[bit & 1][2] + cnt[bit & 1][3]);
            if (tot >= rank) prefix = trial;      // at least `rank` keys >= trial: the rank-th largest has this bit set
        }
        if (tid == 0) thr[row] = __uint_as_float((prefix & 0x80000000u) ? (prefix & 0x7FFFFFFFu) : ~prefix);
        __syncthreads();
    }
}
int k_fused_quantile_rank(int k_keep, int V, int S) {
    const double p = (double)k_keep / (double)V;
    const double q = (double)S * p + 4.5 * sqrt((double)S * p * (1. - p)) + 1.;
    const int r = (int)ceil(q);
    return r < 1 ? 1 : (r > S ? S : r);
}
int k_fused_quantile(hipStream_t s, const float* sub, long ld, int R, int S, int rank, float* thr) {
    if (R <= 0) return MM_OK;
    if (S <= 0 || S > 4096 || rank < 1 || rank > S) return mm_set_error(MM_ERR_SHAPE, "fused_quantile: 1 <= rank <= S <= 4096");
    const int grid = R < 2048 ? R : 2048;
    if (S <= 2048) hipLaunchKernelGGL((quantile_rows_kernel<8>), dim3(grid), dim3(256), 0, s, sub, ld, R, S, rank, thr);
    else hipLaunchKernelGGL((quantile_rows_kernel<16>), dim3(grid), dim3(256), 0, s, sub, ld, R, S, rank, thr);
    return mm_check_launch("quantile_rows_kernel");
}

size_t k_fused_threshold_ws_bytes(int R, int D) { return ((size_t)R * D * 2 + 255) / 256 * 256 + (size_t)R * D * 4 + (size_t)R * 4 + 512; }

int k_fused_threshold(hipStream_t s, const bf16_t* ec, const bf16_t* en, long ld, int R, int D, float cond_scale, const float* wmean, const bf16_t* wcov,
                      float z, void* ws, float* thr) {
    if (R <= 0) return MM_OK;
    if (D <= 0 || (D % 64)) return mm_set_error(MM_ERR_SHAPE, "fused_threshold: D must be a positive multiple of 64");
    unsigned char* w8 = (unsigned char*)ws;
    bf16_t* ebf = (bf16_t*)w8;
    float* tq = (float*)(w8 + ((size_t)R * D * 2 + 255) / 256 * 256);
    float* mu = tq + (size_t)R * D;
    hipLaunchKernelGGL(fused_combine_kernel, dim3((R + 3) / 4), dim3(256), 0, s, ec, en, ld, R, D, cond_scale, wmean, ebf, mu);
    int rc = mm_check_launch("fused_combine_kernel");
    if (rc) return rc;
    GemmArgs a;      // T = E Cov' (Cov is symmetric): [R][D] x [D][D] on the bf16 MFMA GEMM, fp32 out
    memset(&a, 0, sizeof(a));
    a.mode = MODE_DENSE;
    a.W = wcov; a.N = D; a.ldw = D; a.K = D; a.M = R; a.X = ebf; a.ldx = D;
    a.out = tq; a.ldc = D; a.out_kind = OUT_F32;
    rc = mm_gemm_launch(a, s);
    if (rc) return rc;
    hipLaunchKernelGGL(fused_sigma_kernel, dim3((R + 3) / 4), dim3(256), 0, s, ebf, tq, mu, R, D, z, thr);
    return mm_check_launch("fused_sigma_kernel");
}

float* k_fused_threshold_mu(void* ws, int R, int D) {
    unsigned char* w8 = (unsigned char*)ws;
    float* tq = (float*)(w8 + ((size_t)R * D * 2 + 255) / 256 * 256);
    return tq + (size_t)R * D;
}

// the same bound from rows that ARE the operand of the logits GEMM already (k_final_mix: mixed, bf16) with their means in place: T = E Cov on those rows, sigma
int k_fused_threshold_mixed(hipStream_t s, const bf16_t* e, long ld, int R, int D, const bf16_t* wcov, float z, void* ws, float* thr) {
    if (R <= 0) return MM_OK;
    if (D <= 0 || (D % 64) || ld != D) return mm_set_error(MM_ERR_SHAPE, "fused_threshold: D must be a positive multiple of 64, rows dense");
    unsigned char* w8 = (unsigned char*)ws;
    float* tq = (float*)(w8 + ((size_t)R * D * 2 + 255) / 256 * 256);
    float* mu = tq + (size_t)R * D;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = MODE_DENSE;
    a.W = wcov; a.N = D; a.ldw = D; a.K = D; a.M = R; a.X = e; a.ldx = (int)ld;
    a.out = tq; a.ldc = D; a.out_kind = OUT_F32;
    const int rc = mm_gemm_launch(a, s);
    if (rc) return rc;
    hipLaunchKernelGGL(fused_sigma_kernel, dim3((R + 3) / 4), dim3(256), 0, s, e, tq, mu, R, D, z, thr);
    return mm_check_launch("fused_sigma_kernel");
}

int k_fused_emit(hipStream_t s, const float* logits, long ld, int R, int V, const float* thr, float4* stats, float4* cand) {
    if (R <= 0) return MM_OK;
    if (V <= 0 || (V % 256)) return mm_set_error(MM_ERR_SHAPE, "fused_emit: V must be a multiple of 256");
    const long pieces = (long)R * (V / 256);
    hipLaunchKernelGGL(fused_emit_kernel, dim3((unsigned)((pieces + 3) / 4)), dim3(256), 0, s, logits, ld, R, V, thr, stats, cand);
    return mm_check_launch("fused_emit_kernel");
}

int k_sample_fused(hipStream_t s, const FusedSampleArgs& a_in) {
    FusedSampleArgs a = a_in;
    a.debug = g_mm_debug;
    if (a.R <= 0) return MM_OK;
    if (a.V <= 0 || (a.V % 256) || a.V > 65536) return mm_set_error(MM_ERR_SHAPE, "sample_fused: V must be a multiple of 256 and <= 65536");
    if (a.k_keep < 1 || a.k_keep > a.V) return mm_set_error(MM_ERR_SHAPE, "sample_fused: k_keep out of range");
    if ((a.noise_kind == MM_NOISE_GUMBEL || a.noise_kind == MM_NOISE_UNIFORM) && !a.noise) return mm_set_error(MM_ERR_SHAPE, "sample_fused: noise tensor required");
    if (!(a.temperature > 0.f)) return mm_set_error(MM_ERR_SHAPE, "sample_fused: temperature must be > 0");
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sample_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(FusedShared));
        if (e != hipSuccess) return mm_set_hip_error(e, "sample_fused hipFuncSetAttribute");
        attr_set = true;
    }
    const int grid = a.R < 512 ? a.R : 512;       // two workgroups per CU, persistent over the rows
    hipLaunchKernelGGL(sample_fused_kernel, dim3(grid), dim3(FT), sizeof(FusedShared), s, a);
    return mm_check_launch("sample_fused_kernel");
}
