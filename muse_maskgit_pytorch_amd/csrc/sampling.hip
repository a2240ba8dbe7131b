// Sampling tail of MaskGit.generate for gfx950 (muse_maskgit_pytorch.py:558-563, 576-609).
//
//  mask_step   : per sample, the k lowest-confidence tokens (scores.topk(k), ties -> lower index) are set back
//                to the mask id; emits the compact, position-sorted list of masked rows that the rest of the
//                step works on (every sample has exactly k masked tokens at a step).
//  sample_rows : per masked row of the CFG-combined fp32 logits [R][V]:
//                  top_k filter (keep the ceil(0.1 V) largest, mmp.py:413-418)  -> exact k-th largest via a
//                  histogram select on order-preserving integer keys,
//                  Gumbel argmax over the kept entries at the annealed temperature (mmp.py:406-411, 578-580),
//                  confidence score 1 - softmax(logits)[pred] on the UNfiltered logits (mmp.py:603-606),
//                and scatters pred id / score to the token grid.
//                The row (V <= 65536 fp32) is read from HBM exactly ONCE and held in registers by a
//                512-thread workgroup (128 values per lane) -- the reference sweeps this tensor ~14 times.
//                This is the HBM-bound kernel of the path: algorithmic bytes = 4*V per row (+ 4*V noise in
//                parity mode).
#include <math.h>

#include "common.h"
#include "muse_hip_internal.h"

namespace {

constexpr float MASK_FILL = -1e5f;   // mmp.py:609

// ------------------------------------------------------------------------------------------------ mask step
// scores.topk(k) + scatter of the mask id (mmp.py:558-563) with the deterministic tie rule of the oracle (larger score first, then LOWER index): position i is re-masked iff
// rank_i = #{j : s_j > s_i or (s_j == s_i and j < i)} < k.  One workgroup per sample.  Round 6: up to 1024 threads (one position per thread at the super-resolution length), the
// rank loop on 16-byte LDS broadcast reads (four scores per instruction), and the compacted row list's prefix count from wave ballots instead of a second O(n^2) loop:
// 128 -> ~15 us per step at n = 1024 (2.3 ms of a super-resolution generate), 11.5 -> ~7 at n = 256.
__global__ __launch_bounds__(1024) void mask_step_kernel(float* __restrict__ scores, int64_t* __restrict__ ids, int n, int k,
                                                         int64_t mask_id, int32_t* __restrict__ rows_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];          // n scores (padded to a multiple of 4 with -inf: never counted) + per-wave counts
    float* sc = sm;
    const int n4 = (n + 3) & ~3;
    int* wcnt = reinterpret_cast<int*>(sm + n4);
    const int b = blockIdx.x, T = blockDim.x, t = threadIdx.x, lane = t & 63, w = t >> 6, NWV = T >> 6;
    float* srow = scores + (size_t)b * n;
    for (int i = t; i < n4; i += T) sc[i] = i < n ? srow[i] : -__builtin_inff();
    __syncthreads();
    int base = 0;                                   // selected positions in front of the current chunk
    for (int i0 = 0; i0 < n; i0 += T) {
        const int i = i0 + t;
        bool sel = false;
        if (i < n) {
            const float si = sc[i];
            int rank = 0;
            for (int j = 0; j < n4; j += 4) {
                const float4 v = *reinterpret_cast<const float4*>(sc + j);      // (same address in every lane: a broadcast read)
                rank += (int)((v.x > si) || (v.x == si && j < i)) + (int)((v.y > si) || (v.y == si && j + 1 < i)) +
                        (int)((v.z > si) || (v.z == si && j + 2 < i)) + (int)((v.w > si) || (v.w == si && j + 3 < i));
            }
            sel = rank < k;
        }
        const unsigned long long bal = __ballot(sel);
        if (lane == 0) wcnt[w] = __popcll(bal);
        __syncthreads();
        int pre = base;
        for (int q = 0; q < w; ++q) pre += wcnt[q];
        int tot = 0;
        for (int q = 0; q < NWV; ++q) tot += wcnt[q];
        pre += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (i < n) {
            if (sel) {
                ids[(size_t)b * n + i] = mask_id;
                if (rows_out) rows_out[(size_t)b * k + pre] = b * n + i;
            } else {
                srow[i] = MASK_FILL;             // what mmp.py:609 leaves in every unmasked slot
            }
        }
        base += tot;
        __syncthreads();                         // wcnt is rewritten by the next chunk
    }
}

// Philox2x32-10, philox_uniform2, gumbel_of / gumbel_of_unit: common.h (one definition for both sampling paths)

__global__ __launch_bounds__(256) void philox_fill_kernel(uint64_t seed, uint64_t row_offset, uint32_t step, int rows, int V, float* out) {
    const int nv = V >> 2;
    const long total = (long)rows * nv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int row = (int)(i / nv), c = (int)(i - (long)row * nv);
        float u0[2], u1[2];
        philox_uniform2(seed, row_offset + (uint64_t)row, step, (uint32_t)(2 * c), u0);
        philox_uniform2(seed, row_offset + (uint64_t)row, step, (uint32_t)(2 * c + 1), u1);
        *reinterpret_cast<float4*>(out + (long)row * V + c * 4) = make_float4(u0[0], u0[1], u1[0], u1[1]);
    }
}

// ------------------------------------------------------------------------------------------------ sample rows
// Structure (persistent 512-thread workgroups, one row at a time in registers, 128 values per lane at V = 65536):
//   A  max / min / mean / variance of the row                   (4 VALU per value)
//   B  sum exp(x - max)                                         (softmax denominator of mmp.py:603)
//   C  2048-bin value-linear histogram of the row's upper tail  (non-returning LDS atomics; full-range retry pass if the tail
//      estimate missed) -> every wave scans it: bin t that holds the k-th largest value, how many values lie above t
//   D  every value with bin >= t is appended (value, index) to this WAVE's private slice of an LDS list --
//      slot = running wave count + lane prefix of the ballot, so no atomics and no waits
//   -- the row's registers are dead here: the NEXT row's HBM read starts (instalments, see below) --
//   then short loops over the ~k listed entries only: exact k-th largest inside bin t (rank counting), Gumbel
//   noise + argmax over the entries >= that threshold.
// Passes A-D are the only fully unrolled code (the register file cannot be indexed dynamically); keeping them to a few
// instructions per value matters: the first version of this kernel spent its time fetching ~50k instructions per row.
// Rows the fast path cannot take (span 0 / non-finite, > 2048 values inside bin t, a wave slice overflowing -- i.e.
// massive ties) go through slow_threshold(): bisection on the integer keys, re-reading the row from L2.
constexpr int ST = 512;          // threads per row: 8 waves = 2 per SIMD -> up to 256 VGPRs each
constexpr int NW = ST / 64;
constexpr int NB = 2048;         // histogram bins
constexpr int CAND_CAP = 2048;   // exact-select capacity (values inside the threshold bin)
constexpr int WSLICE = 1280;     // per-wave list capacity; k/8 = 820 at V = 65536, sigma ~ 27

// order-preserving map float -> uint32 (larger float <=> larger key)
__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// opaque read: stops the compiler from keeping per-value derived quantities alive across the passes
__device__ __forceinline__ float opaque(float x) {
    asm volatile("" : "+v"(x));
    return x;
}

static_assert(NB == 64 * 32, "the histogram scan assumes 32 bins per lane");

struct SampleShared {
    float redf[NW];
    float redf2[NW];
    float bval[NW];
    float bx[NW];
    float redse[NW];
    float tmax[256], tsum[256];   // per 256-column tile: max and sum exp(x - max) (common.h tile_softmax_stats), V % 256 == 0 only
    int redi[NW];
    uint32_t hist[NB];            // TRANSPOSED: bin b lives at hslot(b), so lane l's 32 consecutive bins form a conflict-free column
    uint32_t cand[CAND_CAP];
    uint2 kv[NW * WSLICE];        // (value bits, vocabulary index), one slice per wave
    int ncand;
    int slow;
    uint32_t thr;
};

__device__ __forceinline__ int hslot(int b) { return ((b & 31) << 6) | (b >> 5); }

// workgroup barrier for LDS-only hand-offs: __syncthreads() also drains vmcnt(0), which would wait for the NEXT row's prefetch
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// inclusive suffix sum over the wave: result[l] = sum of x over lanes >= l
__device__ __forceinline__ uint32_t wave_suffix_sum(uint32_t x, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_down((int)x, o, 64);
        if (lane + o < 64) x += t;
    }
    return x;
}

// exact k-th largest key of the row by bisection over the 32 key bits, reading the row from memory each round
__device__ uint32_t slow_threshold(const float* lr, int V, int k, SampleShared& S) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t prefix = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t trial = prefix | (1u << bit);
        int c = 0;
        for (int i = tid; i < V; i += ST) c += fkey(lr[i]) >= trial;
        c = wave_sum_i(c);
        __syncthreads();
        if (lane == 0) S.redi[wid] = c;
        __syncthreads();
        int tot = 0;
        for (int w = 0; w < NW; ++w) tot += S.redi[w];
        if (tot >= k) prefix = trial;      // at least k keys >= trial: the k-th largest has this bit set
    }
    __syncthreads();
    return prefix;
}

__device__ __forceinline__ float noise_gumbel(const SampleArgs& p, long pos_flat, int idx) {
    if (p.noise_kind == MM_NOISE_GUMBEL) return p.noise[(size_t)pos_flat * p.noise_ld + idx];
    if (p.noise_kind == MM_NOISE_UNIFORM) return gumbel_of(p.noise[(size_t)pos_flat * p.noise_ld + idx]);
    if (p.noise_kind == MM_NOISE_PHILOX) {
        float u[2];
        philox_uniform2(p.seed, p.row_offset + (uint64_t)pos_flat, p.step, (uint32_t)(idx >> 1), u);
        return gumbel_of_unit((idx & 1) ? u[1] : u[0]);
    }
    return 0.f;
}

template <int VEC_IT, bool FULL>
__global__ __launch_bounds__(ST) void sample_kernel(const SampleArgs p_in) {
    SampleArgs p = p_in;
    if (p.seed_dev) { p.seed = p.seed_dev[0]; p.row_offset = p.seed_dev[1] * (uint64_t)p.row_mul; }      // (wave-uniform scalar loads: the keys of a replayed graph)
    __shared__ SampleShared S;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int V = p.V;
    int row = blockIdx.x;                 // persistent: this workgroup samples rows blockIdx.x, + gridDim.x, ...
    const int Rn = p.count_dev ? min(p.R, *p.count_dev) : p.R;      // (fallback launches: the row count lives on the device)
    if (row >= Rn) return;

    // ---- the one HBM read of a row: element index e = (it*ST + tid)*4 + c.  Padding (e >= V) is -inf: neutral for the max and for
    //      exp(); min / histogram / list passes skip it by index.
    float v[VEC_IT * 4];
    // buffer loads: one resource descriptor per row (wave-uniform), ONE per-lane byte offset, the 32 steps as scalar offsets -- plain
    // global loads made the compiler keep 32 separate 64-bit addresses alive across the loop (and spill them)
    const int voff = tid * 16;
#define LOAD_CHUNK(rs_, it0_, it1_)                                                                            \
    _Pragma("unroll") for (int it = (it0_); it < (it1_); ++it) {                                               \
        if (it < VEC_IT) {                                                                                     \
            const int e = (it * ST + tid) * 4;                                                                 \
            u32x4_t x = __builtin_amdgcn_raw_buffer_load_b128(rs_, voff, it * ST * 16, 0);                     \
            const bool ok_ = FULL || e < V;       /* out-of-range reads return 0: make them the -inf padding */ \
            v[it * 4 + 0] = ok_ ? __uint_as_float(x[0]) : -INFINITY; v[it * 4 + 1] = ok_ ? __uint_as_float(x[1]) : -INFINITY; \
            v[it * 4 + 2] = ok_ ? __uint_as_float(x[2]) : -INFINITY; v[it * 4 + 3] = ok_ ? __uint_as_float(x[3]) : -INFINITY; \
        }                                                                                                      \
    }
#define ROW_RSRC(row_) __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.logits + (size_t)(row_) * p.ld), 0, V * 4, 0x00020000)
#define LOAD_ROW(row_) { const __amdgpu_buffer_rsrc_t rs_ = ROW_RSRC(row_); LOAD_CHUNK(rs_, 0, VEC_IT) }
    LOAD_ROW(row)
    for (;;) {
    int tidv = tid;                       // per-row copy the compiler cannot hoist: the 128 element indices derived from it would otherwise
    asm volatile("" : "+v"(tidv));      // be kept in registers across rows (loop-invariant) and spill
    const float* lr = p.logits + (size_t)row * p.ld;
    const int orow = p.src_rows ? p.src_rows[row] : row;      // the row this logits row was computed for
    const long pos_flat = p.rows ? (long)p.rows[orow] : (long)orow;

    // ---- A: row max / min / mean / variance
    float vmax = -INFINITY, vmin = INFINITY, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < VEC_IT; ++it) {
        const bool ok = FULL || (it * ST + tidv) * 4 < V;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float x = v[it * 4 + c];
            vmax = fmaxf(vmax, x);
            vmin = fminf(vmin, ok ? x : INFINITY);
            s1 += ok ? x : 0.f;
            s2 += ok ? x * x : 0.f;
        }
    }
    vmax = wave_max(vmax);
    vmin = -wave_max(-vmin);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) { S.redf[wid] = vmax; S.redf2[wid] = vmin; S.bval[wid] = s1; S.bx[wid] = s2; }
    if (tid == 0) { S.ncand = 0; S.slow = 0; }
    for (int i = tid; i < NB; i += ST) S.hist[i] = 0;
    __syncthreads();
    vmax = S.redf[0]; vmin = S.redf2[0]; s1 = S.bval[0]; s2 = S.bx[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) { vmax = fmaxf(vmax, S.redf[i]); vmin = fminf(vmin, S.redf2[i]); s1 += S.bval[i]; s2 += S.bx[i]; }

    // ---- B: softmax denominator on the unfiltered logits (fast exp: v_exp_f32, ~1e-6 relative per term)
    // V a multiple of 256 (every vocabulary the fused sampler of sampling_fused.hip serves): per-tile statistics with the SAME function and,
    // below, the same combination order as that path -- a wave's 4 values per lane of iteration `it` are exactly the 256-column tile
    // it * 8 + wid in the GEMM emission's lane layout -- so a row gets bit-identical confidences whichever path samples it.
    const bool tilewise = (V & 255) == 0;
    float se = 0.f;
    if (!(p.debug & 128)) {
        if (tilewise) {
#pragma unroll
            for (int it = 0; it < VEC_IT; ++it) {
                const int tile = it * NW + wid;
                if (FULL || tile * 256 < V) {      // wave-uniform
                    float tm, te;
                    tile_softmax_stats(make_float4(opaque(v[it * 4]), opaque(v[it * 4 + 1]), opaque(v[it * 4 + 2]), opaque(v[it * 4 + 3])), tm, te);
                    if (lane == 0) { S.tmax[tile] = tm; S.tsum[tile] = te; }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < VEC_IT * 4; ++i) se += __expf(opaque(v[i]) - vmax);
        }
    }
    se = wave_sum(se);
    if (lane == 0 && !tilewise) S.redse[wid] = se;        // published by the barrier that closes the histogram pass

    // ---- C: value-linear histogram of the row's UPPER TAIL (bin 0 = smallest), then every wave scans it (redundantly: no
    //      broadcast, no extra barrier) for the bin t holding the k-th largest value.
    //      Attempt 0 bins only values >= lo = mean + z_lo * std (z_lo = normal quantile of 1 - k/V minus a margin, from the
    //      host): a bell-shaped row has its k-th largest well above lo, and this cuts the LDS atomics ~5x and makes the bins
    //      finer.  If fewer than k values turn out to be >= lo (heavy-tailed row) attempt 1 repeats the pass over the full
    //      range [min, max] -- the row is still in registers, so a miss costs one more register pass, not a re-read.
    //      Span 0 / inf / NaN or > CAND_CAP values inside bin t (massive ties) -> slow path.
    const float mean = s1 / (float)V;
    const float var = fmaxf(s2 / (float)V - mean * mean, 0.f);
    const int need = p.k_keep;
    float lo = fmaxf(vmin, mean + p.z_lo * sqrtf(var));
    float span, inv_w;
    bool fast, found = false;
    int tbin = 0, above = 0, cnt = 0;
#pragma unroll 1
    for (int attempt = 0; attempt < 2; ++attempt) {
        span = vmax - lo;
        fast = (span >= 1e-30f) && (span < 3.0e38f);
        inv_w = fast ? (float)NB / span : 0.f;
        if (fast && !(p.debug & 16)) {
#pragma unroll
            for (int it = 0; it < VEC_IT; ++it) {
                if (FULL || (it * ST + tidv) * 4 < V) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float x = opaque(v[it * 4 + c]);
                        if (x >= lo) atomicAdd(&S.hist[hslot(min(NB - 1, (int)((x - lo) * inv_w)))], 1u);
                    }
                }
            }
        }
        __syncthreads();                      // histogram complete
        if (fast) {
            uint32_t mine = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) mine += S.hist[j * 64 + lane];      // bins 32*lane .. 32*lane+31
            const uint32_t suf = wave_suffix_sum(mine, lane);
            const uint32_t ab = suf - mine;                                  // values in bins above this lane's range
            if ((uint32_t)__shfl((int)suf, 0, 64) >= (uint32_t)need) {       // else: the tail estimate missed
                const unsigned long long own = __ballot(ab < (uint32_t)need && (uint32_t)need <= suf);
                const int L = __ffsll((long long)own) - 1;
                const uint32_t abL = (uint32_t)__shfl((int)ab, L, 64);
                const uint32_t h = lane < 32 ? S.hist[lane * 64 + L] : 0u;   // bin 32*L + lane
                const uint32_t suf2 = wave_suffix_sum(h, lane);
                const uint32_t ab2 = abL + suf2 - h;
                const unsigned long long own2 = __ballot(lane < 32 && ab2 < (uint32_t)need && (uint32_t)need <= ab2 + h);
                const int J = __ffsll((long long)own2) - 1;
                tbin = L * 32 + J;
                above = __shfl((int)ab2, J, 64);
                cnt = __shfl((int)h, J, 64);
                found = true;
            }
        }
        if (!fast || found || attempt == 1 || !(lo > vmin)) break;
        lo = vmin;                            // retry over the full range
        __syncthreads();                      // every wave is done reading the histogram
        for (int i = tid; i < NB; i += ST) S.hist[i] = 0;
        __syncthreads();
    }
    if (tilewise) {      // sum_t tsum_t * exp(tmax_t - max), thread t = tile t, butterfly per wave, waves in order: sample_fused_kernel's expression
        const float term = wave_sum((tid < (V >> 8) && !(p.debug & 128)) ? S.tsum[tid] * expf(S.tmax[tid] - vmax) : 0.f);
        if (lane == 0) S.redse[wid] = term;      // (the tile statistics were published by the histogram pass's barrier; read at the row's end)
    }
    bool slow = !fast || !found || cnt > CAND_CAP;

    // ---- D: append every value with bin >= tbin to this wave's slice (no atomics: ballot prefix + running count).
    //      Straight-line: with ~10 % kept a ballot is practically never empty, so no branch around the bookkeeping.
    int wcount = 0;                       // wave-uniform
    uint2* mykv = S.kv + wid * WSLICE;
    if (!slow && !(p.debug & 32)) {
        // lower edge of bin tbin, lowered by a relative 1e-6 so that rounding can only ADD a few values of bin tbin-1
        // (harmless: they are neither candidates nor >= the final threshold)
        const float edge = lo + (float)tbin / inv_w - (fabsf(lo) + span) * 1e-6f;
#pragma unroll
        for (int it = 0; it < VEC_IT; ++it) {
            const int e = (it * ST + tidv) * 4;
            const bool ok = FULL || e < V;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float x = opaque(v[it * 4 + c]);
                const bool kp = ok && x >= edge;
                const unsigned long long bal = __ballot(kp);
                const int slot = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, (uint32_t)wcount));
                if (kp) mykv[min(slot, WSLICE - 1)] = make_uint2(__float_as_uint(x), (uint32_t)(e + c));
                wcount += __popcll(bal);
            }
        }
        if (wcount > WSLICE && lane == 0) S.slow = 1;
    }
    // ---- the row's values are dead from here on (everything below works on the LDS lists): start the NEXT row's HBM read now, it
    //      lands in the same registers while the exact select and the Gumbel phase (~40 % of a row's time) run
    const int nrow = row + (int)gridDim.x;
    const bool has_next = nrow < Rn;
    // (issued in instalments -- 8 loads here, 3 per trip of the Gumbel loop below: a wave that issues all 32 at once sits in the
    //  issue queue for about as long as the load takes, measured 13 k cycles of a 75 k-cycle row)
    const __amdgpu_buffer_rsrc_t nrs = ROW_RSRC(has_next ? nrow : row);
    if (has_next) LOAD_CHUNK(nrs, 0, 8)
    const int cw = min(wcount, WSLICE);
    if (!slow) {
        // members of bin tbin in this wave's own slice (a few per wave) -> the shared candidate list
#pragma unroll 2
        for (int i = lane; i < cw; i += 64) {
            const float x = __uint_as_float(mykv[i].x);
            if (x >= lo && min(NB - 1, (int)((x - lo) * inv_w)) == tbin) { const int sl = atomicAdd(&S.ncand, 1); if (sl < CAND_CAP) S.cand[sl] = fkey(x); }
        }
    }
    lds_barrier();
    slow = slow || S.slow != 0;

    uint32_t thr;
    if (p.debug & 64) { thr = 0xFFFFFFFFu; } else
    if (!slow) {
        // ---- exact threshold: the (need - above)-th largest among the members of bin tbin (rank counting)
        const int need_in = need - above;
        const int n_c = min(S.ncand, CAND_CAP);
        for (int i = tid; i < n_c; i += ST) {
            const uint32_t ki = S.cand[i];
            int gt = 0, ge = 0;
#pragma unroll 4
            for (int j = 0; j < n_c; ++j) { const uint32_t kj = S.cand[j]; gt += kj > ki; ge += kj >= ki; }
            if (gt < need_in && need_in <= ge) S.thr = ki;      // every thread that satisfies this holds the same key
        }
        lds_barrier();
        thr = S.thr;
    } else {
        thr = slow_threshold(lr, V, need, S);
    }

    // ---- Gumbel argmax over the kept entries (mmp.py:410-411); ties -> lower index like torch.argmax
    const float T = p.temperature;
    float best = -INFINITY, best_x = 0.f;
    int best_i = 0x7FFFFFFF;
    if (!slow) {
        // each wave walks its own slice, 2 independent entries per lane per trip so the RNG's multiply chains overlap
#pragma unroll
        for (int trip = 0; trip < WSLICE / 128; ++trip) {
            if (has_next) LOAD_CHUNK(nrs, 8 + 3 * trip, 8 + 3 * trip + 3)
            const int base = trip * 128;
            if (base >= cw) continue;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = base + u * 64 + lane;
                if (i < cw) {
                    const uint2 ent = mykv[i];
                    const float x = __uint_as_float(ent.x);
                    if (fkey(x) >= thr) {
                        const int idx = (int)ent.y;
                        const float y = x / T + noise_gumbel(p, pos_flat, idx);      // IEEE division: same bits as torch's CPU kernel
                        if (y > best || (y == best && idx < best_i)) { best = y; best_i = idx; best_x = x; }
                    }
                }
            }
        }
    } else {
        for (int idx = tid; idx < V; idx += ST) {
            const float x = lr[idx];
            if (fkey(x) < thr) continue;
            const float y = x / T + noise_gumbel(p, pos_flat, idx);
            if (y > best || (y == best && idx < best_i)) { best = y; best_i = idx; best_x = x; }
        }
    }
    if (has_next && slow) LOAD_CHUNK(nrs, 8, VEC_IT)          // the fast path issued these inside its loop
    if (has_next && !slow) LOAD_CHUNK(nrs, 8 + 3 * (WSLICE / 128), VEC_IT)
    // wave reduce (value desc, index asc)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        const float ox = __shfl_xor(best_x, o, 64);
        if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; best_x = ox; }
    }
    if (lane == 0) { S.bval[wid] = best; S.redi[wid] = best_i; S.bx[wid] = best_x; }
    lds_barrier();
    if (tid == 0) {
        for (int i = 1; i < NW; ++i)
            if (S.bval[i] > best || (S.bval[i] == best && S.redi[i] < best_i)) { best = S.bval[i]; best_i = S.redi[i]; best_x = S.bx[i]; }
        float sumexp = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) sumexp += S.redse[i];
        const float prob = expf(best_x - vmax) / sumexp;
        const float score = 1.f - prob;
        if (p.ids) p.ids[pos_flat] = (int64_t)best_i;
        if (p.scores) p.scores[pos_flat] = score;
        if (p.pred_out) p.pred_out[orow] = (int64_t)best_i;
        if (p.score_out) p.score_out[orow] = score;
    }
    if (!has_next) break;
    row = nrow;
    lds_barrier();                      // the shared state of this row is done with before the next row's pass A rewrites it
    }
#undef LOAD_ROW
#undef LOAD_CHUNK
#undef ROW_RSRC
}

}  // namespace

int k_mask_step(hipStream_t s, float* scores, int64_t* ids, int B, int n, int k, int64_t mask_id, int32_t* rows_out) {
    if (B <= 0) return MM_OK;
    if (n <= 0 || n > 4096 || k < 0 || k > n) return mm_set_error(MM_ERR_SHAPE, "mask_step: need 0 < n <= 4096 and 0 <= k <= n");
    const int threads = n >= 1024 ? 1024 : (n > 256 ? 512 : 256);
    hipLaunchKernelGGL(mask_step_kernel, dim3(B), dim3(threads), (size_t)((n + 3) & ~3) * 4 + 64, s, scores, ids, n, k, mask_id, rows_out);
    return mm_check_launch("mask_step_kernel");
}

// normal quantile (Acklam's rational approximation, |error| < 1.2e-9 in the central region): used only to place the
// histogram's lower bound, never for a result
static double norm_quantile(double p) {
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    if (p <= 0.0) return -1e9;
    if (p >= 1.0) return 1e9;
    if (p < 0.02425) { const double q = sqrt(-2 * log(p)); return (((((c[0]*q+c[1])*q+c[2])*q+c[3])*q+c[4])*q+c[5]) / ((((d[0]*q+d[1])*q+d[2])*q+d[3])*q+1); }
    if (p > 1 - 0.02425) { const double q = sqrt(-2 * log(1 - p)); return -(((((c[0]*q+c[1])*q+c[2])*q+c[3])*q+c[4])*q+c[5]) / ((((d[0]*q+d[1])*q+d[2])*q+d[3])*q+1); }
    const double q = p - 0.5, r = q * q;
    return (((((a[0]*r+a[1])*r+a[2])*r+a[3])*r+a[4])*r+a[5])*q / (((((b[0]*r+b[1])*r+b[2])*r+b[3])*r+b[4])*r+1);
}

int k_sample_rows(hipStream_t s, const SampleArgs& a_in) {
    SampleArgs a = a_in;
    a.debug = g_mm_debug;
    a.z_lo = (float)(norm_quantile(1.0 - (double)a.k_keep / (double)a.V) - 0.35);
    if (a.R <= 0) return MM_OK;
    if (a.V <= 0 || (a.V % 4) || a.V > 65536) return mm_set_error(MM_ERR_SHAPE, "sample_rows: V must be a multiple of 4 and <= 65536");
    if (a.k_keep < 1 || a.k_keep > a.V) return mm_set_error(MM_ERR_SHAPE, "sample_rows: k_keep out of range");
    if (a.ld % 4 || ((a.noise_kind == MM_NOISE_GUMBEL || a.noise_kind == MM_NOISE_UNIFORM) && (!a.noise || a.noise_ld % 4)))
        return mm_set_error(MM_ERR_ALIGN, "sample_rows: logits/noise strides must be multiples of 4, noise required in tensor modes");
    if (!(a.temperature > 0.f)) return mm_set_error(MM_ERR_SHAPE, "sample_rows: temperature must be > 0 (clamp to 1e-10 like mmp.py:411)");
    const int vec_it = (a.V + ST * 4 - 1) / (ST * 4);
    // balanced persistent grid: at most ~one workgroup per CU, every workgroup the same number of rows (+-1)
    const int rows_per_wg = (a.R + 255) / 256;
    dim3 g((a.R + rows_per_wg - 1) / rows_per_wg), b(ST);
    const bool full = a.V == vec_it * ST * 4;
#define LAUNCH_S(N)                                                                   \
    if (full) hipLaunchKernelGGL((sample_kernel<N, true>), g, b, 0, s, a);            \
    else hipLaunchKernelGGL((sample_kernel<N, false>), g, b, 0, s, a)
    if (vec_it <= 1) { LAUNCH_S(1); }
    else if (vec_it <= 2) { LAUNCH_S(2); }
    else if (vec_it <= 4) { LAUNCH_S(4); }
    else if (vec_it <= 8) { LAUNCH_S(8); }
    else if (vec_it <= 16) { LAUNCH_S(16); }
    else { LAUNCH_S(32); }
#undef LAUNCH_S
    return mm_check_launch("sample_kernel");
}

int k_philox_fill(hipStream_t s, uint64_t seed, uint64_t row_offset, uint32_t step, int rows, int V, float* out) {
    if (rows <= 0) return MM_OK;
    if (V % 4) return mm_set_error(MM_ERR_SHAPE, "philox_fill: V must be a multiple of 4");
    long items = (long)rows * (V / 4);
    long blocks = (items + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(philox_fill_kernel, dim3((int)blocks), dim3(256), 0, s, seed, row_offset, step, rows, V, out);
    return mm_check_launch("philox_fill_kernel");
}

// ------------------------------------------------------------------------------------------------ training-forward losses
// F.cross_entropy(logits (b n c -> b c n), labels, ignore_index) (muse_maskgit_pytorch.py:343): per row
// loss = logsumexp(logits) - logits[label], rows with label == ignore_index contribute nothing; the mean over the other rows
// is taken by ce_finish_kernel.  One 256-thread workgroup streams a row once (online max / sum): HBM-bound, 4*V bytes per row.
namespace {

__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, long ld, int V, const int64_t* __restrict__ labels,
                                                      int64_t ignore_index, float* __restrict__ row_loss) {
    __shared__ float sm[4], ss[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t lab = labels[row];
    if (lab == ignore_index) { if (tid == 0) row_loss[row] = -1.f; return; }      // marker: losses are >= 0
    const float* lr = logits + (size_t)row * ld;
    float m = -INFINITY, s = 0.f;
    for (int i = tid * 4; i < V; i += 256 * 4) {
        const float4 x = *reinterpret_cast<const float4*>(lr + i);
        const float mx = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
        if (mx > m) { s *= expf(m - mx); m = mx; }
        s += expf(x.x - m) + expf(x.y - m) + expf(x.z - m) + expf(x.w - m);
    }
    // combine (m, s) pairs: wave butterfly then across the 4 waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64), os = __shfl_xor(s, o, 64);
        const float nm = fmaxf(m, om);
        s = (nm == -INFINITY) ? 0.f : s * expf(m - nm) + os * expf(om - nm);
        m = nm;
    }
    if (lane == 0) { sm[wid] = m; ss[wid] = s; }
    __syncthreads();
    if (tid == 0) {
        float M = sm[0], S = ss[0];
        for (int w = 1; w < 4; ++w) {
            const float nm = fmaxf(M, sm[w]);
            S = S * expf(M - nm) + ss[w] * expf(sm[w] - nm);
            M = nm;
        }
        const bool ok = lab >= 0 && lab < V;
        row_loss[row] = ok ? (M + logf(S)) - lr[lab] : -1.f;
    }
}

// mean over the rows that are not ignored (row_loss >= 0); a single workgroup: R is a few thousand
__global__ __launch_bounds__(256) void ce_finish_kernel(const float* __restrict__ row_loss, int R, float* __restrict__ out) {
    __shared__ float ssum[4];
    __shared__ int scnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float s = 0.f;
    int c = 0;
    for (int i = tid; i < R; i += 256) { const float v = row_loss[i]; if (v >= 0.f) { s += v; ++c; } }
    s = wave_sum(s);
    c = wave_sum_i(c);
    if (lane == 0) { ssum[wid] = s; scnt[wid] = c; }
    __syncthreads();
    if (tid == 0) {
        const float S = ssum[0] + ssum[1] + ssum[2] + ssum[3];
        const int Cn = scnt[0] + scnt[1] + scnt[2] + scnt[3];
        out[0] = Cn > 0 ? S / (float)Cn : NAN;          // torch returns nan when every target is ignored
    }
}

// F.binary_cross_entropy_with_logits(x, y) (muse_maskgit_pytorch.py:341, 374): mean(max(x,0) - x*y + log1p(exp(-|x|)))
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ x, const float* __restrict__ y, int n, float* __restrict__ out) {
    __shared__ float ssum[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float s = 0.f;
    for (int i = tid; i < n; i += 256) {
        const float v = x[i];
        s += fmaxf(v, 0.f) - v * y[i] + log1pf(expf(-fabsf(v)));
    }
    s = wave_sum(s);
    if (lane == 0) ssum[wid] = s;
    __syncthreads();
    if (tid == 0) out[0] = (ssum[0] + ssum[1] + ssum[2] + ssum[3]) / (float)n;
}

}  // namespace

int k_ce_loss(hipStream_t s, const float* logits, long ld, int R, int V, const int64_t* labels, int64_t ignore_index,
              float* row_loss_ws, float* out) {
    if (R <= 0) return mm_set_error(MM_ERR_SHAPE, "ce_loss: no rows");
    if (V % 4 || ld % 4) return mm_set_error(MM_ERR_ALIGN, "ce_loss: V and the row stride must be multiples of 4");
    hipLaunchKernelGGL(ce_rows_kernel, dim3(R), dim3(256), 0, s, logits, ld, V, labels, ignore_index, row_loss_ws);
    int rc = mm_check_launch("ce_rows_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, s, row_loss_ws, R, out);
    return mm_check_launch("ce_finish_kernel");
}

int k_ce_finish(hipStream_t s, const float* row_loss, int R, float* out) {
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, s, row_loss, R, out);
    return mm_check_launch("ce_finish_kernel");
}

int k_bce_loss(hipStream_t s, const float* x, const float* y, int n, float* out) {
    if (n <= 0) return mm_set_error(MM_ERR_SHAPE, "bce_loss: no elements");
    hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(256), 0, s, x, y, n, out);
    return mm_check_launch("bce_kernel");
}
