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
#include "common.h"
#include "muse_hip_internal.h"

namespace {

constexpr float MASK_FILL = -1e5f;   // mmp.py:609

// ------------------------------------------------------------------------------------------------ mask step
__global__ __launch_bounds__(256) void mask_step_kernel(float* __restrict__ scores, int64_t* __restrict__ ids, int n, int k,
                                                        int64_t mask_id, int32_t* __restrict__ rows_out) {
    extern __shared__ float sm[];          // n scores + n flags
    float* sc = sm;
    int* flag = reinterpret_cast<int*>(sm + n);
    const int b = blockIdx.x;
    float* srow = scores + (size_t)b * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) sc[i] = srow[i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float si = sc[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float sj = sc[j];
            rank += (sj > si) || (sj == si && j < i);
        }
        flag[i] = rank < k;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (flag[i]) {
            int pre = 0;
            for (int j = 0; j < i; ++j) pre += flag[j];
            ids[(size_t)b * n + i] = mask_id;
            if (rows_out) rows_out[(size_t)b * k + pre] = b * n + i;
        } else {
            srow[i] = MASK_FILL;             // what mmp.py:609 leaves in every unmasked slot
        }
    }
}

// ------------------------------------------------------------------------------------------------ philox4x32-10
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void philox_uniform4(uint64_t seed, uint64_t row_global, uint32_t step, uint32_t col4, float (&u)[4]) {
    uint32_t o[4];
    philox4x32_10(col4, step, (uint32_t)row_global, (uint32_t)(row_global >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = (float)(o[i] >> 8) * (1.0f / 16777216.0f);   // 24-bit, [0, 1)
}

__device__ __forceinline__ float gumbel_of(float u) {
    // -log(-log(u)) with the reference's clamps (mmp.py:403-408)
    const float a = logf(fmaxf(u, 1e-20f));
    return -logf(fmaxf(-a, 1e-20f));
}

__global__ __launch_bounds__(256) void philox_fill_kernel(uint64_t seed, uint64_t row_offset, uint32_t step, int rows, int V, float* out) {
    const int nv = V >> 2;
    const long total = (long)rows * nv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int row = (int)(i / nv), c = (int)(i - (long)row * nv);
        float u[4];
        philox_uniform4(seed, row_offset + (uint64_t)row, step, (uint32_t)c, u);
        *reinterpret_cast<float4*>(out + (long)row * V + c * 4) = make_float4(u[0], u[1], u[2], u[3]);
    }
}

// ------------------------------------------------------------------------------------------------ sample rows
constexpr int ST = 512;         // threads per row: 8 waves = 2 per SIMD -> 256 VGPRs each, the row (<= 128 values/lane) stays in registers
constexpr int NW = ST / 64;
constexpr int NB = 2048;        // histogram bins
constexpr int CAND_CAP = 2048;  // exact-select capacity
constexpr int KEPT_CAP = 8192;  // kept-entry list (k = ceil(0.1*65536) = 6554; only threshold TIES can exceed k)

// order-preserving map float -> uint32 (larger float <=> larger key); -0.0 < +0.0 is harmless here
__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct SampleShared {
    float redf[16];
    float redf2[16];
    uint32_t redu[16];
    uint32_t redu2[16];
    int redi[16];
    float bval[16];
    float bx[16];
    uint32_t hist[NB];
    uint32_t cand[CAND_CAP];
    uint32_t lane_sums[64];
    float kx[KEPT_CAP];
    int ki[KEPT_CAP];
    int nkept;
    int ncand;
    int tbin, above, cnt;
    uint32_t thr;
    uint32_t kmin, kmax;
};

// opaque read: stops the compiler from keeping per-element derived values (keys, bins) alive across the passes,
// which is what pushes a 128-values-per-lane kernel into scratch
__device__ __forceinline__ float opaque(float x) {
    asm volatile("" : "+v"(x));
    return x;
}

template <int VEC_IT, bool FULL>
__global__ __launch_bounds__(ST) void sample_kernel(const SampleArgs p) {
    __shared__ SampleShared S;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int row = blockIdx.x;
    const int V = p.V;
    const float* lr = p.logits + (size_t)row * p.ld;
    const long pos_flat = p.rows ? (long)p.rows[row] : (long)row;

    // ---- the one HBM read of the row: element index e = (it*ST + tid)*4 + c
    float v[VEC_IT * 4];
#pragma unroll
    for (int it = 0; it < VEC_IT; ++it) {
        const int e = (it * ST + tid) * 4;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FULL || e < V) x = *reinterpret_cast<const float4*>(lr + e);
        v[it * 4 + 0] = x.x; v[it * 4 + 1] = x.y; v[it * 4 + 2] = x.z; v[it * 4 + 3] = x.w;
    }
    auto valid = [&](int it) { return FULL || (it * ST + tid) * 4 < V; };   // V % 4 == 0: a float4 is all-valid or all-pad

    // ---- phase A: row max / min
    float vmax = -INFINITY, vmin = INFINITY;
#pragma unroll
    for (int it = 0; it < VEC_IT; ++it)
        if (valid(it)) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { vmax = fmaxf(vmax, v[it * 4 + c]); vmin = fminf(vmin, v[it * 4 + c]); }
        }
    vmax = wave_max(vmax);
    vmin = -wave_max(-vmin);
    if (lane == 0) { S.redf[wid] = vmax; S.redf2[wid] = vmin; }
    __syncthreads();
    vmax = S.redf[0]; vmin = S.redf2[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) { vmax = fmaxf(vmax, S.redf[i]); vmin = fminf(vmin, S.redf2[i]); }
    __syncthreads();

    // ---- softmax denominator on the unfiltered logits (mmp.py:603)
    float se = 0.f;
#pragma unroll
    for (int it = 0; it < VEC_IT; ++it)
        if (valid(it)) {
#pragma unroll
            for (int c = 0; c < 4; ++c) se += expf(opaque(v[it * 4 + c]) - vmax);
        }
    se = wave_sum(se);
    if (lane == 0) S.redf[wid] = se;
    __syncthreads();
    float sumexp = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) sumexp += S.redf[i];
    __syncthreads();

    // ---- phase B: exact k-th largest key by iterated histogram select
    uint32_t klo = fkey(vmin), khi = fkey(vmax);
    int need = p.k_keep;             // rank (1-based, from the top) inside [klo, khi]
    uint32_t thr = klo;
    const float span = vmax - vmin;
    bool fbins = (span >= 1e-30f) && (span < 3.0e38f);   // level 0: value-linear bins spread a bell curve evenly
    const float inv_w = fbins ? (float)NB / span : 0.f;
    for (int level = 0; level < 8; ++level) {
        if (klo == khi) { thr = klo; break; }
        const uint32_t range = khi - klo;
        const int sh = max(0, 32 - __clz((int)range) - 11);    // (range >> sh) < 2048
        auto bin_of = [&](float x, uint32_t key) -> int {
            if (fbins) {
                const int bb = (int)((x - vmin) * inv_w);
                return min(NB - 1, max(0, bb));
            }
            return (int)((key - klo) >> sh);
        };
        for (int i = tid; i < NB; i += ST) S.hist[i] = 0;
        if (tid == 0) { S.ncand = 0; S.kmin = 0xFFFFFFFFu; S.kmax = 0u; }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < VEC_IT; ++it)
            if (valid(it)) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float x = opaque(v[it * 4 + c]);
                    const uint32_t key = fkey(x);
                    if (key >= klo && key <= khi) atomicAdd(&S.hist[bin_of(x, key)], 1u);
                }
            }
        __syncthreads();
        // one wave finds the bin holding the need-th largest: each lane owns 32 consecutive bins
        if (wid == 0) {
            uint32_t mine = 0;
            for (int j = 0; j < NB / 64; ++j) mine += S.hist[lane * (NB / 64) + j];
            S.lane_sums[lane] = mine;
            __builtin_amdgcn_wave_barrier();
            uint32_t above = 0;                      // elements in lanes above this one
            for (int l2 = lane + 1; l2 < 64; ++l2) above += S.lane_sums[l2];
            if (above < (uint32_t)need && (uint32_t)need <= above + mine) {
                uint32_t acc = above;
                for (int j = NB / 64 - 1; j >= 0; --j) {
                    const uint32_t hcount = S.hist[lane * (NB / 64) + j];
                    if ((uint32_t)need <= acc + hcount) { S.tbin = lane * (NB / 64) + j; S.above = (int)acc; S.cnt = (int)hcount; break; }
                    acc += hcount;
                }
            }
        }
        __syncthreads();
        const int tbin = S.tbin, cnt = S.cnt;
        need -= S.above;
        const bool small = cnt <= CAND_CAP;
        // gather the target bin's members (exact select) or their key range (refine)
#pragma unroll
        for (int it = 0; it < VEC_IT; ++it)
            if (valid(it)) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float x = opaque(v[it * 4 + c]);
                    const uint32_t key = fkey(x);
                    if (key >= klo && key <= khi && bin_of(x, key) == tbin) {
                        if (small) { const int slot = atomicAdd(&S.ncand, 1); S.cand[slot] = key; }
                        else { atomicMin(&S.kmin, key); atomicMax(&S.kmax, key); }
                    }
                }
            }
        __syncthreads();
        if (small) {
            // value t with  #{> t} < need <= #{>= t}  among the candidates
            for (int i = tid; i < cnt; i += ST) {
                const uint32_t ki = S.cand[i];
                int gt = 0, ge = 0;
                for (int j = 0; j < cnt; ++j) { const uint32_t kj = S.cand[j]; gt += kj > ki; ge += kj >= ki; }
                if (gt < need && need <= ge) S.thr = ki;
            }
            __syncthreads();
            thr = S.thr;
            break;
        }
        klo = S.kmin; khi = S.kmax;
        fbins = false;
        __syncthreads();
    }

    // ---- compact the kept entries (key >= threshold) into LDS: one LDS atomic per wave-instruction
    if (tid == 0) S.nkept = 0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < VEC_IT; ++it) {
        const int e = (it * ST + tid) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float x = opaque(v[it * 4 + c]);
            const bool kp = valid(it) && fkey(x) >= thr;
            const unsigned long long bal = __ballot(kp);
            if (bal != 0ull) {
                const int leader = __ffsll((long long)bal) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(&S.nkept, __popcll(bal));
                base = __shfl(base, leader, 64);
                const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
                if (kp && slot < KEPT_CAP) { S.kx[slot] = x; S.ki[slot] = e + c; }
            }
        }
    }
    __syncthreads();
    const int nkept = min(S.nkept, KEPT_CAP);

    // ---- Gumbel argmax over the kept entries (mmp.py:410-411); ties -> lower index like torch.argmax
    const float T = p.temperature;
    float best = -INFINITY, best_x = 0.f;
    int best_i = 0x7FFFFFFF;
    for (int i = tid; i < nkept; i += ST) {
        const float x = S.kx[i];
        const int idx = S.ki[i];
        float g = 0.f;
        if (p.noise_kind == MM_NOISE_GUMBEL) {
            g = p.noise[(size_t)pos_flat * p.noise_ld + idx];
        } else if (p.noise_kind == MM_NOISE_UNIFORM) {
            g = gumbel_of(p.noise[(size_t)pos_flat * p.noise_ld + idx]);
        } else if (p.noise_kind == MM_NOISE_PHILOX) {
            float u[4];
            philox_uniform4(p.seed, p.row_offset + (uint64_t)pos_flat, p.step, (uint32_t)(idx >> 2), u);
            const int sel = idx & 3;
            g = gumbel_of(sel == 0 ? u[0] : sel == 1 ? u[1] : sel == 2 ? u[2] : u[3]);
        }
        const float y = x / T + g;          // IEEE division: same bits as torch's CPU kernel
        if (y > best || (y == best && idx < best_i)) { best = y; best_i = idx; best_x = x; }
    }
    // wave reduce (value desc, index asc)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        const float ox = __shfl_xor(best_x, o, 64);
        if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; best_x = ox; }
    }
    if (lane == 0) { S.bval[wid] = best; S.redi[wid] = best_i; S.bx[wid] = best_x; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < NW; ++i)
            if (S.bval[i] > best || (S.bval[i] == best && S.redi[i] < best_i)) { best = S.bval[i]; best_i = S.redi[i]; best_x = S.bx[i]; }
        const float prob = expf(best_x - vmax) / sumexp;
        const float score = 1.f - prob;
        if (p.ids) p.ids[pos_flat] = (int64_t)best_i;
        if (p.scores) p.scores[pos_flat] = score;
        if (p.pred_out) p.pred_out[row] = (int64_t)best_i;
        if (p.score_out) p.score_out[row] = score;
    }
}

}  // namespace

int k_mask_step(hipStream_t s, float* scores, int64_t* ids, int B, int n, int k, int64_t mask_id, int32_t* rows_out) {
    if (B <= 0) return MM_OK;
    if (n <= 0 || n > 4096 || k < 0 || k > n) return mm_set_error(MM_ERR_SHAPE, "mask_step: need 0 < n <= 4096 and 0 <= k <= n");
    hipLaunchKernelGGL(mask_step_kernel, dim3(B), dim3(256), (size_t)n * 8, s, scores, ids, n, k, mask_id, rows_out);
    return mm_check_launch("mask_step_kernel");
}

int k_sample_rows(hipStream_t s, const SampleArgs& a) {
    if (a.R <= 0) return MM_OK;
    if (a.V <= 0 || (a.V % 4) || a.V > 65536) return mm_set_error(MM_ERR_SHAPE, "sample_rows: V must be a multiple of 4 and <= 65536");
    if (a.k_keep < 1 || a.k_keep > a.V) return mm_set_error(MM_ERR_SHAPE, "sample_rows: k_keep out of range");
    if (a.ld % 4 || ((a.noise_kind == MM_NOISE_GUMBEL || a.noise_kind == MM_NOISE_UNIFORM) && (!a.noise || a.noise_ld % 4)))
        return mm_set_error(MM_ERR_ALIGN, "sample_rows: logits/noise strides must be multiples of 4, noise required in tensor modes");
    if (!(a.temperature > 0.f)) return mm_set_error(MM_ERR_SHAPE, "sample_rows: temperature must be > 0 (clamp to 1e-10 like mmp.py:411)");
    const int vec_it = (a.V + ST * 4 - 1) / (ST * 4);
    dim3 g(a.R), b(ST);
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
