// Device-side helpers shared by every gfx950 kernel in this library.
// Written for CDNA4 only: wave = 64 lanes, MFMA 16x16x32 bf16, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits; all HBM activations/weights use this
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;  // one MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // one 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;  // 16 raw bytes in registers (native vector: HIP's uint4 class
                                                                  // defeats SROA in conditionally-filled staging arrays)

#define MM_WAVE 64

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even (what torch's .to(bfloat16) applies).  Written as __bf16 casts so hipcc emits the gfx950
// hardware conversion v_cvt_pk_bf16_f32 (one instruction per PAIR) instead of ~5 integer ops per value.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    f[0] = bf16lo(v.x); f[1] = bf16hi(v.x); f[2] = bf16lo(v.y); f[3] = bf16hi(v.y);
    f[4] = bf16lo(v.z); f[5] = bf16hi(v.z); f[6] = bf16lo(v.w); f[7] = bf16hi(v.w);
}

__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
    return v;
}

__device__ __forceinline__ f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mfma16(u32x4_t a, u32x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// The same 16 x 16 x 32 product on fp16 operands (v_mfma_f32_16x16x32_f16: same rate, same fragment layout, fp32 accumulation): the matrix
// instruction of the 'f16x2' precision tier (split.hip), whose operands are fp16 TERMS of fp32 values.  F16 = false is the bf16 instruction above.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
template <bool F16>
__device__ __forceinline__ f32x4_t mfma16t(u32x4_t a, u32x4_t b, f32x4_t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// Phi(x) = (1 + erf(x / sqrt 2)) / 2 for the exact (erf) GELU.  libm's erff costs ~100 instructions with a divergent branch and
// the 1 + erf form cancels for x < 0 (4 % relative error at x = -5); this evaluates the TAIL directly,
//     erfc(z) = t * P7(t) * exp(-z^2),  t = 1 / (1 + 0.4 z),  z = |x| / sqrt 2      (rational-argument form of A&S 7.1.26, degree-8
//     least-squares refit: |error| < 1e-9 in exact arithmetic, < 2e-7 in fp32 on Phi, 4e-6 relative on gelu for |x| < 5)
// with one v_rcp_f32, one v_exp_f32 and nine FMAs, branch-free.  Phi(x < 0) = erfc(z) / 2, Phi(x >= 0) = 1 - erfc(z) / 2.
__device__ __forceinline__ float gelu_phi(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.4f, z, 1.f));
    float q = 0.07745829581367633f;
    q = __builtin_fmaf(q, t, -0.4083556420278116f);
    q = __builtin_fmaf(q, t, 0.7089850599675265f);
    q = __builtin_fmaf(q, t, -0.39673678340693536f);
    q = __builtin_fmaf(q, t, 0.4201735656622941f);
    q = __builtin_fmaf(q, t, 0.1368735691355061f);
    q = __builtin_fmaf(q, t, 0.23662785911393874f);
    q = __builtin_fmaf(q, t, 0.2249740761735375f);
    const float h = 0.5f * (q * t) * __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);
    return x < 0.f ? h : 1.f - h;
}

// exact (erf) GELU of the first GEGLU half times the gate half (muse_maskgit_pytorch.py:72-77)
__device__ __forceinline__ float geglu_f(float x, float gate) {
    return gate * (x * gelu_phi(x));
}
// TWO values at once on the packed fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two lanes' worth of arithmetic per instruction at the full VALU
// rate): the same IEEE operations in the same order as gelu_phi / geglu_f on each element -- bit-identical results, ~10 instead of 17 instructions per value
// (the FF w1 epilogue of gemm_wide.hip evaluates 64 per lane and tile)
typedef float mm_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mm_f32x2_t gelu_phi2(mm_f32x2_t x) {
    const mm_f32x2_t one = {1.f, 1.f};
    const mm_f32x2_t z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
    const mm_f32x2_t d = __builtin_elementwise_fma((mm_f32x2_t){0.4f, 0.4f}, z, one);
    const mm_f32x2_t t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    mm_f32x2_t q = {0.07745829581367633f, 0.07745829581367633f};
    q = __builtin_elementwise_fma(q, t, (mm_f32x2_t){-0.4083556420278116f, -0.4083556420278116f});
    q = __builtin_elementwise_fma(q, t, (mm_f32x2_t){0.7089850599675265f, 0.7089850599675265f});
    q = __builtin_elementwise_fma(q, t, (mm_f32x2_t){-0.39673678340693536f, -0.39673678340693536f});
    q = __builtin_elementwise_fma(q, t, (mm_f32x2_t){0.4201735656622941f, 0.4201735656622941f});
    q = __builtin_elementwise_fma(q, t, (mm_f32x2_t){0.1368735691355061f, 0.1368735691355061f});
    q = __builtin_elementwise_fma(q, t, (mm_f32x2_t){0.23662785911393874f, 0.23662785911393874f});
    q = __builtin_elementwise_fma(q, t, (mm_f32x2_t){0.2249740761735375f, 0.2249740761735375f});
    const mm_f32x2_t a = -(z * z) * 1.4426950408889634f;
    const mm_f32x2_t ex = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    const mm_f32x2_t h = (0.5f * (q * t)) * ex;
    const mm_f32x2_t r = one - h;
    return (mm_f32x2_t){x.x < 0.f ? h.x : r.x, x.y < 0.f ? h.y : r.y};
}
__device__ __forceinline__ mm_f32x2_t geglu_f2(mm_f32x2_t x, mm_f32x2_t gate) { return gate * (x * gelu_phi2(x)); }

// LayerNorm(inner) folded into the FF GEMM pair.  Partial sums (sum, sum of squares) of 64 consecutive GEGLU outputs of one row, taken
// where every GEMM kernel of the family has them as bf16 in registers on their way to HBM: 8 adjacent lanes x 16 B of one row.  The 8
// lanes are combined with DPP (no LDS), every lane of the group ends up with the same result.  ONE definition for all kernels: the same
// values in the same order -> bit-identical statistics whichever kernel a shape is dispatched to.
__device__ __forceinline__ float2 ln_partial_row64(const uint4 v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float f[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u); }
    float s1 = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
    float s2 = ((f[0] * f[0] + f[1] * f[1]) + (f[2] * f[2] + f[3] * f[3])) + ((f[4] * f[4] + f[5] * f[5]) + (f[6] * f[6] + f[7] * f[7]));
    // lanes ^1, ^2 (quad permutes) and the mirrored half row (lane i <-> 7 - i of each 8): commutative pairwise sums, identical in all 8 lanes
#define MM_DPP_ADD(x_, ctrl_) x_ += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x_), ctrl_, 0xF, 0xF, true))
    MM_DPP_ADD(s1, 0xB1); MM_DPP_ADD(s2, 0xB1);      // quad_perm [1,0,3,2]
    MM_DPP_ADD(s1, 0x4E); MM_DPP_ADD(s2, 0x4E);      // quad_perm [2,3,0,1]
    MM_DPP_ADD(s1, 0x141); MM_DPP_ADD(s2, 0x141);    // row_half_mirror
#undef MM_DPP_ADD
    return make_float2(s1, s2);
}
// per-row (mean, rstd) from the np partials, F valid features (eps = 1e-5 like nn.LayerNorm).  Called by TWO ADJACENT LANES per row
// (half = lane & 1): each sums its half of the partials in index order, the halves are added as (first + second) -- one canonical
// order for every kernel; the loads go out in batches of 8 (a plain loop waits for every single load: one L2 round trip per partial).
// Valid in the lane with half == 0.
__device__ __forceinline__ float2 ln_stats_from_partials(const float* part, int np, long row, int F, int half, bool row_ok) {
    const int n0 = (np + 1) >> 1;
    const int cnt = half ? np - n0 : n0;
    const float2* pp = reinterpret_cast<const float2*>(part) + (size_t)row * np + (half ? n0 : 0);
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < n0; i += 8) {
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (row_ok && i + j < cnt) ? pp[i + j] : make_float2(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1 += v[j].x; s2 += v[j].y; }
    }
    const float o1 = __shfl_xor(s1, 1, 64), o2 = __shfl_xor(s2, 1, 64);
    s1 += o1; s2 += o2;                                       // (in the half == 0 lane: first half + second half)
    const float mean = s1 / (float)F;
    const float var = fmaxf(s2 / (float)F - mean * mean, 0.f);
    return make_float2(mean, 1.f / sqrtf(var + 1e-5f));
}

// the same (mean, rstd) from np partials held by ONE lane (pp: the row's np float2 partials, e.g. in LDS): first half summed in index order, second half
// summed in index order, the halves added -- bit-identical to ln_stats_from_partials (adding to 0.f first is exact)
__device__ __forceinline__ float2 ln_stats_seq(const float2* pp, int np, int F) {
    const int n0 = (np + 1) >> 1;
    float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
    for (int i = 0; i < n0; ++i) { const float2 v = pp[i]; a1 += v.x; a2 += v.y; }
    for (int i = n0; i < np; ++i) { const float2 v = pp[i]; b1 += v.x; b2 += v.y; }
    const float s1 = a1 + b1, s2 = a2 + b2;
    const float mean = s1 / (float)F;
    const float var = fmaxf(s2 / (float)F - mean * mean, 0.f);
    return make_float2(mean, 1.f / sqrtf(var + 1e-5f));
}

// LayerNorm(dim) fold: per-row (rstd, -mean) from the np (sum, sum of squares) partials the producer epilogues left, by ONE lane -- summed in index
// order, then multiplications by 1 / F and the hardware reciprocal square root instead of three IEEE divisions and a square root (a dependent chain the
// consumer GEMM waits for once per tile; rsq is within 1 ulp: far inside what the bf16 operands resolve).  One definition for every consumer kernel.
__device__ __forceinline__ float2 ln_rstd_negmean(const float2* pp, int np, float inv_F) {
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < np; ++i) { const float2 v = pp[i]; s1 += v.x; s2 += v.y; }
    const float mean = s1 * inv_F;
    const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * inv_F), 0.f);
    return make_float2(__builtin_amdgcn_rsqf(var + 1e-5f), -mean);
}

// LayerNorm(dim) fold, CONSUMER side: LayerNorm(x) . W^T for one output = rstd * (acc - mean * c1) + c2 with acc = x . Wg^T -- two explicit fused
// multiply-adds (the build runs with -ffp-contract=off; these are written as FMAs so that the 128-value accumulator sweep costs 2 instructions per value)
__device__ __forceinline__ float ln_fold_apply(float acc, float rstd, float neg_mean, float c1, float c2) {
    return __builtin_fmaf(rstd, __builtin_fmaf(neg_mean, c1, acc), c2);
}

// LayerNorm(dim) folded into the GEMMs around it (round 4; GemmArgs::st_part / in_c1).  PRODUCER side: the fp32-residual epilogues (out = x + ...) hand 4
// consecutive columns of one row to 32 adjacent lanes (one 128-column tile row): the row's (sum, sum of squares) over the 64 fp32 values of each 16-lane
// DPP row, identical in its 16 lanes (fixed pairing).
__device__ __forceinline__ float2 row_stats16(float s1, float s2) {      // ... per 16 lanes = 64 columns: two partials per row and 128-column tile, no cross-row exchange
#define MM_DPP_ADD(x_, ctrl_) x_ += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x_), ctrl_, 0xF, 0xF, true))
    MM_DPP_ADD(s1, 0xB1); MM_DPP_ADD(s2, 0xB1);      // quad_perm [1,0,3,2]
    MM_DPP_ADD(s1, 0x4E); MM_DPP_ADD(s2, 0x4E);      // quad_perm [2,3,0,1]
    MM_DPP_ADD(s1, 0x141); MM_DPP_ADD(s2, 0x141);    // row_half_mirror
    MM_DPP_ADD(s1, 0x140); MM_DPP_ADD(s2, 0x140);    // row_mirror
#undef MM_DPP_ADD
    return make_float2(s1, s2);
}

// full-wave (64-lane) butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 16-lane-row reductions on DPP (no LDS crossbar traffic) + 4 readlanes across the rows: every lane ends up with the wave's value.  Fixed order
// -> deterministic.  Used where a reduction rides inside an LDS-issue-bound k-loop (the fused-sampling emission of gemm_cfg.hip).
#define MM_DPP_F(x_, ctrl_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x_), ctrl_, 0xF, 0xF, true))
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, MM_DPP_F(v, 0xB1)); v = fmaxf(v, MM_DPP_F(v, 0x4E)); v = fmaxf(v, MM_DPP_F(v, 0x141)); v = fmaxf(v, MM_DPP_F(v, 0x140));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += MM_DPP_F(v, 0xB1); v += MM_DPP_F(v, 0x4E); v += MM_DPP_F(v, 0x141); v += MM_DPP_F(v, 0x140);
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (a + b) + (c + d);
}
#undef MM_DPP_F

// ------------------------------------------------------------------------------------------------ Philox2x32-10 and the Gumbel transform
// ONE definition for the logits-path sampler (sampling.hip) and the fused finisher (sampling_fused.hip): the two paths must draw bit-identical
// noise.  Counter-based: the uniform for (seed, global token row, decode step, vocabulary index v) is output (v & 1) of Philox2x32-10 with the
// 64-bit counter  row << 24 | (step & 0xFF) << 16 | (v >> 1)  and a 32-bit key mixed from the seed.  One call serves two adjacent vocabulary
// entries; the samplers only evaluate it for the kept entries.
__device__ __forceinline__ void philox2x32_10(uint32_t c0, uint32_t c1, uint32_t k, uint32_t (&out)[2]) {
    // the key schedule lives in vector registers: as scalar values the ten round keys were hoisted out of the samplers' loops, spilled (both kernels
    // are short of scalar registers) and read back with v_readlane + wait states inside every round
    asm volatile("" : "+v"(k));
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p = (uint64_t)0xD256D193u * c0;
        const uint32_t n0 = (uint32_t)(p >> 32) ^ k ^ c1;
        c1 = (uint32_t)p;
        c0 = n0;
        k += 0x9E3779B9u;
    }
    out[0] = c0; out[1] = c1;
}
__device__ __forceinline__ void philox_uniform2(uint64_t seed, uint64_t row_global, uint32_t step, uint32_t col2, float (&u)[2]) {
    const uint64_t ctr = (row_global << 24) | ((uint64_t)(step & 0xFFu) << 16) | (uint64_t)(col2 & 0xFFFFu);
    const uint32_t key = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x85EBCA6Bu) ^ ((step >> 8) * 0xC2B2AE35u);
    uint32_t o[2];
    philox2x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), key, o);
    u[0] = (float)(o[0] >> 8) * (1.0f / 16777216.0f);   // 24-bit, [0, 1)
    u[1] = (float)(o[1] >> 8) * (1.0f / 16777216.0f);
}
// -log(-log(u)) with the reference's clamps (muse_maskgit_pytorch.py:403-408)
__device__ __forceinline__ float gumbel_of(float u) {
    const float a = logf(fmaxf(u, 1e-20f));
    return -logf(fmaxf(-a, 1e-20f));
}
// log(x) for NORMAL, finite, positive x: the operation sequence of the device library's logf on that domain -- v_log_f32, then the product with ln 2
// as a compensated two-term multiply -- without its denormal pre-scaling and infinity pass-through (8 of its 13 instructions).
__device__ __forceinline__ float log_normal_pos(float x) {
    const float r = __builtin_amdgcn_logf(x);
    const float ch = __uint_as_float(0x3F317217u), cl = __uint_as_float(0x3377D1CFu);      // ln 2 = ch + cl
    const float y = r * ch;
    float t = __builtin_fmaf(r, ch, -y);
    t = __builtin_fmaf(r, cl, t);
    return y + t;
}
// gumbel_of for a Philox uniform (0 or k 2^-24, k < 2^24): both logarithms see normal arguments (>= 1e-20 resp. >= 5.9e-8), same values as gumbel_of
__device__ __forceinline__ float gumbel_of_unit(float u) {
    const float a = log_normal_pos(fmaxf(u, 1e-20f));
    return -log_normal_pos(fmaxf(-a, 1e-20f));
}

// Fused sampling (sampling_fused.hip): what leaves the guidance-logits GEMM instead of the logits.  One 256-column piece of a logits row is 128 GRANULES of
// two adjacent columns; a granule whose larger value reaches thr is kept: the kept granules (float2) are stored compacted, in column order, into the
// piece's slot (at most 128 x 8 B: the slot cannot overflow), and the piece's record is two float4s: {max, sum exp(x - max), -, -} and the 128-bit mask of
// the kept granules.  No atomics.
constexpr int FS_SLOT = 64;      // float4 units of a (row, piece) candidate slot = 128 float2 entries
constexpr int FS_REC = 2;        // float4 units of a (row, piece) record
// Softmax statistics of one 256-column piece of a logits row and the order of its candidate granules, in ONE canonical form shared by every
// producer -- the guidance GEMM's epilogue (from its accumulator fragments, gemm_cfg.hip), fused_emit (from materialised logits) and the
// logits-path sampler (sampling.hip) -- so that a row's softmax denominator, its confidence 1 - p and the next step's re-masking are
// bit-identical whichever path sampled the row.  The form is the one the GEMM's accumulator layout gives WITHOUT any lane exchange: a piece is 4
// quarters q of 64 columns (one per wave of the GEMM's vocabulary split); in a quarter, column c = 16 a + 4 f + r is value r of granule (a, f) --
// GEMM: accumulator fragment a of lane group f, which holds the 4 granules a = 0..3 of its f.  Per LANE GROUP (q, f):
//     ml = max of its 16 values;   g(a) = (e0 + e1) + (e2 + e3),  e_r = exp2((x_r - ml) log2 e)  (fs_exp: v_exp_f32);
//     pl = (g(0) + g(1)) + (g(2) + g(3))
// and over the 16 groups of the piece:
//     M = max ml;   t(q, f) = pl * exp(ml - M);   w(q) = (t(q,0) + t(q,1)) + (t(q,2) + t(q,3));   E = (w(0) + w(1)) + (w(2) + w(3))
// Candidate granules (two adjacent columns, kept when the larger value reaches the row's bound) are numbered in column order J = column / 2 =
// 32 q + 8 a + 2 f + h (h: the half of lane group f's 4 values); word q of the piece's 128-bit mask holds quarter q, bit 8 a + 2 f + h.  Granule J sits at
// position (number of kept granules in front of it) of the slot: ONE compacted list per (row, piece).  (Round 3 began with 4-column granules in four
// per-quarter sub-slots -- a wave could store without waiting for the other quarters -- but 40 % of those granules were kept for 12 % of the values and
// every sub-slot cost its own 128-byte line: 0.69 GB written and 0.80 GB fetched per step for 0.16 GB of values.  With 2-column granules 23 % are
// kept and the list is dense; the GEMM's waves rank their granules behind the exchange they need anyway for the statistics.)
__device__ __forceinline__ int fs_pos(const uint4 m, int q, int gq) {      // position of granule gq of quarter q in the slot
    int base = 0;
    if (q > 0) base += __popc(m.x);
    if (q > 1) base += __popc(m.y);
    if (q > 2) base += __popc(m.z);
    const uint32_t mq = q == 0 ? m.x : (q == 1 ? m.y : (q == 2 ? m.z : m.w));
    return base + __popc(mq & ((1u << gq) - 1u));
}
__device__ __forceinline__ uint32_t fs_interleave16(uint32_t even, uint32_t odd) {      // bit 2 i = bit i of `even`, bit 2 i + 1 = bit i of `odd` (16 bits each)
    uint32_t a = even & 0xFFFFu, b = odd & 0xFFFFu;
    a = (a | (a << 8)) & 0x00FF00FFu; a = (a | (a << 4)) & 0x0F0F0F0Fu; a = (a | (a << 2)) & 0x33333333u; a = (a | (a << 1)) & 0x55555555u;
    b = (b | (b << 8)) & 0x00FF00FFu; b = (b | (b << 4)) & 0x0F0F0F0Fu; b = (b | (b << 2)) & 0x33333333u; b = (b | (b << 1)) & 0x55555555u;
    return a | (b << 1);
}
// exp(x - ml) as v_exp_f32((x - ml) * log2 e): the subtraction first -- exact near the maximum, so the largest value contributes exactly 1 (an FMA
// form x * log2 e - ml * log2 e rounds the product of the MAGNITUDES: 2e-6 off at |ml| ~ 40, enough to push a dominant token's 1 - p below 0)
__device__ __forceinline__ float fs_exp(float x, float ml) { return __builtin_amdgcn_exp2f((x - ml) * 1.4426950408889634f); }
// g(a) = (e0 + e1) + (e2 + e3) of one accumulator fragment with the subtraction and the scaling as PACKED fp32 operations (v_pk_add_f32 / v_pk_mul_f32: two
// values per instruction at the full VALU rate) -- the same IEEE operations on the same operands in the same order as four fs_exp calls, bit-identical
typedef float fs_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fs_exp_sum4(float x0, float x1, float x2, float x3, float ml) {
    const fs_f32x2_t m2 = {ml, ml}, l2 = {1.4426950408889634f, 1.4426950408889634f};
    const fs_f32x2_t a = {x0, x1}, b = {x2, x3};
    const fs_f32x2_t da = (a - m2) * l2, db = (b - m2) * l2;
    const fs_f32x2_t e02 = {__builtin_amdgcn_exp2f(da.x), __builtin_amdgcn_exp2f(db.x)}, e13 = {__builtin_amdgcn_exp2f(da.y), __builtin_amdgcn_exp2f(db.y)};
    const fs_f32x2_t s = e02 + e13;                   // (e0 + e1, e2 + e3)
    return s.x + s.y;
}
__device__ __forceinline__ void tile_combine16(const float (&ml)[16], const float (&pl)[16], float& M, float& E) {
    float m = ml[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = fmaxf(m, ml[i]);
    float w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        w[q] = (pl[4 * q] * __expf(ml[4 * q] - m) + pl[4 * q + 1] * __expf(ml[4 * q + 1] - m)) + (pl[4 * q + 2] * __expf(ml[4 * q + 2] - m) + pl[4 * q + 3] * __expf(ml[4 * q + 3] - m));
    M = m;
    E = (w[0] + w[1]) + (w[2] + w[3]);
}
// ROW layout (lane l holds columns 4 l .. 4 l + 3 of the piece: quarter q = l >> 4, granule a = (l >> 2) & 3 of lane group f = l & 3); every
// lane returns the piece's (M, E)
__device__ __forceinline__ void tile_softmax_stats(const float4 x, float& M, float& E) {
#define MM_DPP_F(x_, ctrl_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x_), ctrl_, 0xF, 0xF, true))
    float ml = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
    ml = fmaxf(ml, __shfl_xor(ml, 4, 64));      // over a (lanes f, 4 + f, 8 + f, 12 + f of the 16-lane row)
    ml = fmaxf(ml, __shfl_xor(ml, 8, 64));
    float pl = (fs_exp(x.x, ml) + fs_exp(x.y, ml)) + (fs_exp(x.z, ml) + fs_exp(x.w, ml));      // g(a)
    pl += __shfl_xor(pl, 4, 64);                // g(a) + g(a ^ 1)
    pl += __shfl_xor(pl, 8, 64);                // (g(0) + g(1)) + (g(2) + g(3))
    float mr = fmaxf(ml, MM_DPP_F(ml, 0xB1));   // over f: the row's (quarter's) max
    mr = fmaxf(mr, MM_DPP_F(mr, 0x4E));
    float m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mr), 0));
#pragma unroll
    for (int q = 1; q < 4; ++q) m = fmaxf(m, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mr), 16 * q)));
    float t = pl * __expf(ml - m);
    t += MM_DPP_F(t, 0xB1);                     // t(q, f) + t(q, f ^ 1)
    t += MM_DPP_F(t, 0x4E);                     // w(q)
#undef MM_DPP_F
    float w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 16 * q));
    M = m;
    E = (w[0] + w[1]) + (w[2] + w[3]);
}
__device__ __forceinline__ void fused_emit_piece(const float4 x, int row, int tile, int NT, int lane, float thr, float4* __restrict__ stats,
                                                 float4* __restrict__ cand) {
    float m, e;
    tile_softmax_stats(x, m, e);
    // row layout: lane l holds granules 2 l (x.x, x.y) and 2 l + 1 (x.z, x.w) of the piece
    const bool kp0 = fmaxf(x.x, x.y) >= thr, kp1 = fmaxf(x.z, x.w) >= thr;
    const unsigned long long b0 = __ballot(kp0), b1 = __ballot(kp1);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int pos0 = __popcll(b0 & below) + __popcll(b1 & below);
    float2* slot = reinterpret_cast<float2*>(cand + ((size_t)row * NT + tile) * FS_SLOT);
    if (kp0) slot[pos0] = make_float2(x.x, x.y);
    if (kp1) slot[pos0 + (kp0 ? 1 : 0)] = make_float2(x.z, x.w);
    if (lane == 0) {
        float4* rec = stats + ((size_t)row * NT + tile) * FS_REC;
        rec[0] = make_float4(m, e, 0.f, 0.f);
        rec[1] = make_float4(__uint_as_float(fs_interleave16((uint32_t)b0, (uint32_t)b1)), __uint_as_float(fs_interleave16((uint32_t)(b0 >> 16), (uint32_t)(b1 >> 16))),
                             __uint_as_float(fs_interleave16((uint32_t)(b0 >> 32), (uint32_t)(b1 >> 32))), __uint_as_float(fs_interleave16((uint32_t)(b0 >> 48), (uint32_t)(b1 >> 48))));
    }
}

// ---- 'bf16x3' precision tier (split.hip): an fp32 value as the exact sum of three bf16 terms, x = h + m + l
__device__ __forceinline__ void split3(float x, float& h, float& m, float& l) {
    h = bf16_to_f32(f32_to_bf16(x));
    const float r1 = x - h;
    m = bf16_to_f32(f32_to_bf16(r1));
    l = r1 - m;                           // <= 8 significant bits: its bf16 conversion is exact
}

// which term (0 = h, 1 = m, 2 = l) segment s of X' carries: [h m l h m h]
__device__ __forceinline__ int seg_term(int s) { return s < 3 ? s : (s < 5 ? s - 3 : 0); }

// ---- 'f16x2' precision tier: an fp32 value as the sum of TWO fp16 terms, x ~ h + l (11 + 11 significand bits: relative error <= 2^-22, and
// exact whenever x has <= 22 significant bits; terms below 2^-14 are fp16 subnormals -- the matrix pipe takes them un-flushed -- so the absolute
// floor is 2^-25).  A product of two such sums keeps the pairs h.h, l.h, h.l (the dropped l.l is < 2^-22 of the leading one): X' = [xh | xl | xh]
// against W' = [wh | wh | wl] -- THREE products for general fp32 weights where the bf16 split needs six, TWO ([xh | xl] . [wh | wh]) when every
// weight is a single fp16 term (any bf16-representable checkpoint).  Operand codes: products | MM_SPLIT_F16 (muse_hip.h).
#define MM_SPLIT_F16_BIT 0x100
// Producer-side flag on an operand code (round 5, internal): the repeated segment ([h | l | h]: segment 2) is NOT written -- every consumer of the buffer runs a
// term-sharing k-loop (gemm_terms.hip and the NP forms), which stages the h plane once and never reads the repeat.  Row stride and segment offsets unchanged.
#define MM_SPLIT_NODUP_BIT 0x400
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }      // round-to-nearest-even
__device__ __forceinline__ float f16_bits_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
// Range guard: both terms saturate at the largest finite fp16 (65504) instead of overflowing to infinity -- the pair then still represents |x| up to
// 131008 exactly-ish (h = 65504, l = the rest) and stays FINITE beyond (the value is clipped: an activation of that size is outside what the tier can multiply
// accurately, but one such element no longer turns a whole logits row into NaN through h = inf, l = x - inf = -inf).  One v_med3_f32 per term.
// 16-bit activation storage chosen at compile time (round 6): bf16 (the default engine) or fp16 -- the single-term fp16 VAE decode (vae_model.hip `half`): the
// same MFMA rate, 11 instead of 8 significand bits (decoded pixels 2e-4 instead of 1.6e-3 of the image scale), values saturated at the largest finite fp16.
__device__ __forceinline__ uint32_t pack_f16x2_sat(float a, float b) {
    return (uint32_t)f32_to_f16_bits(__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f)) | ((uint32_t)f32_to_f16_bits(__builtin_amdgcn_fmed3f(b, -65504.f, 65504.f)) << 16);
}
template <bool H> __device__ __forceinline__ uint4 pack8s(const float (&f)[8]) {
    if constexpr (H) return make_uint4(pack_f16x2_sat(f[0], f[1]), pack_f16x2_sat(f[2], f[3]), pack_f16x2_sat(f[4], f[5]), pack_f16x2_sat(f[6], f[7]));
    else return pack8(f);
}
template <bool H> __device__ __forceinline__ void unpack8s(const uint4& v, float (&f)[8]) {
    if constexpr (H) {
        f[0] = f16_bits_to_f32((uint16_t)v.x); f[1] = f16_bits_to_f32((uint16_t)(v.x >> 16)); f[2] = f16_bits_to_f32((uint16_t)v.y); f[3] = f16_bits_to_f32((uint16_t)(v.y >> 16));
        f[4] = f16_bits_to_f32((uint16_t)v.z); f[5] = f16_bits_to_f32((uint16_t)(v.z >> 16)); f[6] = f16_bits_to_f32((uint16_t)v.w); f[7] = f16_bits_to_f32((uint16_t)(v.w >> 16));
    } else {
        unpack8(v, f);
    }
}
template <bool H> __device__ __forceinline__ float ld16s(bf16_t h) {
    if constexpr (H) return f16_bits_to_f32(__builtin_bit_cast(uint16_t, h)); else return bf16_to_f32(h);
}
template <bool H> __device__ __forceinline__ bf16_t st16s(float f) {
    if constexpr (H) return __builtin_bit_cast(bf16_t, f32_to_f16_bits(__builtin_amdgcn_fmed3f(f, -65504.f, 65504.f))); else return f32_to_bf16(f);
}
__device__ __forceinline__ void split2_f16(float x, uint16_t& h, uint16_t& l) {
    h = f32_to_f16_bits(__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f));
    l = f32_to_f16_bits(__builtin_amdgcn_fmed3f(x - f16_bits_to_f32(h), -65504.f, 65504.f));      // (the difference is exact in fp32)
}
__host__ __device__ __forceinline__ int split_count(int code) { return code & 0xff; }            // segments per operand row
__host__ __device__ __forceinline__ bool split_is_f16(int code) { return (code & MM_SPLIT_F16_BIT) != 0; }

// four consecutive values of one row -> the P segments of that row (8-byte stores).  P: 3 / 5 / 6 (bf16 terms) or MM_SPLIT_F16 | 2 / 3 (fp16 terms, [h l h])
__device__ __forceinline__ void store_split4(bf16_t* orow, int K, int P, int col, const float (&v)[4]) {
    if (split_is_f16(P)) {
        uint16_t h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split2_f16(v[j], h[j], l[j]);
        const uint2 hv = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
        *reinterpret_cast<uint2*>(orow + col) = hv;
        *reinterpret_cast<uint2*>(orow + (long)K + col) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
        if (split_count(P) > 2 && !(P & MM_SPLIT_NODUP_BIT)) *reinterpret_cast<uint2*>(orow + 2l * K + col) = hv;
        return;
    }
    float t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split3(v[j], t[0][j], t[1][j], t[2][j]);
#pragma unroll
    for (int s = 0; s < 6; ++s) {          // unrolled: the term index is a compile-time constant (no scratch for t)
        if (s < P) {
            const int k = seg_term(s);
            *reinterpret_cast<uint2*>(orow + (long)s * K + col) = make_uint2(pack_bf16x2(t[k][0], t[k][1]), pack_bf16x2(t[k][2], t[k][3]));
        }
    }
}

// XCD-aware tile order: block b is dispatched to XCD b % 8 (observed, used for speed only), so give
// every XCD one contiguous run of the linear tile index (bijective for any tile count), then walk
// that run in groups of GROUP_M row-tiles so a group's activation tiles stay in the XCD's 4 MiB L2
// while the weight tiles stream past.
__device__ __forceinline__ void xcd_grouped_tile(int bid, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
    const int total = tiles_m * tiles_n;
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int gsz = group_m * tiles_n;
    const int g = lin / gsz;
    const int first_m = g * group_m;
    const int gm = min(tiles_m - first_m, group_m);
    const int in_g = lin - g * gsz;
    tm = first_m + in_g % gm;
    tn = in_g / gm;
}
