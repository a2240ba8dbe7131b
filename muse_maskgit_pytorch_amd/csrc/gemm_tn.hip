// Weight gradients without transposed copies (round 6):   dW [N][K] (fp32) = dY^T X = sum over rows m of dY[m][n] * X[m][k]
// for dY bf16 [rows][ldy] and X bf16 [rows][ldx], both ROW-major with the contraction index m as the SLOW axis ("TN").  Rounds 1-6 ran every dW of the training
// step as transpose(dY) + transpose(X) + an NT GEMM (the MFMA wants 8 consecutive contraction values per lane, i.e. m contiguous): 107 transposes per step, 2.2 ms
// of kernel time on two busy streams.  gfx950's transposing LDS read makes the copies unnecessary: a 32-row x 16-column block of an operand is copied by ONE
// LDS-DMA instruction as [32 m][16 n] (32-byte rows, 1 KiB: lane l fetches the 16 bytes of row l >> 1, half l & 1) and ds_read_b64_tr_b16 hands lane c the column c
// of four rows -- two reads give the 8 contraction slots of a 16 x 16 x 32 MFMA operand.  With the slot <-> row assignment  slot j of lane group fg = row
// (j >> 2) * 16 + 4 fg + (j & 3)  the four lane groups of a read touch 512 contiguous bytes (the layout attention.hip's P.V uses: the one the LDS serves without
// bank conflicts for this instruction), and BOTH operands use it, so the products pair up correctly whatever the order.
// One 256-thread workgroup = a 128 (n) x 128 (k) tile of dW over one split of the rows; wave (wn, wk) owns 64 x 64 (4 x 4 accumulator fragments).  A stage = 64
// rows of both operands (32 KiB), double-buffered, one barrier per stage; rows past the end read as zeros through the buffer descriptor (no padded copies either).
// Output: fp32 slab of the split (the caller sums the slabs in a fixed order, k_colsum), row n, 16 consecutive k per lane group -- 64-byte segments.
#include <string.h>

#include "common.h"
#include "muse_hip_internal.h"

namespace {

constexpr int TT = 128;                         // tile edge (n and k)
constexpr int TSTEP = 64;                       // rows per stage
constexpr int OP_B = TSTEP * TT * 2;            // bytes of one operand's stage: [2 sub-steps][8 column blocks][32 rows][16 columns]
constexpr int TN_SMEM = 2 * 2 * OP_B;           // two operands, two stages: 64 KiB

typedef short v4i16_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 tn_read_tr16(const unsigned char* ptr) {
    typedef __attribute__((address_space(3))) v4i16_t* lds_v4_t;
    const v4i16_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)ptr);
    return __builtin_bit_cast(uint2, r);
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb, int rows, int N, int K,
                                                          int rows_per_split, float* __restrict__ out, long split_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = w >> 1, wk = w & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_k = K / TT;
    const int tile_n = blockIdx.x / tiles_k, tile_k = blockIdx.x % tiles_k;
    const int n0 = tile_n * TT, k0 = tile_k * TT;
    const int m_begin = blockIdx.y * rows_per_split;
    const int m_end = max(m_begin, min(rows, m_begin + rows_per_split));      // (an empty split writes a slab of zeros)
    const int nst = (m_end - m_begin + TSTEP - 1) / TSTEP;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    // buffer descriptors over the split's rows: a row past m_end is out of range -> the DMA writes zeros
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A + (size_t)m_begin * lda), 0,
                                                                         (unsigned)((size_t)(m_end - m_begin) * lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B + (size_t)m_begin * ldb), 0,
                                                                         (unsigned)((size_t)(m_end - m_begin) * ldb * 2), 0x00020000);
    // one DMA instruction = [32 rows][16 columns] of one operand: lane l -> row l >> 1, columns 8 (l & 1) .. + 7.  A stage has 2 sub-steps x 8 column blocks x 2
    // operands = 32 instructions: 8 per wave -- wave w takes column blocks 2 w, 2 w + 1 of both sub-steps of both operands
    const int vrow = lane >> 1, vcol = (lane & 1) * 8;
    int voff_a[2], voff_b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        voff_a[i] = (int)((vrow * lda + n0 + (2 * w + i) * 16 + vcol) * 2);
        voff_b[i] = (int)((vrow * ldb + k0 + (2 * w + i) * 16 + vcol) * 2);
    }
#define TN_ISSUE(st_, stage_)                                                                                                          \
    {                                                                                                                                  \
        unsigned char* sa_ = tsm + (stage_) * 2 * OP_B;                                                                                \
        unsigned char* sb_ = sa_ + OP_B;                                                                                               \
        _Pragma("unroll") for (int ss = 0; ss < 2; ++ss) {                                                                             \
            const int soff_a_ = (int)(((st_) * TSTEP + ss * 32) * lda * 2), soff_b_ = (int)(((st_) * TSTEP + ss * 32) * ldb * 2);      \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                            \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(sa_ + ss * (OP_B / 2) + (2 * w + i) * 1024), 16, voff_a[i], soff_a_, 0, 0); \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(sb_ + ss * (OP_B / 2) + (2 * w + i) * 1024), 16, voff_b[i], soff_b_, 0, 0); \
            }                                                                                                                          \
        }                                                                                                                              \
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (nst > 0) TN_ISSUE(0, 0);
    for (int st = 0; st < nst; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();      // stage st has landed for everybody; everybody is done reading the other stage
        if (st + 1 < nst) TN_ISSUE(st + 1, (st + 1) & 1);
        const unsigned char* sa = tsm + (st & 1) * 2 * OP_B;
        const unsigned char* sb = sa + OP_B;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            uint4 af[4], bf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const unsigned char* pa = sa + ss * (OP_B / 2) + (wn * 4 + a) * 1024 + lane * 8;
                const uint2 lo = tn_read_tr16(pa), hi = tn_read_tr16(pa + 512);
                af[a] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned char* pb = sb + ss * (OP_B / 2) + (wk * 4 + b) * 1024 + lane * 8;
                const uint2 lo = tn_read_tr16(pb), hi = tn_read_tr16(pb + 512);
                bf[b] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(af[a], bf[b], acc[a][b]);
        }
    }
#undef TN_ISSUE
    // accumulator fragment (a, b): lane (fr, fg) holds dW[n0 + 64 wn + 16 a + 4 fg + r][k0 + 64 wk + 16 b + fr], r = 0..3
    float* o = out + (size_t)blockIdx.y * split_stride;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn * 64 + a * 16 + 4 * fg + r;
            float* orow = o + (size_t)n * K + k0 + wk * 64 + fr;
#pragma unroll
            for (int b = 0; b < 4; ++b) orow[b * 16] = acc[a][b][r];
        }
}

}  // namespace

// shapes the kernel takes: whole 128 x 128 tiles, 16-byte aligned rows
bool k_gemm_tn_eligible(int rows, int N, int K, long lda, long ldb) {
    return rows > 0 && N > 0 && K > 0 && (N % TT) == 0 && (K % TT) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 && (size_t)rows * (size_t)(lda > ldb ? lda : ldb) * 2 < ((size_t)1 << 32);
}
// where the step prefers this kernel to transposed copies + the NT GEMM (measured, tools/tn_gemm_timing.py, us per launch incl. the slab sum; NT + the dY transpose
// the step pays): 512 x 512 over 8192 rows 21.5 vs 28.5, the context's k | v 15-17 vs 25, q|k|v 36 vs 41.5, w1 59 vs 62, the head 617 vs 877 (the vocabulary-wide dl
// no longer crosses HBM twice more) -- and w2 (K = 1408) 38 vs 37.6: a tie that keeps its transposes.  Per launch this kernel runs at 0.2-0.6 PFLOP/s against the NT
// kernels' 0.45-0.9: ablations (plain instead of transposing LDS reads: no change; no stores: -3 %; no loads after the first stage: -50 %) put the difference on the
// global -> LDS feed, not on the tr reads: at the head's shape the 2048 workgroups of 128 x 128 tiles pull 5.7 GB through L2 -> LDS in 616 us = 9.3 TB/s, the same
// rate the 256 x 128 NT kernel reaches on 4.3 GB (406 us) -- the tile's 64 flop per staged byte is the limit.  A 256 x 128 tile (512 threads, three stages: 681 us) and a
// four-deep ring of 32-row stages (702 us) measured no better: with one workgroup per CU / half the MFMAs per barrier the waits are exposed instead.
bool k_gemm_tn_prefer(int rows, int N, int K, long lda, long ldb) {
    // (taking q|k|v and w1 as well -- every shape with K <= 512 -- measured 11.49 vs 11.50 ms per step: their launches win in isolation but load the memory system
    //  beside the dependent chain; the small projections + the head alone: 11.32 vs 11.47)
    return !(g_mm_debug2 & 1024) && k_gemm_tn_eligible(rows, N, K, lda, ldb) && ((N <= 1024 && K <= 512 && (long)N * K <= 512 * 1024) || N >= 8192);      // (bit 1024: A/B)
}
// split count: one workgroup per CU at least, at least 512 rows per split, splits of whole 64-row stages
int k_gemm_tn_splits(int rows, int N, int K) {
    const long tiles = (long)(N / TT) * (K / TT);
    int s = 1;
    while (tiles * s < 256 && rows / (s * 2) >= 512) s *= 2;      // (512: twice the slabs for the column sum to read -- 62 / 44 / 43 us instead of 59 / 38 / 36 at the w1 / w2 / q|k|v shapes)
    return s;
}
int k_gemm_tn(hipStream_t s, const bf16_t* A, long lda, const bf16_t* B, long ldb, int rows, int N, int K, int splits, float* out_or_slabs) {
    if (!k_gemm_tn_eligible(rows, N, K, lda, ldb)) return mm_set_error(MM_ERR_SHAPE, "gemm_tn: N and K multiples of 128, strides multiples of 8 elements");
    if (splits < 1) splits = 1;
    int per = (rows + splits - 1) / splits;
    per = (per + TSTEP - 1) / TSTEP * TSTEP;
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TN_SMEM);
        if (e != hipSuccess) return mm_set_hip_error(e, "gemm_tn hipFuncSetAttribute");
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((N / TT) * (K / TT), splits), dim3(256), TN_SMEM, s, A, lda, B, ldb, rows, N, K, per, out_or_slabs, (long)N * K);
    return mm_check_launch("gemm_tn_kernel");
}
