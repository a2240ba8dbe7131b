// fp8 GEMM of the fp8 engine (BASELINE configs[4] "fp8 MFMA weights"; the bias-free Linear layers mmp.py:85,88,118-124,233):
//     out[m][n] = sx[m] * sw[n] * sum_k xq[m][k] * wq[n][k]
// xq / wq are OCP e4m3 rows (activations quantised per token row by the producing kernel, weights per output row at pack time), the products run on
// v_mfma_f32_16x16x128_f8f6f4 (gfx950's K = 128 fp8 instruction: twice the bf16 MFMA rate, half the operand bytes per k), fp32 accumulation, the two
// per-row scales multiply the fp32 accumulators in the epilogue.  Self-defined numerics (the reference has no fp8): the oracle is the fp32 restatement
// on the de-quantised operands (oracle/muse_oracle.py fake-quant hooks).
//
// One 512-thread workgroup = 8 waves (2 token halves x 4 column quarters) computes a 256-token x (64 NF)-column tile (NF = 4: 256 columns, NF = 2:
// 128 columns for the narrow projections); a wave owns 128 tokens x 16 NF columns as 8 x NF accumulator fragments (128 / 64 VGPRs).
//   * k-step = 128 fp8 = 128-byte rows: both operands arrive by LDS-DMA (buffer_load ... lds, 16 B per lane) into two stages of 256 + 64 NF rows,
//     16-byte chunks XOR-swizzled by (row & 7).  A lane's 32 k-values of a fragment are chunks fg and fg + 4 of row fr -- NOT the contiguous 32 bytes at
//     32 fg: the contraction runs over all 128 k whichever lane holds which, as long as both operands use the same assignment, and with this one the two
//     ds_read_b128 per fragment are conflict-free (chunks 2 fg, 2 fg + 1 collide 2-way under the same swizzle: measured, the LDS reads alone then cost
//     more than the MFMAs);
//   * one barrier per k-step: wait for this step's DMA, barrier (everybody is done with the other stage), compute -- the next step's DMA into the other
//     stage is issued behind the first half of the step's MFMAs;
//   * the weight fragment is the MFMA's A operand, so a lane ends up with 4 consecutive output columns of one token: the epilogue scales them, runs
//     GEGLU on the (value, gate) fragment pairs of the interleaved w1 packing, and goes through LDS (the stages are free by then) so that every global
//     store -- and every residual read -- is 16 bytes of one output row per lane.
#include "common.h"
#include "muse_hip_internal.h"

namespace {

typedef int i32x8_t __attribute__((ext_vector_type(8)));

constexpr int TM = 256, BKB = 128;      // tokens per tile, bytes (= fp8 values) per k-step
enum { F8_BF16 = 0, F8_GEGLU = 1, F8_RESID = 2 };

__device__ __forceinline__ int sw128(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
// e4m3 x e4m3 with unit block scales (zero scale operands select the unscaled v_mfma_f32_16x16x128_f8f6f4)
__device__ __forceinline__ f32x4_t mfma_f8(const i32x8_t a, const i32x8_t b, const f32x4_t c) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
}

template <int NF, int EPI>
__global__ __launch_bounds__(512) void gemm_fp8_kernel(const GemmF8Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TN = 64 * NF;                       // weight rows (output columns before GEGLU) of a tile
    constexpr int X_B = TM * BKB, W_B = TN * BKB, STG = X_B + W_B;
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int fr = lane & 15, fg = lane >> 4;
    int tile_m, tile_n;
    xcd_grouped_tile(blockIdx.x, p.tiles_m, p.tiles_n, 8, tile_m, tile_n);
    const int m0 = tile_m * TM, n0 = tile_n * TN;
    const int KT = p.K / BKB;

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int rows_left = p.M - m0;
    const unsigned xbytes = (unsigned)(rows_left < TM ? rows_left : TM) * (unsigned)p.ldx;      // rows beyond M read as zero
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.X + (size_t)m0 * p.ldx), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.W + (size_t)n0 * p.ldw), 0, (unsigned)TN * (unsigned)p.ldw, 0x00020000);
    // a DMA instruction covers 8 rows: lane l fetches row l >> 3, logical chunk (l & 7) ^ (row & 7) into physical chunk l & 7.
    // This wave stages token rows 32 wid .. + 31 (4 instructions) and weight rows 8 NF wid .. (NF instructions).
    const int dchunk = ((lane & 7) ^ (lane >> 3)) * 16;      // (row & 7) == lane >> 3 for every 8-row group
    int voff_x[4], voff_w[4];      // (fixed extents: a template-dependent array type defers the DMA builtin's resolution to instantiation, which the host pass cannot do)
#pragma unroll
    for (int i = 0; i < 4; ++i) voff_x[i] = (32 * wid + 8 * i + (lane >> 3)) * (int)p.ldx + dchunk;
#pragma unroll
    for (int i = 0; i < NF; ++i) voff_w[i] = (8 * NF * wid + 8 * i + (lane >> 3)) * (int)p.ldw + dchunk;
#define ISSUE(kt_, st_)                                                                                                                \
    {                                                                                                                                  \
        unsigned char* xs_ = smem + (st_) * STG + wid * 4096;                                                                          \
        unsigned char* ws_ = smem + (st_) * STG + X_B + wid * (NF * 1024);                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(xs_ + i * 1024), 16, voff_x[i], (kt_) * BKB, 0, 0);              \
        _Pragma("unroll") for (int i = 0; i < NF; ++i)                                                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(ws_ + i * 1024), 16, voff_w[i], (kt_) * BKB, 0, 0);              \
    }

    f32x4_t acc[NF][8];
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    ISSUE(0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        const int st = kt & 1;
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this step's DMA (the only one in flight) has landed
        __builtin_amdgcn_s_barrier();            // ... for everybody, and everybody is done reading the other stage
        const unsigned char* xs = smem + st * STG + (wm * 128) * BKB;
        const unsigned char* ws = smem + st * STG + X_B + (wn * 16 * NF) * BKB;
        i32x8_t wf[NF];
#pragma unroll
        for (int a = 0; a < NF; ++a) {
            const int row = a * 16 + fr;
            const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(ws + sw128(row, fg));
            const u32x4_t hi = *reinterpret_cast<const u32x4_t*>(ws + sw128(row, fg + 4));
            wf[a] = i32x8_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int row = b * 16 + fr;
            const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(xs + sw128(row, fg));
            const u32x4_t hi = *reinterpret_cast<const u32x4_t*>(xs + sw128(row, fg + 4));
            const i32x8_t xf = i32x8_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
#pragma unroll
            for (int a = 0; a < NF; ++a)
                acc[a][b] = mfma_f8(wf[a], xf, acc[a][b]);
            if (b == (NF == 4 ? 3 : 0) && kt + 1 < KT) ISSUE(kt + 1, st ^ 1);      // behind the first 16 (NF = 4) / 2 (NF = 2) MFMAs of the step (placement measured: tools/gemm_harness)
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                // the stages are free: they become the output staging tile

    // ---- epilogue.  Lane (fr, fg) of fragment (a, b) holds columns n = 16 a + 4 fg .. + 3 (of the wave's 16 NF) for token 16 b + fr (of its 128).
    float sxv[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int m = m0 + wm * 128 + b * 16 + fr;
        sxv[b] = m < p.M ? p.sx[m] : 0.f;
    }
    float4 swv[NF];
#pragma unroll
    for (int a = 0; a < NF; ++a) swv[a] = *reinterpret_cast<const float4*>(p.sw + n0 + wn * 16 * NF + a * 16 + 4 * fg);

    if constexpr (EPI == F8_GEGLU) {
        // interleaved w1 packing (as the bf16 engine's): within a wave's 16 NF weight rows the first half are values, the second half their gates
        constexpr int OC = TN / 2;                       // output columns of the tile
        constexpr int ROWB = OC * 2;                     // staging row bytes (bf16)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int row = wm * 128 + b * 16 + fr;
#pragma unroll
            for (int a = 0; a < NF / 2; ++a) {
                const float4 sv = swv[a], sg = swv[a + NF / 2];
                const float s = sxv[b];
                const float o0 = geglu_f(acc[a][b][0] * s * sv.x, acc[a + NF / 2][b][0] * s * sg.x);
                const float o1 = geglu_f(acc[a][b][1] * s * sv.y, acc[a + NF / 2][b][1] * s * sg.y);
                const float o2 = geglu_f(acc[a][b][2] * s * sv.z, acc[a + NF / 2][b][2] * s * sg.z);
                const float o3 = geglu_f(acc[a][b][3] * s * sv.w, acc[a + NF / 2][b][3] * s * sg.w);
                const int col = wn * 8 * NF + a * 16 + 4 * fg;
                const int chunk = col >> 3;
                *reinterpret_cast<uint2*>(smem + row * ROWB + ((chunk ^ (row & 7)) << 4) + (col & 4) * 2) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        constexpr int CPR = ROWB / 16;                   // 16-byte chunks per row
        for (int i = t; i < TM * CPR; i += 512) {
            const int row = i / CPR, c = i % CPR;
            const int m = m0 + row;
            if (m < p.M) {
                const uint4 v = *reinterpret_cast<const uint4*>(smem + row * ROWB + ((c ^ (row & 7)) << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldc + tile_n * OC + c * 8) = v;
            }
        }
    } else if constexpr (EPI == F8_BF16) {
        constexpr int ROWB = TN * 2;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int row = wm * 128 + b * 16 + fr;
#pragma unroll
            for (int a = 0; a < NF; ++a) {
                const float s = sxv[b];
                const int col = wn * 16 * NF + a * 16 + 4 * fg;
                const int chunk = col >> 3;
                *reinterpret_cast<uint2*>(smem + row * ROWB + ((chunk ^ (row & 7)) << 4) + (col & 4) * 2) =
                    make_uint2(pack_bf16x2(acc[a][b][0] * s * swv[a].x, acc[a][b][1] * s * swv[a].y), pack_bf16x2(acc[a][b][2] * s * swv[a].z, acc[a][b][3] * s * swv[a].w));
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        constexpr int CPR = ROWB / 16;
        for (int i = t; i < TM * CPR; i += 512) {
            const int row = i / CPR, c = i % CPR;
            const int m = m0 + row;
            if (m < p.M) {
                const uint4 v = *reinterpret_cast<const uint4*>(smem + row * ROWB + ((c ^ (row & 7)) << 4));
                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldc + n0 + c * 8) = v;
            }
        }
    } else {      // fp32 out = residual + product, one token half (128 rows x TN fp32 <= 128 KiB) at a time
        constexpr int ROWB = TN * 4;
        constexpr int CPR = ROWB / 16;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (wm == half) {
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int row = b * 16 + fr;
#pragma unroll
                    for (int a = 0; a < NF; ++a) {
                        const float s = sxv[b];
                        const int chunk = (wn * 16 * NF + a * 16 + 4 * fg) >> 2;
                        *reinterpret_cast<float4*>(smem + row * ROWB + ((chunk ^ (row & 7)) << 4)) =
                            make_float4(acc[a][b][0] * s * swv[a].x, acc[a][b][1] * s * swv[a].y, acc[a][b][2] * s * swv[a].z, acc[a][b][3] * s * swv[a].w);
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            for (int i = t; i < 128 * CPR; i += 512) {
                const int row = i / CPR, c = i % CPR;
                const int m = m0 + half * 128 + row;
                if (m < p.M) {
                    float4 v = *reinterpret_cast<const float4*>(smem + row * ROWB + ((c ^ (row & 7)) << 4));
                    if (p.resid) {
                        const float4 r = *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ldr + n0 + c * 4);
                        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                    }
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n0 + c * 4) = v;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
        }
    }
#undef ISSUE
}

template <int NF, int EPI>
int launch_f8(GemmF8Args a, hipStream_t stream) {
    constexpr int SMEM = 2 * (TM + 64 * NF) * BKB;      // 128 KiB (NF = 4) / 96 KiB (NF = 2)
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<NF, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return mm_set_hip_error(e, "gemm_fp8 hipFuncSetAttribute");
        attr_set = true;
    }
    a.tiles_m = (a.M + TM - 1) / TM;
    a.tiles_n = a.N / (64 * NF);
    hipLaunchKernelGGL((gemm_fp8_kernel<NF, EPI>), dim3(a.tiles_m * a.tiles_n), dim3(512), SMEM, stream, a);
    return mm_check_launch("gemm_fp8_kernel");
}

}  // namespace

// N: weight rows (GEGLU: 2 x padded inner width, interleaved); out: bf16 [M][ldc] (epi 0 / 1) or fp32 [M][ldc] (+ resid fp32 [M][ldr], may alias out)
int k_gemm_fp8(hipStream_t s, const GemmF8Args& a) {
    if (a.M <= 0 || a.N <= 0) return MM_OK;
    if (!a.X || !a.W || !a.sx || !a.sw || !a.out) return mm_set_error(MM_ERR_SHAPE, "gemm_fp8: NULL pointer");
    if (a.K <= 0 || (a.K % BKB)) return mm_set_error(MM_ERR_SHAPE, "gemm_fp8: K must be a multiple of 128 (zero-pad the quantised rows)");
    if ((a.ldx % 16) || (a.ldw % 16) || a.ldx < a.K || a.ldw < a.K) return mm_set_error(MM_ERR_ALIGN, "gemm_fp8: operand rows must be 16-byte multiples >= K");
    if ((a.N % 128)) return mm_set_error(MM_ERR_SHAPE, "gemm_fp8: N must be a multiple of 128");
    if ((size_t)a.M * (size_t)a.ldx >= (1ull << 31) + (size_t)TM * a.ldx) return mm_set_error(MM_ERR_SHAPE, "gemm_fp8: activation matrix too large for 32-bit row offsets");
    if ((((uintptr_t)a.out) & 15) || (a.epi == F8_RESID ? (a.ldc % 4) : (a.ldc % 8))) return mm_set_error(MM_ERR_ALIGN, "gemm_fp8: output rows must be 16-byte aligned");
    if (a.epi == F8_RESID && a.resid && ((((uintptr_t)a.resid) & 15) || (a.ldr % 4))) return mm_set_error(MM_ERR_ALIGN, "gemm_fp8: residual rows must be 16-byte aligned");
    // 256-column tiles when they still give every CU a workgroup (or N has no 128-column remainder to serve), 128-column tiles otherwise
    const long tiles256 = (a.N % 256) ? 0 : (long)((a.M + TM - 1) / TM) * (a.N / 256);
    const bool wide = tiles256 >= 256 || ((g_mm_debug & (1 << 30)) && (a.N % 256) == 0);      // (debug bit: tests force the 256-column tile on small shapes)
    switch (a.epi) {
        case F8_BF16: return wide ? launch_f8<4, F8_BF16>(a, s) : launch_f8<2, F8_BF16>(a, s);
        case F8_GEGLU:      // the value / gate interleave of w1 is per 64 weight rows = one wave's share of the 256-row tile
            if (a.N % 256) return mm_set_error(MM_ERR_SHAPE, "gemm_fp8: the GEGLU epilogue needs N (= 2 x padded inner width) to be a multiple of 256");
            return launch_f8<4, F8_GEGLU>(a, s);
        case F8_RESID: return wide ? launch_f8<4, F8_RESID>(a, s) : launch_f8<2, F8_RESID>(a, s);
    }
    return mm_set_error(MM_ERR_SHAPE, "gemm_fp8: bad epilogue");
}
