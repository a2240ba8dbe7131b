// Persistent variant of the 256x128 bf16 MFMA GEMM (gemm_big.hip) for the GEMM classes whose epilogue is a pure store:
//   MODE_DENSE, bf16 output (q|k|v projection, FF w1 with the fused GEGLU epilogue)
//   MODE_CFG,   fp32 logits  (to_logits of both guidance passes + the combine)
// Measured on the non-persistent kernels (tools/gemm_bench.py ablation): the output stores are 25-40 % of the kernel and
// overlap with nothing -- every workgroup on the chip reaches its store phase at about the same time (HBM idle during the
// MFMA phase, MFMA idle during the store phase) -- and every tile pays a DMA-latency bubble at its start.  Storing straight
// from the MFMA fragment layout inside the next tile's loop (first attempt) was slower still: 16 rows x 64 B per
// instruction wastes the write path.  So here one workgroup per CU walks its tiles as ONE software pipeline:
//   * two 48 KiB DMA stages + a 64 KiB output tile `ct` in LDS (160 KiB = the whole CU);
//   * at the end of a tile the accumulators are combined (CFG / GEGLU) and transposed into ct (XOR-swizzled rows);
//   * ct is written out row-contiguously, 16 B per lane, ONE 8 KiB piece per k-iteration of the NEXT tile, i.e. inside its
//     MFMA stream; the LDS-DMA of the next tile's first k-tile is already in flight across the tile boundary;
//   * counted s_waitcnt vmcnt(N): behind the DMA we wait for there is at most the one store of the previous iteration.
// Same tile shape and MFMA order as gemm.hip / gemm_big.hip -> bit-identical results.
#include "common.h"
#include "muse_hip_internal.h"

namespace {

constexpr int BMB = 256, BNB = 128, BK = 64;
constexpr int W_BYTES = BNB * BK * 2, X_BYTES = BMB * BK * 2, STAGE_B = W_BYTES + X_BYTES;
constexpr int CT_OFF = 2 * STAGE_B;                 // 96 KiB
constexpr int SMEM_B = CT_OFF + 65536;              // + 64 KiB output tile = 160 KiB

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

struct TilePtrs {
    const bf16_t* w[2];
    const bf16_t* x[4];
    int m0, n0, tile_n;
};

template <int MODE>
__device__ __forceinline__ void tile_setup(const GemmArgs& p, int vb, int wid, int lane, TilePtrs& tp) {
    int tile_m, tile_n;
    xcd_grouped_tile(vb, p.tiles_m, p.tiles_n, 8, tile_m, tile_n);
    tp.tile_n = tile_n;
    tp.n0 = tile_n * BNB;
    tp.m0 = tile_m * (MODE == MODE_CFG ? 128 : BMB);
    const int chunk = (lane & 7) ^ (lane >> 3);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = tp.n0 + 16 * wid + 8 * i + (lane >> 3);
        tp.w[i] = p.W + (size_t)(n < p.N ? n : 0) * p.ldw + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 32 * wid + 8 * i + (lane >> 3);
        if constexpr (MODE == MODE_CFG) {
            const int wm = r >> 6, jj = r & 63;
            const int tok = tp.m0 + wm * 32 + (jj & 31);
            tp.x[i] = ((jj >> 5) ? p.X2 : p.X) + (size_t)(tok < p.M ? tok : 0) * p.ldx + chunk * 8;
        } else {
            const int m = tp.m0 + r;
            tp.x[i] = p.X + (size_t)(m < p.M ? m : 0) * p.ldx + chunk * 8;
        }
    }
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

// one 8 KiB piece of ct -> global, 16 B per lane, consecutive lanes on consecutive addresses of one output row
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
        if (m < p.M) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldc + tile_n * 64 + c * 8) = v;
    } else {
        const int row = piece * 32 + (t >> 4), c = t & 15;
        const uint4 v = *reinterpret_cast<const uint4*>(ct + row * 256 + ((c ^ (row & 15)) << 4));
        const int m = m0 + row;
        if (m < p.M) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldc + n0 + c * 8) = v;
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
#define ISSUE_TILE(tp_, kt_, g_)                                                                              \
    {                                                                                                         \
        const int k0_ = (kt_) * BK;                                                                           \
        const int st_ = (g_) & 1;                                                                             \
        unsigned char* ws_ = smem + st_ * STAGE_B + wid * 2048;                                               \
        unsigned char* xs_ = smem + st_ * STAGE_B + W_BYTES + wid * 4096;                                     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                         \
            __builtin_amdgcn_global_load_lds((tp_).w[i] + k0_, (lds_ptr_t)(ws_ + i * 1024), 16, 0, 0);        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
            __builtin_amdgcn_global_load_lds((tp_).x[i] + k0_, (lds_ptr_t)(xs_ + i * 1024), 16, 0, 0);        \
    }

    f32x4_t acc[4][4];
    TilePtrs cur;
    int prv_m0 = 0, prv_n0 = 0, prv_tile_n = 0;
    int vb = blockIdx.x;
    if (vb >= total) return;
    tile_setup<MODE>(p, vb, wid, lane, cur);
    bool has_next = vb + G < total;
    bool have_prev = false;
    int g = 0;                  // flattened (tile, k-tile) iteration counter: selects the DMA stage
    int st_prev = 0;            // VMEM stores this wave issued in the previous iteration (they sit behind the DMA we wait for)
    ISSUE_TILE(cur, 0, 0);

    // one pipeline iteration: wait for this iteration's operands; DMA_STMT issues the NEXT iteration's; 32 MFMAs per wave;
    // then (PIECE >= 0) one 8 KiB piece of the previous tile's output goes from ct to HBM.
    // A ragged tile may skip a whole store instruction (all lanes predicated off): then count 0 -- under-counting only makes
    // the wait conservative, over-counting would let the DMA we need slip.
#define ITER(PIECE, DMA_STMT)                                                                                                \
    {                                                                                                                        \
        if (st_prev) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
        __builtin_amdgcn_s_barrier();                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        if (!(p.debug & 2)) { DMA_STMT; }                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        const unsigned char* ws = smem + (g & 1) * STAGE_B;                                                                  \
        const unsigned char* xs = ws + W_BYTES;                                                                              \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                   \
            u32x4_t af[4], bfm[4];                                                                                           \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                  \
                af[i] = *reinterpret_cast<const u32x4_t*>(ws + lds_off(wave_n * 64 + i * 16 + fr, ks * 4 + fg));             \
                bfm[i] = *reinterpret_cast<const u32x4_t*>(xs + lds_off(wave_m * 64 + i * 16 + fr, ks * 4 + fg));            \
            }                                                                                                                \
            if (!(p.debug & 4)) {                                                                                            \
            _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                                    \
                _Pragma("unroll") for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(af[a], bfm[b], acc[a][b]);                  \
            } else { _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(af[i]), "v"(bfm[i])); }           \
        }                                                                                                                    \
        st_prev = 0;                                                                                                         \
        if ((PIECE) >= 0 && (PIECE) < npieces && have_prev && !(p.debug & 1)) {                                              \
            store_piece<MODE>(p, ct, (PIECE) < 0 ? 0 : (PIECE), t, prv_m0, prv_n0, prv_tile_n);                              \
            st_prev = (prv_m0 + tile_rows <= p.M) ? 1 : 0;                                                                   \
        }                                                                                                                    \
        ++g;                                                                                                                 \
    }

    while (true) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // head: 7 iterations, pieces 0..6 of the previous tile; middle: K > 512 only; tail: last k-tile, piece 7, and the DMA of
        // the NEXT tile's first k-tile (its pointers are computed here, after this tile's last use of `cur`)
#pragma unroll
        for (int u = 0; u < 7; ++u) ITER(u, ISSUE_TILE(cur, u + 1, g + 1))
        for (int kt = 7; kt < KT - 1; ++kt) ITER(-1, ISSUE_TILE(cur, kt + 1, g + 1))
        const int cur_m0 = cur.m0, cur_n0 = cur.n0, cur_tile_n = cur.tile_n;
        if (has_next) tile_setup<MODE>(p, vb + G, wid, lane, cur);
        ITER(7, if (has_next) ISSUE_TILE(cur, 0, g + 1))
        // tile boundary: everyone is done with the stages' last reads and with reading the previous ct -> overwrite ct
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!(p.debug & 16)) acc_to_ct<MODE>(p, acc, ct, wave_m, wave_n, fr, fg);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        prv_m0 = cur_m0; prv_n0 = cur_n0; prv_tile_n = cur_tile_n; have_prev = true;
        vb += G;
        if (vb >= total) break;
        has_next = vb + G < total;
    }
    // drain the last tile
    if (!(p.debug & 1))
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
