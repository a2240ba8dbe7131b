// bf16 MFMA GEMM family for gfx950:  out[m][n] = sum_k X[m][k] * W[n][k]   (nn.Linear layout, K contiguous
// in both operands), fp32 accumulate.  One main loop serves
//   MODE_DENSE : Linear layers of the transformer            (muse_maskgit_pytorch.py:85,88,118-124,225,233)
//   MODE_CFG   : to_logits on the cond AND null rows of one token at once, epilogue writes
//                null + (cond - null) * cond_scale            (muse_maskgit_pytorch.py:250-254)
//   MODE_CONV  : implicit-GEMM NHWC convolution / one parity class of ConvTranspose2d(4,2,1)
//                                                            (vqgan_vae.py:224-232, 255-261, 271-277)
//
// Tiling (CDNA4): 128 (n) x 128 (m) x 64 (k) per 256-thread workgroup, 4 waves as 2x2, each wave a
// 64x64 patch = 4x4 v_mfma_f32_16x16x32_bf16 fragments.  The WEIGHT tile is the MFMA A operand and the
// activation tile the B operand, so one accumulator fragment holds 4 CONSECUTIVE output features of one
// token: the epilogue stores 16 B (fp32) / 8 B (bf16) per lane instead of 4 scalars.
// LDS: two 32 KiB stages (W tile + X tile, row = 128 B = 8 chunks of 16 B, chunk index XOR (row & 7) so
// every ds_read_b128 lane group hits 16 distinct slots); global->register->LDS staging with the next
// tile's loads issued before the current tile's MFMAs; one barrier per K-tile.
// Grid: 1-D, XCD-aware grouped tile order (common.h).
#include "common.h"
#include "muse_hip_internal.h"

namespace {

constexpr int BT = 128;   // tile edge (both n and m)
constexpr int BK = 64;    // k per stage
constexpr int STAGE_BYTES = 2 * BT * BK * 2;   // W tile + X tile

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <int MODE>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int wave_n = wid >> 1, wave_m = wid & 1;
    int tile_m, tile_n;
    xcd_grouped_tile(blockIdx.x, p.tiles_m, p.tiles_n, 16, tile_m, tile_n);
    const int n0 = tile_n * BT;
    const int m0 = tile_m * (MODE == MODE_CFG ? 64 : BT);   // CFG: 64 tokens x {cond, null} per tile

    // ---- per-thread staging geometry: chunk column t&7, rows (t>>3) + 32*i
    const int chunk = t & 7;
    const int row0 = t >> 3;
    const bf16_t* wptr[4];
    const bf16_t* xptr[4];
    bool wok[4], xok[4];
    int cb[4], cy[4], cx[4];   // conv: batch / output y / output x of the staged rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 32 * i;
        const int n = n0 + r;
        wok[i] = n < p.N;
        wptr[i] = p.W + (size_t)(wok[i] ? n : 0) * p.ldw + chunk * 8;
        if constexpr (MODE == MODE_DENSE) {
            const int m = m0 + r;
            xok[i] = m < p.M;
            xptr[i] = p.X + (size_t)(xok[i] ? m : 0) * p.ldx + chunk * 8;
        } else if constexpr (MODE == MODE_CFG) {
            const int wm = r >> 6, jj = r & 63;
            const int tok = m0 + wm * 32 + (jj & 31);
            xok[i] = tok < p.M;
            xptr[i] = ((jj >> 5) ? p.X2 : p.X) + (size_t)(xok[i] ? tok : 0) * p.ldx + chunk * 8;
        } else {
            const int m = m0 + r;
            xok[i] = m < p.M;
            const int mm = xok[i] ? m : 0;
            const int hw = p.Hv * p.Wv;
            cb[i] = mm / hw;
            const int rem = mm - cb[i] * hw;
            cy[i] = rem / p.Wv;
            cx[i] = rem - cy[i] * p.Wv;
            xptr[i] = p.X;
        }
    }

    u32x4_t wreg[4], xreg[4];

#define LOAD_TILE(kt_)                                                                                             \
    {                                                                                                              \
        const int k0_ = (kt_) * BK;                                                                                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                              \
            wreg[i] = *reinterpret_cast<const u32x4_t*>(wptr[i] + k0_); /* out-of-range rows are clamped to row 0: */ \
        /* they only feed output rows/columns the epilogue never stores, so no zero fill is needed */              \
        if constexpr (MODE != MODE_CONV) {                                                                         \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
                xreg[i] = *reinterpret_cast<const u32x4_t*>(xptr[i] + k0_);                                       \
        } else {                                                                                                   \
            /* im2col on the fly: this thread's 8 channels of K-index k belong to tap k / Cin */                   \
            const int k_ = k0_ + chunk * 8;                                                                        \
            const int tap_ = k_ / p.Cin;                                                                           \
            const int c_ = k_ - tap_ * p.Cin;                                                                      \
            const int ty_ = tap_ / p.TW, tx_ = tap_ - ty_ * p.TW;                                                  \
            const bool kok_ = k_ < p.Ktrue;                                                                        \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                        \
                const int iy_ = cy[i] * p.stride + ty_ + p.off_y;                                                  \
                const int ix_ = cx[i] * p.stride + tx_ + p.off_x;                                                  \
                const bool ok_ = xok[i] && kok_ && iy_ >= 0 && iy_ < p.Hin && ix_ >= 0 && ix_ < p.Win;             \
                const size_t off_ = (((size_t)cb[i] * p.Hin + (ok_ ? iy_ : 0)) * p.Win + (ok_ ? ix_ : 0)) * p.Cin + c_; \
                const u32x4_t ld_ = *reinterpret_cast<const u32x4_t*>(p.X + off_);                                  \
                const unsigned int keep_ = ok_ ? 0xFFFFFFFFu : 0u;            /* zero the padding taps */             \
                xreg[i] = ld_ & keep_;                                                                             \
            }                                                                                                      \
        }                                                                                                          \
    }
#define STORE_TILE(stage_)                                                                                         \
    {                                                                                                              \
        unsigned char* ws_ = smem + (stage_) * STAGE_BYTES;                                                        \
        unsigned char* xs_ = ws_ + BT * BK * 2;                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                            \
            const int r_ = row0 + 32 * i;                                                                          \
            *reinterpret_cast<u32x4_t*>(ws_ + lds_off(r_, chunk)) = wreg[i];                                         \
            *reinterpret_cast<u32x4_t*>(xs_ + lds_off(r_, chunk)) = xreg[i];                                         \
        }                                                                                                          \
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int KT = p.K / BK;
    LOAD_TILE(0);
    STORE_TILE(0);
    __syncthreads();

    const int fr = lane & 15, fg = lane >> 4;
#define COMPUTE_TILE(stage_)                                                                                       \
    {                                                                                                              \
        const unsigned char* ws_ = smem + (stage_) * STAGE_BYTES;                                                  \
        const unsigned char* xs_ = ws_ + BT * BK * 2;                                                              \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                         \
            u32x4_t af[4], bfm[4];                                                                                 \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                        \
                af[i] = *reinterpret_cast<const u32x4_t*>(ws_ + lds_off(wave_n * 64 + i * 16 + fr, ks * 4 + fg));  \
                bfm[i] = *reinterpret_cast<const u32x4_t*>(xs_ + lds_off(wave_m * 64 + i * 16 + fr, ks * 4 + fg)); \
            }                                                                                                      \
            _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                          \
                _Pragma("unroll") for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(af[a], bfm[b], acc[a][b]);        \
        }                                                                                                          \
    }
    // steady state: next tile's global loads are in flight while this tile's MFMAs run; one barrier per tile
    for (int kt = 0; kt < KT - 1; ++kt) {
        LOAD_TILE(kt + 1);
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch issue ahead of the MFMA block (hipcc sinks it otherwise)
        COMPUTE_TILE(kt & 1);
        __builtin_amdgcn_sched_barrier(0);
        STORE_TILE((kt + 1) & 1);
        __syncthreads();
    }
    COMPUTE_TILE((KT - 1) & 1);

    // ---- epilogue: lane holds out[m][n..n+3] per fragment; n = 4 consecutive output features
    constexpr int MT = (MODE == MODE_CFG) ? 2 : 4;
#pragma unroll
    for (int b = 0; b < MT; ++b) {
        int m;
        if constexpr (MODE == MODE_CFG) m = m0 + wave_m * 32 + b * 16 + fr;
        else m = m0 + wave_m * 64 + b * 16 + fr;
        if (m >= p.M) continue;
        size_t orow;        // row index into out / resid
        int ob = 0, oy = 0, ox = 0;
        if constexpr (MODE == MODE_CONV) {
            const int hw = p.Hv * p.Wv;
            ob = m / hw;
            const int rem = m - ob * hw;
            oy = (rem / p.Wv) * p.os + p.py;
            ox = (rem % p.Wv) * p.os + p.px;
            orow = ((size_t)ob * p.Hout + oy) * p.Wout + ox;
        } else {
            orow = (size_t)m;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int n = n0 + wave_n * 64 + a * 16 + fg * 4;
            if (n >= p.N) continue;
            float v[4];
            if constexpr (MODE == MODE_CFG) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float c = acc[a][b][r], nl = acc[a][(b + 2) & 3][r];
                    v[r] = nl + (c - nl) * p.cfg_scale;      // muse_maskgit_pytorch.py:254
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[a][b][r];
            }
            const bool full = n + 3 < p.N;
            if (p.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += p.bias[n + r];
            }
            if (p.act == ACT_LEAKY) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.1f * v[r];   // vqgan_vae.py:103-104
            }
            if (p.resid_f32) {
                const float* rp = p.resid_f32 + orow * p.ldr + n;
                if (full) {
                    const float4 rv = *reinterpret_cast<const float4*>(rp);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += rp[r];
                }
            }
            if (p.resid_bf16) {
                const bf16_t* rp = p.resid_bf16 + orow * p.ldr + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += bf16_to_f32(rp[r]);
            }
            if (p.out_kind == OUT_F32) {
                float* op = reinterpret_cast<float*>(p.out) + orow * p.ldc + n;
                if (full) *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) op[r] = v[r];
                }
            } else if (p.out_kind == OUT_BF16) {
                bf16_t* op = reinterpret_cast<bf16_t*>(p.out) + orow * p.ldc + n;
                if (full) *reinterpret_cast<uint2*>(op) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) op[r] = f32_to_bf16(v[r]);
                }
            } else {   // OUT_NCHW_F32: out[b][n][y][x]
                float* op = reinterpret_cast<float*>(p.out);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) op[(((size_t)ob * p.N + n + r) * p.Hout + oy) * p.Wout + ox] = v[r];
            }
        }
    }
}

template <int MODE>
int launch(const GemmArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
        if (e != hipSuccess) return mm_set_hip_error(e, "gemm hipFuncSetAttribute");
        attr_set = true;
    }
    const int blocks = a.tiles_m * a.tiles_n;
    if (blocks <= 0) return MM_OK;
    hipLaunchKernelGGL(gemm_kernel<MODE>, dim3(blocks), dim3(256), 2 * STAGE_BYTES, stream, a);
    return mm_check_launch("gemm_kernel");
}

}  // namespace

int mm_gemm_launch(GemmArgs a, hipStream_t stream) {
    if (a.K <= 0 || (a.K % BK) != 0) return mm_set_error(MM_ERR_SHAPE, "gemm: K must be a positive multiple of 64 (pad at pack time)");
    if ((a.ldw % 8) != 0 || (a.mode != MODE_CONV && (a.ldx % 8) != 0)) return mm_set_error(MM_ERR_ALIGN, "gemm: row strides must be multiples of 8 elements (16 B)");
    if (a.mode == MODE_CONV && (a.Cin % 8) != 0) return mm_set_error(MM_ERR_SHAPE, "conv: Cin must be a multiple of 8");
    if ((a.ldc % 4) != 0 && a.out_kind != OUT_NCHW_F32 && a.N >= 4) return mm_set_error(MM_ERR_ALIGN, "gemm: ldc must be a multiple of 4");
    a.tiles_n = (a.N + BT - 1) / BT;
    const int tm = a.mode == MODE_CFG ? 64 : BT;
    a.tiles_m = (a.M + tm - 1) / tm;
    switch (a.mode) {
        case MODE_DENSE: return launch<MODE_DENSE>(a, stream);
        case MODE_CFG: return launch<MODE_CFG>(a, stream);
        case MODE_CONV: return launch<MODE_CONV>(a, stream);
    }
    return mm_set_error(MM_ERR_SHAPE, "gemm: bad mode");
}
