// Implicit-GEMM NHWC convolution on the 256-pixel x 256-channel x 64-deep persistent tile of gemm_wide.hip (round 6): the VQGanVAE decoder's convolutions
// (vqgan_vae.py:246-265: GLUResBlock conv3x3 2048 -> 4096, ConvTranspose2d(4, 2, 1) as four parity 2x2 convolutions) for bf16 and for the single-term fp16
// storage of the half-precision decode.
//
// Why: these are LONG contractions (K = taps x Cin = 1024 ... 18432) on which the 256 x 128 three-stage kernel (gemm_big.hip, 32 MFMAs per wave and barrier)
// ran the matrix pipe 0.37-0.44 busy; the 256 x 256 tile puts 64 MFMAs behind every barrier (the k-loop of gemm_wide.hip: 0.6 busy while it runs) and, being
// persistent, requests the next tile's first k-step during the last step of the current one.  With N = 256 a tile holds ALL channels of its 256 pixels, so the
// last up-sampling layer can also take the 1 x 1 head (Conv2d(dim, channels, 1), vqgan_vae.py:232) in its epilogue: the 1 GiB activation it would write and the
// head would read back never exists (EPI_HEAD).
//
// Loader: Cin % 64 == 0, so a 64-deep k-step lies inside ONE filter tap: the tap walk (ty, tx, channel offset) is wave-uniform running state, a staged row's
// source is pixel (y * stride + off_y + ty, x * stride + off_x + tx) of its image, and rows whose tap falls into the zero padding (or beyond M) are requested
// at an offset beyond the buffer's extent -- the LDS-DMA then writes zeros, no branch and no zero page.  A ConvTranspose2d(4, 2, 1) -- four parity classes, each a
// 2 x 2 convolution with its own weight matrix, tap offset and output phase -- runs as ONE launch (GemmArgs::par_w): the parity is the slowest tile coordinate.  Same XOR-swizzled 128-byte LDS rows, same fragment
// reads and MFMA order per output element (k ascending in chunks of 32, one MFMA each) as every other kernel of the family: results are BIT-IDENTICAL to
// gemm_big_kernel<MODE_CONV> (tests/test_gpu_ops.py).
#include "common.h"
#include "muse_hip_internal.h"

namespace {

constexpr int TM = 256, TN = 256, BKB = 128;
constexpr int X_B = TM * BKB, W_B = TN * BKB, STG = X_B + W_B;
constexpr int HEAD_MAX = 8;                                  // image channels of the fused head
constexpr int SMEM = 2 * STG + HEAD_MAX * TN * 4 + 64;      // two stages + the head's fp32 weights [channels][256] and bias

__device__ __forceinline__ int sw128(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

#define MC_VMCNT_IMM(n_) (0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14))
__device__ __forceinline__ void wait_vmcnt_c(int n) {      // wave-uniform n; above 31: wait for 31 (waiting for more is always safe)
    switch (n) {
#define MC_W(n_) case n_: __builtin_amdgcn_s_waitcnt(MC_VMCNT_IMM(n_)); break;
        MC_W(0) MC_W(1) MC_W(2) MC_W(3) MC_W(4) MC_W(5) MC_W(6) MC_W(7) MC_W(8) MC_W(9) MC_W(10) MC_W(11) MC_W(12) MC_W(13) MC_W(14) MC_W(15)
        MC_W(16) MC_W(17) MC_W(18) MC_W(19) MC_W(20) MC_W(21) MC_W(22) MC_W(23) MC_W(24) MC_W(25) MC_W(26) MC_W(27) MC_W(28) MC_W(29) MC_W(30)
#undef MC_W
        default: if (n < 0) __builtin_amdgcn_s_waitcnt(MC_VMCNT_IMM(0)); else __builtin_amdgcn_s_waitcnt(MC_VMCNT_IMM(31)); break;
    }
}

// HEAD: the 1 x 1 head convolution in the epilogue (N == 256 == the tile: tiles_n == 1): p.head_w the head's 16-bit pack [channels][head_ldw], p.head_b fp32 [channels], p.out =
// the NCHW fp32 image [B][channels][Hout][Wout].  The activation is rounded to the 16-bit storage type first (what the unfused sequence stores and reads back).
template <bool F16, bool HEAD>
__global__ __launch_bounds__(512) void gemm_wide_conv_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int per_par = p.tiles_m * p.tiles_n;
    const int total = per_par * (p.par_w[1] ? 4 : 1), G = gridDim.x;
    const int KT = p.K / 64;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int dchunk = ((lane & 7) ^ (lane >> 3)) * 16;
    const int hw = p.Hv * p.Wv;
    constexpr int OOB = 0x7FFF0000;                    // beyond any extent: the DMA writes zeros
    int voff_w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) voff_w[i] = (32 * wid + 8 * i + (lane >> 3)) * p.ldw * 2 + dchunk;
    int vb = blockIdx.x;
    if (vb >= total) return;
    const unsigned x_bytes = (unsigned)(p.M / hw) * (unsigned)p.Hin * (unsigned)p.Win * (unsigned)p.Cin * 2u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X), 0, x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rw;
    int tile_m, tile_n;
    int cur_py = 0, cur_px = 0, cur_offy = 0, cur_offx = 0;      // output parity / tap offset of the tile being SET UP (wave-uniform)
    int cyx[4], cpix[4];      // per staged row of this lane: input (y + 8) << 16 | (x + 8) of tap (0, 0) (packed: the kernel sits at the 256-register limit) and that position's flat pixel index
    int is_ty = 0, is_tx = 0, is_c = 0, is_k = 0;      // the NEXT k-step to issue: its tap, channel offset and weight column (wave-uniform)
#define TILE_SETUP(vb_)                                                                                                                \
    {                                                                                                                                  \
        int par_ = 0, vt_ = (vb_);                                                                                                     \
        if (p.par_w[1]) { par_ = vt_ / per_par; vt_ -= par_ * per_par; }      /* the four parity classes of a ConvTranspose2d(4, 2, 1) in ONE launch */ \
        xcd_grouped_tile(vt_, p.tiles_m, p.tiles_n, 8, tile_m, tile_n);                                                                \
        cur_py = p.par_w[1] ? (par_ >> 1) : p.py; cur_px = p.par_w[1] ? (par_ & 1) : p.px;                                             \
        cur_offy = p.par_w[1] ? cur_py - 1 : p.off_y; cur_offx = p.par_w[1] ? cur_px - 1 : p.off_x;                                    \
        const bf16_t* wb_ = p.par_w[1] ? p.par_w[par_] : p.W;                                                                          \
        rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wb_ + (size_t)tile_n * TN * p.ldw), 0, (unsigned)TN * (unsigned)p.ldw * 2u, 0x00020000); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                                \
            const int m_ = tile_m * TM + 32 * wid + 8 * i + (lane >> 3);                                                               \
            if (m_ < p.M) {                                                                                                            \
                const int cb_ = m_ / hw, rem_ = m_ - cb_ * hw;                                                                         \
                const int cy_ = rem_ / p.Wv, cx_ = rem_ - cy_ * p.Wv;                                                                  \
                const int iy_ = cy_ * p.stride + cur_offy, ix_ = cx_ * p.stride + cur_offx;                                            \
                cyx[i] = ((iy_ + 8) << 16) | (ix_ + 8);                                                                                \
                cpix[i] = (cb_ * p.Hin + iy_) * p.Win + ix_;                                                                           \
            } else {                                                                                                                   \
                cyx[i] = 0x7FFF0000; cpix[i] = 0;      /* rows beyond M: every tap out of bounds -> zeros */                            \
            }                                                                                                                          \
        }                                                                                                                              \
        is_ty = is_tx = is_c = is_k = 0;                                                                                               \
    }
#define ISSUE(st_)                                                                                                                     \
    {                                                                                                                                  \
        unsigned char* xs_ = smem + (st_) * STG + wid * 4096;                                                                          \
        unsigned char* ws_ = smem + (st_) * STG + X_B + wid * 4096;                                                                    \
        const int dpix_ = is_ty * p.Win + is_tx;                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                                \
            const bool ok_ = (unsigned)((cyx[i] >> 16) - 8 + is_ty) < (unsigned)p.Hin && (unsigned)((cyx[i] & 0xFFFF) - 8 + is_tx) < (unsigned)p.Win; \
            const int vo_ = ok_ ? ((cpix[i] + dpix_) * p.Cin + is_c) * 2 + dchunk : OOB;                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(xs_ + i * 1024), 16, vo_, 0, 0, 0);                               \
        }                                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(ws_ + i * 1024), 16, voff_w[i], is_k * 2, 0, 0);                  \
        is_k += 64; is_c += 64;                                                                                                        \
        if (is_c >= p.Cin) { is_c = 0; if (++is_tx == p.TW) { is_tx = 0; ++is_ty; } }                                                  \
    }
    TILE_SETUP(vb);
    ISSUE(0);
    if constexpr (HEAD) {      // the head's weights (its own 16-bit pack, un-scaled) and bias: fp32 into the tail of the LDS, once per workgroup (read in every epilogue)
        float* hwt = reinterpret_cast<float*>(smem + 2 * STG);
        for (int i = t; i < p.head_c * TN; i += 512) {
            const bf16_t w16 = p.head_w[(size_t)(i / TN) * p.head_ldw + (i % TN)];
            hwt[i] = (F16 && p.half_io) ? ld16s<true>(w16) * p.alpha : bf16_to_f32(w16);
        }
        if (t < HEAD_MAX) hwt[HEAD_MAX * TN + t] = t < p.head_c ? p.head_b[t] : 0.f;
    }
    int pending = 0;                 // VMEM stores this wave issued BEHIND the DMA of the coming tile's first step
    unsigned char* stg = smem + STG; // output staging: stage 1
    f32x4_t acc[4][8];               // [channel fragment a][pixel fragment b]: lane (fr, fg) holds channels 64 wn + 16 a + 4 fg .. + 3 of pixel 128 wm + 16 b + fr
    while (true) {
        const int m0 = tile_m * TM, n0 = tile_n * TN, opy = cur_py, opx = cur_px;      // (the next tile's setup overwrites the running values during the last k-step)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT; ++kt) {
            const int st = kt & 1;
            if (kt == 0) wait_vmcnt_c(pending); else __builtin_amdgcn_s_waitcnt(0x0F70);      // this step's DMA has landed (only younger stores may still be in flight)
            __builtin_amdgcn_s_barrier();            // ... for everybody, and everybody is done reading the other stage
            const unsigned char* xs = smem + st * STG + (wm * 128) * BKB;
            const unsigned char* ws = smem + st * STG + X_B + (wn * 64) * BKB;
            u32x4_t wf[4], wf2[4], xr[3];
#pragma unroll
            for (int a = 0; a < 4; ++a) wf[a] = *reinterpret_cast<const u32x4_t*>(ws + sw128(a * 16 + fr, fg));
            xr[0] = *reinterpret_cast<const u32x4_t*>(xs + sw128(fr, fg));
            xr[1] = *reinterpret_cast<const u32x4_t*>(xs + sw128(16 + fr, fg));
#pragma unroll
            for (int it = 0; it < 16; ++it) {        // (the software pipeline of gemm_wide_kernel's fragment reads)
                const int b = it & 7;
                if (it + 2 < 16) xr[(it + 2) % 3] = *reinterpret_cast<const u32x4_t*>(xs + sw128(((it + 2) & 7) * 16 + fr, ((it + 2) >> 3) * 4 + fg));
                if (it == 6) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) wf2[a] = *reinterpret_cast<const u32x4_t*>(ws + sw128(a * 16 + fr, 4 + fg));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a][b] = mfma16t<F16>(it < 8 ? wf[a] : wf2[a], xr[it % 3], acc[a][b]);
                if (it == 7) {
                    if (kt + 1 < KT) {
                        ISSUE(st ^ 1);
                    } else if (vb + G < total) {     // last step (odd: KT is even): the next tile's first step into stage 0
                        TILE_SETUP(vb + G);
                        ISSUE(0);
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();                // everybody is done with stage 1: it becomes the output staging tile
        const bool full = m0 + TM <= p.M;
        int nstore = 0;
        // ---- bias, LeakyReLU(0.1) (vqgan_vae.py:103-104) on the fp32 accumulators (fp16 operands: x alpha first), 16-bit rounding, staging
        const float al = F16 ? p.alpha : 1.f;
        float4 bv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int n = n0 + wn * 64 + a * 16 + 4 * fg;
            bv[a] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const bool leaky = p.act == ACT_LEAKY;
        constexpr int ROWB = TN * 2, CPR = TN / 8;      // staging row bytes, 16-byte chunks per row; one pixel half (128 rows = 64 KiB) at a time
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (wm == half) {
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int row = b * 16 + fr;
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int col = wn * 64 + a * 16 + 4 * fg;
                        float v0 = acc[a][b][0], v1 = acc[a][b][1], v2 = acc[a][b][2], v3 = acc[a][b][3];
                        if constexpr (F16) { v0 *= al; v1 *= al; v2 *= al; v3 *= al; }
                        v0 += bv[a].x; v1 += bv[a].y; v2 += bv[a].z; v3 += bv[a].w;
                        if (leaky) {
                            v0 = v0 > 0.f ? v0 : 0.1f * v0; v1 = v1 > 0.f ? v1 : 0.1f * v1; v2 = v2 > 0.f ? v2 : 0.1f * v2; v3 = v3 > 0.f ? v3 : 0.1f * v3;
                        }
                        const uint2 pk = (F16 && p.half_io) ? make_uint2(pack_f16x2_sat(v0, v1), pack_f16x2_sat(v2, v3)) : make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                        *reinterpret_cast<uint2*>(stg + row * ROWB + (((col >> 3) ^ (row & 7)) << 4) + (col & 4) * 2) = pk;
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            if constexpr (!HEAD) {
#pragma unroll 2
                for (int i = t; i < 128 * CPR; i += 512) {
                    const int row = i / CPR, c = i - row * CPR;
                    const int m = m0 + half * 128 + row;
                    if (m < p.M) {
                        const int ob = m / hw, rem = m - ob * hw;
                        const int oy = (rem / p.Wv) * p.os + opy, ox = (rem % p.Wv) * p.os + opx;
                        const size_t orow = ((size_t)ob * p.Hout + oy) * p.Wout + ox;
                        const uint4 v = *reinterpret_cast<const uint4*>(stg + row * ROWB + ((c ^ (row & 7)) << 4));
                        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + orow * p.ldc + n0 + c * 8) = v;
                    }
                }
                nstore += 128 * CPR / 512;
            } else {
                // four lanes per pixel row, 64 channels each: 8 chunks of 16 bytes from the swizzled staging row, the head's weights as LDS broadcasts
                const float* hwt = reinterpret_cast<const float*>(smem + 2 * STG);
                const int row = t >> 2, q = t & 3;
                float hs[HEAD_MAX];
#pragma unroll
                for (int c = 0; c < HEAD_MAX; ++c) hs[c] = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ch = q * 8 + j;      // 16-byte chunk = channels 8 ch .. 8 ch + 7
                    const uint4 u = *reinterpret_cast<const uint4*>(stg + row * ROWB + ((ch ^ (row & 7)) << 4));
                    float f[8];
                    if (F16 && p.half_io) unpack8s<true>(u, f); else unpack8(u, f);
#pragma unroll
                    for (int c = 0; c < HEAD_MAX; ++c) {
                        if (c < p.head_c) {
                            const float4 w0 = *reinterpret_cast<const float4*>(hwt + c * TN + ch * 8), w1 = *reinterpret_cast<const float4*>(hwt + c * TN + ch * 8 + 4);
                            hs[c] += ((f[0] * w0.x + f[1] * w0.y) + (f[2] * w0.z + f[3] * w0.w)) + ((f[4] * w1.x + f[5] * w1.y) + (f[6] * w1.z + f[7] * w1.w));
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < HEAD_MAX; ++c) {      // the row's four quarters, fixed order
                    hs[c] += __shfl_xor(hs[c], 1, 64);
                    hs[c] += __shfl_xor(hs[c], 2, 64);
                }
                const int m = m0 + half * 128 + row;
                if (q == 0 && m < p.M) {
                    const int ob = m / hw, rem = m - ob * hw;
                    const int oy = (rem / p.Wv) * p.os + opy, ox = (rem % p.Wv) * p.os + opx;
                    float* op = reinterpret_cast<float*>(p.out) + ((size_t)ob * p.head_c * p.Hout + oy) * p.Wout + ox;
#pragma unroll
                    for (int c = 0; c < HEAD_MAX; ++c)
                        if (c < p.head_c) op[(size_t)c * p.Hout * p.Wout] = hs[c] + hwt[HEAD_MAX * TN + c];
                }
                nstore = 0;      // (the store count is not wave-uniform: the next tile's first wait takes everything)
            }
            if (half == 0) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_s_barrier();      // the first half is read out before the second one overwrites it
            }
        }
        pending = (full && !HEAD) ? nstore : 0;
        vb += G;
        if (vb >= total) break;
    }
#undef ISSUE
#undef TILE_SETUP
}

template <bool F16, bool HEAD>
int launch_wide_conv(GemmArgs a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_conv_kernel<F16, HEAD>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return mm_set_hip_error(e, "gemm_wide_conv hipFuncSetAttribute");
        attr_set = true;
    }
    a.tiles_m = (a.M + TM - 1) / TM;
    a.tiles_n = a.N / TN;
    const int total = a.tiles_m * a.tiles_n * (a.par_w[1] ? 4 : 1);
    hipLaunchKernelGGL((gemm_wide_conv_kernel<F16, HEAD>), dim3(total < 256 ? total : 256), dim3(512), SMEM, stream, a);
    return mm_check_launch("gemm_wide_conv_kernel");
}

}  // namespace

// NHWC 16-bit in / out convolutions whose k-steps stay inside one tap (Cin % 64 == 0), with an even number of them (the persistent prefetch relies on the stage
// parity), N a multiple of 256, no residual, at least one tile per CU; fp16 operands only as the single-term half_io form.  With head_w: N == 256 exactly.
bool mm_gemm_wide_conv_eligible(const GemmArgs& a) {
    if (a.mode != MODE_CONV || (a.Cin % 64) || a.K != a.Ktrue || ((a.K / 64) & 1) || a.K < 128 || (a.N % TN) || a.resid_bf16 || a.resid_f32 || a.m_dev || a.splits > 1 ||
        a.epi != EPI_NONE || (a.ldw % 8) || (a.debug & (1 | 2 | 4 | 8)) || (g_mm_debug2 & 8))
        return false;
    if (a.f16 && (!a.half_io || a.terms)) return false;
    if (a.head_w) {
        if (a.N != TN || a.head_c < 1 || a.head_c > HEAD_MAX || !a.head_b) return false;
    } else if (a.out_kind != OUT_BF16 || (a.ldc % 8) || (((uintptr_t)a.out) & 15)) {
        return false;
    }
    if (a.bias && (((uintptr_t)a.bias) & 15)) return false;
    const long hw = (long)a.Hv * a.Wv;
    if (hw <= 0 || (a.M % hw)) return false;
    if ((double)(a.M / hw) * a.Hin * a.Win * a.Cin * 2.0 >= 2147418112.0) return false;      // the input addressed by 32-bit byte offsets below the out-of-bounds marker
    if (a.par_w[1] && (!a.par_w[0] || !a.par_w[2] || !a.par_w[3] || a.TW != 2 || a.K != 4 * a.Cin || a.stride != 1 || a.os != 2)) return false;
    const long tiles = (long)((a.M + TM - 1) / TM) * (a.N / TN) * (a.par_w[1] ? 4 : 1);
    return tiles >= 256;
}

int mm_gemm_wide_conv_launch(GemmArgs a, hipStream_t stream) {
    if (a.head_w) return a.f16 ? launch_wide_conv<true, true>(a, stream) : launch_wide_conv<false, true>(a, stream);
    return a.f16 ? launch_wide_conv<true, false>(a, stream) : launch_wide_conv<false, false>(a, stream);
}
