// The PARITY engine: fp32 storage everywhere + fp32 MFMA (v_mfma_f32_32x32x2_f32) GEMMs / convolutions, fp32 attention, normalisation and
// quantisation kernels -- precision level L0 of SURVEY.md 8c.  The production engine stores activations in bf16 and multiplies in bf16, so
// through 8 layers it cannot meet the north star's "within 1e-3" against the fp32 reference; this engine can (tests/test_gpu_base_size.py)
// and is what `set_precision('parity')` selects.  It follows the reference's operator sequence one to one:
//     mm_f32_gemm          nn.Linear / to_logits                          muse_maskgit_pytorch.py:85,88,118-124,225,233,332
//     mm_f32_conv2d_nhwc   Conv2d / one parity class of ConvTranspose2d   vqgan_vae.py:224-232,255-261,271-277
//     mm_f32_layernorm     LayerNorm                                      muse_maskgit_pytorch.py:63-70
//     mm_f32_geglu         GEGLU (exact erf GELU)                         muse_maskgit_pytorch.py:72-77
//     mm_f32_attend        null kv + l2norm * scale + mask + softmax + AV muse_maskgit_pytorch.py:137-162, attend.py:109-140
//     mm_f32_embed / _text_mask / _axpby                                  muse_maskgit_pytorch.py:322-323, 304, 254
//     mm_f32_glu / _groupnorm / _lfq_* / layout                           vqgan_vae.py:254-276, 424-437
// Speed is secondary here (fp32 MFMA peak is 1/16 of bf16), correctness of every rounding step is the point; the kernels are still tiled
// for gfx950 (LDS-staged 128 x 128 x 16 tiles, 64 accumulator VGPRs per wave) so that the full-size configurations run in milliseconds.
#include <float.h>
#include <string.h>

#include "common.h"
#include "muse_hip_internal.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

namespace {

constexpr int PT = 128;        // tile edge (n and m)
constexpr int PK = 16;         // k per stage
constexpr int PLD = 160;       // LDS row stride in floats: lanes 32..63 of a fragment read (the next k row) land on the other 32 banks

struct F32GemmArgs {
    const float* W; long ldw; int N; int K;
    int M;
    const float* X; long ldx;
    int conv;                                      // 0: dense rows, 1: implicit im2col from an NHWC fp32 image
    int Hin, Win, Cin, TW, stride, off_y, off_x, Hv, Wv, os, py, px, Hout, Wout;
    float* out; long ldc; int out_nchw;
    const float* bias; const float* resid; long ldr; int act;
};

__device__ __forceinline__ float x_elem(const F32GemmArgs& p, int m, int k, int cb, int cy, int cx) {
    if (!p.conv) return p.X[(size_t)m * p.ldx + k];
    const int tap = k / p.Cin, c = k - tap * p.Cin;
    const int ty = tap / p.TW, tx = tap - ty * p.TW;
    const int iy = cy * p.stride + ty + p.off_y, ix = cx * p.stride + tx + p.off_x;
    if (iy < 0 || iy >= p.Hin || ix < 0 || ix >= p.Win) return 0.f;
    return p.X[(((size_t)cb * p.Hin + iy) * p.Win + ix) * p.Cin + c];
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(const F32GemmArgs p) {
    __shared__ float Ws[PK * PLD];
    __shared__ float Xs[PK * PLD];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int wave_n = wid >> 1, wave_m = wid & 1;
    const int tiles_n = (p.N + PT - 1) / PT;
    const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    const int n0 = tile_n * PT, m0 = tile_m * PT;

    // staging: thread t fetches 4 consecutive k of tile rows r0 and r0 + 64
    const int r0 = t >> 2, kq = (t & 3) * 4;
    int cb[2] = {0, 0}, cy[2] = {0, 0}, cx[2] = {0, 0};
    if (p.conv) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + r0 + 64 * i;
            const int mm = m < p.M ? m : 0;
            const int hw = p.Hv * p.Wv;
            cb[i] = mm / hw;
            const int rem = mm - cb[i] * hw;
            cy[i] = rem / p.Wv;
            cx[i] = rem - cy[i] * p.Wv;
        }
    }
    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const bool vec_ok = !p.conv && (p.ldx % 4) == 0 && (((uintptr_t)p.X) & 15) == 0;
    const bool wvec_ok = (p.ldw % 4) == 0 && (((uintptr_t)p.W) & 15) == 0;
    for (int k0 = 0; k0 < p.K; k0 += PK) {
        float wv[2][4], xv[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = n0 + r0 + 64 * i, m = m0 + r0 + 64 * i;
            const int k = k0 + kq;
            if (n < p.N && k + 3 < p.K && wvec_ok) {
                const float4 v = *reinterpret_cast<const float4*>(p.W + (size_t)n * p.ldw + k);
                wv[i][0] = v.x; wv[i][1] = v.y; wv[i][2] = v.z; wv[i][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) wv[i][j] = (n < p.N && k + j < p.K) ? p.W[(size_t)n * p.ldw + k + j] : 0.f;
            }
            if (m < p.M && k + 3 < p.K && vec_ok) {
                const float4 v = *reinterpret_cast<const float4*>(p.X + (size_t)m * p.ldx + k);
                xv[i][0] = v.x; xv[i][1] = v.y; xv[i][2] = v.z; xv[i][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[i][j] = (m < p.M && k + j < p.K) ? x_elem(p, m, k + j, cb[i], cy[i], cx[i]) : 0.f;
            }
        }
        __syncthreads();          // the previous tile's fragments have been read
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Ws[(kq + j) * PLD + r0 + 64 * i] = wv[i][j];
                Xs[(kq + j) * PLD + r0 + 64 * i] = xv[i][j];
            }
        __syncthreads();
        // A operand = weight rows (D rows = output features n), B operand = activation rows (D columns = tokens m):
        // lane l supplies A[n = l % 32][k = l / 32] and B[k = l / 32][m = l % 32]
#pragma unroll
        for (int kk = 0; kk < PK; kk += 2) {
            float af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = Ws[(kk + (lane >> 5)) * PLD + wave_n * 64 + a * 32 + (lane & 31)];
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = Xs[(kk + (lane >> 5)) * PLD + wave_m * 64 + b * 32 + (lane & 31)];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    }
    // epilogue: lane l, register r of block (a, b): n = n0 + wave_n*64 + a*32 + 8*(r/4) + 4*(l/32) + r%4, m = m0 + wave_m*64 + b*32 + l%32
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int m = m0 + wave_m * 64 + b * 32 + (lane & 31);
        if (m >= p.M) continue;
        size_t orow = (size_t)m;
        int ob = 0, oy = 0, ox = 0;
        if (p.conv) {
            const int hw = p.Hv * p.Wv;
            ob = m / hw;
            const int rem = m - ob * hw;
            oy = (rem / p.Wv) * p.os + p.py;
            ox = (rem % p.Wv) * p.os + p.px;
            orow = ((size_t)ob * p.Hout + oy) * p.Wout + ox;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wave_n * 64 + a * 32 + 8 * g + 4 * (lane >> 5);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (n + j >= p.N) continue;
                    float v = acc[a][b][g * 4 + j];
                    if (p.bias) v += p.bias[n + j];
                    if (p.act == ACT_LEAKY) v = v > 0.f ? v : 0.1f * v;       // vqgan_vae.py:103-104
                    if (p.out_nchw) {
                        p.out[(((size_t)ob * p.N + n + j) * p.Hout + oy) * p.Wout + ox] = v;
                    } else {
                        if (p.resid) v += p.resid[orow * p.ldr + n + j];
                        p.out[orow * p.ldc + n + j] = v;
                    }
                }
            }
    }
}

// ---- LayerNorm over the last dim (F.layer_norm, eps 1e-5): two passes over the row in registers / L1, fp32 statistics
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, long ldx, int rows, int D, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ out, long ldo) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)D;
    float v = 0.f;
    for (int c = lane; c < D; c += 64) { const float d = xr[c] - mean; v += d * d; }
    const float rstd = 1.f / sqrtf(wave_sum(v) / (float)D + 1e-5f);
    float* orow = out + (size_t)row * ldo;
    for (int c = lane; c < D; c += 64) orow[c] = (xr[c] - mean) * rstd * gamma[c] + (beta ? beta[c] : 0.f);
}

// ---- GEGLU: out[r][c] = h[r][F + c] * gelu_erf(h[r][c])   (muse_maskgit_pytorch.py:72-77)
__global__ __launch_bounds__(256) void geglu_f32_kernel(const float* __restrict__ h, long ldh, long rows, int F, float* __restrict__ out, long ldo) {
    const long total = rows * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / F;
        const int c = (int)(i - r * F);
        const float xv = h[r * ldh + c], g = h[r * ldh + F + c];
        out[r * ldo + c] = g * (0.5f * xv * (1.f + erff(xv * 0.70710678118654752440f)));
    }
}

// ---- out = b + (a - b) * s   (guidance combine, muse_maskgit_pytorch.py:254)
__global__ __launch_bounds__(256) void cfg_combine_kernel(const float* __restrict__ a, const float* __restrict__ b, float s, long n, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = b[i] + (a[i] - b[i]) * s;
}

// ---- x[row] = tok[ids[row]] + pos[row % n]   (pos may be NULL: plain gather)
__global__ __launch_bounds__(256) void embed_f32_kernel(const int64_t* __restrict__ ids, long rows, int n, const float* __restrict__ tok, int vocab_rows,
                                                        const float* __restrict__ pos, int D, float* __restrict__ x, long ldx) {
    const long total = rows * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / D;
        const int c = (int)(i - r * D);
        long id = ids[r];
        id = id < 0 ? 0 : (id >= vocab_rows ? vocab_rows - 1 : id);
        x[r * ldx + c] = tok[id * D + c] + (pos ? pos[(size_t)(r % n) * D + c] : 0.f);
    }
}

// ---- mask[row] = any(x[row] != 0)   (muse_maskgit_pytorch.py:304)
__global__ __launch_bounds__(256) void text_mask_kernel(const float* __restrict__ x, long rows, int D, uint8_t* __restrict__ mask) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    bool nz = false;
    for (int c = lane; c < D; c += 64) nz |= x[row * D + c] != 0.f;
    const bool any = __ballot(nz) != 0ull;
    if (lane == 0) mask[row] = any ? 1 : 0;
}

// ---- attention: one 256-thread workgroup per (batch, head, 16 queries); extended key 0 is the learned null key / value
struct F32AttnArgs {
    const float* q; long q_sb, q_sh, q_sn;
    const float* k; long k_sb, k_sh, k_sn;
    const float* v; long v_sb, v_sh, v_sn;
    float* out; long o_sb, o_sh, o_sn;
    int B, H, nq, nk;
    const uint8_t* key_mask; long km_sb;
    int normalize;
    const float* q_scale; const float* k_scale; const float* null_k; const float* null_v;
    float scale;
};

__global__ __launch_bounds__(256) void attention_f32_kernel(const F32AttnArgs p) {
    constexpr int QB = 16, KB = 64, LD = 65;
    __shared__ float Qs[QB * LD], Ks[KB * LD], Vs[KB * LD], Ps[QB * LD];
    __shared__ unsigned char valid[KB];
    const int t = threadIdx.x;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB;
    const int qi = t >> 4, j16 = t & 15;      // (query, 16-lane group member)
    {   // Q block: thread (qi, j16) owns dims 4*j16 .. +3
        const int qg = q0 + qi;
        float v4[4] = {0.f, 0.f, 0.f, 0.f};
        if (qg < p.nq) {
            const float* qp = p.q + (size_t)b * p.q_sb + (size_t)h * p.q_sh + (size_t)qg * p.q_sn + 4 * j16;
#pragma unroll
            for (int d = 0; d < 4; ++d) v4[d] = qp[d];
        }
        if (p.normalize) {      // F.normalize(q, dim = -1) * q_scale  (muse_maskgit_pytorch.py:151-153)
            float ss = v4[0] * v4[0] + v4[1] * v4[1] + v4[2] * v4[2] + v4[3] * v4[3];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
            for (int d = 0; d < 4; ++d) v4[d] = v4[d] / den * p.q_scale[4 * j16 + d];
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) Qs[qi * LD + 4 * j16 + d] = v4[d];
    }
    float m_run = -FLT_MAX, l_run = 0.f;
    float o4[4] = {0.f, 0.f, 0.f, 0.f};
    const int has_null = p.null_k ? 1 : 0;
    const int nk_ext = p.nk + has_null;
    for (int kt0 = 0; kt0 < nk_ext; kt0 += KB) {
        __syncthreads();      // previous tile consumed (and Qs written)
        {   // K / V rows: thread (row = t / 4, quarter = t % 4) owns 16 dims
            const int row = t >> 2, qd = (t & 3) * 16;
            const int ke = kt0 + row;                 // extended key index
            float kv[16], vv[16];
            bool ok = ke < nk_ext;
            if (ok && has_null && ke == 0) {
#pragma unroll
                for (int d = 0; d < 16; ++d) { kv[d] = p.null_k[h * 64 + qd + d]; vv[d] = p.null_v[h * 64 + qd + d]; }
            } else if (ok) {
                const int kr = ke - has_null;
                const float* kp = p.k + (size_t)b * p.k_sb + (size_t)h * p.k_sh + (size_t)kr * p.k_sn + qd;
                const float* vp = p.v + (size_t)b * p.v_sb + (size_t)h * p.v_sh + (size_t)kr * p.v_sn + qd;
#pragma unroll
                for (int d = 0; d < 16; ++d) { kv[d] = kp[d]; vv[d] = vp[d]; }
                if (p.key_mask && !p.key_mask[(size_t)b * p.km_sb + kr]) ok = false;
            } else {
#pragma unroll
                for (int d = 0; d < 16; ++d) { kv[d] = 0.f; vv[d] = 0.f; }
            }
            if (p.normalize) {
                float ss = 0.f;
#pragma unroll
                for (int d = 0; d < 16; ++d) ss += kv[d] * kv[d];
                ss += __shfl_xor(ss, 1, 64);
                ss += __shfl_xor(ss, 2, 64);
                const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
                for (int d = 0; d < 16; ++d) kv[d] = kv[d] / den * p.k_scale[qd + d];
            }
#pragma unroll
            for (int d = 0; d < 16; ++d) { Ks[row * LD + qd + d] = kv[d]; Vs[row * LD + qd + d] = vv[d]; }
            if ((t & 3) == 0) valid[row] = ok ? 1 : 0;
        }
        __syncthreads();
        // scores: thread (qi, j16) -> keys 4*j16 .. +3 of the tile
        float s4[4];
        float tmax = -FLT_MAX;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int key = 4 * j16 + c;
            float dot = 0.f;
#pragma unroll 16
            for (int d = 0; d < 64; ++d) dot += Qs[qi * LD + d] * Ks[key * LD + d];
            s4[c] = valid[key] ? dot * p.scale : -FLT_MAX;       // attend.py:126-131 (masked_fill with -finfo.max; such weights are exactly 0)
            tmax = fmaxf(tmax, s4[c]);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = (m_run == -FLT_MAX) ? 0.f : expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float pv = (s4[c] == -FLT_MAX) ? 0.f : expf(s4[c] - m_new);
            Ps[qi * LD + 4 * j16 + c] = pv;
            psum += pv;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) psum += __shfl_xor(psum, o, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        __syncthreads();      // Ps complete
#pragma unroll
        for (int d = 0; d < 4; ++d) o4[d] *= alpha;
        for (int key = 0; key < KB; ++key) {
            const float pv = Ps[qi * LD + key];
#pragma unroll
            for (int d = 0; d < 4; ++d) o4[d] += pv * Vs[key * LD + 4 * j16 + d];
        }
    }
    const int qg = q0 + qi;
    if (qg < p.nq) {
        float* op = p.out + (size_t)b * p.o_sb + (size_t)h * p.o_sh + (size_t)qg * p.o_sn + 4 * j16;
#pragma unroll
        for (int d = 0; d < 4; ++d) op[d] = o4[d] / l_run;
    }
}

// ---- VAE row kernels, NHWC fp32
__global__ __launch_bounds__(256) void glu_f32_kernel(const float* __restrict__ x, long rows, int C, float* __restrict__ out) {      // nn.GLU(dim = channel)
    const long total = rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        const float a = x[r * 2 * C + c], g = x[r * 2 * C + C + c];
        out[i] = a * (1.f / (1.f + expf(-g)));
    }
}

// nn.GroupNorm(groups, C), eps 1e-5, biased variance, optional LeakyReLU(0.1): one workgroup per (image, group), two passes
__global__ __launch_bounds__(256) void groupnorm_f32_kernel(const float* __restrict__ x, int HW, int C, int groups, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int act, float* __restrict__ out) {
    __shared__ float red[8];
    __shared__ float stat[2];
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cg = C / groups;
    const float* xb = x + (size_t)b * HW * C + g * cg;
    const long cnt = (long)HW * cg;
    const int t = threadIdx.x;
    float s = 0.f;
    for (long i = t; i < cnt; i += 256) s += xb[(i / cg) * C + (i % cg)];
    s = wave_sum(s);
    if ((t & 63) == 0) red[t >> 6] = s;
    __syncthreads();
    if (t == 0) stat[0] = (red[0] + red[1] + red[2] + red[3]) / (float)cnt;
    __syncthreads();
    const float mean = stat[0];
    float v = 0.f;
    for (long i = t; i < cnt; i += 256) { const float d = xb[(i / cg) * C + (i % cg)] - mean; v += d * d; }
    v = wave_sum(v);
    if ((t & 63) == 0) red[4 + (t >> 6)] = v;
    __syncthreads();
    if (t == 0) stat[1] = 1.f / sqrtf((red[4] + red[5] + red[6] + red[7]) / (float)cnt + 1e-5f);
    __syncthreads();
    const float rstd = stat[1];
    float* ob = out + (size_t)b * HW * C + g * cg;
    for (long i = t; i < cnt; i += 256) {
        const int c = (int)(i % cg);
        const long o = (i / cg) * C + c;
        float y = (xb[o] - mean) * rstd * gamma[g * cg + c] + beta[g * cg + c];
        if (act) y = y > 0.f ? y : 0.1f * y;
        ob[o] = y;
    }
}

// LFQ.indices_to_codes (+ project_out): out[p][c] = b[c] + sum_j sign_j(id) * w[c][j]   (third-party LFQ as called at vqgan_vae.py:431)
__global__ __launch_bounds__(256) void lfq_decode_f32_kernel(const int64_t* __restrict__ ids, long count, int bits, int C, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out) {
    const long total = count * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / C;
        const int c = (int)(i - pix * C);
        const int64_t id = ids[pix];
        if (!w) { out[i] = ((id >> (bits - 1 - c)) & 1) ? 1.f : -1.f; continue; }
        float a = 0.f;
        for (int j = 0; j < bits; ++j) a += (((id >> (bits - 1 - j)) & 1) ? 1.f : -1.f) * w[(size_t)c * bits + j];
        out[i] = a + bias[c];
    }
}

// LFQ.forward (eval): t = project_in(x) (given, fp32 [count][bits]); ids = sum (t > 0) << (bits-1-j)   (vqgan_vae.py:424)
__global__ __launch_bounds__(256) void lfq_bits_f32_kernel(const float* __restrict__ tin, long count, int bits, int64_t* __restrict__ ids) {
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < count; pix += (long)gridDim.x * blockDim.x) {
        int64_t id = 0;
        for (int j = 0; j < bits; ++j) id |= (int64_t)(tin[pix * bits + j] > 0.f ? 1 : 0) << (bits - 1 - j);
        ids[pix] = id;
    }
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_f32_kernel(const float* __restrict__ in, int B, int C, int HW, float* __restrict__ out) {
    const long total = (long)B * C * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long r = i / C;                 // b * HW + pix
        const long bb = r / HW, pix = r - bb * HW;
        out[i] = in[(bb * C + c) * HW + pix];
    }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_f32_kernel(const float* __restrict__ in, int B, int C, int HW, float* __restrict__ out) {
    const long total = (long)B * C * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i % HW;
        const long r = i / HW;                // b * C + c
        const long bb = r / C;
        const int c = (int)(r - bb * C);
        out[i] = in[(bb * HW + pix) * C + c];
    }
}

inline int grid_for(long items, int cap = 4096) {
    long b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

int launch_gemm(const F32GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return MM_OK;
    if (a.K <= 0) return mm_set_error(MM_ERR_SHAPE, "f32 gemm: K must be positive");
    const long tiles = (long)((a.M + PT - 1) / PT) * ((a.N + PT - 1) / PT);
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)tiles), dim3(256), 0, s, a);
    return mm_check_launch("gemm_f32_kernel");
}

}  // namespace

#define CHKP(p, name) \
    if (!(p)) return mm_set_error(MM_ERR_SHAPE, name " is NULL")

extern "C" {

int mm_f32_gemm(mm_stream_t stream, const float* x, int64_t ldx, const float* w, int64_t ldw, int M, int N, int K, float* out, int64_t ldc,
                const float* bias, int act, const float* resid) {
    if (M == 0 || N == 0) return MM_OK;
    CHKP(x, "x"); CHKP(w, "w"); CHKP(out, "out");
    F32GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.W = w; a.ldw = ldw; a.N = N; a.K = K; a.M = M; a.X = x; a.ldx = ldx;
    a.out = out; a.ldc = ldc; a.bias = bias; a.resid = resid; a.ldr = ldc; a.act = act;
    return launch_gemm(a, (hipStream_t)stream);
}

int mm_f32_conv2d_nhwc(mm_stream_t stream, const float* in, int B, int Hin, int Win, int Cin, const float* w, int Cout, int TH, int TW, int stride,
                       int off_y, int off_x, int Hv, int Wv, int os, int py, int px, int Hout, int Wout, const float* bias, int act,
                       const float* resid, float* out, int out_nchw) {
    CHKP(in, "in"); CHKP(w, "w"); CHKP(out, "out");
    if (out_nchw && resid) return mm_set_error(MM_ERR_UNSUPPORTED, "f32 conv: residual with an NCHW output");
    F32GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.conv = 1;
    a.W = w; a.K = TH * TW * Cin; a.ldw = a.K; a.N = Cout; a.M = B * Hv * Wv; a.X = in;
    a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.TW = TW; a.stride = stride; a.off_y = off_y; a.off_x = off_x; a.Hv = Hv; a.Wv = Wv;
    a.os = os; a.py = py; a.px = px; a.Hout = Hout; a.Wout = Wout;
    a.out = out; a.ldc = Cout; a.out_nchw = out_nchw; a.bias = bias; a.resid = resid; a.ldr = Cout; a.act = act;
    return launch_gemm(a, (hipStream_t)stream);
}

int mm_f32_layernorm(mm_stream_t stream, const float* x, int64_t ldx, int rows, int D, const float* gamma, const float* beta, float* out, int64_t ldo) {
    if (rows == 0) return MM_OK;
    CHKP(x, "x"); CHKP(gamma, "gamma"); CHKP(out, "out");
    hipLaunchKernelGGL(layernorm_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, rows, D, gamma, beta, out, (long)ldo);
    return mm_check_launch("layernorm_f32_kernel");
}

int mm_f32_geglu(mm_stream_t stream, const float* h, int64_t ldh, int64_t rows, int F, float* out, int64_t ldo) {
    if (rows == 0) return MM_OK;
    CHKP(h, "h"); CHKP(out, "out");
    hipLaunchKernelGGL(geglu_f32_kernel, dim3(grid_for(rows * F)), dim3(256), 0, (hipStream_t)stream, h, (long)ldh, (long)rows, F, out, (long)ldo);
    return mm_check_launch("geglu_f32_kernel");
}

int mm_f32_cfg_combine(mm_stream_t stream, const float* cond, const float* null_, float cond_scale, int64_t n, float* out) {
    if (n == 0) return MM_OK;
    CHKP(cond, "cond"); CHKP(null_, "null"); CHKP(out, "out");
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, cond, null_, cond_scale, (long)n, out);
    return mm_check_launch("cfg_combine_kernel");
}

int mm_f32_embed(mm_stream_t stream, const int64_t* ids, int64_t rows, int n, const float* token_emb, int vocab_rows, const float* pos_emb, int D,
                 float* x, int64_t ldx) {
    if (rows == 0) return MM_OK;
    CHKP(ids, "ids"); CHKP(token_emb, "token_emb"); CHKP(x, "x");
    if (n <= 0) return mm_set_error(MM_ERR_SHAPE, "f32 embed: n <= 0");
    hipLaunchKernelGGL(embed_f32_kernel, dim3(grid_for(rows * D)), dim3(256), 0, (hipStream_t)stream, ids, (long)rows, n, token_emb, vocab_rows, pos_emb, D, x,
                       (long)ldx);
    return mm_check_launch("embed_f32_kernel");
}

int mm_f32_text_mask(mm_stream_t stream, const float* text_embeds, int64_t rows, int D, uint8_t* mask) {
    if (rows == 0) return MM_OK;
    CHKP(text_embeds, "text_embeds"); CHKP(mask, "mask");
    hipLaunchKernelGGL(text_mask_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, text_embeds, (long)rows, D, mask);
    return mm_check_launch("text_mask_kernel");
}

// The same attention as fp16 TERM PRODUCTS on the fp16 matrix pipe (attention_x2.hip: the 'f16x2' tier's self-attention kernel), fp32 result -- operator-level
// entry for tests; MM_ERR_UNSUPPORTED outside its shape class (dim_head 64, nk in {128, 192, 256}, nq >= 128, no key mask).
int mm_attend_terms(mm_stream_t stream, const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                    const float* v, int64_t v_sb, int64_t v_sh, int64_t v_sn, float* out, int64_t o_sb, int64_t o_sh, int64_t o_sn, int B, int H, int nq, int nk,
                    int normalize, const float* q_scale, const float* k_scale, const float* null_k, const float* null_v, float scale) {
    if (B == 0 || H == 0 || nq == 0) return MM_OK;
    CHKP(q, "q"); CHKP(k, "k"); CHKP(v, "v"); CHKP(out, "out");
    if (normalize && (!q_scale || !k_scale)) return mm_set_error(MM_ERR_SHAPE, "attend_terms: normalize needs q_scale / k_scale");
    if ((null_k == nullptr) != (null_v == nullptr)) return mm_set_error(MM_ERR_SHAPE, "attend_terms: null_k and null_v come together");
    const int64_t st[] = {q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn, o_sb, o_sh, o_sn};
    bool aligned = ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out)) & 15) == 0;
    for (int64_t x : st) aligned = aligned && (x % 4) == 0;
    if (!aligned) return mm_set_error(MM_ERR_ALIGN, "attend_terms: 16-byte aligned rows required");
    AttnF32Args m;
    memset(&m, 0, sizeof(m));
    m.q = q; m.q_sb = q_sb; m.q_sh = q_sh; m.q_sn = q_sn;
    m.k = k; m.k_sb = k_sb; m.k_sh = k_sh; m.k_sn = k_sn;
    m.v = v; m.v_sb = v_sb; m.v_sh = v_sh; m.v_sn = v_sn;
    m.out = out; m.o_sb = o_sb; m.o_sh = o_sh; m.o_sn = o_sn;
    m.B = B; m.H = H; m.nq = nq; m.nk = nk; m.normalize = normalize;
    m.q_scale = q_scale; m.k_scale = k_scale; m.null_k = null_k; m.null_v = null_v; m.scale = scale; m.dh = 64;
    if (!k_attention_x2_eligible(m)) return mm_set_error(MM_ERR_UNSUPPORTED, "attend_terms: dim_head 64, nk in {128, 192, 256}, nq >= 128");
    return k_attention_x2((hipStream_t)stream, m);
}

int mm_f32_attend(mm_stream_t stream, const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                  const float* v, int64_t v_sb, int64_t v_sh, int64_t v_sn, float* out, int64_t o_sb, int64_t o_sh, int64_t o_sn, int B, int H, int nq, int nk,
                  const uint8_t* key_mask, int64_t km_sb, int normalize, const float* q_scale, const float* k_scale, const float* null_k,
                  const float* null_v, float scale, int dim_head) {
    if (B == 0 || H == 0 || nq == 0) return MM_OK;
    if (dim_head != 32 && dim_head != 64 && dim_head != 128) return mm_set_error(MM_ERR_UNSUPPORTED, "f32 attend: dim_head must be 32, 64 or 128");
    CHKP(q, "q"); CHKP(k, "k"); CHKP(v, "v"); CHKP(out, "out");
    if (normalize && (!q_scale || !k_scale)) return mm_set_error(MM_ERR_SHAPE, "f32 attend: normalize needs q_scale / k_scale");
    if ((null_k == nullptr) != (null_v == nullptr)) return mm_set_error(MM_ERR_SHAPE, "f32 attend: null_k and null_v come together");
    if (nk + (null_k ? 1 : 0) <= 0) return mm_set_error(MM_ERR_SHAPE, "f32 attend: no keys");
    {   // 16-byte aligned rows (every call of parity.py): the fp32-MFMA kernel of attention_f32.hip; anything else: the scalar kernel below
        const int64_t st[] = {q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn, o_sb, o_sh, o_sn};
        bool aligned = ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out)) & 15) == 0;
        for (int64_t x : st) aligned = aligned && (x % 4) == 0;
        if (!aligned && dim_head != 64) return mm_set_error(MM_ERR_ALIGN, "f32 attend: dim_head 32 / 128 needs 16-byte aligned rows");
        if (aligned) {
            AttnF32Args m;
            memset(&m, 0, sizeof(m));
            m.q = q; m.q_sb = q_sb; m.q_sh = q_sh; m.q_sn = q_sn;
            m.k = k; m.k_sb = k_sb; m.k_sh = k_sh; m.k_sn = k_sn;
            m.v = v; m.v_sb = v_sb; m.v_sh = v_sh; m.v_sn = v_sn;
            m.out = out; m.o_sb = o_sb; m.o_sh = o_sh; m.o_sn = o_sn;
            m.B = B; m.H = H; m.nq = nq; m.nk = nk; m.key_mask = key_mask; m.km_sb = km_sb; m.normalize = normalize;
            m.q_scale = q_scale; m.k_scale = k_scale; m.null_k = null_k; m.null_v = null_v; m.scale = scale; m.dh = dim_head;
            return k_attention_f32((hipStream_t)stream, m);
        }
    }
    F32AttnArgs a;
    a.q = q; a.q_sb = q_sb; a.q_sh = q_sh; a.q_sn = q_sn;
    a.k = k; a.k_sb = k_sb; a.k_sh = k_sh; a.k_sn = k_sn;
    a.v = v; a.v_sb = v_sb; a.v_sh = v_sh; a.v_sn = v_sn;
    a.out = out; a.o_sb = o_sb; a.o_sh = o_sh; a.o_sn = o_sn;
    a.B = B; a.H = H; a.nq = nq; a.nk = nk; a.key_mask = key_mask; a.km_sb = km_sb; a.normalize = normalize;
    a.q_scale = q_scale; a.k_scale = k_scale; a.null_k = null_k; a.null_v = null_v; a.scale = scale;
    hipLaunchKernelGGL(attention_f32_kernel, dim3((nq + 15) / 16, H, B), dim3(256), 0, (hipStream_t)stream, a);
    return mm_check_launch("attention_f32_kernel");
}

int mm_f32_glu_nhwc(mm_stream_t stream, const float* x, int64_t rows, int C, float* out) {
    if (rows == 0) return MM_OK;
    CHKP(x, "x"); CHKP(out, "out");
    hipLaunchKernelGGL(glu_f32_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, x, (long)rows, C, out);
    return mm_check_launch("glu_f32_kernel");
}

int mm_f32_groupnorm_nhwc(mm_stream_t stream, const float* x, int B, int HW, int C, int groups, const float* gamma, const float* beta, int act, float* out) {
    if (B == 0) return MM_OK;
    CHKP(x, "x"); CHKP(gamma, "gamma"); CHKP(beta, "beta"); CHKP(out, "out");
    if (groups <= 0 || C % groups) return mm_set_error(MM_ERR_SHAPE, "f32 groupnorm: C must be divisible by groups");
    hipLaunchKernelGGL(groupnorm_f32_kernel, dim3(B * groups), dim3(256), 0, (hipStream_t)stream, x, HW, C, groups, gamma, beta, act, out);
    return mm_check_launch("groupnorm_f32_kernel");
}

int mm_f32_lfq_decode(mm_stream_t stream, const int64_t* ids, int64_t count, int bits, int C, const float* w, const float* b, float* out) {
    if (count == 0) return MM_OK;
    CHKP(ids, "ids"); CHKP(out, "out");
    if (!w && C != bits) return mm_set_error(MM_ERR_SHAPE, "f32 lfq_decode: no projection needs C == bits");
    if (w && !b) return mm_set_error(MM_ERR_SHAPE, "f32 lfq_decode: bias is NULL");
    hipLaunchKernelGGL(lfq_decode_f32_kernel, dim3(grid_for(count * C)), dim3(256), 0, (hipStream_t)stream, ids, (long)count, bits, C, w, b, out);
    return mm_check_launch("lfq_decode_f32_kernel");
}

int mm_f32_lfq_bits(mm_stream_t stream, const float* t_in, int64_t count, int bits, int64_t* ids) {
    if (count == 0) return MM_OK;
    CHKP(t_in, "t_in"); CHKP(ids, "ids");
    if (bits <= 0 || bits > 62) return mm_set_error(MM_ERR_SHAPE, "f32 lfq_bits: bits out of range");
    hipLaunchKernelGGL(lfq_bits_f32_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream, t_in, (long)count, bits, ids);
    return mm_check_launch("lfq_bits_f32_kernel");
}

int mm_f32_nchw_to_nhwc(mm_stream_t stream, const float* in, int B, int C, int HW, float* out) {
    if (B == 0) return MM_OK;
    CHKP(in, "in"); CHKP(out, "out");
    hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel, dim3(grid_for((long)B * C * HW)), dim3(256), 0, (hipStream_t)stream, in, B, C, HW, out);
    return mm_check_launch("nchw_to_nhwc_f32_kernel");
}

int mm_f32_nhwc_to_nchw(mm_stream_t stream, const float* in, int B, int C, int HW, float* out) {
    if (B == 0) return MM_OK;
    CHKP(in, "in"); CHKP(out, "out");
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(grid_for((long)B * C * HW)), dim3(256), 0, (hipStream_t)stream, in, B, C, HW, out);
    return mm_check_launch("nhwc_to_nchw_f32_kernel");
}

}  // extern "C"
