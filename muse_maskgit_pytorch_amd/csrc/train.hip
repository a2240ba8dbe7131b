// Backward kernels of the transformer training step (muse_maskgit_pytorch.py:623-741 runs `loss.backward()` through autograd; here
// every operator of Transformer.forward has a hand-written gradient kernel), gfx950.
//
//   transpose_bf16      : [R][C] -> [C][R]; the NT MFMA GEMM (gemm*.hip) then serves dX = dY * W and dW = dY^T * X
//   layernorm_bwd       : LayerNorm(x; gamma) with fp32 x (mmp.py:63-70); dx is ACCUMULATED into the residual-stream gradient
//   geglu_ln_bwd        : gradient of LayerNorm_inner(gate * gelu(x)) w.r.t. the w1 output h = [x | gate] (mmp.py:72-77, 86)
//   ce_bwd              : d(mean cross-entropy)/d(logits) in bf16 (mmp.py:343)
//   embed_bwd           : token / position embedding gradients (mmp.py:322-323), fixed summation order
//   colsum partial sums : gamma gradients are reduced deterministically (per-workgroup partials, then one pass over them)
// All row kernels keep a row in registers: one wave per row, a lane owns fixed columns, so the gamma partial of a workgroup is a
// plain per-lane accumulation over its rows.
#include <math.h>

#include "common.h"
#include "muse_hip_internal.h"

namespace {

// ------------------------------------------------------------------------------------------------ transpose
// wrows: output columns written per output row -- `rows`, or (mm_train_step's operands) rows rounded up to 64 with the padding written as zeros HERE (the tile rows
// past `rows` are zero-filled on the way in) instead of by a memset of the whole output in front of the launch
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, long rows, long cols, long ldi,
                                                             bf16_t* __restrict__ out, long ldo, long wrows) {
    __shared__ bf16_t tile[64][72];          // row pitch 144 B
    const long r0 = (long)blockIdx.y * 64, c0 = (long)blockIdx.x * 64;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = t + i * 256;         // 512 chunks of 8 elements
        const int r = idx >> 3, c = (idx & 7) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + r < rows) {
            if (c0 + c + 8 <= cols) v = *reinterpret_cast<const uint4*>(in + (r0 + r) * ldi + c0 + c);
            else {
                bf16_t tmp[8];
                for (int j = 0; j < 8; ++j) tmp[j] = (c0 + c + j < cols) ? in[(r0 + r) * ldi + c0 + c + j] : (bf16_t)0;
                v = *reinterpret_cast<const uint4*>(tmp);
            }
        }
        *reinterpret_cast<uint4*>(&tile[r][c]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = t + i * 256;
        const int c = idx >> 3, r = (idx & 7) * 8;       // output row c0 + c, output columns r0 + r .. +7
        if (c0 + c >= cols) continue;
        bf16_t tmp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tmp[j] = tile[r + j][c];
        bf16_t* op = out + (c0 + c) * ldo + r0 + r;
        if (r0 + r + 8 <= wrows) *reinterpret_cast<uint4*>(op) = *reinterpret_cast<const uint4*>(tmp);
        else
            for (int j = 0; j < 8; ++j)
                if (r0 + r + j < wrows) op[j] = tmp[j];
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward (fp32 x)
// y = (x - mean) * rstd * gamma + beta.  With xhat = (x - mean) * rstd and g = dy * gamma:
//   dx = rstd * (g - mean_D(g) - xhat * mean_D(g * xhat)),   dgamma = sum_rows dy * xhat      (beta is a buffer: no gradient)
constexpr int LNB_ROWS = 4;        // rows per wave -> 16 rows per workgroup (512 workgroups at 8192 rows)
template <int NIT>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, long ldx, const bf16_t* __restrict__ dy, long lddy,
                                                            const float* __restrict__ gamma, const int32_t* __restrict__ row_index,
                                                            int rows, int D, float* __restrict__ dx, long lddx, int accumulate,
                                                            float* __restrict__ dgamma_part, bf16_t* __restrict__ dxb) {
    __shared__ float red[4][64 * 4 * NIT];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nvec = D >> 2;
    float4 gm[NIT], dg[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        gm[it] = c < nvec ? *reinterpret_cast<const float4*>(gamma + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int row0 = (blockIdx.x * 4 + wid) * LNB_ROWS;
    for (int rr = 0; rr < LNB_ROWS; ++rr) {
        const int row = row0 + rr;
        if (row >= rows) break;
        // row_index: the forward normalised a GATHERED subset of rows (final norm before to_logits); dy row `row` belongs to x row
        // row_index[row], and dx goes there
        const long src = row_index ? (long)row_index[row] : (long)row;
        const float* xr = x + src * ldx;
        const bf16_t* dyr = dy + (long)row * lddy;
        float4 v[NIT], gy[NIT];
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane;
            v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            gy[it] = v[it];
            if (c < nvec) {
                v[it] = *reinterpret_cast<const float4*>(xr + c * 4);
                const uint2 d2 = *reinterpret_cast<const uint2*>(dyr + c * 4);
                gy[it] = make_float4(bf16lo(d2.x), bf16hi(d2.x), bf16lo(d2.y), bf16hi(d2.y));
                sum += (v[it].x + v[it].y) + (v[it].z + v[it].w);
            }
        }
        const float mean = wave_sum(sum) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane;
            if (c < nvec) {
                const float a = v[it].x - mean, b = v[it].y - mean, cc = v[it].z - mean, d = v[it].w - mean;
                sq += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float rstd = 1.f / sqrtf(wave_sum(sq) / (float)D + 1e-5f);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane;
            if (c < nvec) {
                // v <- xhat, gy stays dy; accumulate the two row means and the gamma partial
                v[it].x = (v[it].x - mean) * rstd; v[it].y = (v[it].y - mean) * rstd;
                v[it].z = (v[it].z - mean) * rstd; v[it].w = (v[it].w - mean) * rstd;
                dg[it].x += gy[it].x * v[it].x; dg[it].y += gy[it].y * v[it].y;
                dg[it].z += gy[it].z * v[it].z; dg[it].w += gy[it].w * v[it].w;
                const float g0 = gy[it].x * gm[it].x, g1 = gy[it].y * gm[it].y, g2 = gy[it].z * gm[it].z, g3 = gy[it].w * gm[it].w;
                s1 += (g0 + g1) + (g2 + g3);
                s2 += (g0 * v[it].x + g1 * v[it].y) + (g2 * v[it].z + g3 * v[it].w);
            }
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
        float* dxr = dx + src * lddx;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane;
            if (c < nvec) {
                float4 o;
                o.x = rstd * (gy[it].x * gm[it].x - c1 - v[it].x * c2);
                o.y = rstd * (gy[it].y * gm[it].y - c1 - v[it].y * c2);
                o.z = rstd * (gy[it].z * gm[it].z - c1 - v[it].z * c2);
                o.w = rstd * (gy[it].w * gm[it].w - c1 - v[it].w * c2);
                if (accumulate) {
                    const float4 old = *reinterpret_cast<const float4*>(dxr + c * 4);
                    o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                }
                *reinterpret_cast<float4*>(dxr + c * 4) = o;
                // (mm_train_step) the bf16 image of the updated gradient row -- the operand of the GEMMs that follow -- in the same pass: the values
                // f32_to_bf16_kernel would produce from dx
                if (dxb) *reinterpret_cast<uint2*>(dxb + src * (long)D + c * 4) = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
            }
        }
    }
    // workgroup partial of dgamma: waves 1..3 -> LDS, wave 0 adds in fixed order and writes
#pragma unroll
    for (int it = 0; it < NIT; ++it) *reinterpret_cast<float4*>(&red[wid][(it * 64 + lane) * 4]) = dg[it];
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane;
            if (c < nvec) {
                float4 a = *reinterpret_cast<const float4*>(&red[0][c * 4]);
                for (int w = 1; w < 4; ++w) {
                    const float4 b = *reinterpret_cast<const float4*>(&red[w][c * 4]);
                    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                }
                *reinterpret_cast<float4*>(dgamma_part + (long)blockIdx.x * D + c * 4) = a;
            }
        }
    }
}

// out[c] = sum over p of part[p][c] in a fixed order (deterministic): 64 columns per workgroup.  The association is that of rounds 1-5 -- eight serial chains per
// column, chain (rg, a) = the parts p with p mod 4 == rg and (p / 4) mod 2 == a in ascending order, out = ((s00 + s01) + (s10 + s11)) + ((s20 + s21) + (s30 + s31)) --
// so every sum is bit-identical to what the 256-thread kernel produced.  Round 6: one WAVE per chain (512 threads) and sixteen loads of a chain in flight per lane:
// the 1024 x 64 partials of a q / k scale gradient took 62 us on one workgroup with two dependent loads per trip (a training step reduces 128 such tables: 2.5 ms
// of its side stream), now 8 trips of latency.
__global__ __launch_bounds__(512) void colsum_kernel(const float* __restrict__ part, int nparts, long D, float* __restrict__ out) {
    __shared__ float red[8][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;      // chain rg = w & 3, a = w >> 2: parts (w & 3) + 4 * (w >> 2) + 8 i
    const long c = (long)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < D) {
        int p = (w & 3) + 4 * (w >> 2);
        constexpr int U = 16;
        for (; p + 8 * (U - 1) < nparts; p += 8 * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = part[(long)(p + 8 * u) * D + c];
#pragma unroll
            for (int u = 0; u < U; ++u) s += v[u];
        }
        for (; p < nparts; p += 8) s += part[(long)p * D + c];
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < D) out[c] = ((red[0][lane] + red[4][lane]) + (red[1][lane] + red[5][lane])) + ((red[2][lane] + red[6][lane]) + (red[3][lane] + red[7][lane]));
}

// The same sums in the same association for WIDE inputs (the split-K slabs of a weight gradient: D = N x K, a handful of parts): one thread owns 4 adjacent
// columns and all 8 accumulators of colsum_kernel's tree -- stream rg = p mod 4, accumulator (p / 4) mod 2 -- so every load of a column quad is in flight
// at once, 16 bytes per lane; out = ((s00 + s01) + (s10 + s11)) + ((s20 + s21) + (s30 + s31)) exactly as above.
__global__ __launch_bounds__(256) void colsum_wide_kernel(const float* __restrict__ part, int nparts, long D, float* __restrict__ out) {
    const long c = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= D) return;
    float4 acc[4][2];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) acc[rg][0] = acc[rg][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p0 = 0; p0 < nparts; p0 += 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p0 + j < nparts ? *reinterpret_cast<const float4*>(part + (long)(p0 + j) * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (p0 + j < nparts) {      // (a missing part adds nothing, as in colsum_kernel: no "+ 0" that could turn -0 into +0)
                float4& a = acc[j & 3][j >> 2];
                a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w;
            }
        }
    }
    float4 r[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) r[rg] = make_float4(acc[rg][0].x + acc[rg][1].x, acc[rg][0].y + acc[rg][1].y, acc[rg][0].z + acc[rg][1].z, acc[rg][0].w + acc[rg][1].w);
    float4 o;
    o.x = (r[0].x + r[1].x) + (r[2].x + r[3].x); o.y = (r[0].y + r[1].y) + (r[2].y + r[3].y);
    o.z = (r[0].z + r[1].z) + (r[2].z + r[3].z); o.w = (r[0].w + r[1].w) + (r[2].w + r[3].w);
    *reinterpret_cast<float4*>(out + c) = o;
}

// ------------------------------------------------------------------------------------------------ GEGLU + inner LayerNorm backward
// forward (norm_act.hip geglu_ln_kernel): a = gate * gelu_erf(x) over the F valid of Fp columns, z = LN(a; gamma).
// gelu'(x) = Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_f(float x) { return x * gelu_phi(x); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return gelu_phi(x) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

template <int NIT>      // 16-byte iterations per lane: 3 (Fp <= 1536) / 6 / 12
__global__ __launch_bounds__(256) void geglu_ln_bwd_kernel(const bf16_t* __restrict__ h, long ldh, const bf16_t* __restrict__ dz, long lddz,
                                                           const float* __restrict__ gamma, int rows, int F, int Fp,
                                                           bf16_t* __restrict__ dh, long lddh, float* __restrict__ dgamma_part) {
    __shared__ float red[4][64 * 8 * NIT];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nch = Fp >> 3;
    float dg[NIT][8];
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) dg[it][j] = 0.f;
    const int row0 = (blockIdx.x * 4 + wid) * LNB_ROWS;
    for (int rr = 0; rr < LNB_ROWS; ++rr) {
        const int row = row0 + rr;
        if (row >= rows) break;
        const bf16_t* hr = h + (long)row * ldh;
        const bf16_t* dzr = dz + (long)row * lddz;
        float a[NIT][8], ph[NIT][8];
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[it][j] = 0.f; ph[it][j] = 0.f; }
            if (c < nch) {
                float xv[8], gv[8];
                unpack8(*reinterpret_cast<const uint4*>(hr + c * 8), xv);
                unpack8(*reinterpret_cast<const uint4*>(hr + Fp + c * 8), gv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float cdf = gelu_phi(xv[j]);      // Phi(x): gelu = x Phi, gelu' = Phi + x phi
                    ph[it][j] = cdf;
                    const float val = (c * 8 + j < F) ? gv[j] * (xv[j] * cdf) : 0.f;
                    a[it][j] = val;
                    sum += val;
                }
            }
        }
        const float mean = wave_sum(sum) / (float)F;
        float sq = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane;
            if (c < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (c * 8 + j < F) { const float d = a[it][j] - mean; sq += d * d; }
            }
        }
        const float rstd = 1.f / sqrtf(wave_sum(sq) / (float)F + 1e-5f);
        float s1 = 0.f, s2 = 0.f;
        float g[NIT][8];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) g[it][j] = 0.f;
            if (c < nch) {
                float dv[8];
                unpack8(*reinterpret_cast<const uint4*>(dzr + c * 8), dv);
                const float4 g0 = *reinterpret_cast<const float4*>(gamma + c * 8), g1 = *reinterpret_cast<const float4*>(gamma + c * 8 + 4);
                const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (c * 8 + j < F) {
                        const float xh = (a[it][j] - mean) * rstd;
                        a[it][j] = xh;                         // a <- xhat
                        dg[it][j] += dv[j] * xh;
                        const float gj = dv[j] * gg[j];
                        g[it][j] = gj;
                        s1 += gj;
                        s2 += gj * xh;
                    }
                }
            }
        }
        const float c1 = wave_sum(s1) / (float)F, c2 = wave_sum(s2) / (float)F;
        bf16_t* dhr = dh + (long)row * lddh;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane;
            if (c < nch) {
                float xv[8], gv[8], ox[8], og[8];
                unpack8(*reinterpret_cast<const uint4*>(hr + c * 8), xv);
                unpack8(*reinterpret_cast<const uint4*>(hr + Fp + c * 8), gv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float da = 0.f;
                    if (c * 8 + j < F) da = rstd * (g[it][j] - c1 - a[it][j] * c2);
                    og[j] = da * (xv[j] * ph[it][j]);                                                                      // d gate
                    ox[j] = da * gv[j] * (ph[it][j] + xv[j] * 0.39894228040143267794f * __expf(-0.5f * xv[j] * xv[j]));     // d x
                }
                *reinterpret_cast<uint4*>(dhr + c * 8) = pack8(ox);
                *reinterpret_cast<uint4*>(dhr + Fp + c * 8) = pack8(og);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wid][(it * 64 + lane) * 8 + j] = dg[it][j];
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane;
            if (c < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float s = red[0][c * 8 + j];
                    for (int w = 1; w < 4; ++w) s += red[w][c * 8 + j];
                    dgamma_part[(long)blockIdx.x * Fp + c * 8 + j] = s;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ cross-entropy backward
// loss = mean over R rows of (logsumexp(l_r) - l_r[label_r]);  dl[r][v] = (softmax(l_r)[v] - [v == label_r]) / R, written in bf16.
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, long ld, int V, const int64_t* __restrict__ labels,
                                                     float scale, bf16_t* __restrict__ dl, long ldd, float* __restrict__ row_loss) {
    __shared__ float sm[4], ss[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* lr = logits + (size_t)row * ld;
    float m = -INFINITY, s = 0.f;
    for (int i = tid * 4; i < V; i += 256 * 4) {
        const float4 x = *reinterpret_cast<const float4*>(lr + i);
        const float mx = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
        if (mx > m) { s *= expf(m - mx); m = mx; }
        s += expf(x.x - m) + expf(x.y - m) + expf(x.z - m) + expf(x.w - m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64), os = __shfl_xor(s, o, 64);
        const float nm = fmaxf(m, om);
        s = (nm == -INFINITY) ? 0.f : s * expf(m - nm) + os * expf(om - nm);
        m = nm;
    }
    if (lane == 0) { sm[wid] = m; ss[wid] = s; }
    __syncthreads();
    float M = sm[0], S = ss[0];
    for (int w = 1; w < 4; ++w) {
        const float nm = fmaxf(M, sm[w]);
        S = S * expf(M - nm) + ss[w] * expf(sm[w] - nm);
        M = nm;
    }
    const float inv = scale / S;
    const int lab = (int)labels[row];
    // (mm_train_step) the row's loss from the same (M, S): what ce_rows_kernel (sampling.hip) computes with the same operations, so the forward pass over
    // the logits is not needed; -1 marks a row without a valid label for ce_finish_kernel
    if (row_loss && tid == 0) row_loss[row] = (lab >= 0 && lab < V) ? (M + logf(S)) - lr[lab] : -1.f;
    bf16_t* dr = dl + (size_t)row * ldd;
    for (int i = tid * 4; i < V; i += 256 * 4) {      // second sweep: the row (<= 256 KiB) is L2-resident
        const float4 x = *reinterpret_cast<const float4*>(lr + i);
        float p0 = expf(x.x - M) * inv, p1 = expf(x.y - M) * inv, p2 = expf(x.z - M) * inv, p3 = expf(x.w - M) * inv;
        if (lab == i) p0 -= scale; else if (lab == i + 1) p1 -= scale; else if (lab == i + 2) p2 -= scale; else if (lab == i + 3) p3 -= scale;
        *reinterpret_cast<uint2*>(dr + i) = make_uint2(pack_bf16x2(p0, p1), pack_bf16x2(p2, p3));
    }
}

// The same gradient with the row held in REGISTERS between the two sweeps (round 6): 1024 threads x 16 float4 = a whole row of up to 65536 logits.  The kernel above reads
// every row twice -- "L2-resident" was wrong at the training head's size: 512 rows of 256 KiB are in flight at once, so the second sweep came from HBM again (2.9 GB read
// + 0.72 GB written per step, 0.72-0.75 ms on the backward's dependent chain).  One read: 2.2 GB.  Per-thread online (max, sum) over its own 64 values, then a fixed-order
// combine over the 16 waves -- a different association of the softmax denominator from the kernel above (both deterministic; the driver and mm_train_step share this one).
// The 131072 exponentials of a row are the hardware exp2 of x log2 e (~1e-6 relative: the gradient leaves as bf16, the loss as M + log S): with libm's expf the kernel
// was bound by them (584 us against 723 for two sweeps), not by its 2.2 GB.
__global__ __launch_bounds__(1024) void ce_bwd_row_kernel(const float* __restrict__ logits, long ld, int V, const int64_t* __restrict__ labels,
                                                          float scale, bf16_t* __restrict__ dl, long ldd, float* __restrict__ row_loss) {
    __shared__ float sm[16], ss[16];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* lr = logits + (size_t)row * ld;
    const int nv = V >> 2;                      // float4 chunks of the row (<= 16384)
    float4 x[16];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = tid + j * 1024;
        x[j] = c < nv ? *reinterpret_cast<const float4*>(lr + (size_t)c * 4) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        m = fmaxf(m, fmaxf(fmaxf(x[j].x, x[j].y), fmaxf(x[j].z, x[j].w)));
    }
    float s = 0.f;
    if (m > -INFINITY) {
#pragma unroll
        for (int j = 0; j < 16; ++j) s += (__expf(x[j].x - m) + __expf(x[j].y - m)) + (__expf(x[j].z - m) + __expf(x[j].w - m));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64), os = __shfl_xor(s, o, 64);
        const float nm = fmaxf(m, om);
        s = (nm == -INFINITY) ? 0.f : s * expf(m - nm) + os * expf(om - nm);
        m = nm;
    }
    if (lane == 0) { sm[wid] = m; ss[wid] = s; }
    __syncthreads();
    float M = sm[0], S = ss[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) {
        const float nm = fmaxf(M, sm[w]);
        S = (nm == -INFINITY) ? 0.f : S * expf(M - nm) + ss[w] * expf(sm[w] - nm);
        M = nm;
    }
    const float inv = scale / S;
    const int lab = (int)labels[row];
    if (row_loss && tid == 0) row_loss[row] = (lab >= 0 && lab < V) ? (M + logf(S)) - lr[lab] : -1.f;      // (-1: no valid label, ce_finish_kernel skips the row)
    bf16_t* dr = dl + (size_t)row * ldd;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = tid + j * 1024;
        if (c >= nv) continue;
        const int i = c * 4;
        float p0 = __expf(x[j].x - M) * inv, p1 = __expf(x[j].y - M) * inv, p2 = __expf(x[j].z - M) * inv, p3 = __expf(x[j].w - M) * inv;
        if (lab == i) p0 -= scale; else if (lab == i + 1) p1 -= scale; else if (lab == i + 2) p2 -= scale; else if (lab == i + 3) p3 -= scale;
        *reinterpret_cast<uint2*>(dr + i) = make_uint2(pack_bf16x2(p0, p1), pack_bf16x2(p2, p3));
    }
}

// ------------------------------------------------------------------------------------------------ embedding backward
// x[b*n + p] = token_emb[ids] + pos_emb[p]  ->  dpos[p] = sum_b dx[b*n + p], dtoken[id] += sum over the rows that carry id.  Both sums run in
// a FIXED order (ascending row index), so the whole backward pass is bit-reproducible: many rows share an id (the mask id covers half the
// batch), and fp32 atomics would add them in whatever order the hardware schedules.
__global__ __launch_bounds__(256) void embed_pos_bwd_kernel(int B, int n, int D, const float* __restrict__ dx, float* __restrict__ dpos) {
    const int p = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += 256) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dx[((long)b * n + p) * D + d];
        dpos[(long)p * D + d] = s;
    }
}
// one workgroup per token row r: if r is the FIRST row with its id, it sums the gradient rows of every row with that id in ascending order and
// is the only writer of dtoken[id]; any other workgroup leaves after the look-back.  Matches are found 256 rows at a time (one ballot per wave).
// blockIdx.y selects a slab of 256 * NC columns: the chain of additions per column is serial (the mask id owns about half of all rows), so the columns are
// spread over as many workgroups as there are 256-column slabs (NC = 1: 32 rows' loads in flight per lane) -- the same sums in the same order.
template <int NC>      // columns per thread: a workgroup covers 256 * NC columns
__global__ __launch_bounds__(256) void embed_token_bwd_kernel(const int64_t* __restrict__ ids, int R, int D, const float* __restrict__ dx,
                                                              float* __restrict__ dtoken) {
    constexpr int U = 32 / NC;      // matching rows per trip: U * NC loads in flight per lane
    __shared__ unsigned long long masks[4];
    __shared__ int seen;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c0 = blockIdx.y * 256 * NC;
    const int64_t id = ids[r];
    if (tid == 0) seen = 0;
    __syncthreads();
    bool hit = false;
    for (int j = tid; j < r; j += 256) hit |= ids[j] == id;
    if (__ballot(hit) != 0ull && lane == 0) seen = 1;
    __syncthreads();
    if (seen) return;
    float acc[NC];     // columns tid, tid + 256, ...
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = 0.f;
    for (int base = r & ~255; base < R; base += 256) {
        const int j = base + tid;
        const unsigned long long m = __ballot(j >= r && j < R && ids[j] == id);
        if (lane == 0) masks[wid] = m;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned long long mm = masks[w];
            while (mm) {      // U matching rows per trip: their loads are in flight together, the additions stay in ascending row order
                int jj[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    jj[u] = mm ? base + w * 64 + __builtin_ctzll(mm) : -1;
                    mm &= mm - 1;      // (0 stays 0)
                }
                float v[U][NC];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int i = 0; i < NC; ++i)
                        v[u][i] = (jj[u] >= 0 && c0 + tid + i * 256 < D) ? dx[(long)jj[u] * D + c0 + tid + i * 256] : 0.f;
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (jj[u] >= 0) {
#pragma unroll
                        for (int i = 0; i < NC; ++i) acc[i] += v[u][i];
                    }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NC; ++i)
        if (c0 + tid + i * 256 < D) dtoken[id * D + c0 + tid + i * 256] += acc[i];
}

// The same sum in TWO levels (round 6, used when the caller brings a workspace): the mask id owns about two thirds of a training batch's rows, and its serial
// chain -- 5500 additions per column at the base size -- made the single-level kernel above 0.41 ms at the very end of the backward's dependent chain.
//   LOCAL = true  : row r acts when it is the first row WITH ITS ID INSIDE ITS BLOCK of 256 rows; it sums the block's rows with that id (ascending) into
//                   part[r] and sets first[r] (every other row clears its flag);
//   LOCAL = false : row r acts when first[r] is set and no earlier flagged row carries its id; it adds part[j] of the flagged rows j >= r with that id
//                   (ascending) to dtoken[id].
// dtoken[id] = sum over blocks (ascending) of (sum over the block's rows, ascending): deterministic, a different association from the single-level kernel.
template <bool LOCAL>
__global__ __launch_bounds__(256) void embed_token_bwd2_kernel(const int64_t* __restrict__ ids, int R, int D, const float* __restrict__ src,
                                                               float* __restrict__ dst, uint8_t* __restrict__ first) {
    constexpr int U = 32;
    __shared__ unsigned long long masks[4];
    __shared__ int seen;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = blockIdx.y * 256 + tid;
    if (!LOCAL && !first[r]) return;
    const int lo = LOCAL ? (r & ~255) : 0;
    const int hi = LOCAL ? (lo + 256 < R ? lo + 256 : R) : R;
    const int64_t id = ids[r];
    if (tid == 0) seen = 0;
    __syncthreads();
    bool hit = false;
    for (int j = lo + tid; j < r; j += 256) hit |= ids[j] == id && (LOCAL || first[j]);
    if (__ballot(hit) != 0ull && lane == 0) seen = 1;
    __syncthreads();
    if (LOCAL && blockIdx.y == 0 && tid == 0) first[r] = seen ? 0 : 1;
    if (seen) return;
    float acc = 0.f;
    for (int base = r & ~255; base < hi; base += 256) {
        const int j = base + tid;
        const unsigned long long m = __ballot(j >= r && j < hi && ids[j] == id && (LOCAL || first[j]));
        if (lane == 0) masks[wid] = m;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned long long mm = masks[w];
            while (mm) {
                int jj[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    jj[u] = mm ? base + w * 64 + __builtin_ctzll(mm) : -1;
                    mm &= mm - 1;
                }
                float v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = (jj[u] >= 0 && c < D) ? src[(long)jj[u] * D + c] : 0.f;
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (jj[u] >= 0) acc += v[u];
            }
        }
        __syncthreads();
    }
    if (c < D) {
        if (LOCAL) dst[(long)r * D + c] = acc;
        else dst[id * D + c] += acc;
    }
}

// ------------------------------------------------------------------------------------------------ BCE head backward (TokenCritic)
// logits x_m = e_m . w (Linear(dim, 1), mmp.py:383-386), loss = mean_m BCEWithLogits(x_m, y_m) (mmp.py:345-346):
//   g_m = (sigmoid(x_m) - y_m) / M,   de_m = g_m * w (bf16),   dw = sum_m g_m * e_m  (per-workgroup partials, reduced by colsum)
template <int NIT>      // 16-byte iterations per lane over D
__global__ __launch_bounds__(256) void bce_head_bwd_kernel(const bf16_t* __restrict__ e, long lde, const float* __restrict__ x,
                                                           const float* __restrict__ y, const float* __restrict__ w, int rows, int D,
                                                           float inv_count, bf16_t* __restrict__ de, long ldde, float* __restrict__ dw_part) {
    __shared__ float red[4][64 * 8 * NIT];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nch = D >> 3;
    float wv[NIT][8], acc[NIT][8];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
#pragma unroll
        for (int j = 0; j < 8; ++j) { wv[it][j] = c < nch ? w[c * 8 + j] : 0.f; acc[it][j] = 0.f; }
    }
    const int row0 = (blockIdx.x * 4 + wid) * LNB_ROWS;
    for (int rr = 0; rr < LNB_ROWS; ++rr) {
        const int row = row0 + rr;
        if (row >= rows) break;
        const float g = (1.f / (1.f + expf(-x[row])) - y[row]) * inv_count;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane;
            if (c < nch) {
                float ev[8], ov[8];
                unpack8(*reinterpret_cast<const uint4*>(e + (long)row * lde + c * 8), ev);
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc[it][j] += g * ev[j]; ov[j] = g * wv[it][j]; }
                *reinterpret_cast<uint4*>(de + (long)row * ldde + c * 8) = pack8(ov);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wid][(it * 64 + lane) * 8 + j] = acc[it][j];
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = it * 64 + lane;
            if (c < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float s2 = red[0][c * 8 + j];
                    for (int w2 = 1; w2 < 4; ++w2) s2 += red[w2][c * 8 + j];
                    dw_part[(long)blockIdx.x * D + c * 8 + j] = s2;
                }
            }
        }
    }
}

// out[i] = bf16(sum_p parts[p][i]) with fp32 accumulation in index order (partial dK / dV of query chunks)
__global__ __launch_bounds__(256) void sum_parts_bf16_kernel(const bf16_t* __restrict__ parts, int P, long n, bf16_t* __restrict__ out) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (long)gridDim.x * 256 * 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < P; ++p) {
            float f[8];
            unpack8(*reinterpret_cast<const uint4*>(parts + (long)p * n + i), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += f[j];
        }
        *reinterpret_cast<uint4*>(out + i) = pack8(acc);
    }
}

// bf16 rows scattered into a zero-initialised [M][D] bf16 buffer (gradient of a row gather)
__global__ __launch_bounds__(256) void scatter_rows_bf16_kernel(const bf16_t* __restrict__ src, const int32_t* __restrict__ row_index, int R, int D,
                                                                bf16_t* __restrict__ dst) {
    const long total = (long)R * (D >> 3);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / (D >> 3)), c = (int)(i % (D >> 3)) * 8;
        *reinterpret_cast<uint4*>(dst + (long)row_index[r] * D + c) = *reinterpret_cast<const uint4*>(src + (long)r * D + c);
    }
}

}  // namespace

int k_transpose_bf16(hipStream_t s, const bf16_t* in, long rows, long cols, long ldi, bf16_t* out, long ldo, int zero_pad64) {
    if (rows <= 0 || cols <= 0) return MM_OK;
    if ((ldi % 8) || (ldo % 8)) return mm_set_error(MM_ERR_ALIGN, "transpose: strides must be multiples of 8 elements");
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64)), dim3(256), 0, s, in, rows, cols, ldi, out, ldo,
                       zero_pad64 ? (rows + 63) / 64 * 64 : rows);
    return mm_check_launch("transpose_bf16_kernel");
}

int k_colsum(hipStream_t s, const float* part, int nparts, long D, float* out) {
    if (D <= 0) return MM_OK;
    if (D >= 65536 && D % 4 == 0 && (((uintptr_t)part | (uintptr_t)out) & 15) == 0) {
        hipLaunchKernelGGL(colsum_wide_kernel, dim3((unsigned)((D / 4 + 255) / 256)), dim3(256), 0, s, part, nparts, D, out);
        return mm_check_launch("colsum_wide_kernel");
    }
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((D + 63) / 64)), dim3(512), 0, s, part, nparts, D, out);
    return mm_check_launch("colsum_kernel");
}

long k_ln_bwd_workspace_floats(int rows, int D) { return (long)((rows + 4 * LNB_ROWS - 1) / (4 * LNB_ROWS)) * D; }

// dgamma == nullptr: only the per-workgroup partials are left in ws ([k_ln_bwd_blocks(rows)][D]) -- the caller reduces them (k_colsum) where and when it likes;
// dxb: optional bf16 image [rows of x][D] of the updated dx rows
int k_ln_bwd_blocks(int rows) { return (rows + 4 * LNB_ROWS - 1) / (4 * LNB_ROWS); }

int k_layernorm_bwd(hipStream_t s, const float* x, long ldx, const bf16_t* dy, long lddy, const float* gamma, const int32_t* row_index,
                    int rows, int D, float* dx, long lddx, int accumulate, float* dgamma, float* ws, bf16_t* dxb) {
    if (rows <= 0) return MM_OK;
    if (D % 4 || D > 2048 || (ldx % 4) || (lddy % 4) || (lddx % 4)) return mm_set_error(MM_ERR_SHAPE, "layernorm_bwd: dim multiple of 4, <= 2048; strides multiples of 4");
    const int blocks = (rows + 4 * LNB_ROWS - 1) / (4 * LNB_ROWS);
    const int nit = (D / 4 + 63) / 64;
    if (nit <= 2) hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(blocks), dim3(256), 0, s, x, ldx, dy, lddy, gamma, row_index, rows, D, dx, lddx, accumulate, ws, dxb);
    else if (nit <= 4) hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(blocks), dim3(256), 0, s, x, ldx, dy, lddy, gamma, row_index, rows, D, dx, lddx, accumulate, ws, dxb);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<8>, dim3(blocks), dim3(256), 0, s, x, ldx, dy, lddy, gamma, row_index, rows, D, dx, lddx, accumulate, ws, dxb);
    int rc = mm_check_launch("layernorm_bwd_kernel");
    if (rc || !dgamma) return rc;
    hipLaunchKernelGGL(colsum_kernel, dim3((D + 63) / 64), dim3(512), 0, s, ws, blocks, (long)D, dgamma);
    return mm_check_launch("colsum_kernel");
}

int k_geglu_ln_bwd(hipStream_t s, const bf16_t* h, long ldh, const bf16_t* dz, long lddz, const float* gamma, int rows, int F, int Fp,
                   bf16_t* dh, long lddh, float* dgamma, float* ws) {
    if (rows <= 0) return MM_OK;
    if (Fp % 8 || Fp < F || Fp > 6144 || (ldh % 8) || (lddz % 8) || (lddh % 8)) return mm_set_error(MM_ERR_SHAPE, "geglu_ln_bwd: bad padded width / strides");
    const int blocks = (rows + 4 * LNB_ROWS - 1) / (4 * LNB_ROWS);
    const int nit = (Fp / 8 + 63) / 64;
    if (nit <= 3) hipLaunchKernelGGL(geglu_ln_bwd_kernel<3>, dim3(blocks), dim3(256), 0, s, h, ldh, dz, lddz, gamma, rows, F, Fp, dh, lddh, ws);
    else if (nit <= 6) hipLaunchKernelGGL(geglu_ln_bwd_kernel<6>, dim3(blocks), dim3(256), 0, s, h, ldh, dz, lddz, gamma, rows, F, Fp, dh, lddh, ws);
    else return mm_set_error(MM_ERR_SHAPE, "geglu_ln_bwd: padded inner width above 3072 is not built");
    int rc = mm_check_launch("geglu_ln_bwd_kernel");
    if (rc || !dgamma) return rc;      // (dgamma == nullptr: partials only, as k_layernorm_bwd)
    hipLaunchKernelGGL(colsum_kernel, dim3((Fp + 63) / 64), dim3(512), 0, s, ws, blocks, (long)Fp, dgamma);
    return mm_check_launch("colsum_kernel");
}

int k_ce_bwd(hipStream_t s, const float* logits, long ld, int R, int V, const int64_t* labels, float scale, bf16_t* dl, long ldd, float* row_loss) {
    if (R <= 0) return MM_OK;
    if (V % 4 || (ld % 4) || (ldd % 4)) return mm_set_error(MM_ERR_SHAPE, "ce_bwd: V and strides must be multiples of 4");
    if (V > 16384 && V <= 65536 && !(g_mm_debug2 & 4096)) {      // a long row: held in registers, read once (bit 4096: the two-sweep kernel, A/B)
        hipLaunchKernelGGL(ce_bwd_row_kernel, dim3(R), dim3(1024), 0, s, logits, ld, V, labels, scale, dl, ldd, row_loss);
        return mm_check_launch("ce_bwd_row_kernel");
    }
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(R), dim3(256), 0, s, logits, ld, V, labels, scale, dl, ldd, row_loss);
    return mm_check_launch("ce_bwd_kernel");
}

int k_bce_head_bwd(hipStream_t s, const bf16_t* e, long lde, const float* x, const float* y, const float* w, int rows, int D, bf16_t* de,
                   long ldde, float* dw, float* ws) {
    if (rows <= 0) return MM_OK;
    if (D % 8 || D > 2048 || (lde % 8) || (ldde % 8)) return mm_set_error(MM_ERR_SHAPE, "bce_head_bwd: D multiple of 8, <= 2048");
    const int blocks = (rows + 4 * LNB_ROWS - 1) / (4 * LNB_ROWS);
    const int nit = (D / 8 + 63) / 64;
    const float inv = 1.f / (float)rows;
    if (nit <= 1) hipLaunchKernelGGL(bce_head_bwd_kernel<1>, dim3(blocks), dim3(256), 0, s, e, lde, x, y, w, rows, D, inv, de, ldde, ws);
    else if (nit <= 2) hipLaunchKernelGGL(bce_head_bwd_kernel<2>, dim3(blocks), dim3(256), 0, s, e, lde, x, y, w, rows, D, inv, de, ldde, ws);
    else hipLaunchKernelGGL(bce_head_bwd_kernel<4>, dim3(blocks), dim3(256), 0, s, e, lde, x, y, w, rows, D, inv, de, ldde, ws);
    int rc = mm_check_launch("bce_head_bwd_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_kernel, dim3((D + 63) / 64), dim3(512), 0, s, ws, blocks, (long)D, dw);
    return mm_check_launch("colsum_kernel");
}

int k_sum_parts_bf16(hipStream_t s, const bf16_t* parts, int P, long n, bf16_t* out) {
    if (n <= 0) return MM_OK;
    if (n % 8) return mm_set_error(MM_ERR_SHAPE, "sum_parts_bf16: element count must be a multiple of 8");
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(sum_parts_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, parts, P, n, out);
    return mm_check_launch("sum_parts_bf16_kernel");
}

size_t k_embed_bwd_workspace_bytes(int B, int n, int D) { return ((size_t)B * n * D * 4 + 255) / 256 * 256 + (size_t)B * n + 256; }

int k_embed_bwd(hipStream_t s, const int64_t* ids, int B, int n, int D, const float* dx, float* dtoken, float* dpos, void* ws) {
    if (B <= 0) return MM_OK;
    if (D > 2048) return mm_set_error(MM_ERR_SHAPE, "embed_bwd: D <= 2048");
    hipLaunchKernelGGL(embed_pos_bwd_kernel, dim3(n), dim3(256), 0, s, B, n, D, dx, dpos);
    int rc = mm_check_launch("embed_pos_bwd_kernel");
    if (rc) return rc;
    if (ws) {      // two levels: per-block partial rows, then the blocks' partials (see embed_token_bwd2_kernel)
        float* part = reinterpret_cast<float*>(ws);
        uint8_t* first = reinterpret_cast<uint8_t*>(ws) + ((size_t)B * n * D * 4 + 255) / 256 * 256;
        const dim3 grid(B * n, (D + 255) / 256);
        hipLaunchKernelGGL(embed_token_bwd2_kernel<true>, grid, dim3(256), 0, s, ids, B * n, D, dx, part, first);
        hipLaunchKernelGGL(embed_token_bwd2_kernel<false>, grid, dim3(256), 0, s, ids, B * n, D, (const float*)part, dtoken, first);
        return mm_check_launch("embed_token_bwd2_kernel");
    }
    hipLaunchKernelGGL(embed_token_bwd_kernel<1>, dim3(B * n, (D + 255) / 256), dim3(256), 0, s, ids, B * n, D, dx, dtoken);
    return mm_check_launch("embed_token_bwd_kernel");
}

int k_scatter_rows_bf16(hipStream_t s, const bf16_t* src, const int32_t* row_index, int R, int D, bf16_t* dst) {
    if (R <= 0) return MM_OK;
    if (D % 8) return mm_set_error(MM_ERR_SHAPE, "scatter_rows: D must be a multiple of 8");
    long blocks = ((long)R * (D / 8) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scatter_rows_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, row_index, R, D, dst);
    return mm_check_launch("scatter_rows_bf16_kernel");
}
