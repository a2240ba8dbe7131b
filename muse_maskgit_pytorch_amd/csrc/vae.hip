// HBM-bound kernels of the VQGanVAE path (gfx950), activations NHWC bf16:
//   LFQ indices -> codes -> project_out   (vqgan_vae.py:427-437 + third-party LFQ.indices_to_codes)
//   LFQ encode: project_in -> sign -> bit-pack -> project_out   (vqgan_vae.py:424 + third-party LFQ.forward, eval)
//   GLU(dim=channel), GroupNorm(16) (+ LeakyReLU 0.1)            (vqgan_vae.py:254-261, 270-276)
//   NCHW fp32 <-> NHWC bf16 layout changes at the module boundary
// The convolutions themselves run on the MFMA implicit-GEMM path in gemm.hip.
#include "common.h"
#include "muse_hip_internal.h"

namespace {

inline int grid_for(long items, int per_block = 256, int cap = 256 * 16) {
    long b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

// out[p][c] = bias[c] + sum_j (bit_j(id_p) ? +1 : -1) * w[c][j],  bit_j = (id >> (bits-1-j)) & 1  (MSB = code dim 0).
// A thread owns 8 consecutive channels: their bits x 8 projection weights stay in registers while it walks a strip of pixels
// (one 16-byte store per pixel); the sum runs j = 0..bits-1 from the bias like the reference's Linear.
template <int BITS, bool H>
__global__ __launch_bounds__(256) void lfq_decode_proj_kernel(const int64_t* __restrict__ ids, long count, int C, int strip,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              bf16_t* __restrict__ out) {
    const int nch = C >> 3;
    const int cg = blockIdx.x * blockDim.x + threadIdx.x;      // channel group
    if (cg >= nch) return;
    const int c0 = cg * 8;
    float wr[8][BITS], br[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        br[k] = bias[c0 + k];
#pragma unroll
        for (int j = 0; j < BITS; ++j) wr[k][j] = w[(c0 + k) * BITS + j];
    }
    const long p0 = (long)blockIdx.y * strip;
    const long p1 = p0 + strip < count ? p0 + strip : count;
    for (long pix = p0; pix < p1; ++pix) {
        const int64_t id = ids[pix];
        float sg[BITS];
#pragma unroll
        for (int j = 0; j < BITS; ++j) sg[j] = ((id >> (BITS - 1 - j)) & 1) ? 1.f : -1.f;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float a = br[k];
#pragma unroll
            for (int j = 0; j < BITS; ++j) a += sg[j] * wr[k][j];      // exact product: same bits as a +/- w
            acc[k] = a;
        }
        *reinterpret_cast<uint4*>(out + pix * C + c0) = pack8s<H>(acc);
    }
}

// general / projection-free form: one thread produces 8 consecutive channels of one pixel
template <bool H>
__global__ __launch_bounds__(256) void lfq_decode_kernel(const int64_t* __restrict__ ids, long count, int bits, int C,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         bf16_t* __restrict__ out) {
    const int nch = C >> 3;
    const long total = count * nch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / nch;
        const int c0 = (int)(i - pix * nch) * 8;
        const int64_t id = ids[pix];
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + k;
            if (w) {
                float a = bias[c];
                for (int j = 0; j < bits; ++j) a += ((id >> (bits - 1 - j)) & 1) ? w[c * bits + j] : -w[c * bits + j];
                acc[k] = a;
            } else {
                acc[k] = ((id >> (bits - 1 - c)) & 1) ? 1.f : -1.f;
            }
        }
        *reinterpret_cast<uint4*>(out + pix * C + c0) = pack8s<H>(acc);
    }
}

// one workgroup per pixel
__global__ __launch_bounds__(256) void lfq_encode_kernel(const bf16_t* __restrict__ x, int C, int bits,
                                                         const float* __restrict__ w, const float* __restrict__ b,
                                                         const float* __restrict__ wo, const float* __restrict__ bo,
                                                         int64_t* __restrict__ ids, bf16_t* __restrict__ out) {
    __shared__ float part[4][32];
    __shared__ float qs[32];
    const long pix = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bf16_t* xr = x + pix * C;
    for (int j = 0; j < bits; ++j) {
        float acc = 0.f;
        if (w) for (int c = tid; c < C; c += 256) acc += bf16_to_f32(xr[c]) * w[j * C + c];
        acc = wave_sum(acc);
        if (lane == 0) part[wid][j] = acc;
    }
    __syncthreads();
    if (tid < bits) {
        float tv;
        if (w) tv = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid] + b[tid];
        else tv = bf16_to_f32(xr[tid]);
        qs[tid] = tv > 0.f ? 1.f : -1.f;
    }
    __syncthreads();
    if (tid == 0) {
        int64_t id = 0;
        for (int j = 0; j < bits; ++j) id |= (int64_t)(qs[j] > 0.f) << (bits - 1 - j);
        ids[pix] = id;
    }
    if (out) {
        for (int c = tid; c < C; c += 256) {
            float acc;
            if (wo) {
                acc = bo[c];
                for (int j = 0; j < bits; ++j) acc += qs[j] * wo[c * bits + j];
            } else acc = qs[c];
            out[pix * C + c] = f32_to_bf16(acc);
        }
    }
}

// F.glu(x, dim=channel): x[:, :C] * sigmoid(x[:, C:])
template <bool H>
__global__ __launch_bounds__(256) void glu_kernel(const bf16_t* __restrict__ x, long rows, int C, bf16_t* __restrict__ out) {
    const int nch = C >> 3;
    const long total = rows * nch;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / nch;
        const int c = (int)(i - row * nch);
        float a[8], g[8], o[8];
        unpack8s<H>(*reinterpret_cast<const uint4*>(x + row * 2 * C + c * 8), a);
        unpack8s<H>(*reinterpret_cast<const uint4*>(x + row * 2 * C + C + c * 8), g);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = a[j] / (1.f + expf(-g[j]));
        *reinterpret_cast<uint4*>(out + row * C + c * 8) = pack8s<H>(o);
    }
}

// GroupNorm statistics: one workgroup per (group, batch); two passes (mean, then centred variance).  Round 6: 16-byte loads (8 channels of one pixel per lane;
// channels-per-group % 8 == 0 -- every VQGanVAE width -- else the element-wise walk): 86 -> ~20 us per launch at the decoder's 256 x 2048 maps.
template <bool H>
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const bf16_t* __restrict__ x, int HW, int C, int groups,
                                                              float* __restrict__ stats) {
    __shared__ float red[4];
    __shared__ float bcast;
    const int g = blockIdx.x, b = blockIdx.y;
    const int cpg = C / groups;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bf16_t* xb = x + (size_t)b * HW * C + g * cpg;
    const long cnt = (long)HW * cpg;
    const bool vec = (cpg % 8) == 0 && (C % 8) == 0;
    const int c8n = cpg >> 3;
    const long cnt8 = (long)HW * c8n;
    float s = 0.f;
    if (vec) {
        for (long i = tid; i < cnt8; i += 256) {
            const long r = i / c8n;
            const int c = (int)(i - r * c8n) * 8;
            float v[8];
            unpack8s<H>(*reinterpret_cast<const uint4*>(xb + r * C + c), v);
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
    } else {
        for (long i = tid; i < cnt; i += 256) {
            const long r = i / cpg;
            const int c = (int)(i - r * cpg);
            s += ld16s<H>(xb[r * C + c]);
        }
    }
    s = wave_sum(s);
    if (lane == 0) red[wid] = s;
    __syncthreads();
    if (tid == 0) bcast = (red[0] + red[1] + red[2] + red[3]) / (float)cnt;
    __syncthreads();
    const float mean = bcast;
    float q = 0.f;
    if (vec) {
        for (long i = tid; i < cnt8; i += 256) {
            const long r = i / c8n;
            const int c = (int)(i - r * c8n) * 8;
            float v[8];
            unpack8s<H>(*reinterpret_cast<const uint4*>(xb + r * C + c), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[j] - mean; q += d * d; }
        }
    } else {
        for (long i = tid; i < cnt; i += 256) {
            const long r = i / cpg;
            const int c = (int)(i - r * cpg);
            const float d = ld16s<H>(xb[r * C + c]) - mean;
            q += d * d;
        }
    }
    q = wave_sum(q);
    __syncthreads();
    if (lane == 0) red[wid] = q;
    __syncthreads();
    if (tid == 0) {
        const float var = (red[0] + red[1] + red[2] + red[3]) / (float)cnt;
        stats[((size_t)b * groups + g) * 2 + 0] = mean;
        stats[((size_t)b * groups + g) * 2 + 1] = 1.f / sqrtf(var + 1e-5f);
    }
}

template <bool H>
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const bf16_t* __restrict__ x, int HW, int C, int groups,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ stats, int act, long total_chunks,
                                                              bf16_t* __restrict__ out) {
    const int nch = C >> 3;
    const int cpg = C / groups;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_chunks; i += (long)gridDim.x * blockDim.x) {
        const long row = i / nch;
        const int c8 = (int)(i - row * nch) * 8;
        const int b = (int)(row / HW);
        float a[8], o[8];
        unpack8s<H>(*reinterpret_cast<const uint4*>(x + row * C + c8), a);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c8 + j;
            const int g = c / cpg;
            const float mean = stats[((size_t)b * groups + g) * 2], rstd = stats[((size_t)b * groups + g) * 2 + 1];
            float y = (a[j] - mean) * rstd * gamma[c] + beta[c];
            if (act == ACT_LEAKY) y = y > 0.f ? y : 0.1f * y;
            o[j] = y;
        }
        *reinterpret_cast<uint4*>(out + row * C + c8) = pack8s<H>(o);
    }
}

__global__ __launch_bounds__(256) void nchw_to_nhwc8_kernel(const float* __restrict__ img, int B, int C, int H, int W,
                                                            bf16_t* __restrict__ out) {
    const long total = (long)B * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / ((long)H * W);
        const long hw = i - b * H * W;
        float o[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = c < C ? img[(b * C + c) * H * W + hw] : 0.f;
        *reinterpret_cast<uint4*>(out + i * 8) = pack8(o);
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_f32_kernel(const bf16_t* __restrict__ x, int B, int C, int HW,
                                                               float* __restrict__ out) {
    const long total = (long)B * C * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / ((long)C * HW);
        const long r = i - b * C * HW;
        const int c = (int)(r / HW);
        const long hw = r - (long)c * HW;
        out[i] = bf16_to_f32(x[(b * HW + hw) * C + c]);
    }
}

}  // namespace

int k_lfq_decode(hipStream_t s, const int64_t* ids, long count, int bits, int C, const float* w, const float* b, bf16_t* out, int half) {
    if (count <= 0) return MM_OK;
    if (!w && C != bits) return mm_set_error(MM_ERR_SHAPE, "lfq_decode: no projection requires C == bits");
    if (C % 8) return mm_set_error(MM_ERR_SHAPE, "lfq_decode: C must be a multiple of 8");
    if (w && (bits == 16 || bits == 13) && C >= 512) {
        // wide projection (the 65536- / 8192-entry codebooks): register-resident weights, strips of pixels
        const int nch = C / 8;
        const int bx = (nch + 255) / 256;
        long strips = 2048 / bx;
        if (strips > count) strips = count;
        const int strip = (int)((count + strips - 1) / strips);
        const dim3 grid(bx, (unsigned)((count + strip - 1) / strip));
        if (bits == 16 && half) hipLaunchKernelGGL((lfq_decode_proj_kernel<16, true>), grid, dim3(256), 0, s, ids, count, C, strip, w, b, out);
        else if (bits == 16) hipLaunchKernelGGL((lfq_decode_proj_kernel<16, false>), grid, dim3(256), 0, s, ids, count, C, strip, w, b, out);
        else if (half) hipLaunchKernelGGL((lfq_decode_proj_kernel<13, true>), grid, dim3(256), 0, s, ids, count, C, strip, w, b, out);
        else hipLaunchKernelGGL((lfq_decode_proj_kernel<13, false>), grid, dim3(256), 0, s, ids, count, C, strip, w, b, out);
        return mm_check_launch("lfq_decode_proj_kernel");
    }
    if (half) hipLaunchKernelGGL(lfq_decode_kernel<true>, dim3(grid_for(count * (C / 8))), dim3(256), 0, s, ids, count, bits, C, w, b, out);
    else hipLaunchKernelGGL(lfq_decode_kernel<false>, dim3(grid_for(count * (C / 8))), dim3(256), 0, s, ids, count, bits, C, w, b, out);
    return mm_check_launch("lfq_decode_kernel");
}

int k_lfq_encode(hipStream_t s, const bf16_t* x, long count, int C, int bits, const float* w, const float* b,
                 const float* wo, const float* bo, int64_t* ids, bf16_t* out) {
    if (count <= 0) return MM_OK;
    if (bits > 32) return mm_set_error(MM_ERR_SHAPE, "lfq_encode: at most 32 code bits");
    hipLaunchKernelGGL(lfq_encode_kernel, dim3((unsigned)count), dim3(256), 0, s, x, C, bits, w, b, wo, bo, ids, out);
    return mm_check_launch("lfq_encode_kernel");
}

int k_glu(hipStream_t s, const bf16_t* x, long rows, int C, bf16_t* out, int half) {
    if (rows <= 0) return MM_OK;
    if (C % 8) return mm_set_error(MM_ERR_SHAPE, "glu: C must be a multiple of 8");
    if (half) hipLaunchKernelGGL(glu_kernel<true>, dim3(grid_for(rows * (C / 8))), dim3(256), 0, s, x, rows, C, out);
    else hipLaunchKernelGGL(glu_kernel<false>, dim3(grid_for(rows * (C / 8))), dim3(256), 0, s, x, rows, C, out);
    return mm_check_launch("glu_kernel");
}

int k_groupnorm(hipStream_t s, const bf16_t* x, int B, int HW, int C, int groups, const float* gamma, const float* beta,
                int act, float* stats_ws, bf16_t* out, int half) {
    if (B <= 0) return MM_OK;
    if (C % 8 || C % groups) return mm_set_error(MM_ERR_SHAPE, "groupnorm: C must be a multiple of 8 and of groups");
    if (half) hipLaunchKernelGGL(groupnorm_stats_kernel<true>, dim3(groups, B), dim3(256), 0, s, x, HW, C, groups, stats_ws);
    else hipLaunchKernelGGL(groupnorm_stats_kernel<false>, dim3(groups, B), dim3(256), 0, s, x, HW, C, groups, stats_ws);
    int rc = mm_check_launch("groupnorm_stats_kernel");
    if (rc) return rc;
    const long chunks = (long)B * HW * (C / 8);
    if (half) hipLaunchKernelGGL(groupnorm_apply_kernel<true>, dim3(grid_for(chunks)), dim3(256), 0, s, x, HW, C, groups, gamma, beta, stats_ws, act, chunks, out);
    else hipLaunchKernelGGL(groupnorm_apply_kernel<false>, dim3(grid_for(chunks)), dim3(256), 0, s, x, HW, C, groups, gamma, beta, stats_ws, act, chunks, out);
    return mm_check_launch("groupnorm_apply_kernel");
}

int k_nchw_to_nhwc8(hipStream_t s, const float* img, int B, int C, int H, int W, bf16_t* out) {
    if (B <= 0) return MM_OK;
    if (C > 8) return mm_set_error(MM_ERR_SHAPE, "nchw_to_nhwc8: at most 8 channels");
    hipLaunchKernelGGL(nchw_to_nhwc8_kernel, dim3(grid_for((long)B * H * W)), dim3(256), 0, s, img, B, C, H, W, out);
    return mm_check_launch("nchw_to_nhwc8_kernel");
}

int k_nhwc_to_nchw_f32(hipStream_t s, const bf16_t* x, int B, int C, int H, int W, float* out) {
    if (B <= 0) return MM_OK;
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(grid_for((long)B * C * H * W)), dim3(256), 0, s, x, B, C, H * W, out);
    return mm_check_launch("nhwc_to_nchw_f32_kernel");
}
