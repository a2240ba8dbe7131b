// Parameter preparation of a training step as ONE launch per group of parameters (round 6).
// mm_train_step needs, of every fp32 master weight W [rows][cols] of the step, (1) its bf16 operand copy, sometimes zero-padded to 64-multiples or placed inside a
// concatenated operand (to_q | to_kv, the two halves of the GEGLU projection), and (2) the transposed bf16 copy the dX = dY W GEMM reads as its NT B operand.  Rounds 4-5
// made them with one f32->bf16 launch per tensor plus memsets, strided copies and one transpose launch per operand: ~190 launches of 2-5 us each at the head of every
// step, enqueued before the forward's first kernel -- 1.3 ms in which the device waits for the host (profiles/r06_train_trace.txt).  Here a job table goes in as the
// kernel argument and every workgroup converts ONE 64 x 64 tile of one job: a single read of the fp32 tile, both bf16 images written from it (the transposed one
// through LDS).  Values: f32_to_bf16 (round to nearest even) of the same fp32 numbers, zeros in the padding -- bit-identical to the launches this replaces.
#include <string.h>

#include "common.h"
#include "muse_hip_internal.h"

namespace {

__global__ __launch_bounds__(256) void train_prep_kernel(PrepArgs a) {
    __shared__ bf16_t tile[64][72];
    // the job this workgroup belongs to: tile0 is ascending, the table sits in scalar registers / the kernel argument segment
    int j = 0;
    const int bid = blockIdx.x;
    while (j + 1 < a.njobs && bid >= a.job[j + 1].tile0) ++j;
    const PrepJob& q = a.job[j];
    const int t = threadIdx.x, lt = bid - q.tile0;
    if (q.kind == 1) {      // fp32 vector copied into a zero-padded fp32 vector: 4096 elements per workgroup
        float* dst = reinterpret_cast<float*>(q.dst);
        for (int i = t; i < 4096; i += 256) {
            const long idx = (long)lt * 4096 + i;
            if (idx < q.cols_p) dst[idx] = idx < q.cols ? q.src[idx] : 0.f;
        }
        return;
    }
    const int tiles_c = (q.cols_p + 63) >> 6;
    const long r0 = (long)(lt / tiles_c) * 64;
    const int c0 = (lt % tiles_c) * 64;
    const int r = t >> 2, cs = (t & 3) * 16;      // this thread: 16 consecutive columns of tile row r
    float v[16];
    const long row = r0 + r;
    const bool row_in = row < q.rows;
    const float* sp = q.src + row * (long)q.cols + c0 + cs;
    if (row_in && c0 + cs + 16 <= q.cols && (q.cols & 3) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 f = *reinterpret_cast<const float4*>(sp + i * 4);
            v[i * 4] = f.x; v[i * 4 + 1] = f.y; v[i * 4 + 2] = f.z; v[i * 4 + 3] = f.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (row_in && c0 + cs + i < q.cols) ? sp[i] : 0.f;
    }
    uint4 lo, hi;
    lo.x = pack_bf16x2(v[0], v[1]); lo.y = pack_bf16x2(v[2], v[3]); lo.z = pack_bf16x2(v[4], v[5]); lo.w = pack_bf16x2(v[6], v[7]);
    hi.x = pack_bf16x2(v[8], v[9]); hi.y = pack_bf16x2(v[10], v[11]); hi.z = pack_bf16x2(v[12], v[13]); hi.w = pack_bf16x2(v[14], v[15]);
    if (q.dst && row < q.rows_p) {
        bf16_t* dp = reinterpret_cast<bf16_t*>(q.dst) + row * (long)q.ld_d + c0 + cs;
        if (c0 + cs + 16 <= q.cols_p) {
            *reinterpret_cast<uint4*>(dp) = lo;
            *reinterpret_cast<uint4*>(dp + 8) = hi;
        } else {      // (cols_p is a multiple of 8 whenever the copy is written: checked on the host)
            if (c0 + cs + 8 <= q.cols_p) *reinterpret_cast<uint4*>(dp) = lo;
        }
    }
    if (!q.dst_t) return;
    *reinterpret_cast<uint4*>(&tile[r][cs]) = lo;
    *reinterpret_cast<uint4*>(&tile[r][cs + 8]) = hi;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = t + i * 256;
        const int c = idx >> 3, rr = (idx & 7) * 8;       // transposed row c0 + c, its columns r0 + rr .. + 7
        if (c0 + c >= q.cols_p || r0 + rr + 8 > q.rows_p) continue;
        bf16_t tmp[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) tmp[k] = tile[rr + k][c];
        *reinterpret_cast<uint4*>(q.dst_t + (long)(c0 + c) * q.ld_t + r0 + rr) = *reinterpret_cast<const uint4*>(tmp);
    }
}

}  // namespace

// ---- host side: a table being filled, flushed as a launch when it is full or when the caller asks
void PrepList::add(const float* src, void* dst, bf16_t* dst_t, int rows, int cols, int rows_p, int cols_p, int ld_d, int ld_t, int kind) {
    if (rc) return;
    if (a.njobs == PREP_MAX_JOBS) flush();
    if (kind == 0 && ((dst && ((ld_d % 8) || (cols_p % 8))) || (dst_t && ((ld_t % 8) || (rows_p % 8))))) {
        rc = mm_set_error(MM_ERR_ALIGN, "train_prep: padded sizes and strides must be multiples of 8");
        return;
    }
    PrepJob& q = a.job[a.njobs++];
    q.src = src; q.dst = dst; q.dst_t = dst_t; q.rows = rows; q.cols = cols; q.rows_p = rows_p; q.cols_p = cols_p; q.ld_d = ld_d; q.ld_t = ld_t; q.kind = kind;
    q.tile0 = tiles;
    tiles += kind == 1 ? (cols_p + 4095) / 4096 : ((rows_p + 63) / 64) * ((cols_p + 63) / 64);
}

int PrepList::flush() {
    if (rc) return rc;
    if (a.njobs == 0) return MM_OK;
    hipLaunchKernelGGL(train_prep_kernel, dim3((unsigned)tiles), dim3(256), 0, s, a);
    rc = mm_check_launch("train_prep_kernel");
    a.njobs = 0;
    tiles = 0;
    return rc;
}
