// tools only: how fast can a CU be fed GEMM operands, and through which path?
// The GEMM family's k-loops stream their operands through the LDS-DMA (buffer_load ... lds) at ~16 B per clock and CU, which bounds them
// (DESIGN.md section 10).  This microbenchmark replays the operand traffic of the 256-token x 256-column tile (FF w1 of the base config:
// M = 16384, N = 2816, K = 512; 704 tiles, 16 k-steps of 32 each, one persistent 512-thread workgroup per CU) WITHOUT the MFMAs and measures:
//   mode 0  A and B through the LDS-DMA (what gemm_cfg2_kernel<WIDE_MIX2> does)                  32 KiB per step
//   mode 1  A only through the LDS-DMA                                                           16 KiB
//   mode 2  A through the LDS-DMA, B as fragment-packed global_load_dwordx4 straight to VGPRs,
//           every wave its own 64 columns (the two wave rows load the same 4 KiB)                16 KiB + 32 KiB issued (16 KiB unique)
//   mode 3  as 2, but every wave loads a distinct 2 KiB (no redundancy: the pure path rate)      16 KiB + 16 KiB
//   mode 4  B only, to VGPRs, as in 2
//   mode 5  A and B to VGPRs (A as 16-row x 64-byte fragment loads, 8 per wave)                  no LDS at all
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/build/feed_rate_bench tools/feed_rate_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int M = 16384, N = 2816, K = 512, BK = 32, KT = K / BK;
constexpr int TM = M / 256, TN = N / 256;
constexpr int NST = 3, STG = 32768;

template <int CNT> __device__ __forceinline__ void wait_vm() {
    // vmcnt is 6 bits: imm[3:0] | imm[15:14]; lgkmcnt / expcnt left at their maxima
    __builtin_amdgcn_s_waitcnt((CNT & 15) | ((CNT >> 4) << 14) | 0x0F70);
}

template <int MODE>
__global__ __launch_bounds__(512) void feed_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wrow, const uint16_t* __restrict__ Wpk,
                                                   unsigned long long* __restrict__ cycles, float* __restrict__ sink, int rounds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    constexpr bool DMA_A = MODE <= 3, DMA_B = MODE == 0;
    constexpr bool VG_B = MODE == 2 || MODE == 3 || MODE == 4 || MODE == 5, VG_A = MODE == 5;
    constexpr int NVB = MODE == 3 ? 2 : 4;                       // B loads per wave and step
    constexpr int NV = (VG_B ? NVB : 0) + (VG_A ? 8 : 0);        // VGPR loads per wave and step
    constexpr int ND = (DMA_A ? 2 : 0) + (DMA_B ? 2 : 0);        // LDS-DMA instructions per wave and step
    constexpr int PER = NV + ND;
    const int total = TM * TN;
    const int c = (lane & 3) ^ ((-(lane >> 4)) & 3);
    int voff_x[2], voff_w[2];
    for (int i = 0; i < 2; ++i) {
        const int xb = 2 * wid + i;
        voff_x[i] = (xb * 16 + (lane >> 2)) * K * 2 + c * 16;
        voff_w[i] = (xb * 16 + (lane >> 2)) * K * 2 + c * 16;
    }
    u32x4_t vb_[3][4], va_[3][8];
    float acc = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    long steps = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
            // tiles of one column walk the rows in groups of 8 workgroups (roughly what xcd_grouped_tile does: neighbours share operands in L2)
            const int tile_m = vb % TM, tile_n = vb / TM;
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(X + (size_t)tile_m * 256 * K), 0, 256u * K * 2u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Wrow + (size_t)tile_n * 256 * K), 0, 256u * K * 2u, 0x00020000);
            const uint16_t* wp = Wpk + ((size_t)(tile_n * 16 + (MODE == 3 ? wid * 2 : wn * 4)) * KT) * 512;      // [16-column block][k-step][lane] 16 B
            const uint16_t* xa = X + (size_t)(tile_m * 256 + wm * 128 + (lane & 15)) * K + (lane >> 4) * 8;
#define ISSUE(kt_, st_)                                                                                                              \
    {                                                                                                                                \
        unsigned char* xs_ = smem + (st_) * STG + wid * 2048;                                                                        \
        if (DMA_A) {                                                                                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(xs_), 16, voff_x[0], (kt_) * 64, 0, 0);                         \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(xs_ + 1024), 16, voff_x[1], (kt_) * 64, 0, 0);                  \
        }                                                                                                                            \
        if (DMA_B) {                                                                                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(xs_ + 16384), 16, voff_w[0], (kt_) * 64, 0, 0);                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(xs_ + 16384 + 1024), 16, voff_w[1], (kt_) * 64, 0, 0);          \
        }                                                                                                                            \
        if (VG_B) {                                                                                                                  \
            _Pragma("unroll") for (int i = 0; i < NVB; ++i)                                                                          \
                vb_[st_][i] = *(reinterpret_cast<const u32x4_t*>(wp + ((size_t)i * KT + (kt_)) * 512) + lane);     \
        }                                                                                                                            \
        if (VG_A) {                                                                                                                  \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                            \
                va_[st_][i] = *reinterpret_cast<const u32x4_t*>(xa + (size_t)i * 16 * K + (kt_) * 32);                               \
        }                                                                                                                            \
    }
#define CONSUME(st_)                                                                                                                 \
    {                                                                                                                                \
        if (VG_B) { _Pragma("unroll") for (int i = 0; i < NVB; ++i) asm volatile("" ::"v"(vb_[st_][i])); }                          \
        if (VG_A) { _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(va_[st_][i])); }                            \
        if (DMA_A) acc += *reinterpret_cast<const float*>(smem + (st_) * STG + t * 4);                                               \
    }
            ISSUE(0, 0);
            ISSUE(1, 1);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int st = kt % 3;
                if (kt + 2 < KT) {
                    if (st == 0) ISSUE(kt + 2, 2) else if (st == 1) ISSUE(kt + 2, 0) else ISSUE(kt + 2, 1);
                    wait_vm<2 * PER>();
                } else if (kt + 1 < KT) {
                    wait_vm<PER>();
                } else {
                    wait_vm<0>();
                }
                if (DMA_A) __builtin_amdgcn_s_barrier();
                if (st == 0) CONSUME(0) else if (st == 1) CONSUME(1) else CONSUME(2);
                if (DMA_A) __builtin_amdgcn_s_barrier();      // the stage is free for the DMA two steps on
                ++steps;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (t == 0) { cycles[blockIdx.x * 2] = t1 - t0; cycles[blockIdx.x * 2 + 1] = (unsigned long long)steps; }
    if (acc == 123.456f) sink[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE> void run(const uint16_t* X, const uint16_t* W, const uint16_t* Wp, unsigned long long* cyc, float* sink, const char* what, double bytes_per_step) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(feed_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, NST * STG));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rounds = 4;
    hipLaunchKernelGGL(feed_kernel<MODE>, dim3(256), dim3(512), NST * STG, 0, X, W, Wp, cyc, sink, 1);      // warm-up
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(feed_kernel<MODE>, dim3(256), dim3(512), NST * STG, 0, X, W, Wp, cyc, sink, rounds);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(512);
    CK(hipMemcpy(h.data(), cyc, 512 * 8, hipMemcpyDeviceToHost));
    double cs = 0, st = 0;
    for (int i = 0; i < 256; ++i) { cs += (double)h[2 * i]; st += (double)h[2 * i + 1]; }
    const double cyc_per_step = cs / st, us = ms * 1e3 / rounds;
    printf("mode %d  %-58s %8.1f us per pass  %7.0f cycles/step  %6.1f B/clk/CU  %6.2f TB/s chip  (clock %.2f GHz)\n", MODE, what, us, cyc_per_step,
           bytes_per_step / cyc_per_step, bytes_per_step * st / rounds / (us * 1e-6) / 1e12, cs / 256 / rounds / (us * 1e-6) / 1e9);
}

int main() {
    uint16_t *X, *W, *Wp;
    unsigned long long* cyc;
    float* sink;
    CK(hipMalloc(&X, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&Wp, (size_t)N * K * 2));
    CK(hipMalloc(&cyc, 512 * 8)); CK(hipMalloc(&sink, 16));
    CK(hipMemset(X, 0x11, (size_t)M * K * 2)); CK(hipMemset(W, 0x22, (size_t)N * K * 2)); CK(hipMemset(Wp, 0x33, (size_t)N * K * 2));
    run<0>(X, W, Wp, cyc, sink, "A + B via LDS-DMA (32 KiB/step)", 32768);
    run<1>(X, W, Wp, cyc, sink, "A via LDS-DMA only (16 KiB/step)", 16384);
    run<2>(X, W, Wp, cyc, sink, "A via LDS-DMA + B to VGPRs, wave rows redundant (16+32)", 49152);
    run<3>(X, W, Wp, cyc, sink, "A via LDS-DMA + B to VGPRs, distinct (16+16)", 32768);
    run<4>(X, W, Wp, cyc, sink, "B to VGPRs only, redundant (32 KiB issued)", 32768);
    run<5>(X, W, Wp, cyc, sink, "A (fragment loads) + B to VGPRs, no LDS (64+32 issued)", 98304);
    return 0;
}
