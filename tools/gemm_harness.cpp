// tools only: times the GEMM classes of the decode loop through the C ABI alone (no Python, no torch: a gpurun call of this binary costs
// seconds of GPU budget instead of minutes) and prints a checksum of every output, so kernel variants (mm_debug_set bits, rebuilt libraries)
// can be compared bit for bit.  Shapes: BASELINE configs[1] at B = 32, both guidance halves (16384 rows).
//   build:  hipcc --offload-arch=gfx950 -O2 -Iinclude -o tools/build/gemm_harness tools/gemm_harness.cpp -Lmuse_maskgit_pytorch_amd -lmuse_hip \
//                 -Wl,-rpath,'$ORIGIN/../../muse_maskgit_pytorch_amd'
//   run:    tools/build/gemm_harness [debug_bits] [case ...]      cases: qkv w1 out w2 xq logits sample fp8check fp8 (default: all)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "muse_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define MK(x) do { int r_ = (x); if (r_) { printf("mm error %d (%s) at line %d\n", r_, mm_last_error(), __LINE__); exit(1); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }
static inline float gauss() {      // sum of 4 uniforms, variance 1
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += (float)(rnd() & 0xFFFFFF) / 16777216.f - 0.5f;
    return s * 1.7320508f;
}
static inline uint16_t bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

static void* dev_bf16(size_t n, float scale) {
    std::vector<uint16_t> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = bf16(gauss() * scale);
    void* d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}
static void* dev_f32(size_t n, float scale) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = gauss() * scale;
    void* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}
static uint64_t checksum(const void* d, size_t bytes) {
    std::vector<uint64_t> h(bytes / 8);
    CK(hipMemcpy(h.data(), d, bytes / 8 * 8, hipMemcpyDeviceToHost));
    uint64_t a = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < h.size(); ++i) a = (a ^ h[i]) * 0x100000001b3ull + (a >> 29);
    return a;
}

template <typename F> static double time_us(F&& f, int iters = 20) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / iters;
}

static void report(const char* name, double us, double flops, uint64_t sum) {
    printf("%-8s %9.1f us  %8.1f TFLOP/s  (%.3f of 2.5 PF)  checksum %016llx\n", name, us, flops / us * 1e-6, flops / us * 1e-6 / 2500., (unsigned long long)sum);
}

int main(int argc, char** argv) {
    unsigned debug = 0;
    std::vector<std::string> cases;
    for (int i = 1; i < argc; ++i) {
        if (argv[i][0] >= '0' && argv[i][0] <= '9') debug = (unsigned)strtoul(argv[i], nullptr, 0);
        else cases.push_back(argv[i]);
    }
    auto want = [&](const char* c) { if (cases.empty()) return true; for (auto& s : cases) if (s == c) return true; return false; };
    MK(mm_device_check());
    mm_debug_set(debug);
    if (getenv("MM_DEBUG2")) mm_debug_set2((int)strtoul(getenv("MM_DEBUG2"), nullptr, 0));      // (tools: the second debug word)
    const int M = 16384, D = 512, I = 512, F = 1365, Fp = 1408, V = 65536, R = 5140;
    if (want("qkv")) {
        void* x = dev_bf16((size_t)M * D, 1.f); void* w = dev_bf16((size_t)3 * I * D, 0.04f);
        void* o; CK(hipMalloc(&o, (size_t)M * 3 * I * 2));
        const double us = time_us([&] { MK(mm_gemm_bf16(nullptr, x, D, w, D, M, 3 * I, D, o, 3 * I, 0, nullptr)); });
        report("qkv", us, 2.0 * M * 3 * I * D, checksum(o, (size_t)M * 3 * I * 2));
    }
    if (want("w1")) {
        void* x = dev_bf16((size_t)M * D, 1.f); void* w = dev_bf16((size_t)2 * Fp * D, 0.04f);
        void* o; CK(hipMalloc(&o, (size_t)M * Fp * 2));
        const double us = time_us([&] { MK(mm_gemm_geglu(nullptr, x, D, w, D, M, Fp, D, o, Fp)); });
        report("w1", us, 2.0 * M * 2 * Fp * D, checksum(o, (size_t)M * Fp * 2));
    }
    if (want("out")) {      // self-attention output projection + fp32 residual
        void* x = dev_bf16((size_t)M * I, 1.f); void* w = dev_bf16((size_t)D * I, 0.04f);
        float* r = (float*)dev_f32((size_t)M * D, 1.f);
        float* o; CK(hipMalloc(&o, (size_t)M * D * 4));
        const double us = time_us([&] { MK(mm_gemm_bf16(nullptr, x, I, w, I, M, D, I, o, D, 1, r)); });
        report("out", us, 2.0 * M * D * I, checksum(o, (size_t)M * D * 4));
    }
    if (want("w2")) {       // FF w2 (K = padded inner) + fp32 residual, LayerNorm(inner) as its own pass (the folded form needs the model entry points)
        void* x = dev_bf16((size_t)M * Fp, 1.f); void* w = dev_bf16((size_t)D * Fp, 0.03f);
        float* r = (float*)dev_f32((size_t)M * D, 1.f);
        float* o; CK(hipMalloc(&o, (size_t)M * D * 4));
        const double us = time_us([&] { MK(mm_gemm_bf16(nullptr, x, Fp, w, Fp, M, D, Fp, o, D, 1, r)); });
        report("w2", us, 2.0 * M * D * Fp, checksum(o, (size_t)M * D * 4));
    }
    if (want("xq")) {       // cross-attention q / output projection: the cond half only
        const int Mc = M / 2;
        void* x = dev_bf16((size_t)Mc * D, 1.f); void* w = dev_bf16((size_t)I * D, 0.04f);
        void* o; CK(hipMalloc(&o, (size_t)Mc * I * 2));
        const double us = time_us([&] { MK(mm_gemm_bf16(nullptr, x, D, w, D, Mc, I, D, o, I, 0, nullptr)); });
        report("xq", us, 2.0 * Mc * I * D, checksum(o, (size_t)Mc * I * 2));
        float* r = (float*)dev_f32((size_t)Mc * D, 1.f);
        float* o2; CK(hipMalloc(&o2, (size_t)Mc * D * 4));
        const double us2 = time_us([&] { MK(mm_gemm_bf16(nullptr, x, I, w, I, Mc, D, I, o2, D, 1, r)); });
        report("xout", us2, 2.0 * Mc * D * I, checksum(o2, (size_t)Mc * D * 4));
    }
    for (int pass = 0; pass < 2 && want("fp8check"); ++pass) {
        mm_debug_set(pass ? (1u << 30) : 0u);      // second pass: the 256-column tile forced on the small shape      // the fp8 GEMM against a host reference (validates the operand layout of v_mfma_f32_16x16x128_f8f6f4 as used), all three epilogues
        auto dec = [](uint8_t c) { const int s_ = c >> 7, e = (c >> 3) & 15, m = c & 7; const float v = e ? ldexpf(1.f + m / 8.f, e - 7) : ldexpf(m / 8.f, -6); return s_ ? -v : v; };
        const int M2 = 300, N2 = 512, K2 = 384;
        std::vector<uint8_t> hx((size_t)M2 * K2), hw((size_t)N2 * K2);
        for (auto& c : hx) { c = (uint8_t)(rnd() & 0xFF); if (((c >> 3) & 15) > 9) c &= 0xB7; }
        for (auto& c : hw) { c = (uint8_t)(rnd() & 0xFF); if (((c >> 3) & 15) > 9) c &= 0xB7; }
        std::vector<float> hsx(M2), hsw(N2), hr((size_t)M2 * N2);
        for (auto& v : hsx) v = 0.5f + (rnd() & 255) / 256.f;
        for (auto& v : hsw) v = 0.01f + (rnd() & 255) / 4096.f;
        for (auto& v : hr) v = gauss();
        std::vector<double> ref((size_t)M2 * N2);
        for (int m = 0; m < M2; ++m)
            for (int n = 0; n < N2; ++n) {
                double a = 0;
                for (int k = 0; k < K2; ++k) a += (double)dec(hx[(size_t)m * K2 + k]) * dec(hw[(size_t)n * K2 + k]);
                ref[(size_t)m * N2 + n] = a * hsx[m] * hsw[n];
            }
        void *dx, *dw; float *dsx, *dsw, *dr;
        CK(hipMalloc(&dx, hx.size())); CK(hipMalloc(&dw, hw.size())); CK(hipMalloc(&dsx, M2 * 4)); CK(hipMalloc(&dsw, N2 * 4)); CK(hipMalloc(&dr, hr.size() * 4));
        CK(hipMemcpy(dx, hx.data(), hx.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dsx, hsx.data(), M2 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsw, hsw.data(), N2 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
        auto bf2f = [](uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; };
        {   // epilogue 2: fp32 + residual
            float* o; CK(hipMalloc(&o, (size_t)M2 * N2 * 4));
            MK(mm_gemm_fp8(nullptr, dx, K2, dsx, dw, K2, dsw, M2, N2, K2, o, N2, 2, dr));
            std::vector<float> h((size_t)M2 * N2); CK(hipMemcpy(h.data(), o, h.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0, scale = 0;
            for (size_t i = 0; i < h.size(); ++i) { worst = fmax(worst, fabs(h[i] - (ref[i] + hr[i]))); scale = fmax(scale, fabs(ref[i])); }
            printf("fp8check fp32+resid: max abs err %.3g on scale %.3g  %s\n", worst, scale, worst <= 2e-4 * scale ? "OK" : "FAIL");
        }
        {   // epilogue 0: bf16
            void* o; CK(hipMalloc(&o, (size_t)M2 * N2 * 2));
            MK(mm_gemm_fp8(nullptr, dx, K2, dsx, dw, K2, dsw, M2, N2, K2, o, N2, 0, nullptr));
            std::vector<uint16_t> h((size_t)M2 * N2); CK(hipMemcpy(h.data(), o, h.size() * 2, hipMemcpyDeviceToHost));
            double worst = 0, scale = 0;
            for (size_t i = 0; i < h.size(); ++i) { worst = fmax(worst, fabs(bf2f(h[i]) - ref[i])); scale = fmax(scale, fabs(ref[i])); }
            printf("fp8check bf16:       max abs err %.3g on scale %.3g  %s\n", worst, scale, worst <= 4.5e-3 * scale ? "OK" : "FAIL");
        }
        {   // epilogue 1: GEGLU, rows interleaved in 64-row blocks (32 values | 32 gates)
            void* o; CK(hipMalloc(&o, (size_t)M2 * (N2 / 2) * 2));
            MK(mm_gemm_fp8(nullptr, dx, K2, dsx, dw, K2, dsw, M2, N2, K2, o, N2 / 2, 1, nullptr));
            std::vector<uint16_t> h((size_t)M2 * (N2 / 2)); CK(hipMemcpy(h.data(), o, h.size() * 2, hipMemcpyDeviceToHost));
            double worst = 0, scale = 0;
            for (int m = 0; m < M2; ++m)
                for (int c = 0; c < N2 / 2; ++c) {
                    const int blk = c / 32, j = c % 32;
                    const double x = ref[(size_t)m * N2 + blk * 64 + j], g = ref[(size_t)m * N2 + blk * 64 + 32 + j];
                    const double want_ = g * 0.5 * x * (1.0 + erf(x / sqrt(2.0)));
                    worst = fmax(worst, fabs(bf2f(h[(size_t)m * (N2 / 2) + c]) - want_)); scale = fmax(scale, fabs(want_));
                }
            printf("fp8check geglu:      max abs err %.3g on scale %.3g  %s\n", worst, scale, worst <= 4.5e-3 * scale ? "OK" : "FAIL");
        }
    }
    mm_debug_set(debug);
    if (want("fp8")) {
        struct Shape { const char* name; int M, N, K, epi; };
        const Shape shapes[] = {{"c5 qkv", 16384, 3072, 1024, 0}, {"c5 out", 16384, 1024, 1024, 2}, {"c5 w1", 16384, 5632, 1024, 1}, {"c5 w2", 16384, 1024, 2816, 2}, {"long k", 16384, 1024, 8192, 0}, {"nf2 lk", 16384, 512, 8192, 0}, {"nf4 lk2", 32768, 1024, 8192, 0}, {"nf2 lk2", 32768, 512, 8192, 0},
                                {"c2 qkv", 16384, 1536, 512, 0}, {"c2 out", 16384, 512, 512, 2}, {"c2 w1", 16384, 2816, 512, 1}, {"c2 w2", 16384, 512, 1408, 2}};
        for (const Shape& sh : shapes) {
            std::vector<uint8_t> hx((size_t)sh.M * sh.K), hw((size_t)sh.N * sh.K);
            for (auto& c : hx) { c = (uint8_t)(rnd() & 0xFF); if (((c >> 3) & 15) > 9) c &= 0xB7; }
            for (auto& c : hw) { c = (uint8_t)(rnd() & 0xFF); if (((c >> 3) & 15) > 9) c &= 0xB7; }
            void *dx, *dw, *o; float *dsx, *dsw, *dr = nullptr;
            CK(hipMalloc(&dx, hx.size())); CK(hipMalloc(&dw, hw.size()));
            CK(hipMemcpy(dx, hx.data(), hx.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size(), hipMemcpyHostToDevice));
            dsx = (float*)dev_f32(sh.M, 0.f); dsw = (float*)dev_f32(sh.N, 0.f);
            const int oc = sh.epi == 1 ? sh.N / 2 : sh.N;
            CK(hipMalloc(&o, (size_t)sh.M * oc * (sh.epi == 2 ? 4 : 2)));
            if (sh.epi == 2) dr = (float*)dev_f32((size_t)sh.M * oc, 1.f);
            const double us = time_us([&] { MK(mm_gemm_fp8(nullptr, dx, sh.K, dsx, dw, sh.K, dsw, sh.M, sh.N, sh.K, o, oc, sh.epi, dr)); });
            printf("fp8 %-7s %9.1f us  %8.1f TFLOP/s  (%.3f of 5 PF)\n", sh.name, us, 2.0 * sh.M * sh.N * sh.K / us * 1e-6, 2.0 * sh.M * sh.N * sh.K / us * 1e-6 / 5000.);
            CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(o)); CK(hipFree(dsx)); CK(hipFree(dsw)); if (dr) CK(hipFree(dr));
        }
    }
    if (want("logits") || want("sample") || want("stamps") || want("pairs")) {
        void* x = dev_bf16((size_t)R * D, 1.f); void* w = dev_bf16((size_t)V * D, 0.02f);
        // logits ~ N(0, sigma^2), sigma = sqrt(D) * 0.02; thr_lo at the 12 % quantile + margin as in the decode loop
        const float sigma = sqrtf((float)D) * 0.02f;
        const char* thr_env = getenv("MM_THR");      // bound in sigmas (default: the decode loop's 12 % quantile); larger = fewer candidates (rows then fail the count check: gather-only timing)
        std::vector<float> th(R, (thr_env ? (float)atof(thr_env) : 1.17f) * sigma);
        float* thr; CK(hipMalloc(&thr, R * 4)); CK(hipMemcpy(thr, th.data(), R * 4, hipMemcpyHostToDevice));
        void *stats, *cand;
        CK(hipMalloc(&stats, (size_t)R * (V / 256) * 32)); CK(hipMalloc(&cand, (size_t)R * (V / 256) * MM_FUSED_SLOT * 16));
        if (want("logits")) {
            // one run on zeroed buffers first: the candidate slots are only partly written, so their checksum needs a defined background
            CK(hipMemset(stats, 0, (size_t)R * (V / 256) * 32)); CK(hipMemset(cand, 0, (size_t)R * (V / 256) * MM_FUSED_SLOT * 16));
            MK(mm_gemm_cfg_logits_fused(nullptr, x, nullptr, D, w, D, R, V, D, 3.f, thr, stats, cand));
            CK(hipDeviceSynchronize());
            const uint64_t cs_stats = checksum(stats, (size_t)R * (V / 256) * 32), cs_cand = checksum(cand, (size_t)R * (V / 256) * MM_FUSED_SLOT * 16);
            const double us = time_us([&] { MK(mm_gemm_cfg_logits_fused(nullptr, x, nullptr, D, w, D, R, V, D, 3.f, thr, stats, cand)); }, getenv("MM_ITERS") ? atoi(getenv("MM_ITERS")) : 20);
            report("logits", us, 2.0 * R * (double)V * D, cs_stats);
            printf("         candidates checksum %016llx  (MM_PP=%s)\n", (unsigned long long)cs_cand, getenv("MM_PP") ? getenv("MM_PP") : "0");
        }
        if (want("stamps")) {
            // in-kernel s_memtime stamps of workgroup 0 (a library built with -DMM_GEMM_TIMING: tools/build_timing.sh): per tile {start, [k-steps of the 4th tile], k-loop end,
            // exchange barrier passed, stores issued}, for wave 0 (group 0) and wave 4 (group 1)
            typedef int (*stamps_fn)(unsigned long long*, int);
            const bool pp = getenv("MM_PP") && atoi(getenv("MM_PP")) != 0;
            stamps_fn fn = (stamps_fn)dlsym(RTLD_DEFAULT, pp ? "mm_debug_pp_stamps" : "mm_debug_wide_stamps");
            if (!fn) { printf("stamps: the library was not built with MM_GEMM_TIMING\n"); return 1; }
            for (int i = 0; i < 3; ++i) MK(mm_gemm_cfg_logits_fused(nullptr, x, nullptr, D, w, D, R, V, D, 3.f, thr, stats, cand));
            CK(hipDeviceSynchronize());
            const double us = time_us([&] { MK(mm_gemm_cfg_logits_fused(nullptr, x, nullptr, D, w, D, R, V, D, 3.f, thr, stats, cand)); }, 5);
            std::vector<unsigned long long> st(2 * 2048);
            if (fn(st.data(), 2048)) { printf("stamps: copy failed\n"); return 1; }
            const int extra = getenv("MM_STAMP_EXTRA") ? atoi(getenv("MM_STAMP_EXTRA")) : (pp ? D / 32 : D / 64 - 1);      // stamps inside the 4th tile's k-loop
            printf("stamps (MM_PP=%s, %.1f us per launch, %d in-loop stamps in tile 3)\n", getenv("MM_PP") ? getenv("MM_PP") : "0", us, extra);
            for (int g = 0; g < 2; ++g) {
                const unsigned long long* t = st.data() + g * 2048;
                int i = 0, tile = 0;
                double sk = 0, s1 = 0, s2 = 0, sg = 0; int cnt = 0;
                unsigned long long prev_end = 0;
                while (i + 4 <= 2048 && t[i] && tile < 40) {
                    const unsigned long long t0 = t[i];
                    const int ex = tile == 3 ? extra : 0;
                    if (i + 4 + ex > 2048 || !t[i + 3 + ex]) break;
                    const unsigned long long te = t[i + 1 + ex], tb = t[i + 2 + ex], ts_ = t[i + 3 + ex];
                    if (tile == 3) {
                        printf("  group %d tile 3 in-loop deltas:", g);
                        unsigned long long p0 = t0;
                        for (int j = 0; j < ex; ++j) { printf(" %llu", t[i + 1 + j] - p0); p0 = t[i + 1 + j]; }
                        printf(" | last -> loop end %llu\n", te - p0);
                    }
                    if (tile >= 1 && tile != 3) { sk += (double)(te - t0); s1 += (double)(tb - te); s2 += (double)(ts_ - tb); if (prev_end) sg += (double)(t0 - prev_end); ++cnt; }
                    if (tile < 3 || tile == 3) printf("  group %d tile %d: k-loop %llu  stats+exchange %llu  stores+records %llu  (gap before %lld)\n", g, tile, te - t0, tb - te, ts_ - tb, prev_end ? (long long)(t0 - prev_end) : 0ll);
                    prev_end = ts_;
                    i += 4 + ex; ++tile;
                }
                if (cnt) printf("  group %d mean over %d tiles: k-loop %.0f  stats+exchange %.0f  stores+records %.0f  gap %.0f  -> tile %.0f cycles; %d tiles seen, first->last %.0f cycles\n", g, cnt, sk / cnt, s1 / cnt, s2 / cnt, sg / cnt,
                                (sk + s1 + s2 + sg) / cnt, tile, (double)(prev_end - t[0]));
            }
        }
        if (want("pairs")) {
            // which workgroups share a CU, and how their k-loops / emissions interleave (timing build, MM_PP=13x): per workgroup {hw id, start, per tile: k-loop end, emission end}
            typedef int (*tfn)(unsigned long long*);
            tfn fn = (tfn)dlsym(RTLD_DEFAULT, "mm_debug_pp_tile_stamps");
            if (!fn) { printf("pairs: the library was not built with MM_GEMM_TIMING\n"); return 1; }
            for (int i = 0; i < 2; ++i) MK(mm_gemm_cfg_logits_fused(nullptr, x, nullptr, D, w, D, R, V, D, 3.f, thr, stats, cand));
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> ts(512 * 128);
            if (fn(ts.data())) { printf("pairs: copy failed\n"); return 1; }
            auto cu_key = [&](int b) { const unsigned long long h = ts[(size_t)b * 128]; return (unsigned)(((h >> 32) & 15) << 12) | (unsigned)((h >> 8) & 0xFF); };      // xcc | se, sh, cu
            int shown = 0, odd_second_half = 0, pairs_half = 0, npairs = 0;
            double both_k = 0, total = 0;
            for (int b = 0; b < 512; ++b) {
                for (int c = b + 1; c < 512; ++c) {
                    if (cu_key(b) != cu_key(c)) continue;
                    ++npairs;
                    if ((b < 256) != (c < 256)) ++pairs_half;
                    const unsigned long long* A = &ts[(size_t)b * 128]; const unsigned long long* B = &ts[(size_t)c * 128];
                    if (shown < 3) {
                        printf("  CU %05x: workgroups %d (wave slot %llu) and %d (slot %llu); start %lld apart\n", cu_key(b), b, A[0] & 15, c, B[0] & 15, (long long)(B[1] - A[1]));
                        for (int i = 0; i < 6; ++i) printf("    tile %d: wg %d k-loop [%lld, %lld) emission -> %lld | wg %d k-loop [%lld, %lld) emission -> %lld\n", i,
                                                           b, (long long)((i ? A[1 + 2 * i] : A[1]) - A[1]), (long long)(A[2 + 2 * i] - A[1]), (long long)(A[3 + 2 * i] - A[1]),
                                                           c, (long long)((i ? B[1 + 2 * i] : B[1]) - A[1]), (long long)(B[2 + 2 * i] - A[1]), (long long)(B[3 + 2 * i] - A[1]));
                        ++shown;
                    }
                    // overlap of the two k-loops over tiles 1 .. 15
                    for (int i = 1; i < 16; ++i) {
                        const long long a0 = A[1 + 2 * i], a1 = A[2 + 2 * i];
                        for (int j = 0; j < 18; ++j) {
                            const long long b0 = j ? B[1 + 2 * j] : B[1], b1 = B[2 + 2 * j];
                            const long long lo = a0 > b0 ? a0 : b0, hi = a1 < b1 ? a1 : b1;
                            if (hi > lo) both_k += (double)(hi - lo);
                        }
                        total += (double)(a1 - a0);
                    }
                }
                if ((ts[(size_t)b * 128] & 1) && b >= 256) ++odd_second_half;
            }
            printf("pairs: %d co-resident pairs found, %d of them (first half, second half); odd-slot workgroups in the second half: %d; k-loop time of a workgroup spent beside its partner's k-loop: %.2f\n",
                   npairs, pairs_half, odd_second_half, total > 0 ? both_k / total : 0.);
        }
        if (want("sample")) {
            MK(mm_gemm_cfg_logits_fused(nullptr, x, nullptr, D, w, D, R, V, D, 3.f, thr, stats, cand));
            int64_t* ids; float* sc; int32_t* flag;
            CK(hipMalloc(&ids, (size_t)R * 8)); CK(hipMalloc(&sc, (size_t)R * 4)); CK(hipMalloc(&flag, 8)); CK(hipMemset(flag, 0, 8));
            const double us = time_us([&] { MK(mm_fused_sample(nullptr, thr, stats, cand, R, V, V / 10, nullptr, 1.f, MM_NOISE_PHILOX, nullptr, 0, 1234, 0, 3, nullptr, nullptr, ids, sc, flag, nullptr, nullptr, 0)); });
            int32_t hf[2]; CK(hipMemcpy(hf, flag, 8, hipMemcpyDeviceToHost));
            printf("sample   %9.1f us  flag %d  checksum ids %016llx scores %016llx\n", us, hf[0], (unsigned long long)checksum(ids, (size_t)R * 8), (unsigned long long)checksum(sc, (size_t)R * 4));
        }
    }
    return 0;
}
