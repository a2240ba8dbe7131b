// Composite VQGanVAE entry points (mm_vae_encode / mm_vae_decode_from_ids): the layer list of ResnetEncDec
// (vqgan_vae.py:223-249) + the LFQ quantizer (:424, :430-437) sequenced in C on one stream -- one call per encode / decode, no allocation,
// no synchronisation (hipGraph-capturable), instead of ~40 operator calls from Python.  Pure launch sequencing over the operators of
// gemm*.hip (implicit-GEMM convolutions) and vae.hip; activations NHWC bf16, image in / out NCHW fp32 like the reference.
#include <new>
#include <string.h>
#include <vector>

#include "muse_hip_internal.h"

struct mm_vae {
    mm_vae_desc d;
    std::vector<mm_vae_layer> enc, dec;
};

namespace {

#define RC(x)                \
    do {                     \
        int _rc = (x);       \
        if (_rc) return _rc; \
    } while (0)

inline size_t al(size_t b) { return (b + 255) & ~(size_t)255; }

// shapes after a layer (NHWC): returns false on an unknown kind
bool next_shape(const mm_vae_layer& l, int& H, int& W, int& C) {
    switch (l.kind) {
        case MM_VAE_STEM: C = l.cout; return true;
        case MM_VAE_DOWN: H /= 2; W /= 2; C = l.cout; return true;
        case MM_VAE_RES: case MM_VAE_GLU: return true;
        case MM_VAE_UP: H *= 2; W *= 2; C = l.cout; return true;
        case MM_VAE_HEAD: C = l.cout; return true;
    }
    return false;
}

// workspace: two ping-pong activation buffers + two block temporaries (the 2C-wide conv output and the C-wide GLU / GroupNorm output of a
// residual block) + GroupNorm statistics
struct Plan { size_t act, tmp_wide, tmp_c, stats; };
Plan plan(const std::vector<mm_vae_layer>& layers, int B, int H, int W, int C) {
    Plan p = {0, 0, 0, 0};
    p.act = (size_t)B * H * W * (C < 8 ? 8 : C) * 2;
    for (const mm_vae_layer& l : layers) {
        if (l.kind == MM_VAE_RES || l.kind == MM_VAE_GLU) {
            const size_t wide = (size_t)B * H * W * (l.kind == MM_VAE_GLU ? 2 * C : C) * 2, c = (size_t)B * H * W * C * 2;
            if (wide > p.tmp_wide) p.tmp_wide = wide;
            if (c > p.tmp_c) p.tmp_c = c;
            const size_t st = (size_t)B * (l.groups > 0 ? l.groups : 1) * 2 * sizeof(float);
            if (st > p.stats) p.stats = st;
        }
        next_shape(l, H, W, C);
        const size_t a = (size_t)B * H * W * C * 2;
        if (l.kind != MM_VAE_HEAD && a > p.act) p.act = a;
    }
    return p;
}

// (hf: the handle's storage -- 0 bf16, 1 fp16 with the weights' inverse scale al)
int conv(hipStream_t s, const bf16_t* in, int B, int H, int W, int Cin, const void* w, int Cout, int k, int stride, int off, int Hv, int Wv, int os, int py,
         int px, int Hout, int Wout, const float* bias, int act, const bf16_t* resid, void* out, int nchw, int hf, float al) {
    if (hf) return mm_conv2d_nhwc_half((mm_stream_t)s, in, B, H, W, Cin, w, Cout, k, k, stride, off, off, Hv, Wv, os, py, px, Hout, Wout, bias, act, resid, out, nchw, al);
    return mm_conv2d_nhwc((mm_stream_t)s, in, B, H, W, Cin, w, Cout, k, k, stride, off, off, Hv, Wv, os, py, px, Hout, Wout, bias, act, resid, out, nchw);
}

// runs `layers` on x (NHWC bf16, B x H x W x C); the result is left in *out_buf (one of the two ping-pong buffers) or, for the head, in image_out
int run_layers(const std::vector<mm_vae_layer>& layers, hipStream_t s, int B, int& H, int& W, int& C, bf16_t* cur, bf16_t* other, bf16_t* tw, bf16_t* tc,
               float* stats, float* image_out, bf16_t** result, int hf = 0, float al = 1.f) {
    for (size_t li = 0; li < layers.size(); ++li) {
        const mm_vae_layer& l = layers[li];
        switch (l.kind) {
            case MM_VAE_STEM:      // Conv2d(channels, dim, k, padding k // 2) on the 8-channel padded image
                RC(conv(s, cur, B, H, W, 8, l.w[0], l.cout, l.k, 1, -(l.k / 2), H, W, 1, 0, 0, H, W, l.b[0], 0, nullptr, other, 0, hf, al));
                break;
            case MM_VAE_DOWN:      // Conv2d(4, stride 2, pad 1) + LeakyReLU(0.1)
                RC(conv(s, cur, B, H, W, C, l.w[0], l.cout, 4, 2, -1, H / 2, W / 2, 1, 0, 0, H / 2, W / 2, l.b[0], 1, nullptr, other, 0, hf, al));
                break;
            case MM_VAE_RES:       // ResBlock (vqgan_vae.py:267-281): conv3 -> GN + LeakyReLU -> conv3 -> GN + LeakyReLU -> conv1 + x
                RC(conv(s, cur, B, H, W, C, l.w[0], C, 3, 1, -1, H, W, 1, 0, 0, H, W, l.b[0], 0, nullptr, tw, 0, hf, al));
                RC(k_groupnorm(s, tw, B, H * W, C, l.groups, l.gn_g[0], l.gn_b[0], ACT_LEAKY, stats, tc, hf));
                RC(conv(s, tc, B, H, W, C, l.w[1], C, 3, 1, -1, H, W, 1, 0, 0, H, W, l.b[1], 0, nullptr, tw, 0, hf, al));
                RC(k_groupnorm(s, tw, B, H * W, C, l.groups, l.gn_g[1], l.gn_b[1], ACT_LEAKY, stats, tc, hf));
                RC(conv(s, tc, B, H, W, C, l.w[2], C, 1, 1, 0, H, W, 1, 0, 0, H, W, l.b[2], 0, cur, other, 0, hf, al));
                break;
            case MM_VAE_GLU:       // GLUResBlock (vqgan_vae.py:251-265): conv3 (C -> 2C) -> GLU -> GN -> conv3 -> GLU -> GN -> conv1 + x
                RC(conv(s, cur, B, H, W, C, l.w[0], 2 * C, 3, 1, -1, H, W, 1, 0, 0, H, W, l.b[0], 0, nullptr, tw, 0, hf, al));
                RC(k_glu(s, tw, (long)B * H * W, C, tc, hf));
                RC(k_groupnorm(s, tc, B, H * W, C, l.groups, l.gn_g[0], l.gn_b[0], ACT_NONE, stats, tc, hf));
                RC(conv(s, tc, B, H, W, C, l.w[1], 2 * C, 3, 1, -1, H, W, 1, 0, 0, H, W, l.b[1], 0, nullptr, tw, 0, hf, al));
                RC(k_glu(s, tw, (long)B * H * W, C, tc, hf));
                RC(k_groupnorm(s, tc, B, H * W, C, l.groups, l.gn_g[1], l.gn_b[1], ACT_NONE, stats, tc, hf));
                RC(conv(s, tc, B, H, W, C, l.w[2], C, 1, 1, 0, H, W, 1, 0, 0, H, W, l.b[2], 0, cur, other, 0, hf, al));
                break;
            case MM_VAE_UP: {      // ConvTranspose2d(4, 2, 1) + LeakyReLU(0.1) as four parity 2x2 convolutions
                // Round 6: when the head Conv2d(dim, channels, 1) follows a 256-channel up-sampling layer, it rides in that layer's epilogue (gemm_wide_conv.hip:
                // a 256 x 256 tile holds every channel of its pixels) -- the largest activation of the decoder is neither written nor read back.  The first parity
                // decides: MM_ERR_UNSUPPORTED (shape outside that kernel's class) before anything ran -> the separate sequence below.
                const bool head_next = li + 1 < layers.size() && layers[li + 1].kind == MM_VAE_HEAD && image_out && l.cout == 256 && layers[li + 1].cout <= 8 &&
                                       !(g_mm_debug2 & 16);
                // ... and the four parity classes run as ONE launch of that kernel (the parity is its slowest tile coordinate): MM_ERR_UNSUPPORTED before anything ran
                // (shape outside the kernel's class, or mm_debug_set2(8)) -> the sequences below
                if (!(g_mm_debug2 & 32)) {
                    const mm_vae_layer* hd = head_next ? &layers[li + 1] : nullptr;
                    const int rc4 = mm_convT2d_nhwc_4((mm_stream_t)s, cur, B, H, W, C, l.w, l.cout, l.b[0], 1, hd ? (void*)image_out : (void*)other,
                                                      hd ? hd->w[0] : nullptr, (l.cout + 63) / 64 * 64, hd ? hd->b[0] : nullptr, hd ? hd->cout : 0, hf, al);
                    if (rc4 == MM_OK) {
                        next_shape(l, H, W, C);
                        if (hd) {
                            next_shape(*hd, H, W, C);
                            *result = nullptr;
                            return MM_OK;
                        }
                        bf16_t* t4_ = cur; cur = other; other = t4_;
                        continue;
                    }
                    if (rc4 != MM_ERR_UNSUPPORTED) return rc4;
                }
                if (head_next) {
                    const mm_vae_layer& hd = layers[li + 1];
                    int rc = MM_OK;
                    bool fused = true;
                    for (int py = 0; py < 2 && fused; ++py)
                        for (int px = 0; px < 2; ++px) {
                            rc = mm_conv2d_nhwc_head((mm_stream_t)s, cur, B, H, W, C, l.w[py * 2 + px], l.cout, 2, 2, 1, py - 1, px - 1, H, W, 2, py, px, 2 * H, 2 * W, l.b[0], 1,
                                                     hd.w[0], (l.cout + 63) / 64 * 64, hd.b[0], hd.cout, image_out, hf, al);
                            if (rc == MM_ERR_UNSUPPORTED && py == 0 && px == 0) { fused = false; break; }
                            if (rc) return rc;
                        }
                    if (fused) {
                        next_shape(l, H, W, C);
                        next_shape(hd, H, W, C);
                        *result = nullptr;
                        return MM_OK;
                    }
                }
                for (int py = 0; py < 2; ++py)
                    for (int px = 0; px < 2; ++px)
                        RC(hf ? mm_conv2d_nhwc_half((mm_stream_t)s, cur, B, H, W, C, l.w[py * 2 + px], l.cout, 2, 2, 1, py - 1, px - 1, H, W, 2, py, px, 2 * H, 2 * W, l.b[0], 1,
                                                     nullptr, other, 0, al)
                              : mm_conv2d_nhwc((mm_stream_t)s, cur, B, H, W, C, l.w[py * 2 + px], l.cout, 2, 2, 1, py - 1, px - 1, H, W, 2, py, px, 2 * H, 2 * W, l.b[0], 1,
                                               nullptr, other, 0));
                break;
            }
            case MM_VAE_HEAD:      // Conv2d(dim, channels, 1) -> NCHW fp32 image
                if (!image_out) return mm_set_error(MM_ERR_SHAPE, "vae: head layer without an image output");
                RC(conv(s, cur, B, H, W, C, l.w[0], l.cout, 1, 1, 0, H, W, 1, 0, 0, H, W, l.b[0], 0, nullptr, image_out, 1, hf, al));
                next_shape(l, H, W, C);
                *result = nullptr;
                return MM_OK;
            default:
                return mm_set_error(MM_ERR_UNSUPPORTED, "vae: unknown layer kind");
        }
        next_shape(l, H, W, C);
        bf16_t* t_ = cur; cur = other; other = t_;
    }
    *result = cur;
    return MM_OK;
}

// in-place GroupNorm is safe: k_groupnorm computes the statistics in a first kernel and normalises element-wise in a second

}  // namespace

extern "C" {

int mm_vae_create(const mm_vae_desc* desc, mm_vae_t** out) {
    if (!desc || !out) return mm_set_error(MM_ERR_SHAPE, "vae_create: NULL argument");
    if (desc->n_enc < 0 || desc->n_dec < 0 || (desc->n_enc && !desc->enc) || (desc->n_dec && !desc->dec)) return mm_set_error(MM_ERR_SHAPE, "vae_create: layer lists");
    if (desc->channels <= 0 || desc->channels > 8) return mm_set_error(MM_ERR_UNSUPPORTED, "vae_create: 1..8 image channels");
    if (desc->bits <= 0 || desc->bits > 62 || desc->encoded_dim % 8) return mm_set_error(MM_ERR_SHAPE, "vae_create: bits / encoded_dim");
    if (desc->half && desc->n_enc) return mm_set_error(MM_ERR_UNSUPPORTED, "vae_create: an fp16-storage handle is decode-only (n_enc = 0)");
    mm_vae* v = new (std::nothrow) mm_vae();
    if (!v) return mm_set_error(MM_ERR_HIP, "out of host memory");
    v->d = *desc;
    v->enc.assign(desc->enc, desc->enc + desc->n_enc);
    v->dec.assign(desc->dec, desc->dec + desc->n_dec);
    v->d.enc = v->enc.data(); v->d.dec = v->dec.data();
    *out = v;
    return MM_OK;
}

void mm_vae_destroy(mm_vae_t* vae) { delete vae; }

size_t mm_vae_decode_workspace_bytes(const mm_vae_t* v, int B, int h, int w) {
    if (!v) return 0;
    const Plan p = plan(v->dec, B, h, w, v->d.encoded_dim);
    return 2 * al(p.act) + al(p.tmp_wide) + al(p.tmp_c) + al(p.stats) + 256;
}

size_t mm_vae_encode_workspace_bytes(const mm_vae_t* v, int B, int H, int W) {
    if (!v) return 0;
    const Plan p = plan(v->enc, B, H, W, 8);
    return 2 * al(p.act) + al(p.tmp_wide) + al(p.tmp_c) + al(p.stats) + 256;
}

int mm_vae_decode_from_ids(const mm_vae_t* v, mm_stream_t stream, const int64_t* ids, int B, int h, int w, float* image, void* workspace, size_t workspace_bytes) {
    if (!v) return mm_set_error(MM_ERR_SHAPE, "vae handle is NULL");
    if (!ids || !image || !workspace) return mm_set_error(MM_ERR_SHAPE, "vae_decode_from_ids: NULL argument");
    if (B <= 0 || h <= 0 || w <= 0) return mm_set_error(MM_ERR_SHAPE, "vae_decode_from_ids: bad sizes");
    if (workspace_bytes < mm_vae_decode_workspace_bytes(v, B, h, w)) return mm_set_error(MM_ERR_WORKSPACE, "vae_decode_from_ids: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const Plan p = plan(v->dec, B, h, w, v->d.encoded_dim);
    unsigned char* base = (unsigned char*)workspace;
    bf16_t* a0 = (bf16_t*)base; bf16_t* a1 = (bf16_t*)(base + al(p.act));
    bf16_t* tw = (bf16_t*)(base + 2 * al(p.act)); bf16_t* tc = (bf16_t*)(base + 2 * al(p.act) + al(p.tmp_wide));
    float* stats = (float*)(base + 2 * al(p.act) + al(p.tmp_wide) + al(p.tmp_c));
    // LFQ.indices_to_codes + project_out (vqgan_vae.py:430-432) straight into NHWC
    const int hf = v->d.half ? 1 : 0;
    RC(k_lfq_decode(s, ids, (long)B * h * w, v->d.bits, v->d.encoded_dim, v->d.lfq_wo, v->d.lfq_bo, a0, hf));
    int H = h, W = w, C = v->d.encoded_dim;
    bf16_t* res = nullptr;
    return run_layers(v->dec, s, B, H, W, C, a0, a1, tw, tc, stats, image, &res, hf, v->d.alpha != 0.f ? v->d.alpha : 1.f);
}

int mm_vae_encode(const mm_vae_t* v, mm_stream_t stream, const float* image, int B, int H, int W, float* fmap_out, int64_t* ids_out, void* workspace,
                  size_t workspace_bytes) {
    if (!v) return mm_set_error(MM_ERR_SHAPE, "vae handle is NULL");
    if (!image || !ids_out || !workspace) return mm_set_error(MM_ERR_SHAPE, "vae_encode: NULL argument");
    if (B <= 0 || H <= 0 || W <= 0) return mm_set_error(MM_ERR_SHAPE, "vae_encode: bad sizes");
    if (v->d.half) return mm_set_error(MM_ERR_UNSUPPORTED, "vae_encode: this handle is the fp16-storage decoder");
    if (workspace_bytes < mm_vae_encode_workspace_bytes(v, B, H, W)) return mm_set_error(MM_ERR_WORKSPACE, "vae_encode: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const Plan p = plan(v->enc, B, H, W, 8);
    unsigned char* base = (unsigned char*)workspace;
    bf16_t* a0 = (bf16_t*)base; bf16_t* a1 = (bf16_t*)(base + al(p.act));
    bf16_t* tw = (bf16_t*)(base + 2 * al(p.act)); bf16_t* tc = (bf16_t*)(base + 2 * al(p.act) + al(p.tmp_wide));
    float* stats = (float*)(base + 2 * al(p.act) + al(p.tmp_wide) + al(p.tmp_c));
    RC(k_nchw_to_nhwc8(s, image, B, v->d.channels, H, W, a0));
    int h = H, w = W, C = 8;
    bf16_t* res = nullptr;
    RC(run_layers(v->enc, s, B, h, w, C, a0, a1, tw, tc, stats, nullptr, &res));
    if (!res || C != v->d.encoded_dim) return mm_set_error(MM_ERR_SHAPE, "vae_encode: the encoder does not end at encoded_dim channels");
    // LFQ.forward in eval mode (vqgan_vae.py:424): ids + quantized features; the quantized map goes to the buffer the features are not in
    bf16_t* q = (res == a0) ? a1 : a0;
    RC(k_lfq_encode(s, res, (long)B * h * w, C, v->d.bits, v->d.lfq_wi, v->d.lfq_bi, v->d.lfq_wo, v->d.lfq_bo, ids_out, q));
    if (fmap_out) RC(k_nhwc_to_nchw_f32(s, q, B, C, h, w, fmap_out));
    return MM_OK;
}

}  // extern "C"
