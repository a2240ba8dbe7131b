"""precision = 'parity': the reference's operator sequence on the fp32 engine of libmuse_hip.so (csrc/parity.hip: fp32 storage, fp32 MFMA).

`Transformer.set_precision('parity')` / `VQGanVAE.set_precision('parity')` / `MaskGit.set_precision('parity')` route the forward, the guidance
pass, the decode loop (stepwise, full-vocabulary logits at every position -- what the reference computes) and the VAE through this module.
It exists to PROVE the arithmetic: logits and pixels within 1e-3 of the reference's fp32 run and bit-equal token ids at full size
(tests/test_gpu_base_size.py), where the production bf16 engine can only be compared through rounding-point oracles.  Inference only.

torch is used for device memory and data movement (allocation, views, one concatenation of the context rows); every arithmetic operation
is a kernel behind the C ABI (`mm_f32_*`, include/muse_hip.h).  Reference lines: muse_maskgit_pytorch.py (mmp) / vqgan_vae.py (vae).
"""
import torch
from torch import nn

from . import _lib as L

f32 = torch.float32


def _w(p):
    return p.detach().to(f32).contiguous()


# ------------------------------------------------------------------------------------------------ operator wrappers (C ABI)
def gemm(x, w, bias=None, act=False, resid=None):
    """x [M, K] @ w [N, K]^T (+ bias, LeakyReLU(0.1), + resid) -> fp32 [M, N]"""
    assert x.dtype == f32 and w.dtype == f32 and x.stride(-1) == 1 and w.stride(-1) == 1 and x.shape[1] == w.shape[1]
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=f32, device=x.device)
    if resid is not None:
        assert resid.shape == out.shape and resid.is_contiguous()
    L.check(L.lib().mm_f32_gemm(L.stream(), L.ptr(x), x.stride(0), L.ptr(w), w.stride(0), M, N, K, L.ptr(out), N, L.ptr(bias), int(act), L.ptr(resid)),
            'mm_f32_gemm')
    return out


def layernorm(x, gamma, beta=None):
    rows, D = x.shape
    out = torch.empty(rows, D, dtype=f32, device=x.device)
    L.check(L.lib().mm_f32_layernorm(L.stream(), L.ptr(x), x.stride(0), rows, D, L.ptr(gamma), L.ptr(beta), L.ptr(out), D), 'mm_f32_layernorm')
    return out


def geglu(h):
    rows, two_f = h.shape
    F = two_f // 2
    out = torch.empty(rows, F, dtype=f32, device=h.device)
    L.check(L.lib().mm_f32_geglu(L.stream(), L.ptr(h), h.stride(0), rows, F, L.ptr(out), F), 'mm_f32_geglu')
    return out


def cfg_combine(cond, null, scale):
    out = torch.empty_like(cond)
    L.check(L.lib().mm_f32_cfg_combine(L.stream(), L.ptr(cond), L.ptr(null), float(scale), cond.numel(), L.ptr(out)), 'mm_f32_cfg_combine')
    return out


def embed(ids, tok, pos=None):
    ids = ids.contiguous()
    rows, D = ids.numel(), tok.shape[1]
    n = ids.shape[-1]
    out = torch.empty(rows, D, dtype=f32, device=ids.device)
    L.check(L.lib().mm_f32_embed(L.stream(), L.ptr(ids), rows, n, L.ptr(tok), tok.shape[0], L.ptr(pos), D, L.ptr(out), D), 'mm_f32_embed')
    return out


def text_mask(te):
    b, Lt, D = te.shape
    m = torch.empty(b, Lt, dtype=torch.uint8, device=te.device)
    L.check(L.lib().mm_f32_text_mask(L.stream(), L.ptr(te), b * Lt, D, L.ptr(m)), 'mm_f32_text_mask')
    return m


def attend(q, k, v, b, heads, nq, nk, q_strides, k_strides, v_strides, key_mask=None, q_scale=None, k_scale=None, null_k=None, null_v=None, scale=8.0, dh=64):
    """q / k / v: fp32 storage addressed by element strides (batch, head, token), d (dh = 32 / 64 / 128) contiguous; returns [b * nq, heads * dh]"""
    I = heads * dh
    out = torch.empty(b * nq, I, dtype=f32, device=q.device)
    L.check(L.lib().mm_f32_attend(L.stream(), L.ptr(q), *q_strides, L.ptr(k), *k_strides, L.ptr(v), *v_strides, L.ptr(out), nq * I, dh, I, b, heads, nq, nk,
                                  L.ptr(key_mask), key_mask.stride(0) if key_mask is not None else 0, int(q_scale is not None), L.ptr(q_scale), L.ptr(k_scale),
                                  L.ptr(null_k), L.ptr(null_v), float(scale), dh), 'mm_f32_attend')
    return out


def attend_terms(q, k, v, b, heads, nq, nk, q_strides, k_strides, v_strides, q_scale=None, k_scale=None, null_k=None, null_v=None, scale=8.0):
    """the same attention as fp16 term products on the fp16 matrix pipe (csrc/attention_x2.hip, the 'f16x2' tier's self-attention kernel); dim_head 64"""
    I = heads * 64
    out = torch.empty(b * nq, I, dtype=f32, device=q.device)
    L.check(L.lib().mm_attend_terms(L.stream(), L.ptr(q), *q_strides, L.ptr(k), *k_strides, L.ptr(v), *v_strides, L.ptr(out), nq * I, 64, I, b, heads, nq, nk,
                                    int(q_scale is not None), L.ptr(q_scale), L.ptr(k_scale), L.ptr(null_k), L.ptr(null_v), float(scale)), 'mm_attend_terms')
    return out


class _View:
    """a data_ptr() at an element offset inside a tensor (k / v halves of a fused projection)"""

    def __init__(self, t, offset):
        self.t, self.offset = t, offset

    def data_ptr(self):
        return self.t.data_ptr() + self.offset * self.t.element_size()


# ------------------------------------------------------------------------------------------------ Transformer (mmp.py:63-335)
def _feed_forward(ff, x_in, resid):
    """mmp.py:79-89: LN -> Linear(D, 2F) -> GEGLU -> LN(F) -> Linear(F, D), + resid"""
    u = layernorm(x_in, _w(ff[0].gamma), _w(ff[0].beta))
    a = geglu(gemm(u, _w(ff[1].weight)))
    z = layernorm(a, _w(ff[3].gamma), _w(ff[3].beta))
    return gemm(z, _w(ff[4].weight), resid=resid)


def _attention(a, x, b, n, heads, context=None, m=0, key_mask=None):
    """mmp.py:126-162 (+ the residual of :189-191): x [b*n, D]; context [b*m, D] or None (self-attention)"""
    dh = a.to_q.weight.shape[0] // heads
    I = heads * dh
    xn = layernorm(x, _w(a.norm.gamma), _w(a.norm.beta))
    q = gemm(xn, _w(a.to_q.weight))                                  # [b*n, I]
    kv_in, nk = (xn, n) if context is None else (context, m)
    kv = gemm(kv_in, _w(a.to_kv.weight))                             # [b*nk, 2I]: k | v  (chunk(2, dim=-1), :137)
    nkv = _w(a.null_kv)                                              # (2, h, 1, dh)
    o = attend(q, kv, _View(kv, I), b, heads, n, nk, (n * I, dh, I), (nk * 2 * I, dh, 2 * I), (nk * 2 * I, dh, 2 * I), key_mask=key_mask,
               q_scale=_w(a.q_scale), k_scale=_w(a.k_scale), null_k=nkv[0].reshape(heads, dh).contiguous(), null_v=nkv[1].reshape(heads, dh).contiguous(),
               scale=float(a.scale), dh=dh)
    return gemm(o, _w(a.to_out.weight), resid=x)


def transformer_run(tr, ids, text_embeds, cond_drop_prob=0., conditioning_token_ids=None, self_cond_embed=None, want_logits=True):
    """Transformer.forward without the loss branch (mmp.py:293-335) -> (embed fp32 [b*n, D], logits fp32 [b*n, dim_out] or None)"""
    L.require_device()
    dev = tr.token_emb.weight.device
    if dev.type != 'cuda':
        raise L.MuseHipError('Transformer parameters are not on the GPU; the MI355X path has no CPU fallback')
    ids = ids.to(device=dev, dtype=torch.long).contiguous()
    b, n = ids.shape
    assert n <= tr.seq_len                                                                     # mmp.py:293
    te = text_embeds.to(device=dev, dtype=f32).contiguous()
    Lt = te.shape[1]
    cfgb = tr.transformer_blocks.cfg
    heads, D = cfgb['heads'], tr.dim
    if cfgb['dim_head'] not in (32, 64, 128):
        raise L.MuseHipError('dim_head must be 32, 64 or 128')
    ctx = te.reshape(b * Lt, -1)
    if isinstance(tr.text_embed_proj, nn.Linear):                                              # mmp.py:233, 302
        ctx = gemm(ctx, _w(tr.text_embed_proj.weight))
    mask = text_mask(te)                                                                       # mmp.py:304 (on the RAW embeds)
    if cond_drop_prob == 1.:                                                                   # mmp.py:308-310
        mask.zero_()
    elif cond_drop_prob > 0.:
        mask *= (torch.rand((b, 1), device=dev) < (1. - cond_drop_prob)).to(torch.uint8)
    m = Lt
    if conditioning_token_ids is not None:                                                     # mmp.py:314-318
        cids = conditioning_token_ids.reshape(b, -1).to(device=dev, dtype=torch.long).contiguous()
        nc = cids.shape[1]
        cctx = embed(cids, _w(tr.token_emb.weight))
        ctx = torch.cat((ctx.reshape(b, Lt, D), cctx.reshape(b, nc, D)), dim=1).reshape(b * (Lt + nc), D).contiguous()      # data movement only
        mask = torch.cat((mask, torch.ones(b, nc, dtype=torch.uint8, device=dev)), dim=1).contiguous()
        m = Lt + nc
    x = embed(ids, _w(tr.token_emb.weight), _w(tr.pos_emb.weight))                             # mmp.py:322-323
    if tr.self_cond:                                                                           # mmp.py:325-328
        sce = torch.zeros_like(x) if self_cond_embed is None else self_cond_embed.to(device=dev, dtype=f32).reshape(b * n, D).contiguous()
        x = _feed_forward(tr.self_cond_to_init_embed, sce, x)
    for sa, ca, ff in tr.transformer_blocks.layers:                                            # mmp.py:187-195
        x = _attention(sa, x, b, n, heads)
        x = _attention(ca, x, b, n, heads, context=ctx, m=m, key_mask=mask)
        x = _feed_forward(ff, x, x)
    emb = layernorm(x, _w(tr.transformer_blocks.norm.gamma), _w(tr.transformer_blocks.norm.beta))
    logits = gemm(emb, _w(tr.to_logits.weight)) if want_logits else None                        # mmp.py:332
    return emb, logits


def cfg_logits(tr, emb_a, emb_b, cond_scale):
    """to_logits on both passes + null + (cond - null) * cond_scale (mmp.py:250-254)"""
    w = _w(tr.to_logits.weight)
    return cfg_combine(gemm(emb_a.to(f32), w), gemm(emb_b.to(f32), w), cond_scale)


def linear_head(embeds, lin):
    """SelfCritic.to_pred (mmp.py:359, 372): Linear(dim, 1) with bias"""
    return gemm(embeds, _w(lin.weight), bias=_w(lin.bias))


# ------------------------------------------------------------------------------------------------ VQGanVAE (vae.py:185-281, 422-441)
def conv(x, w_packed, cout, th, tw, stride=1, off=(0, 0), out_hw=None, os_=1, parity=(0, 0), full_hw=None, bias=None, act=False, resid=None, out=None,
         out_nchw=False):
    B, H, W, Cin = x.shape
    Hv, Wv = out_hw if out_hw is not None else (H, W)
    Hout, Wout = full_hw if full_hw is not None else (Hv * os_, Wv * os_)
    if out is None:
        out = torch.empty((B, cout, Hout, Wout) if out_nchw else (B, Hout, Wout, cout), dtype=f32, device=x.device)
    L.check(L.lib().mm_f32_conv2d_nhwc(L.stream(), L.ptr(x), B, H, W, Cin, L.ptr(w_packed), cout, th, tw, stride, off[0], off[1], Hv, Wv, os_, parity[0],
                                       parity[1], Hout, Wout, L.ptr(bias), int(act), L.ptr(resid), L.ptr(out), int(out_nchw)), 'mm_f32_conv2d_nhwc')
    return out


# ---- 'bf16x3' form of the convolutions (precision tier, decode path): the fp32 NHWC activation is split into P exact bf16 term segments per pixel
#      (mm_split_rows), the weight is the matching per-tap segment pack, and the product runs on the bf16 MFMA implicit-GEMM kernels with fp32 output:
#      exact to fp32 accumulation like the transformer's tier (csrc/split.hip), at bf16-MFMA rate x P instead of the 1/16-rate fp32 MFMA.
# 'f16x2' (round 4): the same form on fp16 terms -- MM_SPLIT_F16 | 2 / 3 segments per pixel, weight terms scaled by a power of two, fp16 MFMA (mm_conv2d_nhwc_f16).
# The packed weights are cached per VAE and parameter version (`cache`): repacking 200 M decoder parameters on every decode cost more than the convolutions.
_X3 = {'P': 0, 'code': 0, 'scale': 1.0, 'cache': None}          # P > 0 while a precision-tier decode runs (set by vae_decode_*): conv() then takes the split form


def _pack_x3(w2d_taps, code, scale=1.0):
    """list of per-tap fp32 [Cout][Cin] matrices -> 16-bit [Cout][Kp], k = tap * (P * Cin) + segment * Cin + ci, zero-padded to a multiple of 64"""
    from . import ops
    return ops.pad_cols(torch.cat([ops.split_pack_weight(wt.contiguous(), code, 1, scale) for wt in w2d_taps], dim=1), 64)


def _cached_pack(key, make_taps):
    """the packed term segments of one convolution weight (key: the weight Parameter + a tag), built once per parameter version of the VAE"""
    cache = _X3['cache']
    if cache is None:
        return _pack_x3(make_taps(), _X3['code'], _X3['scale'])
    if key not in cache:
        cache[key] = _pack_x3(make_taps(), _X3['code'], _X3['scale'])
    return cache[key]


def split_nhwc(x):
    """the term-segment form of an fp32 NHWC activation for the current tier (what conv_x3 multiplies): [B, H, W, P * C]"""
    from . import ops
    B, H, W, Cin = x.shape
    return ops.split_rows(x.reshape(-1, Cin), _X3['code']).reshape(B, H, W, _X3['P'] * Cin)


def conv_x3(x, w_taps, cout, th, tw, stride=1, off=(0, 0), out_hw=None, os_=1, parity=(0, 0), full_hw=None, bias=None, act=False, resid=None, out=None,
            out_nchw=False, xs=None):
    """xs (optional): split_nhwc(x) made by the caller -- the four output parities of a ConvTranspose2d read the SAME input (round 5: split once, not four times)"""
    from . import ops
    P, code = _X3['P'], _X3['code']
    B, H, W, Cin = x.shape
    if xs is None:
        xs = split_nhwc(x)
    wp = w_taps if torch.is_tensor(w_taps) else _pack_x3(w_taps, code, _X3['scale'])      # (already packed: _cached_pack)
    Hv, Wv = out_hw if out_hw is not None else (H, W)
    Hout, Wout = full_hw if full_hw is not None else (Hv * os_, Wv * os_)
    if out is None:
        out = torch.empty((B, cout, Hout, Wout) if out_nchw else (B, Hout, Wout, cout), dtype=f32, device=x.device)
    if out_nchw:
        assert resid is None
    if ops.split_is_f16(code):
        # (round 5) the packs are genuine term segments per pixel / per tap: the 256 x 128 kernel may stage every term plane once (MM_SPLIT_SHARED)
        L.check(L.lib().mm_conv2d_nhwc_terms(L.stream(), L.ptr(xs), B, H, W, P * Cin, L.ptr(wp), cout, th, tw, stride, off[0], off[1], Hv, Wv, os_,
                                             parity[0], parity[1], Hout, Wout, L.ptr(bias), int(act), L.ptr(resid), L.ptr(out), 1 if out_nchw else 2,
                                             1.0 / _X3['scale'], int(code) | ops.MM_SPLIT_SHARED), 'mm_conv2d_nhwc_terms')
        return out
    L.check(L.lib().mm_conv2d_nhwc(L.stream(), L.ptr(xs), B, H, W, P * Cin, L.ptr(wp), cout, th, tw, stride, off[0], off[1], Hv, Wv, os_,
                                   parity[0], parity[1], Hout, Wout, L.ptr(bias), int(act), L.ptr(resid), L.ptr(out), 1 if out_nchw else 2), 'mm_conv2d_nhwc')
    return out


def _taps_conv(w):
    """Conv2d weight [Cout, Cin, TH, TW] -> per-tap fp32 [Cout][Cin] matrices in (ty, tx) order"""
    w = w.detach().to(f32)
    return [w[:, :, ty, tx] for ty in range(w.shape[2]) for tx in range(w.shape[3])]


def _taps_convT(w):
    """ConvTranspose2d(4,2,1) weight [Cin, Cout, 4, 4] -> per output parity the four taps' fp32 [Cout][Cin] matrices (the tap algebra of pack_convT)"""
    w = w.detach().to(f32)
    return {(py, px): [w[:, :, 3 - py - 2 * ty, 3 - px - 2 * tx].t() for ty in range(2) for tx in range(2)] for py in range(2) for px in range(2)}


def pack_conv(w):
    """Conv2d weight [Cout, Cin, TH, TW] -> fp32 [Cout, TH*TW*Cin], k = (ty*TW + tx)*Cin + ci"""
    return w.detach().to(f32).permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


def pack_convT(w):
    """ConvTranspose2d(4,2,1) weight [Cin, Cout, 4, 4] -> four fp32 [Cout, 4*Cin] matrices, one per output parity (same tap algebra as ops.pack_convT_weight)"""
    w = w.detach().to(f32)
    packs = {}
    for py in range(2):
        for px in range(2):
            taps = [w[:, :, 3 - py - 2 * ty, 3 - px - 2 * tx].t() for ty in range(2) for tx in range(2)]
            packs[(py, px)] = torch.cat(taps, dim=1).contiguous()
    return packs


def _glu(x):
    C = x.shape[-1] // 2
    out = torch.empty(*x.shape[:-1], C, dtype=f32, device=x.device)
    L.check(L.lib().mm_f32_glu_nhwc(L.stream(), L.ptr(x), x.numel() // (2 * C), C, L.ptr(out)), 'mm_f32_glu_nhwc')
    return out


def _groupnorm(x, gn, act=False):
    B, H, W, C = x.shape
    out = torch.empty_like(x)
    L.check(L.lib().mm_f32_groupnorm_nhwc(L.stream(), L.ptr(x), B, H * W, C, gn.num_groups, L.ptr(_w(gn.weight)), L.ptr(_w(gn.bias)), int(act), L.ptr(out)),
            'mm_f32_groupnorm_nhwc')
    return out


def _conv_module(x, c, **kw):
    k = c.kernel_size[0]
    if _X3['P']:
        return conv_x3(x, _cached_pack((c.weight, 'conv'), lambda: _taps_conv(c.weight)), c.out_channels, k, k, 1, (-(k // 2), -(k // 2)), bias=_w(c.bias), **kw)
    return conv(x, pack_conv(c.weight), c.out_channels, k, k, 1, (-(k // 2), -(k // 2)), bias=_w(c.bias), **kw)


def _vae_layer(x, m, first=False, last=False):
    from .vqgan_vae import ResBlock, GLUResBlock
    if isinstance(m, ResBlock):                                                               # vae.py:267-281
        net = m.net
        h = _groupnorm(_conv_module(x, net[0]), net[1], act=True)
        h = _groupnorm(_conv_module(h, net[3]), net[4], act=True)
        return _conv_module(h, net[6], resid=x)
    if isinstance(m, GLUResBlock):                                                            # vae.py:251-265
        net = m.net
        h = _groupnorm(_glu(_conv_module(x, net[0])), net[2])
        h = _groupnorm(_glu(_conv_module(h, net[3])), net[5])
        return _conv_module(h, net[6], resid=x)
    if isinstance(m, nn.Sequential) and isinstance(m[0], nn.ConvTranspose2d):                 # ConvTranspose2d(4,2,1) + LeakyReLU(0.1)
        ct = m[0]
        B, H, W, _ = x.shape
        out = torch.empty(B, 2 * H, 2 * W, ct.out_channels, dtype=f32, device=x.device)
        if _X3['P']:
            xs = split_nhwc(x)          # one split of the input for the four output parities
            for py in range(2):
                for px in range(2):
                    wp = _cached_pack((ct.weight, 'convT', py, px), lambda: _taps_convT(ct.weight)[(py, px)])
                    conv_x3(x, wp, ct.out_channels, 2, 2, 1, (py - 1, px - 1), out_hw=(H, W), os_=2, parity=(py, px), full_hw=(2 * H, 2 * W), bias=_w(ct.bias),
                            act=True, out=out, xs=xs)
            return out
        for (py, px), wp in pack_convT(ct.weight).items():
            conv(x, wp, ct.out_channels, 2, 2, 1, (py - 1, px - 1), out_hw=(H, W), os_=2, parity=(py, px), full_hw=(2 * H, 2 * W), bias=_w(ct.bias), act=True,
                 out=out)
        return out
    if isinstance(m, nn.Sequential) and isinstance(m[0], nn.Conv2d):                          # Conv2d(4, 2, 1) + LeakyReLU(0.1)
        c = m[0]
        B, H, W, _ = x.shape
        return conv(x, pack_conv(c.weight), c.out_channels, 4, 4, 2, (-1, -1), out_hw=(H // 2, W // 2), bias=_w(c.bias), act=True)
    if isinstance(m, nn.Conv2d):
        return _conv_module(x, m, out_nchw=last)
    raise NotImplementedError(type(m).__name__)


def _nchw_to_nhwc(img):
    B, C, H, W = img.shape
    out = torch.empty(B, H, W, C, dtype=f32, device=img.device)
    L.check(L.lib().mm_f32_nchw_to_nhwc(L.stream(), L.ptr(img.contiguous()), B, C, H * W, L.ptr(out)), 'mm_f32_nchw_to_nhwc')
    return out


def _nhwc_to_nchw(x):
    B, H, W, C = x.shape
    out = torch.empty(B, C, H, W, dtype=f32, device=x.device)
    L.check(L.lib().mm_f32_nhwc_to_nchw(L.stream(), L.ptr(x.contiguous()), B, C, H * W, L.ptr(out)), 'mm_f32_nhwc_to_nchw')
    return out


def vae_decode_nhwc(vae, x):
    """ResnetEncDec.decode (vae.py:246-249): NHWC fp32 feature map -> NCHW fp32 image.  vae.precision 'bf16x3': the convolutions take the split form above
    (P from the checkpoint: 3 when every convolution weight is bf16-representable, else 5 / 6), everything else is the fp32 engine's."""
    dec = list(vae.enc_dec.decoders)
    if getattr(vae, 'precision', 'bf16') in ('bf16x3', 'f16x2'):
        _X3['code'], _X3['scale'] = vae.x3_code(), vae.x3_scale()
        _X3['P'] = _X3['code'] & 0xff
        _X3['cache'] = vae.x3_pack_cache()
    try:
        for i, m in enumerate(dec):
            x = _vae_layer(x, m, last=(i == len(dec) - 1))
    finally:
        _X3['P'], _X3['code'], _X3['scale'], _X3['cache'] = 0, 0, 1.0, None
    return x


def vae_decode_from_ids(vae, ids):
    """vae.py:427-438 (LFQ branch)"""
    L.require_device()
    q = vae.quantizer
    ids = ids.to(vae.device).contiguous()
    B, h, w = ids.shape
    C = vae.enc_dec.encoded_dim
    codes = torch.empty(B, h, w, C, dtype=f32, device=ids.device)
    has_proj = isinstance(q.project_out, nn.Linear)
    L.check(L.lib().mm_f32_lfq_decode(L.stream(), L.ptr(ids), ids.numel(), q.codebook_dim, C, L.ptr(_w(q.project_out.weight)) if has_proj else None,
                                      L.ptr(_w(q.project_out.bias)) if has_proj else None, L.ptr(codes)), 'mm_f32_lfq_decode')
    return vae_decode_nhwc(vae, codes)


def vae_decode(vae, fmap):
    L.require_device()
    return vae_decode_nhwc(vae, _nchw_to_nhwc(fmap.to(device=vae.device, dtype=f32)))


def vae_encode(vae, img):
    """vae.py:422-425 -> (quantized fmap NCHW fp32, ids (B, h, w) int64, aux loss 0)"""
    L.require_device()
    q = vae.quantizer
    x = _nchw_to_nhwc(img.to(device=vae.device, dtype=f32))
    for i, m in enumerate(vae.enc_dec.encoders):
        x = _vae_layer(x, m, first=(i == 0))
    B, h, w, C = x.shape
    bits = q.codebook_dim
    has_proj = isinstance(q.project_in, nn.Linear)
    t = gemm(x.reshape(B * h * w, C), _w(q.project_in.weight), bias=_w(q.project_in.bias)) if has_proj else x.reshape(B * h * w, C)
    ids = torch.empty(B, h, w, dtype=torch.long, device=x.device)
    L.check(L.lib().mm_f32_lfq_bits(L.stream(), L.ptr(t), B * h * w, bits, L.ptr(ids)), 'mm_f32_lfq_bits')
    codes = torch.empty(B, h, w, C, dtype=f32, device=x.device)
    L.check(L.lib().mm_f32_lfq_decode(L.stream(), L.ptr(ids), ids.numel(), bits, C, L.ptr(_w(q.project_out.weight)) if has_proj else None,
                                      L.ptr(_w(q.project_out.bias)) if has_proj else None, L.ptr(codes)), 'mm_f32_lfq_decode')
    return _nhwc_to_nchw(codes), ids, torch.zeros((), device=x.device)
