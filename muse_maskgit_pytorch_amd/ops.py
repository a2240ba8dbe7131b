"""Tensor-level wrappers over the C ABI (include/muse_hip.h).  torch supplies device memory and the stream only;
every computation below happens inside libmuse_hip.so.  All tensors must live on the current gfx950 device."""
import math

import torch

from . import _lib as L

bf16 = torch.bfloat16


def _chk_cuda(*ts):
    L.require_device()
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.MuseHipError('tensor is not on the GPU: the muse_maskgit_pytorch_amd ops have no CPU path')


def pad_cols(t, mult):
    """zero-pad the last dim to a multiple of `mult` (contiguous copy)."""
    k = t.shape[-1]
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return t.contiguous()
    out = torch.zeros(*t.shape[:-1], kp, dtype=t.dtype, device=t.device)
    out[..., :k] = t
    return out


def gemm(x, w, out_f32=False, resid=None, out=None):
    """x bf16 [M,K] @ w bf16 [N,K]^T -> [M,N] (bf16 or fp32); resid fp32 [M,N] added in the epilogue."""
    _chk_cuda(x, w, resid)
    assert x.dtype == bf16 and w.dtype == bf16 and x.stride(-1) == 1 and w.stride(-1) == 1
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        ldc = (N + 7) // 8 * 8                 # 16-byte aligned rows for the vectorised write-out
        out = torch.empty(M, ldc, dtype=torch.float32 if out_f32 else bf16, device=x.device)[:, :N]
    L.check(L.lib().mm_gemm_bf16(L.stream(), L.ptr(x), x.stride(0), L.ptr(w), w.stride(0), M, N, K, L.ptr(out),
                                 out.stride(0), int(out_f32), L.ptr(resid)), 'mm_gemm_bf16')
    return out


def gemm_cfg_logits(x_cond, x_null, w, cond_scale, out=None):
    _chk_cuda(x_cond, x_null, w)
    M, K = x_cond.shape
    N = w.shape[0]
    assert x_null.shape == x_cond.shape and x_cond.stride(0) == x_null.stride(0)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=w.device)
    L.check(L.lib().mm_gemm_cfg_logits(L.stream(), L.ptr(x_cond), L.ptr(x_null), x_cond.stride(0), L.ptr(w), w.stride(0),
                                       M, N, K, L.ptr(out), out.stride(0), float(cond_scale)), 'mm_gemm_cfg_logits')
    return out


def embed(ids, token_emb, pos_emb):
    _chk_cuda(ids, token_emb, pos_emb)
    b, n = ids.shape
    D = token_emb.shape[1]
    x = torch.empty(b * n, D, dtype=torch.float32, device=ids.device)
    L.check(L.lib().mm_embed(L.stream(), L.ptr(ids.contiguous()), b * n, n, L.ptr(token_emb), token_emb.shape[0],
                             L.ptr(pos_emb), D, L.ptr(x)), 'mm_embed')
    return x


def layernorm(x, gamma, beta=None, row_index=None):
    _chk_cuda(x, gamma, beta, row_index)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    rows = x.shape[0] if row_index is None else row_index.numel()
    D = x.shape[1]
    out = torch.empty(rows, D, dtype=bf16, device=x.device)
    L.check(L.lib().mm_layernorm(L.stream(), L.ptr(x), x.stride(0), rows, D, L.ptr(gamma), L.ptr(beta), L.ptr(row_index),
                                 L.ptr(out), D), 'mm_layernorm')
    return out


def geglu_ln(h, F, gamma, beta=None):
    _chk_cuda(h, gamma, beta)
    rows, two_fp = h.shape
    Fp = two_fp // 2
    out = torch.empty(rows, Fp, dtype=bf16, device=h.device)
    gamma, beta = pad_cols(gamma.float(), Fp), (pad_cols(beta.float(), Fp) if beta is not None else None)
    L.check(L.lib().mm_geglu_ln(L.stream(), L.ptr(h), h.stride(0), rows, F, Fp, L.ptr(gamma), L.ptr(beta), L.ptr(out), Fp),
            'mm_geglu_ln')
    return out


def pack_w1_geglu(w1, Fp, dtype=bf16):
    """FeedForward's first Linear weight [2F, D] (rows [0,F) = gelu half, [F,2F) = gate half, mmp.py:72-77,85) -> `dtype` (bf16; fp32 for
    the fp8 engine, which quantises the interleaved rows itself: no bf16 rounding in between)
    [2*Fp, D] in the tile order the GEGLU-fused GEMM epilogue expects: per 128-row tile t and wave half w, 32 gelu-half
    rows (output columns 64t+32w .. +31) followed by the 32 gate-half rows of the same columns; columns >= F are zero."""
    F2, D = w1.shape
    F = F2 // 2
    out = torch.zeros(2 * Fp, D, dtype=dtype, device=w1.device)
    r = torch.arange(2 * Fp, device=w1.device)
    t, rem = r // 128, r % 128
    w, rem2 = rem // 64, rem % 64
    is_gate = rem2 >= 32
    col = 64 * t + 32 * w + (rem2 % 32)                 # output column of this packed row
    src = torch.where(is_gate, col + F, col)
    valid = col < F
    out[valid] = w1[src[valid]].to(dtype)
    return out


def gemm_geglu(x, w1_packed, out=None):
    """x bf16 [M, K] @ GEGLU-interleaved w1 [2Fp, K] -> bf16 [M, Fp] = gate * gelu(x-half)."""
    _chk_cuda(x, w1_packed)
    M, K = x.shape
    Fp = w1_packed.shape[0] // 2
    if out is None:
        out = torch.empty(M, Fp, dtype=bf16, device=x.device)
    L.check(L.lib().mm_gemm_geglu(L.stream(), L.ptr(x), x.stride(0), L.ptr(w1_packed), w1_packed.stride(0), M, Fp, K, L.ptr(out),
                                  out.stride(0)), 'mm_gemm_geglu')
    return out


def layernorm_inner(a, F, gamma, beta=None):
    """LayerNorm over the first F of Fp columns of a bf16 [rows, Fp]; padding columns are written as zero."""
    _chk_cuda(a, gamma, beta)
    rows, Fp = a.shape
    out = torch.empty(rows, Fp, dtype=bf16, device=a.device)
    gamma, beta = pad_cols(gamma.float(), Fp), (pad_cols(beta.float(), Fp) if beta is not None else None)
    L.check(L.lib().mm_layernorm_inner(L.stream(), L.ptr(a), a.stride(0), rows, F, Fp, L.ptr(gamma), L.ptr(beta), L.ptr(out), Fp),
            'mm_layernorm_inner')
    return out


def attend(q, k, v, key_mask=None, scale=8.0, normalize=False, q_scale=None, k_scale=None, null_k=None, null_v=None, out_rows=False):
    """q (b,h,n,d), k/v (b,h,j,d) bf16, d = 32 / 64 / 128, with arbitrary batch/head/token strides (d contiguous); key_mask (b,j) bool/uint8.
    out_rows: return the output as [b*n, h*d] rows (heads merged, 'b h n d -> b n (h d)', mmp.py:161) instead of (b,h,n,64)."""
    _chk_cuda(q, k, v, key_mask)
    assert q.dtype == bf16 and k.dtype == bf16 and v.dtype == bf16
    b, h, n, d = q.shape
    j = k.shape[2]
    assert d in (32, 64, 128) and q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1, 'dim_head must be 32, 64 or 128' 
    if out_rows:
        out = torch.empty(b * n, h * d, dtype=bf16, device=q.device)
        ost = (n * h * d, d, h * d)
    else:
        out = torch.empty(b, h, n, d, dtype=bf16, device=q.device)
        ost = (out.stride(0), out.stride(1), out.stride(2))
    km = None
    if key_mask is not None:
        km = key_mask.to(torch.uint8).contiguous()
        assert km.shape == (b, j)
    L.check(L.lib().mm_attend(L.stream(), L.ptr(q), q.stride(0), q.stride(1), q.stride(2), L.ptr(k), k.stride(0), k.stride(1),
                              k.stride(2), L.ptr(v), v.stride(0), v.stride(1), v.stride(2), L.ptr(out), ost[0], ost[1], ost[2],
                              b, h, n, j, L.ptr(km), j, int(normalize), L.ptr(q_scale),
                              L.ptr(k_scale), L.ptr(null_k), L.ptr(null_v), float(scale), d), 'mm_attend')
    return out


def mask_step(scores, ids, k, mask_id, want_rows=True):
    """in place on scores / ids; returns int32 [B*k] flat positions of the masked tokens."""
    _chk_cuda(scores, ids)
    B, n = scores.shape
    assert scores.is_contiguous() and ids.is_contiguous() and ids.dtype == torch.long and scores.dtype == torch.float32
    rows = torch.empty(B * k, dtype=torch.int32, device=scores.device) if want_rows else None
    L.check(L.lib().mm_mask_step(L.stream(), L.ptr(scores), L.ptr(ids), B, n, k, int(mask_id), L.ptr(rows)), 'mm_mask_step')
    return rows


def sample_rows(logits, k_keep, temperature, rows=None, noise_kind=L.MM_NOISE_NONE, noise=None, seed=0, row_offset=0,
                step=0, ids=None, scores=None):
    """logits fp32 [R,V]; returns (pred int64 [R], score fp32 [R]) and scatters into ids/scores when given."""
    _chk_cuda(logits, rows, noise, ids, scores)
    R, V = logits.shape
    pred = torch.empty(R, dtype=torch.long, device=logits.device)
    sc = torch.empty(R, dtype=torch.float32, device=logits.device)
    noise_ld = 0
    if noise is not None:
        assert noise.dtype == torch.float32 and noise.stride(-1) == 1
        noise_ld = noise.shape[-1]
    L.check(L.lib().mm_sample_rows(L.stream(), L.ptr(logits), logits.stride(0), R, V, int(k_keep), L.ptr(rows),
                                   float(temperature), int(noise_kind), L.ptr(noise), noise_ld, int(seed), int(row_offset),
                                   int(step), L.ptr(ids), L.ptr(scores), L.ptr(pred), L.ptr(sc)), 'mm_sample_rows')
    return pred, sc


def ce_loss(logits, labels, ignore_index):
    """logits fp32 [R, V], labels int64 [R] -> scalar fp32 tensor: F.cross_entropy(..., ignore_index), mean reduction."""
    _chk_cuda(logits, labels)
    R, V = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and labels.dtype == torch.long and labels.is_contiguous()
    ws = torch.empty(R, dtype=torch.float32, device=logits.device)
    out = torch.empty(1, dtype=torch.float32, device=logits.device)
    L.check(L.lib().mm_ce_loss(L.stream(), L.ptr(logits), logits.stride(0), R, V, L.ptr(labels), int(ignore_index), L.ptr(ws), L.ptr(out)),
            'mm_ce_loss')
    return out[0]


def bce_loss(x, y):
    _chk_cuda(x, y)
    x, y = x.float().contiguous().reshape(-1), y.float().contiguous().reshape(-1)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    L.check(L.lib().mm_bce_loss(L.stream(), L.ptr(x), L.ptr(y), x.numel(), L.ptr(out)), 'mm_bce_loss')
    return out[0]


def philox_uniform(seed, row_offset, step, rows, V, device):
    L.require_device()
    out = torch.empty(rows, V, dtype=torch.float32, device=device)
    L.check(L.lib().mm_philox_uniform(L.stream(), int(seed), int(row_offset), int(step), rows, V, L.ptr(out)), 'mm_philox_uniform')
    return out


# ------------------------------------------------------------------------------------------------ VAE ops (NHWC bf16)

def conv2d_nhwc(x, w_packed, cout, th, tw, stride=1, off=(0, 0), out_hw=None, os_=1, parity=(0, 0), full_hw=None, bias=None,
                act=False, resid=None, out=None, out_nchw_f32=False):
    """x bf16 [B,H,W,Cin]; w_packed bf16 [Cout, Kp] (see pack_conv_weight).  Returns NHWC bf16 (or NCHW fp32)."""
    _chk_cuda(x, w_packed, bias, resid, out)
    B, H, W, Cin = x.shape
    Hv, Wv = out_hw if out_hw is not None else (H, W)
    Hout, Wout = full_hw if full_hw is not None else (Hv * os_, Wv * os_)
    if out is None:
        if out_nchw_f32:
            out = torch.empty(B, cout, Hout, Wout, dtype=torch.float32, device=x.device)
        else:
            out = torch.empty(B, Hout, Wout, cout, dtype=bf16, device=x.device)
    L.check(L.lib().mm_conv2d_nhwc(L.stream(), L.ptr(x), B, H, W, Cin, L.ptr(w_packed), cout, th, tw, stride, off[0], off[1],
                                   Hv, Wv, os_, parity[0], parity[1], Hout, Wout, L.ptr(bias), int(act), L.ptr(resid),
                                   L.ptr(out), int(out_nchw_f32)), 'mm_conv2d_nhwc')
    return out


def conv2d_nhwc_half(x, w_packed, cout, th, tw, stride=1, off=(0, 0), out_hw=None, os_=1, parity=(0, 0), full_hw=None, bias=None,
                     act=False, resid=None, out=None, out_nchw_f32=False, alpha=1.0):
    """mm_conv2d_nhwc_half (round 6): x fp16 [B,H,W,Cin]; w_packed fp16 [Cout, Kp] = pack_conv_weight(w, torch.float16, 1 / alpha); single fp16 terms on the
    fp16 MFMA, fp32 accumulation.  Returns NHWC fp16 (resid NHWC fp16) or NCHW fp32."""
    _chk_cuda(x, w_packed, bias, resid, out)
    f16 = torch.float16
    assert x.dtype == f16 and w_packed.dtype == f16 and (resid is None or resid.dtype == f16)
    B, H, W, Cin = x.shape
    Hv, Wv = out_hw if out_hw is not None else (H, W)
    Hout, Wout = full_hw if full_hw is not None else (Hv * os_, Wv * os_)
    if out is None:
        out = (torch.empty(B, cout, Hout, Wout, dtype=torch.float32, device=x.device) if out_nchw_f32
               else torch.empty(B, Hout, Wout, cout, dtype=f16, device=x.device))
    L.check(L.lib().mm_conv2d_nhwc_half(L.stream(), L.ptr(x), B, H, W, Cin, L.ptr(w_packed), cout, th, tw, stride, off[0], off[1],
                                        Hv, Wv, os_, parity[0], parity[1], Hout, Wout, L.ptr(bias), int(act), L.ptr(resid),
                                        L.ptr(out), int(out_nchw_f32), float(alpha)), 'mm_conv2d_nhwc_half')
    return out


def glu_nhwc(x):
    _chk_cuda(x)
    C2 = x.shape[-1]
    rows = x.numel() // C2
    out = torch.empty(*x.shape[:-1], C2 // 2, dtype=bf16, device=x.device)
    L.check(L.lib().mm_glu_nhwc(L.stream(), L.ptr(x), rows, C2 // 2, L.ptr(out)), 'mm_glu_nhwc')
    return out


def groupnorm_nhwc(x, groups, gamma, beta, act=False):
    _chk_cuda(x, gamma, beta)
    B, H, W, Cc = x.shape
    out = torch.empty_like(x)
    ws = torch.empty(B * groups * 2, dtype=torch.float32, device=x.device)
    L.check(L.lib().mm_groupnorm_nhwc(L.stream(), L.ptr(x), B, H * W, Cc, groups, L.ptr(gamma), L.ptr(beta), int(act), L.ptr(ws),
                                      L.ptr(out)), 'mm_groupnorm_nhwc')
    return out


def lfq_decode(ids, bits, C, w=None, b=None):
    _chk_cuda(ids, w, b)
    ids = ids.contiguous()
    out = torch.empty(*ids.shape, C, dtype=bf16, device=ids.device)
    L.check(L.lib().mm_lfq_decode(L.stream(), L.ptr(ids), ids.numel(), bits, C, L.ptr(w), L.ptr(b), L.ptr(out)), 'mm_lfq_decode')
    return out


def lfq_encode(x, bits, w_in=None, b_in=None, w_out=None, b_out=None):
    """x bf16 [..., C] -> (ids int64 [...], quantized bf16 [..., C])."""
    _chk_cuda(x)
    C = x.shape[-1]
    count = x.numel() // C
    ids = torch.empty(x.shape[:-1], dtype=torch.long, device=x.device)
    out = torch.empty_like(x)
    L.check(L.lib().mm_lfq_encode(L.stream(), L.ptr(x), count, C, bits, L.ptr(w_in), L.ptr(b_in), L.ptr(w_out), L.ptr(b_out),
                                  L.ptr(ids), L.ptr(out)), 'mm_lfq_encode')
    return ids, out


def nchw_to_nhwc8(img):
    _chk_cuda(img)
    B, Cc, H, W = img.shape
    out = torch.empty(B, H, W, 8, dtype=bf16, device=img.device)
    L.check(L.lib().mm_nchw_f32_to_nhwc8_bf16(L.stream(), L.ptr(img.contiguous().float()), B, Cc, H, W, L.ptr(out)), 'nchw_to_nhwc8')
    return out


def nhwc_to_nchw_f32(x):
    _chk_cuda(x)
    B, H, W, Cc = x.shape
    out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=x.device)
    L.check(L.lib().mm_nhwc_bf16_to_nchw_f32(L.stream(), L.ptr(x), B, Cc, H, W, L.ptr(out)), 'nhwc_to_nchw')
    return out


# ------------------------------------------------------------------------------------------------ weight packing (host side, once)

def pack_conv_weight(w, dtype=bf16, scale=1.0):
    """Conv2d weight [Cout, Cin, TH, TW] -> bf16 [Cout, Kp], k = (ty*TW + tx)*Cin + ci, Kp = ceil64(TH*TW*Cin).
    dtype=torch.float16, scale=2^j (round 6): the single-term fp16 pack of the half-precision VAE decode (the kernels multiply their accumulators by 1 / scale)."""
    cout = w.shape[0]
    return pad_cols((w.permute(0, 2, 3, 1).reshape(cout, -1).float() * float(scale)).to(dtype), 64)


def pack_conv_weight_cin8(w):
    """first conv (Cin = image channels <= 8): pad Cin to 8 to match the NHWC8 image layout."""
    cout, cin, th, tw = w.shape
    wp = torch.zeros(cout, 8, th, tw, dtype=w.dtype, device=w.device)
    wp[:, :cin] = w
    return pack_conv_weight(wp)


def pack_convT_weight(w, dtype=bf16, scale=1.0):
    """ConvTranspose2d(4,2,1) weight [Cin, Cout, 4, 4] -> four bf16 [Cout, Kp] matrices, one per output parity (py,px):
    out[2y+py, 2x+px] = sum_{ty,tx in 0..1} in[y+ty-1+py, x+tx-1+px] . w[:, :, 3-py-2ty, 3-px-2tx]."""
    packs = {}
    for py in range(2):
        for px in range(2):
            taps = []
            for ty in range(2):
                for tx in range(2):
                    taps.append(w[:, :, 3 - py - 2 * ty, 3 - px - 2 * tx].t())      # [Cout, Cin]
            packs[(py, px)] = pad_cols((torch.cat(taps, dim=1).float() * float(scale)).to(dtype), 64)
    return packs


def mask_counts(timesteps, seq_len):
    """muse_maskgit_pytorch.py:556-559 evaluated on the host once (the reference syncs the device every step for this):
    same fp32 linspace / cos, python int() truncation, max(., 1)."""
    out = []
    for t in torch.linspace(0, 1, timesteps):
        out.append(max(int((torch.cos(t * math.pi * 0.5) * seq_len).item()), 1))
    return out


def step_temperatures(timesteps, temperature):
    """muse_maskgit_pytorch.py:578 + the clamp of :411."""
    return [max(temperature * (s / timesteps), 1e-10) for s in reversed(range(timesteps))]


# ------------------------------------------------------------------------------------------------ backward operators (training)
def transpose(x, pad_to=None):
    """bf16 [R,C] (row stride % 8 == 0) -> [C,R].  pad_to: return the contiguous [C, Rp] buffer instead, Rp = R rounded up to
    `pad_to`, padding columns zero (a GEMM contraction dimension)."""
    _chk_cuda(x)
    assert x.dtype == bf16 and x.dim() == 2 and x.stride(1) == 1
    R, C = x.shape
    mult = pad_to or 8
    Rp = (R + mult - 1) // mult * mult
    out = torch.empty(C, Rp, dtype=bf16, device=x.device)
    if Rp != R:
        out.zero_()
    L.check(L.lib().mm_transpose_bf16(L.stream(), L.ptr(x), R, C, x.stride(0), L.ptr(out), Rp), 'mm_transpose_bf16')
    return out if pad_to else out[:, :R]


def gemm_wgrad(xt, wt):
    """fp32 [M, N] = xt [M, K] @ wt [N, K]^T for a weight gradient (K = padded token count): split-K over the contraction."""
    _chk_cuda(xt, wt)
    M, K = xt.shape
    N = wt.shape[0]
    assert wt.shape[1] == K and xt.dtype == bf16 and wt.dtype == bf16
    lib = L.lib()
    splits = lib.mm_gemm_wgrad_splits(M, N, K) if N % 4 == 0 else 1
    if splits <= 1:
        return gemm(xt, wt, out_f32=True)
    ws = torch.empty(splits, M, N, dtype=torch.float32, device=xt.device)
    out = torch.empty(M, N, dtype=torch.float32, device=xt.device)
    L.check(lib.mm_gemm_wgrad(L.stream(), L.ptr(xt), xt.stride(0), L.ptr(wt), wt.stride(0), M, N, K, splits, L.ptr(ws), L.ptr(out)), 'mm_gemm_wgrad')
    return out


def gemm_wgrad_tn(dy, x):
    """fp32 [N, K] = dy^T @ x for row-major bf16 dy [rows, N], x [rows, K] (strided rows allowed): no transposed copies (csrc/gemm_tn.hip).  N, K multiples of 128."""
    _chk_cuda(dy, x)
    rows, N = dy.shape
    K = x.shape[1]
    assert x.shape[0] == rows and dy.dtype == bf16 and x.dtype == bf16 and dy.stride(1) == 1 and x.stride(1) == 1
    lib = L.lib()
    splits = lib.mm_gemm_wgrad_tn_splits(rows, N, K)
    ws = torch.empty(splits, N, K, dtype=torch.float32, device=dy.device) if splits > 1 else None
    out = torch.empty(N, K, dtype=torch.float32, device=dy.device)
    L.check(lib.mm_gemm_wgrad_tn(L.stream(), L.ptr(dy), dy.stride(0), L.ptr(x), x.stride(0), rows, N, K, L.ptr(ws), L.ptr(out)), 'mm_gemm_wgrad_tn')
    return out


def to_bf16(x):
    _chk_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(x.shape, dtype=bf16, device=x.device)
    L.check(L.lib().mm_f32_to_bf16(L.stream(), L.ptr(x), L.ptr(out), x.numel()), 'mm_f32_to_bf16')
    return out


def colsum(part):
    """fp32 [P, D] -> [D], rows added in index order."""
    _chk_cuda(part)
    assert part.dtype == torch.float32 and part.is_contiguous()
    P, D = part.shape
    out = torch.empty(D, dtype=torch.float32, device=part.device)
    L.check(L.lib().mm_colsum_f32(L.stream(), L.ptr(part), P, D, L.ptr(out)), 'mm_colsum_f32')
    return out


def layernorm_bwd(x, dy, gamma, dx, accumulate=True, row_index=None):
    """dx (fp32 [M,D], in place) += / = LayerNorm backward; returns dgamma fp32 [D]."""
    _chk_cuda(x, dy, gamma, dx, row_index)
    rows, D = dy.shape
    assert x.dtype == torch.float32 and dy.dtype == bf16 and dx.dtype == torch.float32 and x.stride(1) == 1 and dy.stride(1) == 1
    ws = torch.empty(L.lib().mm_ln_bwd_workspace_floats(rows, D), dtype=torch.float32, device=x.device)
    dg = torch.empty(D, dtype=torch.float32, device=x.device)
    L.check(L.lib().mm_layernorm_bwd(L.stream(), L.ptr(x), x.stride(0), L.ptr(dy), dy.stride(0), L.ptr(gamma), L.ptr(row_index), rows, D,
                                     L.ptr(dx), dx.stride(0), int(accumulate), L.ptr(dg), L.ptr(ws)), 'mm_layernorm_bwd')
    return dg


def geglu_ln_bwd(h, dz, F, gamma):
    """h bf16 [rows, 2Fp] = [x | gate], dz bf16 [rows, Fp]; returns (dh bf16 [rows, 2Fp], dgamma fp32 [F])."""
    _chk_cuda(h, dz, gamma)
    rows, two_fp = h.shape
    Fp = two_fp // 2
    gp = pad_cols(gamma.float(), Fp)
    dh = torch.empty(rows, two_fp, dtype=bf16, device=h.device)
    ws = torch.empty(L.lib().mm_ln_bwd_workspace_floats(rows, Fp), dtype=torch.float32, device=h.device)
    dg = torch.empty(Fp, dtype=torch.float32, device=h.device)
    L.check(L.lib().mm_geglu_ln_bwd(L.stream(), L.ptr(h), h.stride(0), L.ptr(dz), dz.stride(0), L.ptr(gp), rows, F, Fp, L.ptr(dh), two_fp,
                                    L.ptr(dg), L.ptr(ws)), 'mm_geglu_ln_bwd')
    return dh, dg[:F]


def ce_bwd(logits, labels, scale, pad_to=1):
    """d(mean CE)/d(logits) as bf16 [R, Vp]: Vp = V rounded up to `pad_to`, the padding columns zero (a GEMM contraction dim downstream)."""
    _chk_cuda(logits, labels)
    R, V = logits.shape
    assert logits.dtype == torch.float32 and labels.dtype == torch.long and logits.stride(1) == 1
    Vp = (V + pad_to - 1) // pad_to * pad_to
    dl = torch.empty(R, Vp, dtype=bf16, device=logits.device) if Vp == V else torch.zeros(R, Vp, dtype=bf16, device=logits.device)
    L.check(L.lib().mm_ce_bwd(L.stream(), L.ptr(logits), logits.stride(0), R, V, L.ptr(labels.contiguous()), float(scale), L.ptr(dl), Vp), 'mm_ce_bwd')
    return dl


def bce_head_bwd(e, x, y, w):
    """TokenCritic head + BCE backward: e bf16 [rows, D], x / y fp32 [rows], w fp32 [D] -> (de bf16 [rows, D], dw fp32 [D])."""
    _chk_cuda(e, x, y, w)
    rows, D = e.shape
    de = torch.empty(rows, D, dtype=bf16, device=e.device)
    dw = torch.empty(D, dtype=torch.float32, device=e.device)
    ws = torch.empty(L.lib().mm_ln_bwd_workspace_floats(rows, D), dtype=torch.float32, device=e.device)
    L.check(L.lib().mm_bce_head_bwd(L.stream(), L.ptr(e), e.stride(0), L.ptr(x.contiguous()), L.ptr(y.contiguous()), L.ptr(w.contiguous()), rows, D,
                                    L.ptr(de), D, L.ptr(dw), L.ptr(ws)), 'mm_bce_head_bwd')
    return de, dw


def embed_bwd(ids, dx, table_rows, dtoken=None, two_level=True):
    """ids int64 [B,n], dx fp32 [B*n, D] -> (dtoken fp32 [table_rows, D], dpos fp32 [n, D]); dtoken: accumulate into this table.
    two_level=False: the workspace-free single chain per id (same sum, another association)."""
    _chk_cuda(ids, dx, dtoken)
    B, n = ids.shape
    D = dx.shape[1]
    assert dx.is_contiguous() and dx.dtype == torch.float32
    dtok = dtoken if dtoken is not None else torch.zeros(table_rows, D, dtype=torch.float32, device=dx.device)
    dpos = torch.empty(n, D, dtype=torch.float32, device=dx.device)
    # (round 6) the two-level sum: per 256-row block, then over the blocks -- the mask id of a training batch owns most rows, one serial chain was 0.4 ms
    wsb = L.lib().mm_embed_bwd_workspace_bytes(B, n, D) if two_level else 0
    ws = torch.empty(int(wsb), dtype=torch.uint8, device=dx.device) if two_level else None
    L.check(L.lib().mm_embed_bwd(L.stream(), L.ptr(ids.contiguous()), B, n, D, L.ptr(dx), L.ptr(dtok), L.ptr(dpos), L.ptr(ws), wsb), 'mm_embed_bwd')
    return dtok, dpos


def scatter_rows(src, row_index, M):
    _chk_cuda(src, row_index)
    R, D = src.shape
    dst = torch.zeros(M, D, dtype=bf16, device=src.device)
    L.check(L.lib().mm_scatter_rows_bf16(L.stream(), L.ptr(src.contiguous()), L.ptr(row_index), R, D, L.ptr(dst)), 'mm_scatter_rows_bf16')
    return dst


def attention_bwd(q, k, v, o, dout, q_scale, k_scale, null_k, null_v, key_mask=None, scale=8.0):
    """Muse attention backward.  q/o/dout (b,h,n,64), k/v (b,h,j,64) bf16 strided views.  Returns dqn (b,n,h*64), dkn, dv (b,j,h*64)
    bf16 -- gradients w.r.t. the normalised q / k and v -- and dnk, dnv fp32 (b*h, 64) for the null key / value.
    One kernel call per block of <= 256 queries (blocks shorter than the kernel's 64 / 128 / 256 are padded), the blocks' partial key /
    value gradients are summed afterwards."""
    _chk_cuda(q, k, v, o, dout, key_mask)
    b, h, n, d = q.shape
    j = k.shape[2]
    dev = q.device
    jj = max(j, 1)
    chunks = (n + 255) // 256
    dqn = torch.empty(b, n, h * 64, dtype=bf16, device=dev)
    dkn = torch.empty(chunks, b, jj, h * 64, dtype=bf16, device=dev)
    dv = torch.empty(chunks, b, jj, h * 64, dtype=bf16, device=dev)
    dnk = torch.empty(chunks, b * h, 64, dtype=torch.float32, device=dev)
    dnv = torch.empty(chunks, b * h, 64, dtype=torch.float32, device=dev)
    km = key_mask.to(torch.uint8).contiguous() if key_mask is not None else None

    def st(t):
        return (L.ptr(t), t.stride(0), t.stride(1), t.stride(2))

    for c in range(chunks):
        lo = c * 256
        nq = min(256, n - lo)
        nqp = 64 if nq <= 64 else (128 if nq <= 128 else 256)        # the kernel's query-block sizes
        qs = slice(lo, lo + nq)
        qc, oc, dc, dq_c, dq_sb = q[:, :, qs], o[:, :, qs], dout[:, :, qs], dqn[:, qs], n * h * 64
        if nqp != nq:
            # pad the query block: the extra queries repeat query 0 with a zero output gradient, so dS = P * (dP - delta) is exactly 0 for
            # them and they add nothing to the key / value gradients; their own dq rows are dropped
            def padq(t, zero=False):
                out = torch.zeros(b, h, nqp, 64, dtype=bf16, device=dev) if zero else t[:, :, :1].expand(b, h, nqp, 64).contiguous()
                out[:, :, :nq] = t
                return out
            qc, oc, dc = padq(qc), padq(oc), padq(dc, zero=True)
            dq_c = torch.empty(b, nqp, h * 64, dtype=bf16, device=dev)
            dq_sb = nqp * h * 64
        L.check(L.lib().mm_attention_bwd(L.stream(), *st(qc), *st(k), *st(v), *st(oc), *st(dc),
                                         L.ptr(dq_c), dq_sb, 64, h * 64, L.ptr(dkn[c]), jj * h * 64, 64, h * 64,
                                         L.ptr(dv[c]), jj * h * 64, 64, h * 64, L.ptr(dnk[c]), L.ptr(dnv[c]), b, h, nqp, j, L.ptr(km), j,
                                         L.ptr(q_scale), L.ptr(k_scale), L.ptr(null_k), L.ptr(null_v), float(scale)), 'mm_attention_bwd')
        if nqp != nq:
            dqn[:, qs] = dq_c[:, :nq]
    if chunks == 1:
        return dqn, dkn[0][:, :j], dv[0][:, :j], dnk[0], dnv[0]
    per = b * jj * h * 64
    dkn_s = torch.empty(b, jj, h * 64, dtype=bf16, device=dev)
    dv_s = torch.empty(b, jj, h * 64, dtype=bf16, device=dev)
    L.check(L.lib().mm_sum_parts_bf16(L.stream(), L.ptr(dkn), chunks, per, L.ptr(dkn_s)), 'mm_sum_parts_bf16')
    L.check(L.lib().mm_sum_parts_bf16(L.stream(), L.ptr(dv), chunks, per, L.ptr(dv_s)), 'mm_sum_parts_bf16')
    return (dqn, dkn_s[:, :j], dv_s[:, :j], colsum(dnk.reshape(chunks, b * h * 64)).reshape(b * h, 64),
            colsum(dnv.reshape(chunks, b * h * 64)).reshape(b * h, 64))


def qk_norm_bwd(x, dy, scale, heads, x_f32=None, dy_f32=None):
    """l2norm * scale backward on `heads` vectors of 64 per row.  bf16 form: x, dy [rows, >= heads*64] -> (dx bf16 [rows, heads*64],
    dscale fp32 [64]).  fp32 form (null key): x_f32 [H, 64] broadcast over dy_f32 [rows*H, 64] -> (dx fp32 [rows*H, 64], dscale)."""
    lib = L.lib()
    if x_f32 is not None:
        _chk_cuda(x_f32, dy_f32, scale)
        nvec = dy_f32.shape[0]
        H = x_f32.shape[0]
        dx = torch.empty(nvec, 64, dtype=torch.float32, device=dy_f32.device)
        part = torch.empty(lib.mm_qk_norm_bwd_blocks(nvec), 64, dtype=torch.float32, device=dy_f32.device)
        L.check(lib.mm_qk_norm_bwd(L.stream(), None, 0, L.ptr(x_f32.contiguous()), H, None, 0, L.ptr(dy_f32), L.ptr(scale), nvec, 1, None, 0,
                                   L.ptr(dx), L.ptr(part)), 'mm_qk_norm_bwd')
        return dx, colsum(part)
    _chk_cuda(x, dy, scale)
    rows = x.shape[0]
    dx = torch.empty(rows, heads * 64, dtype=bf16, device=x.device)
    part = torch.empty(lib.mm_qk_norm_bwd_blocks(rows * heads), 64, dtype=torch.float32, device=x.device)
    L.check(lib.mm_qk_norm_bwd(L.stream(), L.ptr(x), x.stride(0), None, heads, L.ptr(dy), dy.stride(0), None, L.ptr(scale), rows, heads,
                               L.ptr(dx), heads * 64, None, L.ptr(part)), 'mm_qk_norm_bwd')
    return dx, colsum(part)


# ------------------------------------------------------------------------------------------------ vector quantisation (extension)
def vq_nearest(x, codebook, cosine=False):
    """x fp32 [N, C], codebook fp32 [K, C] -> int64 [N]: nearest code (L2), or most cosine-similar code.  Ties -> lower index."""
    _chk_cuda(x, codebook)
    assert x.dtype == torch.float32 and codebook.dtype == torch.float32 and x.stride(1) == 1 and codebook.is_contiguous()
    N, C = x.shape
    K = codebook.shape[0]
    ids = torch.empty(N, dtype=torch.long, device=x.device)
    aux = torch.empty(K, dtype=torch.float32, device=x.device)
    L.check(L.lib().mm_vq_nearest(L.stream(), L.ptr(x), x.stride(0), N, C, L.ptr(codebook), K, int(cosine), L.ptr(aux), L.ptr(ids)), 'mm_vq_nearest')
    return ids


def vq_gather(ids, codebook):
    _chk_cuda(ids, codebook)
    N, C = ids.numel(), codebook.shape[1]
    out = torch.empty(N, C, dtype=torch.float32, device=codebook.device)
    L.check(L.lib().mm_vq_gather(L.stream(), L.ptr(ids.reshape(-1).contiguous()), N, C, L.ptr(codebook.contiguous()), L.ptr(out)), 'mm_vq_gather')
    return out.reshape(*ids.shape, C)


# ------------------------------------------------------------------------------------------------ fp8 weights (W8A16)
def quantize_e4m3_rows(w):
    """fp32/bf16 [N, K] -> (wq uint8 [N, Kp] OCP-e4m3 bytes, Kp = K rounded up to 128 -- the fp8 MFMA's k-step --, zero padded; scale fp32 [N])."""
    _chk_cuda(w)
    w = w.detach().float().contiguous()
    N, K = w.shape
    Kp = (K + 127) // 128 * 128
    wq = torch.empty(N, Kp, dtype=torch.uint8, device=w.device)
    scale = torch.empty(N, dtype=torch.float32, device=w.device)
    L.check(L.lib().mm_quantize_e4m3_rows(L.stream(), L.ptr(w), K, N, K, Kp, L.ptr(wq), L.ptr(scale)), 'mm_quantize_e4m3_rows')
    return wq, scale


def quantize_act_e4m3(x, Kp=None):
    """activation rows (bf16 or fp32 [M, K]) -> (xq uint8 [M, Kp] OCP-e4m3 bytes, scale fp32 [M]); Kp defaults to K rounded up to 128"""
    _chk_cuda(x)
    x = x.contiguous()
    M, K = x.shape
    Kp = Kp or (K + 127) // 128 * 128
    if x.dtype == bf16 and K % 8:          # bf16 rows are read 16 bytes at a time
        x = pad_cols(x, 8)
    xq = torch.empty(M, Kp, dtype=torch.uint8, device=x.device)
    scale = torch.empty(M, dtype=torch.float32, device=x.device)
    assert x.dtype in (bf16, torch.float32)
    L.check(L.lib().mm_quantize_act_e4m3(L.stream(), L.ptr(x), int(x.dtype == torch.float32), x.stride(0), M, K, Kp, L.ptr(xq), L.ptr(scale)), 'mm_quantize_act_e4m3')
    return xq, scale


def gemm_fp8(xq, x_scale, wq, w_scale, epilogue=0, resid=None):
    """e4m3 x e4m3 on the K = 128 fp8 MFMA: out[m][n] = x_scale[m] * w_scale[n] * sum_k xq[m][k] wq[n][k].  epilogue 0: bf16 [M, N]; 1: GEGLU over
    w1 rows interleaved in 64-row blocks (32 values | 32 gates), bf16 [M, N / 2]; 2: fp32 [M, N] (+ resid)"""
    _chk_cuda(xq, x_scale, wq, w_scale, resid)
    M, K = xq.shape
    N = wq.shape[0]
    assert xq.dtype == torch.uint8 and wq.dtype == torch.uint8 and wq.shape[1] == K and K % 128 == 0
    oc = N // 2 if epilogue == 1 else N
    out = torch.empty(M, oc, dtype=torch.float32 if epilogue == 2 else bf16, device=xq.device)
    L.check(L.lib().mm_gemm_fp8(L.stream(), L.ptr(xq), xq.stride(0), L.ptr(x_scale), L.ptr(wq), wq.stride(0), L.ptr(w_scale), M, N, K, L.ptr(out), out.stride(0),
                                int(epilogue), L.ptr(resid)), 'mm_gemm_fp8')
    return out


# ------------------------------------------------------------------------------------------------ fused sampling (no logits round trip)
FUSED_SLOT = 64


def fused_buffers(R, V, device):
    """caller-owned scratch of the fused sampling path for R rows of a V-entry vocabulary (include/muse_hip.h, mm_fused_*): per (row, 256-column piece) a
    record of 2 float4s ({max, sum exp, -, -} and the 128-bit mask of the kept 2-column granules) and a slot of up to 128 float2 candidates"""
    NT = V // 256
    return dict(stats=torch.empty(R, NT, 8, dtype=torch.float32, device=device), cand=torch.empty(R, NT, FUSED_SLOT, 4, dtype=torch.float32, device=device),
                fail=torch.zeros(1, dtype=torch.int32, device=device))


def fused_quantile(sub, k_keep, V):
    """thr[r] = the (S k / V + 4.5 sigma)-th largest of the sampled logits sub fp32 [R][S] (the distribution-free bound of the fused sampler)"""
    sub = sub.contiguous()
    R, S = sub.shape
    rank = L.lib().mm_fused_quantile_rank(int(k_keep), int(V), int(S))
    thr = torch.empty(R, dtype=torch.float32, device=sub.device)
    L.check(L.lib().mm_fused_quantile(L.stream(), L.ptr(sub), S, R, S, rank, L.ptr(thr)), 'mm_fused_quantile')
    return thr, rank


def fused_z(k_keep, V, margin=0.20):
    return float(L.lib().mm_fused_z(int(k_keep), int(V), float(margin)))


def fused_threshold(emb_cond, emb_null, cond_scale, wmean, wcov_bf16, z):
    _chk_cuda(emb_cond, emb_null, wmean, wcov_bf16)
    assert wcov_bf16.dtype == bf16 and wmean.dtype == torch.float32
    R, D = emb_cond.shape
    thr = torch.empty(R, dtype=torch.float32, device=emb_cond.device)
    ws = torch.empty(int(L.lib().mm_fused_threshold_workspace_bytes(R, D)), dtype=torch.uint8, device=emb_cond.device)
    L.check(L.lib().mm_fused_threshold(L.stream(), L.ptr(emb_cond), L.ptr(emb_null), emb_cond.stride(0), R, D, float(cond_scale), L.ptr(wmean), L.ptr(wcov_bf16),
                                       float(z), L.ptr(ws), L.ptr(thr)), 'mm_fused_threshold')
    return thr


def gemm_cfg_logits_fused(x_cond, x_null, w, cond_scale, thr, fb):
    """x_null None: x_cond are mixed embeddings (cfg_mix), one pass"""
    _chk_cuda(x_cond, x_null, w, thr)
    M, K = x_cond.shape
    N = w.shape[0]
    L.check(L.lib().mm_gemm_cfg_logits_fused(L.stream(), L.ptr(x_cond), L.ptr(x_null), x_cond.stride(0), L.ptr(w), w.stride(0), M, N, K, float(cond_scale),
                                             L.ptr(thr), L.ptr(fb['stats']), L.ptr(fb['cand'])), 'mm_gemm_cfg_logits_fused')


def fused_emit(logits, thr, fb):
    _chk_cuda(logits, thr)
    R, V = logits.shape
    L.check(L.lib().mm_fused_emit(L.stream(), L.ptr(logits), logits.stride(0), R, V, L.ptr(thr), L.ptr(fb['stats']), L.ptr(fb['cand'])), 'mm_fused_emit')


def fused_sample(fb, thr, R, V, k_keep, temperature, rows=None, noise_kind=L.MM_NOISE_NONE, noise=None, seed=0, row_offset=0, step=0, fail_list=None):
    """-> (pred int64 [R], score fp32 [R]); fb['fail'] is set to 1 if a row's candidates could not be proven complete -- or, with
    fail_list = (rows int32 [cap], count int32 [1]), such rows are appended to the list instead (entries of pred / score undefined for them)"""
    dev = fb['stats'].device
    pred = torch.empty(R, dtype=torch.long, device=dev)
    score = torch.empty(R, dtype=torch.float32, device=dev)
    fr, fc = fail_list if fail_list is not None else (None, None)
    L.check(L.lib().mm_fused_sample(L.stream(), L.ptr(thr), L.ptr(fb['stats']), L.ptr(fb['cand']), R, V, int(k_keep), L.ptr(rows), float(temperature),
                                    int(noise_kind), L.ptr(noise), noise.stride(0) if noise is not None else 0, int(seed), int(row_offset), int(step), None, None,
                                    L.ptr(pred), L.ptr(score), L.ptr(fb['fail']), L.ptr(fr), L.ptr(fc), fr.numel() if fr is not None else 0), 'mm_fused_sample')
    return pred, score


# ---- precision tier 'bf16x3' (csrc/split.hip): fp32 values as sums of bf16 terms
SPLIT_W_SEGMENTS = (0, 0, 0, 1, 1, 2)      # which weight term (0 = h, 1 = m, 2 = l) each segment of W' carries; X' carries [h m l h m h]


def split_terms(w):
    """The exact three-term bf16 split of an fp32 tensor: w = h + m + l (h = bf16(w), m = bf16(w - h), l = w - h - m).  Layout / dtype
    preparation of weights at pack time (plain torch casts); activations are split by the kernels of csrc/split.hip."""
    w = w.detach().float()
    h = w.to(bf16)
    r = w - h.float()
    m = r.to(bf16)
    l = (r - m.float()).to(bf16)
    return h, m, l


def weight_terms(w):
    """1 if every value of w is bf16-representable, 2 if two bf16 terms are exact, else 3."""
    h, m, l = split_terms(w)
    if not bool(m.float().abs().max() > 0):
        return 1
    return 2 if not bool(l.float().abs().max() > 0) else 3


def products_for_terms(terms):
    """term pairs (activation term i, weight term j) with i + j <= 2 that exist for a weight of `terms` bf16 terms: 3, 5 or 6"""
    return {1: 3, 2: 5, 3: 6}[int(terms)]


# ---- precision tier 'f16x2' (round 4): the same engine on fp16 terms.  x ~ h + l (11 + 11 significand bits), products h.h + l.h (+ h.l): X' = [xh|xl|xh][:P]
#      against W' = [wh|wh|wl][:P] -- P = 3 for general fp32 weights, 2 when one fp16 term holds every weight.  Operand code: MM_SPLIT_F16 | P.
MM_SPLIT_F16 = 0x100
MM_SPLIT_SHARED = 0x200      # mm_gemm_split: the operands are genuine term-segment packs -> every term plane may be staged once (csrc/gemm_terms.hip)
SPLIT_W_SEGMENTS_F16 = (0, 0, 1)


def split_count(products):
    """segments per operand row of an operand code (3 / 5 / 6 bf16 terms, MM_SPLIT_F16 | 2 / 3 fp16 terms)"""
    return int(products) & 0xff


def split_is_f16(products):
    return bool(int(products) & MM_SPLIT_F16)


def split_terms_f16(w, scale=1.0):
    """two fp16 terms of scale * w (scale a power of two: exact): h = fp16(v), l = fp16(v - h); returned as fp16 tensors"""
    v = w.detach().float() * float(scale)
    h = v.to(torch.float16)
    assert bool(torch.isfinite(h.float()).all()), 'fp16 overflow: the weight scale is too large'
    l = (v - h.float()).to(torch.float16)
    return h, l


def f16_weight_scale(weights):
    """the power of two that puts the largest |w| of `weights` into [2^13, 2^14): every low term of a weight within 2^-10 of the largest is then a
    NORMAL fp16 number, and the absolute floor of the two-term split (2^-25 after scaling) is 2^-38 of the largest weight.  The GEMMs multiply
    their accumulators by the inverse (mm_transformer_desc.split_alpha / GemmArgs.alpha): exact."""
    import math
    mx = max(float(w.detach().abs().max()) for w in weights)
    if not mx > 0. or not math.isfinite(mx):
        return 1.0
    return 2.0 ** (13 - math.floor(math.log2(mx)))


def weight_terms_f16(w, scale=1.0):
    """1 if one fp16 term holds scale * w to within 2^-24 of its largest value (any bf16-representable checkpoint: 8 significant bits), else 2"""
    v = w.detach().float() * float(scale)
    r = (v - v.to(torch.float16).float()).abs().max()
    return 1 if float(r) <= float(v.abs().max()) * 2.0 ** -24 else 2


def split_pack_weight(w, products, pad_k=1, scale=1.0):
    """nn.Linear weight fp32 [N][K] -> bf16 [N][products * Kp], the segments [wh|wh|wh|wm|wm|wl][:products] (each zero-padded to Kp =
    K rounded up to pad_k); multiplies the activation pack [xh|xm|xl|xh|xm|xh] written by the split kernels.
    products = MM_SPLIT_F16 | 2 / 3: fp16 terms of scale * w, segments [wh|wh|wl][:P] (returned as a 16-bit container of dtype bfloat16)."""
    N, K = w.shape
    Kp = (K + pad_k - 1) // pad_k * pad_k
    P = split_count(products)
    if split_is_f16(products):
        terms = split_terms_f16(w, scale)
        out = torch.zeros(N, P, Kp, dtype=torch.float16, device=w.device)
        for s in range(P):
            out[:, s, :K] = terms[SPLIT_W_SEGMENTS_F16[s]]
        return out.reshape(N, P * Kp).contiguous().view(bf16)
    assert scale == 1.0
    terms = split_terms(w)
    out = torch.zeros(N, products, Kp, dtype=bf16, device=w.device)
    for s in range(products):
        out[:, s, :K] = terms[SPLIT_W_SEGMENTS[s]]
    return out.reshape(N, products * Kp).contiguous()


def gemm_split(xs, ws, products, alpha=1.0, resid=None, shared=False):
    """fp32 [M][N] = X' . W'^T over term-segment packs (mm_gemm_split): xs / ws from split_rows / split_pack_weight with the same operand code.
    shared=True (fp16 terms): the packs are genuine ([xh|xl|xh] / [wh|wh|wl]) -- the term-sharing kernels may read every term plane once (MM_SPLIT_SHARED)"""
    _chk_cuda(xs, ws)
    M, K = xs.shape
    N = ws.shape[0]
    assert ws.shape[1] == K and xs.stride(1) == 1 and ws.stride(1) == 1
    ldc = (N + 3) // 4 * 4 if resid is None else N      # (fp32 rows are written 16 bytes at a time)
    out = torch.empty(M, ldc, dtype=torch.float32, device=xs.device)
    code = int(products) | (MM_SPLIT_SHARED if shared and split_is_f16(products) else 0)
    L.check(L.lib().mm_gemm_split(L.stream(), L.ptr(xs), xs.stride(0), L.ptr(ws), ws.stride(0), M, N, K, code, float(alpha), L.ptr(out), ldc,
                                  L.ptr(resid)), 'mm_gemm_split')
    return out[:, :N]


def gemm_split_geglu(xs, w1s, products, alpha=1.0, stats=True):
    """FF w1 of the 'f16x2' tier (mm_gemm_split_geglu, csrc/gemm_terms.hip): xs term-segment rows [M][P*K], w1s = split_pack_weight of the GEGLU-interleaved
    w1 (pack_w1_geglu(..., dtype=float32)) [2Fp][P*K] -> (term-segment pack [M][P*Fp] of gate * gelu(x), fp32 [M][Fp/32][2] LayerNorm(inner) partial sums or None)"""
    _chk_cuda(xs, w1s)
    M, K = xs.shape
    N = w1s.shape[0]
    P = split_count(products)
    assert w1s.shape[1] == K and xs.stride(1) == 1 and w1s.stride(1) == 1 and split_is_f16(products)
    out = torch.empty(M, P * (N // 2), dtype=bf16, device=xs.device)
    part = torch.empty(M, N // 64, 2, dtype=torch.float32, device=xs.device) if stats else None
    L.check(L.lib().mm_gemm_split_geglu(L.stream(), L.ptr(xs), xs.stride(0), L.ptr(w1s), w1s.stride(0), M, N, K, int(products), float(alpha), L.ptr(out),
                                        out.stride(0), L.ptr(part)), 'mm_gemm_split_geglu')
    return out, part


def split_rows(x, products):
    """fp32 [rows][K] -> bf16 [rows][products * K] (mm_split_rows): the GEMM operand form of the precision tier"""
    _chk_cuda(x)
    x = x.float()
    assert x.dim() == 2 and x.stride(1) == 1 and x.shape[1] % 4 == 0 and x.stride(0) % 4 == 0
    rows, K = x.shape
    out = torch.empty(rows, split_count(products) * K, dtype=bf16, device=x.device)
    L.check(L.lib().mm_split_rows(L.stream(), L.ptr(x), x.stride(0), rows, K, int(products), L.ptr(out)), 'mm_split_rows')
    return out


def unsplit_rows(xs, products, K):
    """the fp32 value of a segment pack: h + m + l (exact); fp16 terms: h + l (the value to within 2^-22)"""
    P = split_count(products)
    if split_is_f16(products):
        v = xs.view(torch.float16).reshape(xs.shape[0], P, K)[:, :2].float()
        return v[:, 0] + v[:, 1]
    v = xs.reshape(xs.shape[0], P, K)[:, :3].float()
    return (v[:, 0] + v[:, 1]) + v[:, 2]


def cfg_mix(emb_cond, emb_null, cond_scale, D, products=None):
    """e = e_null + (e_cond - e_null) * cond_scale on the final embeddings (mm_cfg_mix): bf16 [R, D] rows, or term-segment packs [R, P*D] of the
    precision tiers (mixed as fp32 values and re-split; `products`: the operand code -- inferred from the width for bf16 terms).
    to_logits of the result is the guidance-combined logits of mmp.py:254."""
    _chk_cuda(emb_cond, emb_null)
    assert emb_cond.dtype == bf16 and emb_null.dtype == bf16 and emb_cond.shape == emb_null.shape and emb_cond.stride(0) == emb_null.stride(0)
    R, W = emb_cond.shape
    P = (W // D if W != D else 0) if products is None else int(products)
    assert W == max(split_count(P), 1) * D and emb_cond.stride(1) == 1 and emb_null.stride(1) == 1
    out = torch.empty(R, W, dtype=bf16, device=emb_cond.device)
    L.check(L.lib().mm_cfg_mix(L.stream(), L.ptr(emb_cond), L.ptr(emb_null), emb_cond.stride(0), R, D, P, float(cond_scale), L.ptr(out)), 'mm_cfg_mix')
    return out
