"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

A CPU model of the ROUNDING POINTS of the bf16 engine's general forward path (csrc/model.hip mm_transformer_forward, round 6), so that what is left between the
engine and this model is accumulation order (and the rare bf16 rounding that an accumulation-order difference flips) -- SURVEY.md 8c precision ladder, level L1:
"bf16 kernels vs a bf16-cast oracle with the same rounding points".  `muse_oracle.transformer_forward(rp=bf16_round)` was that model for round 1's engine; since
then the engine moved its rounding points (LayerNorm folds, cross-attention with the output projection folded into the values) and that oracle no longer describes
it: the engine sat 2-3e-2 of the logits' scale from it.  This module restates, operator by operator, what the kernels multiply and where they round:

  weights        every nn.Linear weight is packed as bf16(W) (muse_maskgit.py _pack_attn / _pack_ff); a LayerNorm folded into the GEMM behind it uses
                 bf16(W . diag gamma) with c1[o] = sum_k of those bf16 values and c2 = W beta in fp32 (:296-299, :264-273)
  embeddings     bf16 tables, summed in fp32 (norm_act.hip embed_kernel); text context -> bf16 -> projection (bf16 W) -> bf16; k|v of the context -> bf16
  LayerNorm(dim) layer 0's self-attention / feed-forward inputs and the final LayerNorm: a LayerNorm pass on the fp32 stream, output bf16.  Everywhere else (model.hip
                 fold flags: self-attention l > 0, every cross-attention q, feed-forward l > 0) the FOLD: the GEMM multiplies bf16(x) -- the RAW residual row -- by
                 bf16(W gamma) and applies  rstd * (acc - mean * c1) + c2  on its fp32 accumulator, mean / variance (E[x^2] - mean^2) from the fp32 row
  attention      q|k|v stored bf16; q^ = normalize(q) * q_scale and k^ likewise rounded to bf16; scores fp32 x 8; p = exp(s - max) rounded to bf16 for P V, the
                 normaliser is the fp32 sum of the UNROUNDED p; output bf16  (attention.hip; == muse_oracle.attend(rp))
  cross_fold     (dim 512, 8 heads, <= 79 context tokens) q stays fp32 until q^ is rounded; the softmax is NORMALISED before its bf16 rounding; the values are
                 VW_h = bf16(V_h W_o,h^T) (V bf16, W_o bf16, fp32 product), the null value enters through the same fold; x += sum_h bf16(P_h) VW_h   (cross_fold.hip)
  feed-forward   w1 (fold or LayerNorm pass) -> GEGLU in fp32 -> h stored bf16; LayerNorm(inner) folded into w2: bf16(bf16(W2) gamma2), statistics of the bf16 h,
                 out = rstd * acc - (rstd * mean) * c1 + c2 + x
  logits         bf16(final LayerNorm) . bf16(to_logits)^T in fp32

Cites: muse_maskgit_pytorch.py:63-162, 187-195, 279-335 for the operators; the csrc files named above for the rounding points.
"""
import torch
import torch.nn.functional as F

import muse_oracle as O

R = O.bf16_round
EPS = 1e-5


def _stats(x):
    """per-row (mean, rstd) the way the fold's consumers form them: fp32 sums, variance = E[x^2] - mean^2 clamped at 0 (common.h ln_rstd_negmean)"""
    d = x.shape[-1]
    mean = x.sum(dim=-1, keepdim=True) / d
    var = ((x * x).sum(dim=-1, keepdim=True) / d - mean * mean).clamp_min(0.)
    return mean, torch.rsqrt(var + EPS)


def _fold_linear(x, w, gamma, beta):
    """LayerNorm(x) @ w^T with the LayerNorm folded into the GEMM (csrc/common.h ln_fold_apply): bf16(x) @ bf16(w gamma)^T, then rstd * (acc - mean * c1) + c2"""
    wg = R(w * gamma[None, :])
    c1 = wg.sum(dim=1)
    c2 = w @ beta
    mean, rstd = _stats(x)
    acc = R(x) @ wg.t()
    return rstd * (acc - mean * c1) + c2


def _ln_linear(x, w, gamma, beta):
    """LayerNorm pass (bf16 out) + bf16 GEMM"""
    return R(O.layer_norm(x, gamma, beta)) @ R(w).t()


def _split(t, h):
    return t.reshape(t.shape[0], t.shape[1], h, -1).permute(0, 2, 1, 3)


def _self_attention(x, sd, p, heads, fold):
    b, n, _ = x.shape
    g, be = sd[p + 'norm.gamma'], sd[p + 'norm.beta']
    wqkv = torch.cat([sd[p + 'to_q.weight'], sd[p + 'to_kv.weight']], dim=0)
    qkv = R(_fold_linear(x, wqkv, g, be) if fold else _ln_linear(x, wqkv, g, be))
    q, k, v = qkv.chunk(3, dim=-1)
    q, k, v = _split(q, heads), _split(k, heads), _split(v, heads)
    nk, nv = sd[p + 'null_kv']
    k = torch.cat((nk[None].expand(b, -1, -1, -1), k), dim=-2)
    v = torch.cat((R(nv)[None].expand(b, -1, -1, -1), v), dim=-2)
    q = R(F.normalize(q, dim=-1) * sd[p + 'q_scale'])
    k = R(F.normalize(k, dim=-1) * sd[p + 'k_scale'])
    out = R(O.attend(q, k, v, rp=R)).permute(0, 2, 1, 3).reshape(b, n, -1)
    return out @ R(sd[p + 'to_out.weight']).t()


def _cross_attention(x, sd, p, heads, ctx, ctx_mask, fold, one_kernel):
    b, n, _ = x.shape
    g, be = sd[p + 'norm.gamma'], sd[p + 'norm.beta']
    wq = sd[p + 'to_q.weight']
    q = _fold_linear(x, wq, g, be) if fold else _ln_linear(x, wq, g, be)
    kv = R(ctx @ R(sd[p + 'to_kv.weight']).t())
    k, v = kv.chunk(2, dim=-1)
    k, v = _split(k, heads), _split(v, heads)
    nk, nv = sd[p + 'null_kv']
    k = torch.cat((nk[None].expand(b, -1, -1, -1), k), dim=-2)
    k = R(F.normalize(k, dim=-1) * sd[p + 'k_scale'])
    mask = F.pad(ctx_mask[:, None, None, :].expand(b, heads, n, -1), (1, 0), value=True)
    wo = R(sd[p + 'to_out.weight'])                                   # [D][I]
    if one_kernel:
        # cross_fold.hip: q is not rounded on its way to q^; P is normalised before its rounding; the output projection is folded into the values
        qh = R(F.normalize(_split(q, heads), dim=-1) * sd[p + 'q_scale'])
        sim = torch.einsum('bhid,bhjd->bhij', qh, k) * O.ATTN_SCALE
        sim = sim.masked_fill(~mask, -torch.finfo(sim.dtype).max)
        pr = R(sim.softmax(dim=-1))
        vfull = torch.cat((nv[None].expand(b, -1, -1, -1), v), dim=-2)  # the null value enters the pack in fp32 (k_cross_fold_pack), the context's values are bf16
        dh = vfull.shape[-1]
        woh = wo.reshape(wo.shape[0], heads, dh)                       # [D][h][dh]
        vw = R(torch.einsum('bhjd,ohd->bhjo', vfull, woh))            # VW_h [keys][D], bf16
        return torch.einsum('bhij,bhjo->bio', pr, vw)
    q = R(q)
    qh = R(F.normalize(_split(q, heads), dim=-1) * sd[p + 'q_scale'])
    v = torch.cat((R(nv)[None].expand(b, -1, -1, -1), v), dim=-2)
    out = R(O.attend(qh, k, v, mask=mask, rp=R)).permute(0, 2, 1, 3).reshape(b, n, -1)
    return out @ wo.t()


def _feed_forward(x, sd, p, fold):
    g1, b1 = sd[p + '0.gamma'], sd[p + '0.beta']
    w1, w2 = sd[p + '1.weight'], sd[p + '4.weight']
    u = _fold_linear(x, w1, g1, b1) if fold else _ln_linear(x, w1, g1, b1)
    h = R(O.geglu(u))                                                  # GEGLU on the fp32 accumulators, stored bf16
    g2, b2 = sd[p + '3.gamma'], sd[p + '3.beta']
    w2b = R(w2)
    w2g = R(w2b * g2[None, :])
    c1 = w2g.sum(dim=1)
    c2 = (w2b * b2[None, :]).sum(dim=1)
    mean, rstd = _stats(h)                                             # statistics of the bf16 values the GEMM multiplies (ln_partial_row64)
    acc = h @ w2g.t()
    return rstd * acc - (rstd * mean) * c1 + c2


def bf16_engine_forward(sd, cfg, ids, text_embeds, cond_drop_prob=0., conditioning_token_ids=None, return_embed=False, fold=True, one_kernel_cross=None):
    """Transformer.forward (muse_maskgit_pytorch.py:279-335, the non-loss returns) with the bf16 engine's rounding points.  cfg: dict(depth, heads).
    `fold`: the LayerNorm(dim) fold is on (set False for Transformer.set_layernorm_fold(False)); `one_kernel_cross` None = the engine's own rule
    (dim == inner == 512, 8 heads of 64, <= 79 context keys, fold on)."""
    assert cond_drop_prob in (0., 1.) and not cfg.get('self_cond', False)
    b, n = ids.shape
    depth, heads = cfg['depth'], cfg['heads']
    ctx = R(text_embeds)
    if 'text_embed_proj.weight' in sd:
        ctx = R(ctx @ R(sd['text_embed_proj.weight']).t())
    ctx_mask = (text_embeds != 0).any(dim=-1)
    if cond_drop_prob > 0.:
        ctx_mask = ctx_mask & torch.zeros((b, 1), dtype=torch.bool)
    if conditioning_token_ids is not None:
        cids = conditioning_token_ids.reshape(b, -1)
        ctx = torch.cat((ctx, R(sd['token_emb.weight'])[cids]), dim=-2)
        ctx_mask = F.pad(ctx_mask, (0, cids.shape[-1]), value=True)
    D = sd['pos_emb.weight'].shape[1]
    inner = sd['transformer_blocks.layers.0.0.to_q.weight'].shape[0]
    if one_kernel_cross is None:
        one_kernel_cross = fold and D == 512 and inner == 512 and heads == 8 and ctx.shape[1] <= 79
    x = R(sd['token_emb.weight'])[ids] + R(sd['pos_emb.weight'])[torch.arange(n)]
    for l in range(depth):
        p = f'transformer_blocks.layers.{l}.'
        x = _self_attention(x, sd, p + '0.', heads, fold and l > 0) + x
        x = _cross_attention(x, sd, p + '1.', heads, ctx, ctx_mask, fold, one_kernel_cross) + x
        x = _feed_forward(x, sd, p + '2.', fold and l > 0) + x
    emb = R(O.layer_norm(x, sd['transformer_blocks.norm.gamma'], sd['transformer_blocks.norm.beta']))
    logits = emb @ R(sd['to_logits.weight']).t()
    return (logits, emb) if return_embed else logits
