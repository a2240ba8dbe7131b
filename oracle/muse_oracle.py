"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatement (torch fp32, functional, state_dict-driven) of the reference hot path:
``Transformer.forward`` / ``forward_with_cond_scale`` / ``MaskGit.generate`` and the
``VQGanVAE`` encode / LFQ / decode.  Every function cites the reference file:line it follows
(paths relative to /root/reference/muse_maskgit_pytorch/).  Only tests/, bench.py's
``cpu_baseline`` leg and __graft_entry__.smoke() may import this module -- as the checker.

Pinning: validated against the reference's own executable code run in the build container
(oracle/make_golden.py -> tests/golden/*.pt; tests/test_oracle_vs_golden.py).  The two
third-party dependencies (LFQ, FlashAttentionFunction) are absent from /root/reference, so for
those two operators parity is UNPINNED (see oracle/third_party_restatement.py).

Two knobs exist that the reference does not have, both default to reference behaviour:

* ``rp`` (rounding points): a callable applied where the HIP pipeline stores bf16
  (LayerNorm out, q/k/v, normalised q/k, softmax weights, attention out, GEGLU+LN out).
  ``rp=None`` is plain fp32 = the reference.  ``rp=bf16_round`` models the HIP path's
  storage precision so that the remaining difference is accumulation order only.
* tie rule: ``torch.topk`` leaves the order of equal scores implementation-defined
  (SURVEY.md Appendix B).  The restatement uses the deterministic rule the HIP path
  implements -- larger score first, then LOWER index -- and exposes ``boundary_ties`` so a
  test can tell whether the reference's own choice was forced.
"""
import math

import torch
import torch.nn.functional as F

MASK_FILL = -1e5            # muse_maskgit_pytorch.py:609
ATTN_SCALE = 8.0            # muse_maskgit_pytorch.py:98, attend.py:37


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _rp(rp, t):
    return t if rp is None else rp(t)


def e4m3_rows(t):
    """per-row fake quantisation to OCP e4m3 (the fp8 engine's rule for weights and activations: scale = max |row| / 448, 1 for a zero row;
    row / scale rounded to nearest even; the product uses value * scale)"""
    amax = t.abs().amax(dim=-1, keepdim=True)
    scale = torch.where(amax > 0, amax / 448., torch.ones_like(amax))
    return (t / scale).to(torch.float8_e4m3fn).to(torch.float32) * scale


class Fp8Rounding:
    """rounding points of the fp8 engine (self-defined numerics, SURVEY 8c "L2"; csrc/gemm_fp8.hip, fp8_act.hip): called like bf16_round wherever the
    HIP pipeline stores bf16, and `gq` is applied to the activation that feeds one of the layers' Linear weights -- the fp32 LayerNorm output (NOT
    rounded to bf16 first: the LayerNorm kernels quantise their fp32 values) or the bf16 attention output.  Use with the de-quantised weights
    (Transformer.fp8_dequantized_state_dict)."""

    def __call__(self, t):
        return bf16_round(t)

    @staticmethod
    def gq(t):
        return e4m3_rows(t)


def _lin_in(rp, t, stored_bf16=False):
    """the activation a Linear multiplies: fp8 engine -> fake-quantised rows (of the bf16-stored value if the producer stores bf16), else rp(t)"""
    gq = getattr(rp, 'gq', None)
    if gq is None:
        return _rp(rp, t)
    return gq(_rp(rp, t)) if stored_bf16 else gq(t)


# ----------------------------------------------------------------------------- operators

def layer_norm(x, gamma, beta):
    """muse_maskgit_pytorch.py:63-70 -- F.layer_norm over the last dim, eps 1e-5, beta is a zero buffer."""
    return F.layer_norm(x, x.shape[-1:], gamma, beta)


def geglu(x):
    """muse_maskgit_pytorch.py:72-77 -- first half is gelu'd (exact erf), second half is the gate."""
    a, gate = x.chunk(2, dim=-1)
    return gate * F.gelu(a)


def feed_forward(x, sd, prefix, rp=None):
    """muse_maskgit_pytorch.py:79-89 -- LN(D) -> Linear(D,2F) -> GEGLU -> LN(F) -> Linear(F,D), no bias."""
    h = _lin_in(rp, layer_norm(x, sd[prefix + '0.gamma'], sd[prefix + '0.beta']))
    h = h @ sd[prefix + '1.weight'].t()
    h = _rp(rp, geglu(h))          # HIP path: GEGLU runs on the fp32 accumulators in the GEMM epilogue, its result is stored bf16
    h = _lin_in(rp, layer_norm(h, sd[prefix + '3.gamma'], sd[prefix + '3.beta']))
    return h @ sd[prefix + '4.weight'].t()


def attend(q, k, v, mask=None, scale=ATTN_SCALE, rp=None):
    """attend.py:123-140 (math branch == what the flash branch attend.py:66-107 computes with its
    q,k pre-scaling by sqrt(scale/dh**-.5)).  mask True = keep; fill -finfo.max."""
    sim = torch.einsum('bhid,bhjd->bhij', q, k) * scale
    if mask is not None:
        sim = sim.masked_fill(~mask, -torch.finfo(sim.dtype).max)
    attn = sim.softmax(dim=-1)
    if rp is not None:
        # HIP path: weights are rounded to bf16 for the PV product, the normaliser is the
        # fp32 sum of the UNROUNDED weights
        m = sim.amax(dim=-1, keepdim=True)
        p = torch.exp(sim - m)
        l = p.sum(dim=-1, keepdim=True)
        return torch.einsum('bhij,bhjd->bhid', rp(p), v) / l
    return torch.einsum('bhij,bhjd->bhid', attn, v)


def attention(x, sd, prefix, heads, context=None, context_mask=None, rp=None):
    """muse_maskgit_pytorch.py:126-162."""
    b, n, _ = x.shape
    h = heads
    xn = _lin_in(rp, layer_norm(x, sd[prefix + 'norm.gamma'], sd[prefix + 'norm.beta']))
    kv_in = context if context is not None else xn          # (fp8 engine: the context's k | v projection stays bf16 -- context is not quantised)
    q = _rp(rp, xn @ sd[prefix + 'to_q.weight'].t())
    kv = _rp(rp, kv_in @ sd[prefix + 'to_kv.weight'].t())
    k, v = kv.chunk(2, dim=-1)

    def split(t):
        return t.reshape(t.shape[0], t.shape[1], h, -1).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    nk, nv = sd[prefix + 'null_kv']                         # (h, 1, dh) each   :145-149
    k = torch.cat((nk[None].expand(b, -1, -1, -1), k), dim=-2)
    v = torch.cat((_rp(rp, nv)[None].expand(b, -1, -1, -1), v), dim=-2)
    q = F.normalize(q, dim=-1) * sd[prefix + 'q_scale']     # :151-153
    k = F.normalize(k, dim=-1) * sd[prefix + 'k_scale']
    q, k = _rp(rp, q), _rp(rp, k)
    mask = None
    if context_mask is not None:                            # :155-157
        mask = F.pad(context_mask[:, None, None, :].expand(b, h, n, -1), (1, 0), value=True)
    out = _rp(rp, attend(q, k, v, mask=mask, rp=rp))
    out = out.permute(0, 2, 1, 3).reshape(b, n, -1)
    if getattr(rp, 'gq', None) is not None:
        out = rp.gq(out)                                    # fp8 engine: the bf16 attention output is quantised per token row in front of to_out
    return out @ sd[prefix + 'to_out.weight'].t()


def transformer_blocks(x, sd, depth, heads, context, context_mask, rp=None, prefix='transformer_blocks.'):
    """muse_maskgit_pytorch.py:187-195."""
    for i in range(depth):
        p = f'{prefix}layers.{i}.'
        x = attention(x, sd, p + '0.', heads, rp=rp) + x
        x = attention(x, sd, p + '1.', heads, context=context, context_mask=context_mask, rp=rp) + x
        x = feed_forward(x, sd, p + '2.', rp=rp) + x
    return layer_norm(x, sd[prefix + 'norm.gamma'], sd[prefix + 'norm.beta'])


def transformer_forward(sd, cfg, ids, text_embeds, cond_drop_prob=0., conditioning_token_ids=None,
                        self_cond_embed=None, rp=None, return_embed=False):
    """muse_maskgit_pytorch.py:279-335 (the non-loss returns).  cfg: dict(depth, heads, self_cond).
    cond_drop_prob must be 0 or 1 (the only values generate() uses; :393-399 is then deterministic)."""
    assert cond_drop_prob in (0., 1.)
    b, n = ids.shape
    context = text_embeds
    if 'text_embed_proj.weight' in sd:                      # :233, :302
        context = text_embeds @ sd['text_embed_proj.weight'].t()
    context = _rp(rp, context)
    context_mask = (text_embeds != 0).any(dim=-1)           # :304 (on the RAW embeds)
    if cond_drop_prob > 0.:                                 # :308-310
        context_mask = context_mask & torch.zeros((b, 1), dtype=torch.bool)
    if conditioning_token_ids is not None:                  # :314-318
        cids = conditioning_token_ids.reshape(b, -1)
        context = torch.cat((context, sd['token_emb.weight'][cids]), dim=-2)
        context_mask = F.pad(context_mask, (0, cids.shape[-1]), value=True)
    x = sd['token_emb.weight'][ids] + sd['pos_emb.weight'][torch.arange(n)]    # :322-323
    if cfg.get('self_cond', False):                         # :325-328
        if self_cond_embed is None:
            self_cond_embed = torch.zeros_like(x)
        x = x + feed_forward(self_cond_embed, sd, 'self_cond_to_init_embed.', rp=rp)
    embed = transformer_blocks(x, sd, cfg['depth'], cfg['heads'], context, context_mask, rp=rp)
    embed = _rp(rp, embed)
    logits = embed @ sd['to_logits.weight'].t()             # :332
    if return_embed:
        return logits, embed
    return logits


def forward_with_cond_scale(sd, cfg, ids, text_embeds, cond_scale=3., rp=None, return_embed=False, **kw):
    """muse_maskgit_pytorch.py:240-259."""
    if cond_scale == 1:
        return transformer_forward(sd, cfg, ids, text_embeds, 0., rp=rp, return_embed=return_embed, **kw)
    logits, embed = transformer_forward(sd, cfg, ids, text_embeds, 0., rp=rp, return_embed=True, **kw)
    null_logits = transformer_forward(sd, cfg, ids, text_embeds, 1., rp=rp, **kw)
    scaled = null_logits + (logits - null_logits) * cond_scale
    if return_embed:
        return scaled, embed
    return scaled


def forward_with_neg_prompt(sd, cfg, ids, text_embed, neg_text_embed, cond_scale=3., rp=None, return_embed=False, **kw):
    """SELF-DEFINED, PARITY UNPINNED: muse_maskgit_pytorch.py:261-277 cannot execute (undefined `*args` / `scaled_logits`; the
    caller at :544 passes a keyword the signature does not have).  This restates its evident intent -- the classifier-free
    formula of :254 with the negative prompt's pass in the place of the null pass, both passes with cond_drop_prob 0."""
    neg = transformer_forward(sd, cfg, ids, neg_text_embed, 0., rp=rp, **kw)
    pos, embed = transformer_forward(sd, cfg, ids, text_embed, 0., rp=rp, return_embed=True, **kw)
    out = neg + (pos - neg) * cond_scale
    return (out, embed) if return_embed else out


def transformer_loss(sd, cfg, ids, text_embeds, labels, ignore_index=0, cond_drop_prob=0., rp=None, **kw):
    """muse_maskgit_pytorch.py:337-348: CE over the vocabulary with ignore_index, or BCE-with-logits when dim_out == 1."""
    logits = transformer_forward(sd, cfg, ids, text_embeds, cond_drop_prob, rp=rp, **kw)
    if logits.shape[-1] == 1:
        return F.binary_cross_entropy_with_logits(logits[..., 0], labels)
    return F.cross_entropy(logits.permute(0, 2, 1), labels, ignore_index=ignore_index)


# ----------------------------------------------------------------------------- sampling tail

def cosine_schedule(t):
    """muse_maskgit_pytorch.py:422-423."""
    return torch.cos(t * math.pi * 0.5)


def mask_counts(timesteps, seq_len):
    """muse_maskgit_pytorch.py:556-559 -- the per-step number of tokens to (re)mask, same fp32
    arithmetic (linspace / cos in fp32, python int() truncation, max(.,1))."""
    out = []
    for t in torch.linspace(0, 1, timesteps):
        out.append(max(int((cosine_schedule(t) * seq_len).item()), 1))
    return out


def step_temperatures(timesteps, temperature=1.):
    """muse_maskgit_pytorch.py:578 with :411's clamp -- the divisor actually used per step."""
    return [max(temperature * (s / timesteps), 1e-10) for s in reversed(range(timesteps))]


def log_clamped(t, eps=1e-20):
    """muse_maskgit_pytorch.py:403-404."""
    return torch.log(t.clamp(min=eps))


def gumbel_from_uniform(u):
    """muse_maskgit_pytorch.py:406-408."""
    return -log_clamped(-log_clamped(u))


def topk_threshold(logits, thres=0.9):
    """muse_maskgit_pytorch.py:413-418 -- returns (k, kth-largest value per row).  The reference keeps
    exactly k entries (topk + scatter into -inf); keeping every entry >= the k-th value is
    identical unless several entries EQUAL the k-th value (see ``threshold_ties``)."""
    k = math.ceil((1 - thres) * logits.shape[-1])
    kth = logits.topk(k, dim=-1).values[..., -1:]
    return k, kth


def threshold_ties(logits, thres=0.9):
    """Rows where #(x >= kth) != k, i.e. the reference's kept set depends on topk's tie order."""
    k, kth = topk_threshold(logits, thres)
    return (logits >= kth).sum(dim=-1) != k


def top_k_filter(logits, thres=0.9):
    k, kth = topk_threshold(logits, thres)
    return torch.where(logits >= kth, logits, torch.full_like(logits, float('-inf')))


def gumbel_sample(filtered, gumbel, temperature):
    """muse_maskgit_pytorch.py:410-411 with the noise tensor injected (already -log(-log(u)))."""
    return ((filtered / max(temperature, 1e-10)) + gumbel).argmax(dim=-1)


def select_topk_stable(scores, k):
    """scores.topk(k).indices (muse_maskgit_pytorch.py:561) with the deterministic tie rule:
    larger score first, then lower index.  Returns a bool mask (b, n) of selected positions."""
    order = torch.sort(scores, dim=-1, descending=True, stable=True).indices
    sel = torch.zeros_like(scores, dtype=torch.bool)
    sel.scatter_(1, order[:, :k], True)
    return sel


def boundary_ties(scores, k):
    """True per row if the k-th and (k+1)-th largest scores are equal (topk's choice not forced)."""
    n = scores.shape[-1]
    if k >= n:
        return torch.zeros(scores.shape[0], dtype=torch.bool)
    s = torch.sort(scores, dim=-1, descending=True).values
    return s[:, k - 1] == s[:, k]


def sample_step(logits, gumbel, ids, mask_id, temperature, thres=0.9, can_remask_prev_masked=False):
    """muse_maskgit_pytorch.py:576-609 for one step, given the CFG-combined logits (b,n,V),
    the injected Gumbel noise (b,n,V) and ids AFTER the re-mask scatter.
    Returns (new_ids, new_scores, pred_ids)."""
    filtered = top_k_filter(logits, thres)
    pred = gumbel_sample(filtered, gumbel, temperature)
    is_mask = ids == mask_id
    new_ids = torch.where(is_mask, pred, ids)
    probs = logits.softmax(dim=-1)
    scores = 1 - probs.gather(2, pred[..., None])[..., 0]
    if not can_remask_prev_masked:                          # :608-609
        scores = scores.masked_fill(~is_mask, MASK_FILL)
    return new_ids, scores, pred


def generate_ids(demask_fn, batch, seq_len, mask_id, gumbel_fn, timesteps=18, temperature=1.,
                 thres=0.9, trace=None, can_remask_prev_masked=False, critic_fn=None, critic_uniform_fn=None,
                 critic_noise_scale=1.):
    """muse_maskgit_pytorch.py:507-615 without text encoding / VAE.  ``demask_fn(ids, step)`` returns the
    CFG-combined logits (b,n,V) fp32 (a stateful closure carries the self-conditioning embed, :574);
    ``gumbel_fn(step, shape)`` returns the Gumbel noise.  With ``critic_fn(ids, step) -> (b,n)`` the next step's
    scores come from the token critic plus annealed uniform noise ``critic_uniform_fn(step, shape)`` (:590-601)."""
    ids = torch.full((batch, seq_len), mask_id, dtype=torch.long)
    scores = torch.zeros((batch, seq_len), dtype=torch.float32)
    counts = mask_counts(timesteps, seq_len)
    temps = step_temperatures(timesteps, temperature)
    for step in range(timesteps):
        sel = select_topk_stable(scores, counts[step])
        tie = boundary_ties(scores, counts[step])
        ids = torch.where(sel, torch.full_like(ids, mask_id), ids)
        logits = demask_fn(ids, step)
        g = gumbel_fn(step, logits.shape)
        masked_ids = ids
        ids, scores, pred = sample_step(logits, g, ids, mask_id, temps[step], thres, can_remask_prev_masked)
        if critic_fn is not None:
            steps_until_x0 = timesteps - (step + 1)
            scores = critic_fn(ids, step)
            scores = scores + (critic_uniform_fn(step, scores.shape) - 0.5) * critic_noise_scale * (steps_until_x0 / timesteps)
        if trace is not None:
            trace.append(dict(step=step, k=counts[step], sel=sel, tie=tie, masked_ids=masked_ids,
                              ids=ids.clone(), scores=scores.clone(), pred=pred))
    return ids


# ----------------------------------------------------------------------------- VQGanVAE

def lfq_indices_to_codes(sd, ids, prefix='quantizer.'):
    """third-party LFQ.indices_to_codes as called at vqgan_vae.py:431 -- (B,N) -> (B,N,C)."""
    mask = sd[prefix + 'mask']
    bits = ((ids[..., None].long() & mask) != 0).float()
    codes = bits * 2 - 1
    if prefix + 'project_out.weight' in sd:
        codes = codes @ sd[prefix + 'project_out.weight'].t() + sd[prefix + 'project_out.bias']
    return codes


def lfq_encode(sd, fmap, prefix='quantizer.'):
    """third-party LFQ.forward (eval) as called at vqgan_vae.py:424 -- (B,C,h,w) -> (fmap, ids (B,h,w))."""
    b, c, h, w = fmap.shape
    t = fmap.permute(0, 2, 3, 1).reshape(b, h * w, c)
    if prefix + 'project_in.weight' in sd:
        t = t @ sd[prefix + 'project_in.weight'].t() + sd[prefix + 'project_in.bias']
    pos = t > 0
    idx = (pos.long() * sd[prefix + 'mask'].long()).sum(dim=-1)
    q = torch.where(pos, torch.ones_like(t), -torch.ones_like(t))
    if prefix + 'project_out.weight' in sd:
        q = q @ sd[prefix + 'project_out.weight'].t() + sd[prefix + 'project_out.bias']
    return q.reshape(b, h, w, c).permute(0, 3, 1, 2), idx.reshape(b, h, w)


def _conv(sd, key, x, rp=None, **kw):
    return F.conv2d(x, sd[key + '.weight'], sd[key + '.bias'], **kw)


def glu_res_block(sd, p, x, groups=16, rp=None):
    """vqgan_vae.py:251-265."""
    h = _rp(rp, _conv(sd, p + 'net.0', x, padding=1))
    h = _rp(rp, F.glu(h, dim=1))
    h = _rp(rp, F.group_norm(h, groups, sd[p + 'net.2.weight'], sd[p + 'net.2.bias']))
    h = _rp(rp, _conv(sd, p + 'net.3', h, padding=1))
    h = _rp(rp, F.glu(h, dim=1))
    h = _rp(rp, F.group_norm(h, groups, sd[p + 'net.5.weight'], sd[p + 'net.5.bias']))
    h = _conv(sd, p + 'net.6', h)
    return _rp(rp, h + x)


def res_block(sd, p, x, groups=16, rp=None):
    """vqgan_vae.py:267-281."""
    h = _rp(rp, _conv(sd, p + 'net.0', x, padding=1))
    h = _rp(rp, F.leaky_relu(F.group_norm(h, groups, sd[p + 'net.1.weight'], sd[p + 'net.1.bias']), 0.1))
    h = _rp(rp, _conv(sd, p + 'net.3', h, padding=1))
    h = _rp(rp, F.leaky_relu(F.group_norm(h, groups, sd[p + 'net.4.weight'], sd[p + 'net.4.bias']), 0.1))
    h = _conv(sd, p + 'net.6', h)
    return _rp(rp, h + x)


def vae_decode(sd, fmap, layers=4, rp=None, prefix='enc_dec.decoders.'):
    """vqgan_vae.py:246-249 with the decoder list built at :223-232:
    GLUResBlock -> layers x [ConvTranspose2d(4,2,1) + LeakyReLU(0.1)] -> Conv2d(d, channels, 1)."""
    x = glu_res_block(sd, prefix + '0.', fmap, rp=rp)
    for i in range(1, layers + 1):
        x = F.conv_transpose2d(x, sd[f'{prefix}{i}.0.weight'], sd[f'{prefix}{i}.0.bias'], stride=2, padding=1)
        x = _rp(rp, F.leaky_relu(x, 0.1))
    return _conv(sd, f'{prefix}{layers + 1}', x)


def vae_decode_from_ids(sd, ids, layers=4, rp=None):
    """vqgan_vae.py:427-438 (LFQ branch)."""
    b, h, w = ids.shape
    codes = lfq_indices_to_codes(sd, ids.reshape(b, h * w))
    fmap = _rp(rp, codes.reshape(b, h, w, -1).permute(0, 3, 1, 2))
    return vae_decode(sd, fmap, layers=layers, rp=rp)


def vae_encode(sd, img, layers=4, rp=None, prefix='enc_dec.encoders.'):
    """vqgan_vae.py:422-425 / :241-244 with the encoder list built at :223-231."""
    x = _rp(rp, _conv(sd, prefix + '0', img, padding=sd[prefix + '0.weight'].shape[-1] // 2))
    for i in range(1, layers + 1):
        x = _rp(rp, F.leaky_relu(_conv(sd, f'{prefix}{i}.0', x, stride=2, padding=1), 0.1))
    x = res_block(sd, f'{prefix}{layers + 1}.', x, rp=rp)
    return lfq_encode(sd, x)
