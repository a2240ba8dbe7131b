"""Transformer / MaskGitTransformer / TokenCritic / MaskGit / Muse drop-ins for MI355X.

Same constructor + method signatures and the same state_dict key names as the reference
(muse_maskgit_pytorch.py:63-791), so reference checkpoints load unchanged.  The nn.Modules hold fp32 parameters
only; all arithmetic of `Transformer.forward`, `forward_with_cond_scale` and `MaskGit.generate` runs in
libmuse_hip.so (bf16 MFMA kernels, fp32 residual stream / logits / sampling) through the C ABI in
include/muse_hip.h.  There is no eager fallback: without the library or without a gfx950 device these raise.
"""
import ctypes as C
import weakref
import math
from functools import partial
from pathlib import Path
from typing import Callable, List, Optional

import torch
from torch import nn

from . import _lib as L
from . import ops
from . import parity as P32
from .attend import Attend
from .t5 import DEFAULT_T5_NAME, get_encoded_dim, t5_encode_text
from .vqgan_vae import VQGanVAE

bf16 = torch.bfloat16


SPLIT_TIERS = ('bf16x3', 'f16x2')      # the fp32-grade precision tiers inside the C entry points: bf16 terms (3 / 5 / 6 products), fp16 terms (2 / 3)


def exists(v):
    return v is not None


def default(v, d):
    return v if exists(v) else d


def eval_decorator(fn):
    def inner(model, *args, **kwargs):
        was_training = model.training
        model.eval()
        out = fn(model, *args, **kwargs)
        model.train(was_training)
        return out
    return inner


def cosine_schedule(t):
    return torch.cos(t * math.pi * 0.5)


def _get_mask_subset_prob(mask, prob, min_mask=0):
    """mmp.py:46-59 (BERT-style: a random subset of the masked positions keeps its token)."""
    batch, seq = mask.shape
    num_to_mask = (mask.sum(dim=-1, keepdim=True) * prob).clamp(min=min_mask)
    logits = torch.rand((batch, seq), device=mask.device).masked_fill(~mask, -1)
    randperm = logits.argsort(dim=-1).argsort(dim=-1).float()
    randperm -= (~mask).sum(dim=-1, keepdim=True)
    subset_mask = randperm < num_to_mask
    subset_mask.masked_fill_(~mask, False)
    return subset_mask


# ------------------------------------------------------------------------------------------------ parameter containers

class LayerNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer('beta', torch.zeros(dim))

    def forward(self, x):
        shp = x.shape
        out = ops.layernorm(x.reshape(-1, shp[-1]).float().contiguous(), self.gamma.detach().float(), self.beta.float())
        return out.reshape(shp).to(x.dtype)


class GEGLU(nn.Module):
    def forward(self, x):
        raise NotImplementedError('GEGLU is fused with the following LayerNorm on MI355X (mm_geglu_ln); call the Transformer')


def FeedForward(dim, mult=4):
    inner_dim = int(dim * mult * 2 / 3)
    return nn.Sequential(LayerNorm(dim), nn.Linear(dim, inner_dim * 2, bias=False), GEGLU(), LayerNorm(inner_dim),
                         nn.Linear(inner_dim, dim, bias=False))


class Attention(nn.Module):
    def __init__(self, dim, dim_head=64, heads=8, cross_attend=False, scale=8, flash=True, dropout=0.):
        super().__init__()
        self.scale = scale
        self.heads = heads
        inner_dim = dim_head * heads
        self.cross_attend = cross_attend
        self.norm = LayerNorm(dim)
        self.attend = Attend(flash=flash, dropout=dropout, scale=scale)
        self.null_kv = nn.Parameter(torch.randn(2, heads, 1, dim_head))
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim, inner_dim * 2, bias=False)
        self.q_scale = nn.Parameter(torch.ones(dim_head))
        self.k_scale = nn.Parameter(torch.ones(dim_head))
        self.to_out = nn.Linear(inner_dim, dim, bias=False)


class TransformerBlocks(nn.Module):
    def __init__(self, *, dim, depth, dim_head=64, heads=8, ff_mult=4, flash=True):
        super().__init__()
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                Attention(dim=dim, dim_head=dim_head, heads=heads, flash=flash),
                Attention(dim=dim, dim_head=dim_head, heads=heads, cross_attend=True, flash=flash),
                FeedForward(dim=dim, mult=ff_mult)]))
        self.norm = LayerNorm(dim)
        self.cfg = dict(dim=dim, depth=depth, dim_head=dim_head, heads=heads, ff_mult=ff_mult)


# ------------------------------------------------------------------------------------------------ Transformer

FUSED_BOUND_SAMPLES = 2048      # vocabulary rows of to_logits the distribution-free bound of the fused sampler samples (mm_transformer_desc.logits_wsub)


class _Handle:
    """Owns the packed device weights (torch tensors) and the C model handle built from them."""

    def __init__(self):
        self.keep = []
        self.ptr = C.c_void_p()
        self.desc = None           # the mm_transformer_desc the handle was created from (re-used by ensure_logits_stats)
        self.packed = None
        self.stats_src = None      # callable -> fp32 [V][D] to_logits weights for the vocabulary statistics, or None when they do not apply
        self.auto_bound = 'gaussian'   # Transformer.fused_bound == 'auto': the bound this packed model currently uses (switched to 'quantile' by the first generate the Gaussian one fails)
        self.owner = None          # the Transformer this handle was packed for (keeps the LayerNorm-fold probe's verdict)
        self.ln_probe = None       # device float[1] while the LayerNorm(dim) fold of this handle is being probed (Transformer.set_layernorm_fold('auto'))
        self.ln_ratio = None       # the probe's result: max |row mean| / (row standard deviation) over every folded LayerNorm input of the first call

    def create(self, d, packed):
        self.desc, self.packed = d, packed
        L.check(L.lib().mm_transformer_create(C.byref(d), C.byref(self.ptr)), 'mm_transformer_create')

    def ensure_logits_stats(self, mode='quantile'):
        """What sampling without materialised logits needs besides the weights: a per-row LOWER BOUND of the k-th largest logit before the logits exist.
        mode 'quantile' (default, round 5; mm_transformer_desc.logits_wsub): FUSED_BOUND_SAMPLES rows of to_logits at vocabulary indices drawn once (seeded);
        per step a small GEMM gives every row's logits at those columns and the bound is their (S k / V + 4.5 sigma)-th largest -- distribution-free.
        mode 'gaussian' (rounds 2-4; logits_wmean / _wcov): mean and covariance of the weight rows; a row's logits over the vocabulary have mean <e, wmean>
        and variance e' wcov e, the bound is the Gaussian quantile -- fails every row of a peaky / heavy-tailed checkpoint (the call then repeats on the logits
        path).  Computed on the first generate() that wants the fused sampler (a forward()-only user never pays), then the C handle is re-created from the
        same packed tensors.  Either bound is verified per row on the device: the ids never depend on it."""
        if self.stats_src is None:
            return self
        d = self.desc
        if mode == 'quantile':
            if self.packed.get('wsub') is None:
                w = self.stats_src()                                    # [V][D] fp32
                V = w.shape[0]
                S = min(FUSED_BOUND_SAMPLES, V // 2)
                idx = torch.randperm(V, generator=torch.Generator().manual_seed(0x5EED))[:S].sort().values.to(w.device)
                self.packed['wsub'], self.packed['wsub_idx'] = w[idx].to(bf16).contiguous(), idx
            want = (int(self.packed['wsub'].data_ptr()), int(self.packed['wsub'].shape[0]))
        else:
            if self.packed.get('wcov') is None:
                wt = self.stats_src().t().contiguous()                  # [D][V]
                V = wt.shape[1]
                self.packed['wmean'] = wt.mean(dim=1).contiguous()
                self.packed['wcov'] = (P32.gemm(wt, wt) / float(V) - torch.outer(self.packed['wmean'], self.packed['wmean'])).to(bf16).contiguous()
                del wt
            d.logits_wmean, d.logits_wcov = L.ptr(self.packed['wmean']), L.ptr(self.packed['wcov'])
            want = (None, 0)
        if (d.logits_wsub, d.logits_wsub_rows) != want or (mode != 'quantile' and getattr(self, '_bound_mode', None) != mode):      # (a c_void_p field reads back as int / None)
            d.logits_wsub, d.logits_wsub_rows = want
            self._recreate()
        self._bound_mode = mode
        return self

    @property
    def fused_ready(self):
        return self.packed is not None and (self.packed.get('wsub') is not None or self.packed.get('wcov') is not None)

    def _recreate(self):
        L.lib().mm_transformer_destroy(self.ptr)
        self.ptr = C.c_void_p()
        L.check(L.lib().mm_transformer_create(C.byref(self.desc), C.byref(self.ptr)), 'mm_transformer_create')

    def begin_ln_probe(self, dev):
        self.ln_probe = torch.zeros(1, dtype=torch.float32, device=dev)
        self.desc.ln_probe = L.ptr(self.ln_probe)

    def finish_ln_probe(self):
        """Reads what the first call through this handle recorded (one host synchronisation, once per packed model) and re-creates the C handle without
        the probe -- with the LayerNorm(dim) fold OFF when some folded LayerNorm input had |mean| > MM_LN_FOLD_MAX_RATIO x its standard deviation
        (include/muse_hip.h, mm_transformer_desc.ln_fold_off).  Returns True in that case: the caller recomputes what it just computed."""
        if self.ln_probe is None:
            return False
        self.ln_ratio = float(self.ln_probe.item())
        off = self.ln_ratio > L.MM_LN_FOLD_MAX_RATIO
        owner = self.owner() if self.owner is not None else None
        if owner is not None:
            owner._ln_fold_auto = not off           # the verdict outlives this packed copy (Transformer._model)
            owner._ln_fold_ratio = self.ln_ratio
        self.desc.ln_probe, self.desc.ln_fold_off = None, int(off)
        self.ln_probe = None
        self._recreate()
        return off

    def __del__(self):
        try:
            if self.ptr:
                L.lib().mm_transformer_destroy(self.ptr)
        except Exception:
            pass


class Transformer(nn.Module):
    def __init__(self, *, num_tokens, dim, seq_len, dim_out=None, t5_name=DEFAULT_T5_NAME, self_cond=False,
                 add_mask_id=False, **kwargs):
        super().__init__()
        self.dim = dim
        self.mask_id = num_tokens if add_mask_id else None
        self.num_tokens = num_tokens
        self.token_emb = nn.Embedding(num_tokens + int(add_mask_id), dim)
        self.pos_emb = nn.Embedding(seq_len, dim)
        self.seq_len = seq_len
        self.transformer_blocks = TransformerBlocks(dim=dim, **kwargs)
        self.norm = LayerNorm(dim)            # allocated but never applied, as in the reference (mmp.py:222)
        self.dim_out = default(dim_out, num_tokens)
        self.to_logits = nn.Linear(dim, self.dim_out, bias=False)
        self.encode_text = partial(t5_encode_text, name=t5_name)
        text_embed_dim = get_encoded_dim(t5_name)
        self.text_embed_dim = text_embed_dim
        self.text_embed_proj = nn.Linear(text_embed_dim, dim, bias=False) if text_embed_dim != dim else nn.Identity()
        self.self_cond = self_cond
        self.self_cond_to_init_embed = FeedForward(dim)
        self._handle = None
        self._handle_key = None
        self._handle_x3 = None         # packed weights + C handle of the 'bf16x3' precision tier (built on first use)
        self._handle_x3_key = None
        self._x3_min_products = 0      # MaskGit raises it so that a generator and its token critic are packed with the same number of term products
        self._x3_extra_weights = ()    # MaskGit: the self-critic head's weight, packed with this model's term scale ('f16x2')
        self._ws = None
        self.grad_sync = None          # optional parallel.GradBucketer: data-parallel gradient averaging inside the backward
        self._handle_f8 = None         # packed weights + C handle of the fp8 engine (precision 'fp8', built on first use)
        self._handle_f8_key = None
        self.precision = 'bf16'        # 'bf16x3': fp32-grade tier on the bf16 matrix pipe inside the same C loop; 'parity': fp32 MFMA, operator by operator (set_precision)
        self.fused_bound = 'auto'      # fused sampling, per-row bound of the k-th largest logit: 'gaussian' (rounds 2-4: from the vocabulary statistics of to_logits; ~1 ms per
                                       # generate cheaper), 'quantile' (round 5: from sampled vocabulary columns, distribution-free), or 'auto' (default): Gaussian until a
                                       # generate shows it failing on this checkpoint, then the sampled bound for good (_Handle.ensure_logits_stats, MaskGit.generate)
        self._pack_gen = 0             # bumped whenever the packed weights are dropped or the engine choice changes (part of MaskGit's hipGraph cache key)
        self._ln_fold_auto = None      # the 'auto' LayerNorm-fold verdict for the current parameter values (None: not probed yet)
        self.layernorm_fold = 'auto'   # bf16 engine: LayerNorm(dim) folded into the GEMMs around it -- 'auto' (probed on the first call per packed model) | True | False (set_layernorm_fold)

    # ---- packing (once per parameter version / device)
    def _pack_ff(self, ff, keep):
        w1, w2 = ff[1].weight.detach(), ff[4].weight.detach()
        F = w2.shape[1]
        Fp = (F + 63) // 64 * 64
        D = w1.shape[1]
        w1p = ops.pack_w1_geglu(w1, Fp)          # GEGLU-interleaved tile order (the GEMM epilogue emits gate*gelu(x))
        w2p = ops.pad_cols(w2.to(bf16), 64)
        assert w2p.shape[1] == Fp
        t = dict(g1=ff[0].gamma.detach().float().contiguous(), b1=ff[0].beta.float().contiguous(), w1=w1p,
                 g2=ops.pad_cols(ff[3].gamma.detach().float(), Fp), b2=ops.pad_cols(ff[3].beta.float(), Fp), w2=w2p)
        # LayerNorm(inner) folded into w2 (mm_ff_weights.w2_folded): gains into the weights, the mean / bias terms as two [D] vectors
        w2f = (w2p.float() * t['g2']).to(bf16).contiguous()
        t['w2f'] = w2f
        t['c1'] = w2f.float().sum(dim=1).contiguous()
        t['c2'] = (w2p.float() * t['b2']).sum(dim=1).contiguous()
        # LayerNorm(dim) in front of w1 folded into w1 (mm_ff_weights.w1_ln): gains into the (GEGLU-interleaved) weight rows, mean / bias terms per packed row
        g1, b1 = t['g1'], t['b1']
        w1l = ops.pack_w1_geglu((w1.float() * g1[None, :]).to(bf16), Fp)
        t['w1l'] = w1l
        t['l1c1'] = w1l.float().sum(dim=1).contiguous()
        t['l1c2'] = ops.pack_w1_geglu((w1.float() @ b1)[:, None], Fp, dtype=torch.float32)[:, 0].contiguous() if bool((b1 != 0).any()) else None
        keep.append(t)
        fw = L.FFWeights(L.ptr(t['g1']), L.ptr(t['b1']), L.ptr(t['w1']), L.ptr(t['g2']), L.ptr(t['b2']), L.ptr(t['w2']),
                         L.ptr(t['w2f']), L.ptr(t['c1']), L.ptr(t['c2']), None, None, L.ptr(t['w1l']), L.ptr(t['l1c1']), L.ptr(t['l1c2']))
        return fw, F, Fp

    def _pack_attn(self, a, keep, fused):
        f32c = lambda x: x.detach().float().contiguous()
        I = a.to_q.weight.shape[0]
        D = a.to_q.weight.shape[1]
        if fused:
            wqkv = torch.cat([a.to_q.weight.detach(), a.to_kv.weight.detach()], dim=0).to(bf16).contiguous()
            wq_ptr = wqkv.data_ptr()
            wkv_ptr = wq_ptr + I * D * 2
            wkeep = wqkv
        else:
            wq = a.to_q.weight.detach().to(bf16).contiguous()
            wkv = a.to_kv.weight.detach().to(bf16).contiguous()
            wq_ptr, wkv_ptr, wkeep = wq.data_ptr(), wkv.data_ptr(), (wq, wkv)
        t = dict(g=f32c(a.norm.gamma), b=f32c(a.norm.beta), w=wkeep, wo=a.to_out.weight.detach().to(bf16).contiguous(),
                 nk=f32c(a.null_kv[0, :, 0, :]), nv=f32c(a.null_kv[1, :, 0, :]), qs=f32c(a.q_scale), ks=f32c(a.k_scale))
        # the block's LayerNorm folded into its first projection (mm_attn_weights.w_q_ln): self-attention q|k|v (normalises the kv input too,
        # mmp.py:137-141), cross-attention q only (its k|v read the context, which is not normalised)
        wf = (torch.cat([a.to_q.weight.detach(), a.to_kv.weight.detach()], dim=0) if fused else a.to_q.weight.detach()).float()
        t['wl'] = (wf * t['g'][None, :]).to(bf16).contiguous()
        t['lc1'] = t['wl'].float().sum(dim=1).contiguous()
        t['lc2'] = (wf @ t['b']).contiguous() if bool((t['b'] != 0).any()) else None
        keep.append(t)
        return L.AttnWeights(L.ptr(t['g']), L.ptr(t['b']), C.c_void_p(wq_ptr), C.c_void_p(wkv_ptr), L.ptr(t['wo']),
                             L.ptr(t['nk']), L.ptr(t['nv']), L.ptr(t['qs']), L.ptr(t['ks']), None, None, None, L.ptr(t['wl']), L.ptr(t['lc1']), L.ptr(t['lc2']))

    def _model(self):
        L.require_device()
        dev = self.token_emb.weight.device
        if dev.type != 'cuda':
            raise L.MuseHipError('Transformer parameters are not on the GPU; the MI355X path has no CPU fallback')
        key = self._pack_key()
        if self.precision in SPLIT_TIERS:
            key = key + (self.precision, self._x3_min_products, tuple((w.data_ptr(), w._version) for w in self._x3_extra_weights))
            if self._handle_x3 is None or self._handle_x3_key != key:
                self._handle_x3, self._handle_x3_key = self._model_x3(), key
            return self._handle_x3
        if self.precision == 'fp8':
            if self._handle_f8 is None or self._handle_f8_key != key:
                self._handle_f8, self._handle_f8_key = self._model_fp8(), key
            return self._handle_f8
        if self._handle is not None and self._handle_key == key:
            return self._handle
        h = _Handle()
        tb = self.transformer_blocks
        cfg = tb.cfg
        layers = (L.LayerWeights * cfg['depth'])()
        F = Fp = 0
        for i, (sa, ca, ff) in enumerate(tb.layers):
            layers[i].self_attn = self._pack_attn(sa, h.keep, fused=True)
            layers[i].cross_attn = self._pack_attn(ca, h.keep, fused=False)
            layers[i].ff, F, Fp = self._pack_ff(ff, h.keep)
        sc_ff, _, _ = self._pack_ff(self.self_cond_to_init_embed, h.keep)
        f32c = lambda x: x.detach().float().contiguous()
        t = dict(tok=self.token_emb.weight.detach().to(bf16).contiguous(), pos=self.pos_emb.weight.detach().to(bf16).contiguous(),
                 fg=f32c(tb.norm.gamma), fb=f32c(tb.norm.beta), wl=self.to_logits.weight.detach().to(bf16).contiguous(),
                 tp=self.text_embed_proj.weight.detach().to(bf16).contiguous() if isinstance(self.text_embed_proj, nn.Linear) else None)
        t['wmean'] = t['wcov'] = None                                  # filled by _Handle.ensure_logits_stats on the first fused generate()
        if self.dim_out % 256 == 0 and self.dim_out >= 4096:
            h.stats_src = lambda t=t: t['wl'].float()
        h.keep.append(t)
        h.keep.append(layers)
        d = L.TransformerDesc()
        d.dim, d.depth, d.heads, d.dim_head = self.dim, cfg['depth'], cfg['heads'], cfg['dim_head']
        d.ff_inner, d.ff_inner_padded = F, Fp
        d.seq_len, d.num_tokens, d.vocab_rows, d.dim_out = self.seq_len, self.num_tokens, self.token_emb.weight.shape[0], self.dim_out
        d.text_dim, d.self_cond = self.text_embed_dim, int(bool(self.self_cond))
        d.token_emb, d.pos_emb, d.text_proj = L.ptr(t['tok']), L.ptr(t['pos']), L.ptr(t['tp'])
        d.layers = C.cast(layers, C.POINTER(L.LayerWeights))
        d.final_gamma, d.final_beta, d.to_logits = L.ptr(t['fg']), L.ptr(t['fb']), L.ptr(t['wl'])
        d.self_cond_ff = sc_ff
        d.logits_wmean, d.logits_wcov = L.ptr(t['wmean']), L.ptr(t['wcov'])
        # 'auto' (ADVICE r5): the probe's verdict is kept PER MODEL OBJECT (`_ln_fold_auto`), not per packed copy -- an optimizer step repacks the weights every
        # iteration, and re-probing each time cost a host synchronisation + a handle re-creation per step and could flip the engine between steps.  It is
        # forgotten only where the VALUES can change wholesale: load_state_dict / invalidate_packed_weights / .to().
        decided = getattr(self, '_ln_fold_auto', None)
        d.ln_fold_off = int(self.layernorm_fold is False or (self.layernorm_fold == 'auto' and decided is False))
        h.desc = d
        if self.layernorm_fold == 'auto' and decided is None:
            h.owner = weakref.ref(self)
            h.begin_ln_probe(dev)
        h.create(d, t)
        self._handle, self._handle_key = h, key
        return h

    def set_layernorm_fold(self, mode='auto'):
        """bf16 engine.  The three LayerNorm(dim)s of a block ride in the GEMMs around them (csrc/model.hip ln_fold_on): the projections multiply the bf16 image of
        the RAW residual row and apply rstd * (acc - mean * c1) + c2 on their accumulators.  The bf16 rounding then acts on x instead of LayerNorm(x): its error in
        normalised units grows like |x^ + mean / sigma|, i.e. with the row's DC offset -- invisible for random-init-like statistics (|mean| / sigma ~ 0.05), a
        silent loss of accuracy for a checkpoint whose residual stream carries massive-activation channels or a large mean.
        'auto' (default): the first call through the model records max |mean| / sigma over every folded LayerNorm input (one extra host
        synchronisation); above MM_LN_FOLD_MAX_RATIO the model falls back to the LayerNorm kernels and that call is recomputed.  The verdict is kept for the
        model object until its values are replaced (load_state_dict / invalidate_packed_weights / .to()); repacks after optimizer steps reuse it.  It depends
        on the FIRST inputs the model sees: two processes that must produce identical ids should pin the engine with True / False.
        True / False force the choice.  `layernorm_fold_ratio` holds the probe's result."""
        if mode not in ('auto', True, False):
            raise ValueError("layernorm fold mode must be 'auto', True or False")
        if mode != self.layernorm_fold:
            self.layernorm_fold = mode
            self._handle, self._handle_key = None, None
            self._pack_gen = getattr(self, '_pack_gen', 0) + 1
        return self

    @property
    def layernorm_fold_ratio(self):
        r = self._handle.ln_ratio if self._handle is not None else None
        return r if r is not None else getattr(self, '_ln_fold_ratio', None)

    def linear_weights(self):
        """every nn.Linear weight of the hot path (what the precision tier packs as bf16 term segments)"""
        ws = [self.to_logits.weight]
        if isinstance(self.text_embed_proj, nn.Linear):
            ws.append(self.text_embed_proj.weight)
        ffs = [ff for _, _, ff in self.transformer_blocks.layers] + [self.self_cond_to_init_embed]
        for ff in ffs:
            ws += [ff[1].weight, ff[4].weight]
        for sa, ca, _ in self.transformer_blocks.layers:
            for a in (sa, ca):
                ws += [a.to_q.weight, a.to_kv.weight, a.to_out.weight]
        return ws

    def split_scale(self):
        """'f16x2' tier: the power of two the weight terms are packed with (ops.f16_weight_scale over every Linear weight of the hot path and the
        self-critic head, when one rides along); every GEMM multiplies its accumulators by the inverse (mm_transformer_desc.split_alpha)"""
        return ops.f16_weight_scale(self.linear_weights() + list(self._x3_extra_weights))

    def split_products(self):
        """term pairs per product of the precision tier for THIS checkpoint.  'bf16x3': 3 when every Linear weight is bf16-representable (a
        checkpoint trained / stored in bf16), 5 for two-term weights, 6 for general fp32 weights.  'f16x2': 2 when one fp16 term holds every
        weight (any bf16-representable checkpoint), 3 for general fp32 weights (csrc/split.hip, common.h split2_f16)"""
        f16 = self.precision == 'f16x2'
        key = self._pack_key() + (f16,) + tuple((w.data_ptr(), w._version) for w in self._x3_extra_weights)      # (the extra weights move the fp16 term scale: ADVICE r4)
        if getattr(self, '_x3_terms', None) is None or self._x3_terms[0] != key:      # one pass over the weights per parameter version
            if f16:
                sc = self.split_scale()
                self._x3_terms = (key, max(ops.weight_terms_f16(w, sc) for w in self.linear_weights()))
            else:
                self._x3_terms = (key, max(ops.weight_terms(w) for w in self.linear_weights()))
        need = (1 + self._x3_terms[1]) if f16 else ops.products_for_terms(self._x3_terms[1])
        return max(need, self._x3_min_products)

    def split_code(self):
        """the operand code of the current precision tier (mm_transformer_desc.split_products): the product count, | MM_SPLIT_F16 for fp16 terms"""
        return self.split_products() | (ops.MM_SPLIT_F16 if self.precision == 'f16x2' else 0)

    def _model_x3(self):
        """weights of the 'bf16x3' precision tier (mm_transformer_desc.split_products): Linear weights as term-segment packs [out][P*in], fp32
        tables / norms, plain (not GEGLU-interleaved) w1 with both halves padded to Fp"""
        h = _Handle()
        tb = self.transformer_blocks
        cfg = tb.cfg
        PC = self.split_code()
        P = ops.split_count(PC)
        scale = self.split_scale() if ops.split_is_f16(PC) else 1.0
        f32c = lambda x: x.detach().float().contiguous()
        pack = lambda w, pad_k=1: ops.split_pack_weight(w, PC, pad_k, scale)

        def pack_ff(ff):
            w1, w2 = ff[1].weight.detach().float(), ff[4].weight.detach().float()
            F, D = w2.shape[1], w1.shape[1]
            Fp = (F + 63) // 64 * 64
            w1p = torch.zeros(2 * Fp, D, dtype=torch.float32, device=w1.device)
            w1p[:F] = w1[:F]
            w1p[Fp:Fp + F] = w1[F:]
            t = dict(g1=f32c(ff[0].gamma), b1=ff[0].beta.float().contiguous(), w1=pack(w1p), g2=ops.pad_cols(f32c(ff[3].gamma), Fp),
                     b2=ops.pad_cols(ff[3].beta.float(), Fp), w2=pack(w2, 64))
            t['w1g'] = t['w2f'] = t['c1'] = t['c2'] = None
            if ops.split_is_f16(PC):
                # round 5 (csrc/gemm_terms.hip, model.hip ff_block): w1 as term segments of its GEGLU-interleaved rows (GEGLU + the term split + the LayerNorm(inner)
                # partial sums ride in its epilogue) and LayerNorm(inner) folded into w2 -- gains into the weight terms, mean / bias terms as two [D] vectors
                # taken from the terms the GEMM really multiplies.  Only when the gain-folded weight fits the model's term count and fp16 range.
                w2g = ops.pad_cols(w2, 64) * t['g2'][None, :]
                if float(w2g.abs().max()) * scale <= 65504. and ops.weight_terms_f16(w2g, scale) <= P - 1:
                    hh, ll = ops.split_terms_f16(w2g, scale)
                    rep = hh.double() + (ll.double() if P > 2 else 0.)
                    t['w2f'] = ops.split_pack_weight(w2g, PC, 64, scale)
                    t['c1'] = (rep.sum(dim=1) / scale).float().contiguous()
                    t['c2'] = (ops.pad_cols(w2, 64).double() * t['b2'].double()[None, :]).sum(dim=1).float().contiguous()
                    t['w1g'] = pack(ops.pack_w1_geglu(ff[1].weight.detach().float(), Fp, dtype=torch.float32))
            h.keep.append(t)
            return L.FFWeights(ln1_gamma=L.ptr(t['g1']), ln1_beta=L.ptr(t['b1']), w1=L.ptr(t['w1']), ln2_gamma=L.ptr(t['g2']), ln2_beta=L.ptr(t['b2']), w2=L.ptr(t['w2']),
                               w2_folded=L.ptr(t['w2f']), ln2_c1=L.ptr(t['c1']), ln2_c2=L.ptr(t['c2']), w1_terms_geglu=L.ptr(t['w1g'])), F, Fp

        def pack_attn(a, fused):
            I, D = a.to_q.weight.shape
            if fused:
                wqkv = pack(torch.cat([a.to_q.weight.detach(), a.to_kv.weight.detach()], dim=0))
                wq_ptr = wqkv.data_ptr()
                wkv_ptr, wkeep = wq_ptr + I * P * D * 2, wqkv
            else:
                wq, wkv = pack(a.to_q.weight), pack(a.to_kv.weight)
                wq_ptr, wkv_ptr, wkeep = wq.data_ptr(), wkv.data_ptr(), (wq, wkv)
            t = dict(g=f32c(a.norm.gamma), b=f32c(a.norm.beta), w=wkeep, wo=pack(a.to_out.weight), nk=f32c(a.null_kv[0, :, 0, :]),
                     nv=f32c(a.null_kv[1, :, 0, :]), qs=f32c(a.q_scale), ks=f32c(a.k_scale))
            h.keep.append(t)
            return L.AttnWeights(L.ptr(t['g']), L.ptr(t['b']), C.c_void_p(wq_ptr), C.c_void_p(wkv_ptr), L.ptr(t['wo']), L.ptr(t['nk']), L.ptr(t['nv']),
                                 L.ptr(t['qs']), L.ptr(t['ks']))
        layers = (L.LayerWeights * cfg['depth'])()
        F = Fp = 0
        for i, (sa, ca, ff) in enumerate(tb.layers):
            layers[i].self_attn = pack_attn(sa, True)
            layers[i].cross_attn = pack_attn(ca, False)
            layers[i].ff, F, Fp = pack_ff(ff)
        sc_ff, _, _ = pack_ff(self.self_cond_to_init_embed)
        t = dict(tok=f32c(self.token_emb.weight), pos=f32c(self.pos_emb.weight), fg=f32c(tb.norm.gamma), fb=f32c(tb.norm.beta), wl=pack(self.to_logits.weight),
                 tp=pack(self.text_embed_proj.weight) if isinstance(self.text_embed_proj, nn.Linear) else None, P=P, PC=PC, scale=scale, alpha=1.0 / scale)
        t['wmean'] = t['wcov'] = None
        if self.dim_out % 256 == 0 and self.dim_out >= 4096:      # (an estimate: the fp32 weights as they are)
            h.stats_src = lambda: self.to_logits.weight.detach().float()
        h.keep.append(t)
        h.keep.append(layers)
        d = L.TransformerDesc()
        d.dim, d.depth, d.heads, d.dim_head = self.dim, cfg['depth'], cfg['heads'], cfg['dim_head']
        d.ff_inner, d.ff_inner_padded = F, Fp
        d.seq_len, d.num_tokens, d.vocab_rows, d.dim_out = self.seq_len, self.num_tokens, self.token_emb.weight.shape[0], self.dim_out
        d.text_dim, d.self_cond = self.text_embed_dim, int(bool(self.self_cond))
        d.token_emb, d.pos_emb, d.text_proj = L.ptr(t['tok']), L.ptr(t['pos']), L.ptr(t['tp'])
        d.layers = C.cast(layers, C.POINTER(L.LayerWeights))
        d.final_gamma, d.final_beta, d.to_logits = L.ptr(t['fg']), L.ptr(t['fb']), L.ptr(t['wl'])
        d.self_cond_ff = sc_ff
        d.logits_wmean, d.logits_wcov = L.ptr(t['wmean']), L.ptr(t['wcov'])
        d.split_products = PC
        d.split_alpha = 1.0 / scale
        h.create(d, t)
        return h

    def _pack_key(self):
        """identity of the packed (bf16 / fp8, kernel-layout) weight copies: device + (storage pointer, in-place version) of every parameter and
        buffer.  `.data` surgery that keeps the storage and does not bump the version counter (EMA `p.data.copy_`, `p.data.lerp_`) is NOT
        visible here: call `invalidate_packed_weights()` after such edits."""
        ts = list(self.parameters()) + list(self.buffers())
        return (str(self.token_emb.weight.device),) + tuple((t.data_ptr(), t._version) for t in ts)

    def invalidate_packed_weights(self):
        """Drop the packed device copies of the weights (rebuilt on the next call).  Needed only after edits the version counters cannot
        see (`param.data.copy_(...)`, raw pointer writes); `load_state_dict`, `.to()`, optimizer steps and in-place ops are detected."""
        self._drop_packed()
        return self

    def _drop_packed(self):
        """every cache derived from the parameter VALUES: the three engines' handles and the term count of the precision tier (keyed on
        `_pack_key()`, which by its own docstring cannot see `.data` surgery -- so an explicit invalidation must not leave it behind)"""
        self._handle, self._handle_key = None, None
        self._handle_f8, self._handle_f8_key = None, None
        self._handle_x3, self._handle_x3_key = None, None
        self._x3_terms = None
        self._ln_fold_auto = None      # (the 'auto' LayerNorm-fold verdict belongs to the values that were just replaced)
        self._pack_gen = getattr(self, '_pack_gen', 0) + 1      # (part of the hipGraph cache key -- `_pack_key()` cannot see `.data` surgery)

    def _apply(self, fn, *args, **kwargs):
        self._drop_packed()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._drop_packed()
        return super().load_state_dict(*args, **kwargs)

    def set_precision(self, precision):
        """'bf16' (default): the production engine -- bf16 operands, fp32 accumulation / residual stream / logits.
        'bf16x3': the tolerance-meeting tier INSIDE the same C entry points (mm_transformer_forward / mm_generate): activations that feed a
        Linear are kept as exact three-term bf16 splits and multiplied as 3 (bf16-representable checkpoint) / 5 / 6 (general fp32 weights)
        term products on the bf16 matrix pipe, fp32 everywhere else, attention on the fp32 MFMA (csrc/split.hip, attention_f32.hip);
        logits within 1e-3 of the reference's fp32 run and bit-equal ids at full size.  Inference only.
        'f16x2' (round 4): the same tier on fp16 TERMS and the fp16 MFMA -- two terms per value (22 significand bits), 3 term products for
        GENERAL fp32 weights and 2 for a bf16-representable checkpoint (half of what 'bf16x3' needs), weight terms scaled by a power of two so that
        low terms stay normal numbers; logits within 1e-3 (measured 4e-6) and bit-equal ids at full size on both kinds of checkpoint.  Inference only.
        'parity': precision level L0 -- fp32 storage and fp32 MFMA through the reference's exact operator sequence, one
        operator call at a time from Python (parity.py / csrc/parity.hip): the verification baseline of the tier above.  Inference only.
        'fp8': the fp8 engine (BASELINE configs[4]) inside the same C entry points -- e4m3 weights and activations on the K = 128 fp8 MFMA for the
        Linear layers of the blocks (quantize_weights_fp8); self-defined numerics.  Inference only."""
        if precision not in ('bf16', 'bf16x3', 'f16x2', 'parity', 'fp8'):
            raise ValueError(f"precision must be 'bf16', 'f16x2', 'bf16x3', 'parity' or 'fp8', got {precision!r}")
        self.precision = precision
        return self

    def _workspace(self, nbytes, device):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            # zero-filled once per (re)allocation: every buffer carved from it is written before it is read, but no result may ever
            # depend on what a previous owner of the memory left there
            self._ws = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
        return self._ws

    # ---- context (mmp.py:302-318)
    def _context(self, text_embeds, conditioning_token_ids, cond_drop_prob):
        h = self._model()
        dev = self.token_emb.weight.device
        te = text_embeds.to(device=dev, dtype=torch.float32).contiguous()
        b, Lt, td = te.shape
        assert td == self.text_embed_dim, f'text embeds have dim {td}, transformer expects {self.text_embed_dim}'
        cids, nc = None, 0
        if exists(conditioning_token_ids):
            cids = conditioning_token_ids.reshape(b, -1).to(device=dev, dtype=torch.long).contiguous()
            nc = cids.shape[1]
        m = Lt + nc
        seg = h.packed['P'] if self.precision in SPLIT_TIERS else 1          # precision tier: P term segments per context row
        ctx = torch.empty(b, m, seg * self.dim, dtype=bf16, device=dev)
        mask = torch.empty(b, m, dtype=torch.uint8, device=dev)
        wsb = L.lib().mm_context_workspace_bytes(h.ptr, b, Lt)
        ws = torch.empty(int(wsb), dtype=torch.uint8, device=dev)
        L.check(L.lib().mm_transformer_context(h.ptr, L.stream(), L.ptr(te), b, Lt, L.ptr(cids), nc, int(cond_drop_prob == 1),
                                               L.ptr(ctx), L.ptr(mask), L.ptr(ws), wsb), 'mm_transformer_context')
        if 0. < cond_drop_prob < 1.:       # per-sample text drop for training-time CFG (mmp.py:308-310, 393-399)
            keep = (torch.rand((b, 1), device=dev) < (1. - cond_drop_prob)).to(torch.uint8)
            mask[:, :Lt] *= keep
        return ctx, mask

    def _run(self, ids, ctx, mask, self_cond_embed=None, want_embed=True, want_logits=True):
        assert self.precision in ('bf16', 'fp8') + SPLIT_TIERS
        h = self._model()
        dev = self.token_emb.weight.device
        ids = ids.to(device=dev, dtype=torch.long).contiguous()
        b, n = ids.shape
        assert n <= self.seq_len                                              # mmp.py:293
        m = ctx.shape[1]
        seg = h.packed['P'] if self.precision in SPLIT_TIERS else 1          # precision tier: the embed leaves as P term segments per row (see _embed_f32)
        embed = torch.empty(b * n, seg * self.dim, dtype=bf16, device=dev) if want_embed else None
        logits = torch.empty(b * n, self.dim_out, dtype=torch.float32, device=dev) if want_logits else None
        sce = None
        if self.self_cond and exists(self_cond_embed):
            sce = self_cond_embed.to(device=dev, dtype=torch.float32).reshape(b * n, self.dim).contiguous()
        wsb = L.lib().mm_transformer_workspace_bytes(h.ptr, b, n, m)
        ws = self._workspace(wsb, dev)
        L.check(L.lib().mm_transformer_forward(h.ptr, L.stream(), L.ptr(ids), b, n, L.ptr(ctx), L.ptr(mask), m, L.ptr(sce),
                                               L.ptr(embed), L.ptr(logits), L.ptr(ws), ws.numel()), 'mm_transformer_forward')
        if h.ln_probe is not None and not torch.cuda.is_current_stream_capturing() and h.finish_ln_probe():
            # the LayerNorm(dim) fold was probed on this call and found unsafe for this checkpoint: once more on the LayerNorm kernels
            L.check(L.lib().mm_transformer_forward(h.ptr, L.stream(), L.ptr(ids), b, n, L.ptr(ctx), L.ptr(mask), m, L.ptr(sce),
                                                   L.ptr(embed), L.ptr(logits), L.ptr(ws), ws.numel()), 'mm_transformer_forward')
        return embed, logits

    def cross_attention_block(self, layer, x, context, context_mask=None):
        """The cross-attention block of `layer` as an operator (mmp.py:139-162, 191): returns x + CrossAttention(LayerNorm(x), context) for x fp32 [b, n, dim],
        context [b, m, dim] (rounded to bf16, what the engine's context is), context_mask bool [b, m] or None -- computed exactly as the model's forward computes
        the block (one-kernel form on the headline shape class).  bf16 engine only; a test surface (tests/test_gpu_ops.py)."""
        assert self.precision == 'bf16'
        h = self._model()
        dev = self.token_emb.weight.device
        xs = x.to(device=dev, dtype=torch.float32).contiguous().clone()
        b, n, _ = xs.shape
        ctx = context.to(device=dev, dtype=bf16).contiguous()
        m = ctx.shape[1]
        km = context_mask.to(device=dev, dtype=torch.uint8).contiguous() if exists(context_mask) else None
        wsb = L.lib().mm_cross_attention_block_workspace_bytes(h.ptr, b, n, m)
        ws = torch.zeros(int(wsb), dtype=torch.uint8, device=dev)
        L.check(L.lib().mm_cross_attention_block(h.ptr, L.stream(), int(layer), L.ptr(xs), b, n, L.ptr(ctx), L.ptr(km), m, L.ptr(ws), ws.numel()), 'mm_cross_attention_block')
        return xs

    def _embed_f32(self, embed):
        """fp32 [rows, dim] view of what `_run` / `forward(_embed_only=True)` returned: the bf16 embed, or -- precision tier -- the exact sum of
        its three bf16 terms"""
        if self.precision in SPLIT_TIERS:
            return ops.unsplit_rows(embed, self._model().packed['PC'], self.dim)
        return embed.float()

    # ---- fp8 engine (BASELINE configs[4] "fp8 MFMA weights"): e4m3 weights AND activations on the K = 128 fp8 MFMA, inside the same C entry points
    def quantize_weights_fp8(self, enabled=True):
        """Switch inference to the fp8 engine (`set_precision('fp8')`): the Linear weights of the blocks are quantised per output row to OCP e4m3
        (`mm_quantize_e4m3_rows`), their input activations per token row inside the producing kernels, and the products run on
        v_mfma_f32_16x16x128_f8f6f4 (`mm_gemm_fp8`) inside `mm_transformer_forward` / `mm_generate`.  Embeddings, norms, the text projection, the
        cross-attention's key/value projection of the context, attention, to_logits and the sampling are the bf16 engine's.  Self-defined oracle
        (SURVEY 8c "L2"): the fp32 restatement with the same per-row fake quantisation at the same Linear inputs, run on `fp8_dequantized_state_dict()`
        (tests/test_gpu_fp8_engine.py)."""
        return self.set_precision('fp8' if enabled else 'bf16')

    FP8_LINEARS = ('0.to_q.weight', '0.to_kv.weight', '0.to_out.weight', '1.to_q.weight', '1.to_out.weight', '2.1.weight', '2.4.weight')

    def fp8_dequantized_state_dict(self):
        """state_dict with every Linear weight the fp8 engine quantises replaced by its de-quantised value (what the engine multiplies by)."""
        sd = {k: v.detach().clone() for k, v in self.state_dict().items()}

        def dq(w):
            wq, sc = ops.quantize_e4m3_rows(w)
            return (wq[:, :w.shape[1]].view(torch.float8_e4m3fn).float() * sc[:, None]).to(w.dtype)
        for k in list(sd):
            if (k.startswith('transformer_blocks.layers.') and k.split('.', 3)[3] in self.FP8_LINEARS) or \
                    (k.startswith('self_cond_to_init_embed.') and k.endswith('.weight')):
                sd[k] = dq(sd[k])
        return sd

    def _model_fp8(self):
        """weights of the fp8 engine (mm_transformer_desc.fp8): the layers' Linear weights as e4m3 rows + per-row scales, everything else as the bf16
        engine packs it; the feed-forward's inner width is padded to a multiple of 128 (the fp8 MFMA's k-step)"""
        h = _Handle()
        tb = self.transformer_blocks
        cfg = tb.cfg
        if self.dim % 128 or (cfg['heads'] * cfg['dim_head']) % 128:
            raise NotImplementedError('the fp8 engine needs dim and heads * dim_head to be multiples of 128')
        f32c = lambda x: x.detach().float().contiguous()
        q = ops.quantize_e4m3_rows

        def pack_ff(ff):
            w1, w2 = ff[1].weight.detach(), ff[4].weight.detach()
            F = w2.shape[1]
            Fp = (F + 127) // 128 * 128
            w1q, w1s = q(ops.pack_w1_geglu(w1.float(), Fp, dtype=torch.float32))   # the bf16 engine's GEGLU interleave as a row permutation of the fp32 weight, quantised row by row
            w2p = torch.zeros(w2.shape[0], Fp, dtype=torch.float32, device=w2.device)
            w2p[:, :F] = w2.float()
            w2q, w2s = q(w2p)
            t = dict(g1=f32c(ff[0].gamma), b1=ff[0].beta.float().contiguous(), w1=w1q, w1s=w1s, g2=ops.pad_cols(f32c(ff[3].gamma), Fp),
                     b2=ops.pad_cols(ff[3].beta.float(), Fp), w2=w2q, w2s=w2s)
            h.keep.append(t)
            return L.FFWeights(L.ptr(t['g1']), L.ptr(t['b1']), L.ptr(t['w1']), L.ptr(t['g2']), L.ptr(t['b2']), L.ptr(t['w2']), None, None, None,
                               L.ptr(t['w1s']), L.ptr(t['w2s'])), F, Fp

        def pack_attn(a, fused):
            I, D = a.to_q.weight.shape
            t = dict(g=f32c(a.norm.gamma), b=f32c(a.norm.beta), nk=f32c(a.null_kv[0, :, 0, :]), nv=f32c(a.null_kv[1, :, 0, :]), qs=f32c(a.q_scale),
                     ks=f32c(a.k_scale))
            t['wo'], t['wos'] = q(a.to_out.weight)
            if fused:      # q | k | v as one e4m3 matrix with one scale vector: a single GEMM
                t['w'], t['ws'] = q(torch.cat([a.to_q.weight.detach(), a.to_kv.weight.detach()], dim=0))
                wq_ptr, wkv_ptr = t['w'].data_ptr(), t['w'].data_ptr() + I * D
                wqs_ptr, wkvs_ptr = t['ws'].data_ptr(), t['ws'].data_ptr() + I * 4
            else:          # cross-attention: q on the fp8 MFMA; the context's k | v projection stays bf16 (once per generate)
                t['w'], t['ws'] = q(a.to_q.weight)
                t['wkv'] = a.to_kv.weight.detach().to(bf16).contiguous()
                wq_ptr, wkv_ptr, wqs_ptr, wkvs_ptr = t['w'].data_ptr(), t['wkv'].data_ptr(), t['ws'].data_ptr(), None
            h.keep.append(t)
            return L.AttnWeights(L.ptr(t['g']), L.ptr(t['b']), C.c_void_p(wq_ptr), C.c_void_p(wkv_ptr), L.ptr(t['wo']), L.ptr(t['nk']), L.ptr(t['nv']),
                                 L.ptr(t['qs']), L.ptr(t['ks']), C.c_void_p(wqs_ptr), C.c_void_p(wkvs_ptr), L.ptr(t['wos']))
        layers = (L.LayerWeights * cfg['depth'])()
        F = Fp = 0
        for i, (sa, ca, ff) in enumerate(tb.layers):
            layers[i].self_attn = pack_attn(sa, True)
            layers[i].cross_attn = pack_attn(ca, False)
            layers[i].ff, F, Fp = pack_ff(ff)
        sc_ff, _, _ = pack_ff(self.self_cond_to_init_embed)
        t = dict(tok=self.token_emb.weight.detach().to(bf16).contiguous(), pos=self.pos_emb.weight.detach().to(bf16).contiguous(),
                 fg=f32c(tb.norm.gamma), fb=f32c(tb.norm.beta), wl=self.to_logits.weight.detach().to(bf16).contiguous(),
                 tp=self.text_embed_proj.weight.detach().to(bf16).contiguous() if isinstance(self.text_embed_proj, nn.Linear) else None)
        t['wmean'] = t['wcov'] = None
        if self.dim_out % 256 == 0 and self.dim_out >= 4096:
            h.stats_src = lambda t=t: t['wl'].float()
        h.keep.append(t)
        h.keep.append(layers)
        d = L.TransformerDesc()
        d.dim, d.depth, d.heads, d.dim_head = self.dim, cfg['depth'], cfg['heads'], cfg['dim_head']
        d.ff_inner, d.ff_inner_padded = F, Fp
        d.seq_len, d.num_tokens, d.vocab_rows, d.dim_out = self.seq_len, self.num_tokens, self.token_emb.weight.shape[0], self.dim_out
        d.text_dim, d.self_cond = self.text_embed_dim, int(bool(self.self_cond))
        d.token_emb, d.pos_emb, d.text_proj = L.ptr(t['tok']), L.ptr(t['pos']), L.ptr(t['tp'])
        d.layers = C.cast(layers, C.POINTER(L.LayerWeights))
        d.final_gamma, d.final_beta, d.to_logits = L.ptr(t['fg']), L.ptr(t['fb']), L.ptr(t['wl'])
        d.self_cond_ff = sc_ff
        d.logits_wmean, d.logits_wcov = L.ptr(t['wmean']), L.ptr(t['wcov'])
        d.fp8 = 1
        h.create(d, t)
        return h

    def _cfg_logits(self, emb_a, emb_b, cond_scale):
        """the guidance-combined logits b + (a - b) * cond_scale of two passes (mmp.py:250-254, 332) from their final embeddings."""
        if self.precision == 'parity':
            return P32.cfg_logits(self, emb_a, emb_b, cond_scale)
        # guidance in the embedding: to_logits is linear, so b + (a - b) * s of the logits is to_logits(e_b + (e_a - e_b) * s): mix, then ONE GEMM
        # (what mm_generate does; every GEMM kernel of the family accumulates in the same order, so the two agree bit for bit)
        if self.precision in SPLIT_TIERS:
            pk = self._model().packed
            return ops.gemm_split(ops.cfg_mix(emb_a, emb_b, cond_scale, self.dim, pk['PC']), pk['wl'], pk['PC'], pk['alpha'], shared=True)
        return ops.gemm(ops.cfg_mix(emb_a, emb_b, cond_scale, self.dim), self._model().packed['wl'], out_f32=True)

    # ---- reference surface
    def forward_with_cond_scale(self, *args, cond_scale=3., return_embed=False, **kwargs):
        """mmp.py:240-259.  Both passes run to the final LayerNorm; the guidance combine null + (cond - null) * cond_scale is applied to the two
        embeddings (mm_cfg_mix; to_logits is linear) and to_logits runs once on the result."""
        if cond_scale == 1:
            return self.forward(*args, return_embed=return_embed, cond_drop_prob=0., **kwargs)
        x = args[0] if args else kwargs.pop('x')
        b, n = x.shape
        emb_c = self.forward(x, *args[1:], _embed_only=True, cond_drop_prob=0., **kwargs)
        emb_n = self.forward(x, *args[1:], _embed_only=True, cond_drop_prob=1., **kwargs)
        scaled = self._cfg_logits(emb_c, emb_n, cond_scale).reshape(b, n, self.dim_out)
        if return_embed:
            return scaled, self._embed_f32(emb_c).reshape(b, n, self.dim)
        return scaled

    def forward_with_neg_prompt(self, x, text_embed: torch.Tensor, neg_text_embed: torch.Tensor, cond_scale=3., return_embed=False, **kwargs):
        """EXTENSION, no reference parity possible: mmp.py:261-277 cannot run (undefined `*args` / `scaled_logits`, and generate
        passes `neg_text_embeds` where the signature says `neg_text_embed`, mmp.py:544).  This is its evident intent: guidance
        away from the negative prompt, neg + (pos - neg) * cond_scale, returning the POSITIVE pass's embed; both passes run with
        their text attended (cond_drop_prob 0) and share the fused to_logits + combine GEMM.  `x` (the token ids) is an added
        first argument -- the broken original never received them."""
        b, n = x.shape
        emb_p = self.forward(x, _embed_only=True, cond_drop_prob=0., text_embeds=text_embed, **kwargs)
        emb_n = self.forward(x, _embed_only=True, cond_drop_prob=0., text_embeds=neg_text_embed, **kwargs)
        scaled = self._cfg_logits(emb_p, emb_n, cond_scale).reshape(b, n, self.dim_out)
        if return_embed:
            return scaled, self._embed_f32(emb_p).reshape(b, n, self.dim)
        return scaled

    def forward(self, x, return_embed=False, return_logits=False, labels=None, ignore_index=0, self_cond_embed=None,
                cond_drop_prob=0., conditioning_token_ids: Optional[torch.Tensor] = None, texts: Optional[List[str]] = None,
                text_embeds: Optional[torch.Tensor] = None, _embed_only=False):
        """mmp.py:279-348.  Returns fp32 logits (b, n, dim_out) [and the fp32 view of the bf16 embed]; `_embed_only` (internal)
        returns the bf16 [b*n, dim] embed for the fused CFG GEMM.  With `labels`, autograd enabled and trainable parameters the
        cross-entropy comes from the differentiable MI355X training path (training.py: hand-written backward); otherwise the loss
        is computed forward-only."""
        # (the reference returns (logits, embed) before it looks at labels, mmp.py:334-335; the ignore_index emptiness test only applies to the
        #  cross-entropy head: a TokenCritic's float 0/1 labels may legitimately all equal the default ignore_index 0)
        if (exists(labels) and not return_logits and not return_embed and torch.is_grad_enabled() and self.to_logits.weight.requires_grad
                and (self.dim_out == 1 or bool((labels != ignore_index).any()))):      # CE with all rows ignored: NaN like F.cross_entropy, nothing to differentiate
            assert exists(texts) ^ exists(text_embeds)
            if self.precision != 'bf16':
                raise NotImplementedError(f"precision {self.precision!r} is an inference mode: train with set_precision('bf16')")
            if exists(texts):
                text_embeds = self.encode_text(texts)
            from .training import transformer_loss
            return transformer_loss(self, x, text_embeds, labels, ignore_index, cond_drop_prob, grad_sync=getattr(self, 'grad_sync', None),
                                    self_cond_embed=self_cond_embed, conditioning_token_ids=conditioning_token_ids)
        with torch.no_grad():
            return self._forward_no_grad(x, return_embed, return_logits, labels, ignore_index, self_cond_embed, cond_drop_prob,
                                         conditioning_token_ids, texts, text_embeds, _embed_only)

    def _forward_no_grad(self, x, return_embed, return_logits, labels, ignore_index, self_cond_embed, cond_drop_prob,
                         conditioning_token_ids, texts, text_embeds, _embed_only):
        b, n = x.shape
        assert n <= self.seq_len
        assert exists(texts) ^ exists(text_embeds)
        if exists(texts):
            text_embeds = self.encode_text(texts)
        if self.precision == 'parity':
            embed, logits = P32.transformer_run(self, x, text_embeds, cond_drop_prob, conditioning_token_ids, self_cond_embed, want_logits=not _embed_only)
            if _embed_only:
                return embed
            return self._finish_forward(embed, logits, b, n, return_embed, return_logits, labels, ignore_index)
        ctx, mask = self._context(text_embeds, conditioning_token_ids, cond_drop_prob)
        if _embed_only:
            embed, _ = self._run(x, ctx, mask, self_cond_embed, want_embed=True, want_logits=False)
            return embed
        embed, logits = self._run(x, ctx, mask, self_cond_embed)
        return self._finish_forward(embed, logits, b, n, return_embed, return_logits, labels, ignore_index)

    def _finish_forward(self, embed, logits, b, n, return_embed, return_logits, labels, ignore_index):
        if return_embed:
            return logits.reshape(b, n, self.dim_out), self._embed_f32(embed).reshape(b, n, self.dim)
        if not exists(labels):
            return logits.reshape(b, n, self.dim_out)
        # training-forward losses (mmp.py:340-348), forward only: no autograd graph is built on this path
        dev = logits.device
        if self.dim_out == 1:
            loss = ops.bce_loss(logits.reshape(-1), labels.to(dev))
        else:
            loss = ops.ce_loss(logits, labels.to(device=dev, dtype=torch.long).reshape(-1).contiguous(), ignore_index)
        if not return_logits:
            return loss
        return loss, logits.reshape(b, n, self.dim_out)


class SelfCritic(nn.Module):
    """mmp.py:352-374: a Linear(dim, 1) head on the generator's own (cond-pass) embedding."""

    def __init__(self, net):
        super().__init__()
        self.net = net
        self.to_pred = nn.Linear(net.dim, 1)

    def _pred(self, embeds):
        b, n, d = embeds.shape
        if self.net.precision == 'parity':
            return P32.linear_head(embeds.reshape(b * n, d).float().contiguous(), self.to_pred).reshape(b, n, 1)
        if self.net.precision == 'bf16x3':      # Linear(dim, 1) as term products; the head's own weight decides how many
            P = max(self.net._model().packed['P'], ops.products_for_terms(ops.weight_terms(self.to_pred.weight)))
            x = ops.split_rows(embeds.reshape(b * n, d).float().contiguous(), P)
            out = ops.gemm(x, ops.split_pack_weight(self.to_pred.weight, P), out_f32=True)
            return (out + self.to_pred.bias.detach().float()).reshape(b, n, 1)
        if self.net.precision == 'f16x2':       # ... as fp16 term products (all three: the head is tiny), with its own power-of-two scale
            PC, sc = ops.MM_SPLIT_F16 | 3, ops.f16_weight_scale([self.to_pred.weight])
            x = ops.split_rows(embeds.reshape(b * n, d).float().contiguous(), PC)
            out = ops.gemm_split(x, ops.split_pack_weight(self.to_pred.weight, PC, 1, sc), PC, 1.0 / sc, shared=True)
            return (out + self.to_pred.bias.detach().float()).reshape(b, n, 1)
        w = ops.pad_cols(self.to_pred.weight.detach().to(bf16), 64)                    # [1, D] as a 1x1 conv weight
        x = embeds.reshape(b * n, 1, 1, d).to(bf16).contiguous()
        out = ops.conv2d_nhwc(x, w, 1, 1, 1, 1, (0, 0), bias=self.to_pred.bias.detach().float().contiguous(), out_nchw_f32=True)
        return out.reshape(b, n, 1)

    @torch.no_grad()
    def forward_with_cond_scale(self, x, *args, **kwargs):
        _, embeds = self.net.forward_with_cond_scale(x, *args, return_embed=True, **kwargs)
        return self._pred(embeds)

    @torch.no_grad()
    def forward_with_neg_prompt(self, x, *args, **kwargs):
        """EXTENSION (see Transformer.forward_with_neg_prompt): the critic head on the negative-prompt-guided generator embed."""
        _, embeds = self.net.forward_with_neg_prompt(x, *args, return_embed=True, **kwargs)
        return self._pred(embeds)

    def forward(self, x, *args, labels=None, **kwargs):
        """mmp.py:364-374.  With labels and autograd enabled the BCE is differentiable w.r.t. the head AND the generator (training.py)."""
        if exists(labels) and torch.is_grad_enabled() and self.to_pred.weight.requires_grad and not args:
            from .training import transformer_loss
            text_embeds = kwargs.get('text_embeds')
            if not exists(text_embeds):
                text_embeds = self.net.encode_text(kwargs['texts'])
            return transformer_loss(self.net, x, text_embeds, labels, None, kwargs.get('cond_drop_prob', 0.), grad_sync=self.net.grad_sync,
                                    self_cond_embed=kwargs.get('self_cond_embed'), conditioning_token_ids=kwargs.get('conditioning_token_ids'),
                                    head=self.to_pred)
        with torch.no_grad():
            _, embeds = self.net(x, *args, return_embed=True, **kwargs)
            logits = self._pred(embeds)
            if not exists(labels):
                return logits
            return ops.bce_loss(logits.reshape(-1), labels.to(logits.device))


class MaskGitTransformer(Transformer):
    def __init__(self, *args, **kwargs):
        assert 'add_mask_id' not in kwargs
        super().__init__(*args, add_mask_id=True, **kwargs)


class TokenCritic(Transformer):
    def __init__(self, *args, **kwargs):
        assert 'dim_out' not in kwargs
        super().__init__(*args, dim_out=1, **kwargs)


# ------------------------------------------------------------------------------------------------ MaskGit

class MaskGit(nn.Module):
    def __init__(self, image_size, transformer: MaskGitTransformer, noise_schedule: Callable = cosine_schedule,
                 token_critic: Optional[TokenCritic] = None, self_token_critic=False, vae: Optional[VQGanVAE] = None,
                 cond_vae: Optional[VQGanVAE] = None, cond_image_size=None, cond_drop_prob=0.5, self_cond_prob=0.9,
                 no_mask_token_prob=0., critic_loss_weight=1.):
        super().__init__()
        if not isinstance(transformer, MaskGitTransformer):
            raise TypeError('transformer must be a MaskGitTransformer')        # what @beartype enforces (mmp.py:427-432)
        if exists(vae) and not isinstance(vae, VQGanVAE):
            raise TypeError('vae must be a VQGanVAE')
        self.vae = vae.copy_for_eval() if exists(vae) else None
        self.cond_vae = cond_vae.eval() if exists(cond_vae) else self.vae
        assert not (exists(cond_vae) and not exists(cond_image_size)), 'cond_image_size must be specified if conditioning'
        self.image_size = image_size
        self.cond_image_size = cond_image_size
        self.resize_image_for_cond_image = exists(cond_image_size)
        self.cond_drop_prob = cond_drop_prob
        self.transformer = transformer
        self.self_cond = transformer.self_cond
        if exists(self.vae):
            assert self.vae.codebook_size == self.cond_vae.codebook_size == transformer.num_tokens, \
                'transformer num_tokens must be set to be equal to the vae codebook size'
        self.mask_id = transformer.mask_id
        self.noise_schedule = noise_schedule
        assert not (self_token_critic and exists(token_critic))
        self.token_critic = token_critic
        if self_token_critic:
            self.token_critic = SelfCritic(transformer)
        self.critic_loss_weight = critic_loss_weight
        self.self_cond_prob = self_cond_prob
        self.no_mask_token_prob = no_mask_token_prob
        self._gen_ws = None
        self._graphs = {}                      # generate(graph=True): one captured hipGraph per call signature (see _generate_graphed)
        self.fused_sampling_fallbacks = 0      # generate() calls repeated on the logits path (more than 128 rows of one step failed the candidate bound)
        self.fused_bound_switches = 0          # Transformer.fused_bound == 'auto': times a packed model was moved from the Gaussian to the sampled bound
        self.fused_row_fallbacks = 0           # rows whose bound could not be verified and that the on-device per-row fallback finished

    def save(self, path):
        torch.save(self.state_dict(), path)

    def load(self, path):
        path = Path(path)
        assert path.exists()
        self.load_state_dict(torch.load(str(path)))

    def set_precision(self, precision):
        """'bf16' | 'bf16x3' | 'parity' | 'fp8' for the transformer and the token critic (see Transformer.set_precision); the VAEs take 'bf16' or their
        fp32 engine ('parity'); under 'bf16x3' they decode with exact bf16 term products on the bf16 MFMA and encode on the fp32 engine (exact LFQ ids of
        a super-resolution condition image)."""
        self.transformer.set_precision(precision)
        if isinstance(self.token_critic, Transformer):
            self.token_critic.set_precision(precision)
        for v in (self.vae, self.cond_vae):
            if exists(v):
                v.set_precision({'fp8': 'bf16'}.get(precision, precision))      # (the fp8 engine covers the transformer's Linear layers)
        return self

    def _mask_counts(self, timesteps, seq_len, device='cpu'):
        """mmp.py:556-559 with the user's noise_schedule, evaluated up front instead of one host sync per step."""
        out = []
        for t in torch.linspace(0, 1, timesteps):
            out.append(max(int((self.noise_schedule(t) * seq_len).item()), 1))
        return out

    @torch.no_grad()
    @eval_decorator
    def generate(self, texts: List[str], negative_texts: Optional[List[str]] = None, cond_images: Optional[torch.Tensor] = None,
                 fmap_size=None, temperature=1., topk_filter_thres=0.9, can_remask_prev_masked=False,
                 force_not_use_token_critic=False, timesteps=18, cond_scale=3, critic_noise_scale=1,
                 *, text_embeds: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None, noise_kind: str = 'philox',
                 seed: Optional[int] = None, row_offset: int = 0, return_ids: bool = False, trace: Optional[dict] = None,
                 critic_noise: Optional[torch.Tensor] = None, neg_text_embeds: Optional[torch.Tensor] = None, fused_sampling: bool = True,
                 stepwise: bool = False, loop_end_event=None, graph: bool = False, _seed_dev: Optional[torch.Tensor] = None):
        """mmp.py:491-621.  Keyword-only extras (not in the reference): `text_embeds` bypasses the T5 call,
        `noise` (+ `noise_kind` 'gumbel' | 'uniform') injects the per-step noise tensor [T,B,n,V] for parity runs,
        `seed` / `row_offset` key the on-device Philox stream (row_offset = global index of this shard's first
        sample, so sharded runs reproduce the unsharded ids), `trace` receives per-step states, `critic_noise` [T,B,n]
        injects the U(0,1) draws of the token-critic score annealing (mmp.py:601).  `negative_texts` (or `neg_text_embeds`) is an
        EXTENSION: the reference's negative-prompt path cannot run (see Transformer.forward_with_neg_prompt).  `fused_sampling=False` forces
        the decode loop to materialise the logits (same ids; tests / A-B timing); `fused_sampling='deferred'` leaves the device status flag in
        `self.fused_status` instead of reading it (hipGraph capture of the fused path: check it after every replay).  Every decode variant of the reference -- token critic / self critic
        scores, self-conditioning, can_remask_prev_masked, cond_scale == 1 -- runs inside the one mm_generate call; `stepwise=True` runs the same
        loop one operator call at a time from Python instead (tests: the two must agree bit for bit).
        `graph=True` (round 5): the whole call -- decode loop and VAE decode, ~1190 launches -- is captured ONCE per call signature in a hipGraph and replayed
        on later calls; the Philox keys are read from a small device buffer at execution time, so every replay draws fresh noise and its ids / images are
        bit-identical to the eager call with the same `seed` (tests/test_gpu_model.py).  Signatures the graph path does not take (injected noise, critics,
        negative prompts, traces, the stepwise / parity engines) run eagerly."""
        if graph and not torch.cuda.is_current_stream_capturing():
            if (not exists(noise) and not exists(trace) and not exists(negative_texts) and not exists(neg_text_embeds) and not stepwise and not exists(loop_end_event)
                    and not exists(critic_noise) and self.transformer.precision != 'parity' and fused_sampling is True
                    and not (exists(self.token_critic) and not force_not_use_token_critic)):
                return self._generate_graphed(texts, cond_images, fmap_size, temperature, topk_filter_thres, can_remask_prev_masked, timesteps, cond_scale,
                                              text_embeds, seed, row_offset, return_ids, force_not_use_token_critic, critic_noise_scale)
        tr = self.transformer
        use_token_critic = exists(self.token_critic) and not force_not_use_token_critic
        if exists(negative_texts) or exists(neg_text_embeds):
            assert exists(neg_text_embeds) or len(texts) == len(negative_texts)       # mmp.py:541
        critic = self.token_critic if use_token_critic else None
        critic_net = critic.net if isinstance(critic, SelfCritic) else critic
        if (exists(negative_texts) or exists(neg_text_embeds) or stepwise or tr.precision == 'parity'
                or (exists(critic_net) and critic_net.precision == 'parity')):
            # the negative-prompt extension and the fp8 / parity engines run the loop one step at a time over the same C-ABI operators
            return self._generate_stepwise(texts, cond_images, fmap_size, temperature, topk_filter_thres, can_remask_prev_masked,
                                           use_token_critic, timesteps, cond_scale, critic_noise_scale, text_embeds, noise, noise_kind,
                                           seed, row_offset, return_ids, trace, critic_noise, negative_texts, neg_text_embeds)
        if can_remask_prev_masked and not use_token_critic:                    # mmp.py:548-549
            assert self.no_mask_token_prob > 0., 'without training with some of the non-masked tokens forced to predict, not sure if the logits will be meaningful for these token'
        if exists(fmap_size):
            fmap = fmap_size
        else:
            fmap = self.vae.get_encoded_fmap_size(self.image_size)
        dev = tr.token_emb.weight.device
        seq_len = fmap ** 2
        if not exists(text_embeds):
            text_embeds = tr.encode_text(texts)
        te = text_embeds.to(device=dev, dtype=torch.float32).contiguous()
        B, Lt, _ = te.shape
        cond_ids, nc = None, 0
        if self.resize_image_for_cond_image:
            assert exists(cond_images), 'conditioning image must be passed in to generate for super res maskgit'
            _, cond_ids, _ = self.cond_vae.encode(cond_images)
            cond_ids = cond_ids.reshape(B, -1).contiguous()
            nc = cond_ids.shape[1]
        if tr.precision in SPLIT_TIERS and exists(critic):      # one term-product count for the generator and its critic (token critic network or self-critic head)
            tr._x3_min_products = 0
            tr._x3_extra_weights = ()
            if isinstance(critic, Transformer):
                critic._x3_min_products = 0
                need = critic.split_products()
            elif tr.precision == 'f16x2':      # the head rides on the generator's term scale
                tr._x3_extra_weights = (critic.to_pred.weight,)
                need = 1 + ops.weight_terms_f16(critic.to_pred.weight, tr.split_scale())
            else:
                need = ops.products_for_terms(ops.weight_terms(critic.to_pred.weight))
            tr._x3_min_products = max(tr.split_products(), need)
            if isinstance(critic, Transformer):
                critic._x3_min_products = tr._x3_min_products
        elif tr.precision in SPLIT_TIERS:
            tr._x3_extra_weights = ()
        h = tr._model()
        if fused_sampling and not torch.cuda.is_current_stream_capturing():
            h.ensure_logits_stats(h.auto_bound if tr.fused_bound == 'auto' else tr.fused_bound)      # what the bound of the k-th largest logit needs: first fused generate() only (not inside a capture: it allocates)
        elif fused_sampling == 'deferred' and h.stats_src is not None and not h.fused_ready:
            raise RuntimeError("generate(fused_sampling='deferred') under stream capture needs the vocabulary statistics of to_logits: run one eager "
                               'generate() with the same weights before capturing')
        counts = self._mask_counts(timesteps, seq_len)
        temps = ops.step_temperatures(timesteps, temperature)
        V = tr.dim_out
        k_keep = math.ceil((1 - topk_filter_thres) * V)                       # mmp.py:414
        p = L.GenerateParams()
        p.batch, p.n, p.timesteps, p.k_keep, p.nc, p.L = B, seq_len, timesteps, k_keep, nc, Lt
        p.cond_scale = float(cond_scale)
        kinds = dict(none=L.MM_NOISE_NONE, gumbel=L.MM_NOISE_GUMBEL, uniform=L.MM_NOISE_UNIFORM, philox=L.MM_NOISE_PHILOX)
        p.noise_kind = kinds[noise_kind] if not exists(noise) or noise_kind != 'philox' else L.MM_NOISE_UNIFORM
        if exists(noise):
            noise = noise.to(device=dev, dtype=torch.float32).contiguous()
            assert noise.shape == (timesteps, B, seq_len, V), f'noise must be [T,B,n,V], got {tuple(noise.shape)}'
            assert p.noise_kind in (L.MM_NOISE_GUMBEL, L.MM_NOISE_UNIFORM)
        if not exists(seed):
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())                # follows torch.manual_seed
        p.seed, p.row_offset = seed, row_offset
        if exists(_seed_dev):                                                  # (graph capture: the sampling kernels read {seed, row_offset} from this buffer at execution time)
            p.seed_dev = L.ptr(_seed_dev)
        cnt_arr = (C.c_int32 * timesteps)(*counts)
        tmp_arr = (C.c_float * timesteps)(*temps)
        p.mask_counts, p.temperatures = cnt_arr, tmp_arr
        ids = torch.empty(B, seq_len, dtype=torch.long, device=dev)
        scores = torch.empty(B, seq_len, dtype=torch.float32, device=dev)
        p.text_embeds, p.cond_ids, p.noise, p.ids, p.scores = L.ptr(te), L.ptr(cond_ids), L.ptr(noise), L.ptr(ids), L.ptr(scores)
        keep = []                                                              # tensors the C call reads must outlive it
        if can_remask_prev_masked:
            p.flags |= L.MM_GEN_CAN_REMASK
        if exists(critic):
            if isinstance(critic, SelfCritic):
                if tr.precision in SPLIT_TIERS:
                    hw = ops.split_pack_weight(critic.to_pred.weight.detach().to(dev), h.packed['PC'], 1, h.packed['scale']).reshape(-1).contiguous()
                else:
                    hw = ops.pad_cols(critic.to_pred.weight.detach().to(device=dev, dtype=bf16), 64).reshape(-1).contiguous()
                keep += [hw, critic.to_pred.bias.detach().to(device=dev, dtype=torch.float32).contiguous()]
                p.critic_head_w, p.critic_head_b = L.ptr(keep[-2]), L.ptr(keep[-1])
                ch = h
            else:
                ch = critic._model()
                p.critic = ch.ptr
            cn = critic_noise if exists(critic_noise) else torch.rand(timesteps, B, seq_len, device=dev)      # mmp.py:601
            cn = cn.to(device=dev, dtype=torch.float32).reshape(timesteps, B, seq_len).contiguous()
            keep.append(cn)
            p.critic_noise, p.critic_noise_scale = L.ptr(cn), float(critic_noise_scale)
            cwsb = L.lib().mm_generate_critic_workspace_bytes(ch.ptr, B, seq_len, Lt, nc)
            cws = getattr(self, '_critic_ws', None)
            if cws is None or cws.numel() < cwsb or cws.device != dev:
                cws = self._critic_ws = torch.zeros(int(cwsb), dtype=torch.uint8, device=dev)
            p.critic_workspace, p.critic_workspace_bytes = L.ptr(cws), cws.numel()
        if trace is not None:
            trace['masked_ids'] = torch.empty(timesteps, B, seq_len, dtype=torch.long, device=dev)
            trace['ids'] = torch.empty(timesteps, B, seq_len, dtype=torch.long, device=dev)
            trace['scores'] = torch.empty(timesteps, B, seq_len, dtype=torch.float32, device=dev)
            trace['counts'], trace['temperatures'] = counts, temps
            p.trace_masked_ids, p.trace_ids, p.trace_scores = L.ptr(trace['masked_ids']), L.ptr(trace['ids']), L.ptr(trace['scores'])
        wsb = L.lib().mm_generate_workspace_bytes(h.ptr, B, seq_len, Lt, nc)
        if self._gen_ws is None or self._gen_ws.numel() < wsb or self._gen_ws.device != dev:
            self._gen_ws = torch.zeros(int(wsb), dtype=torch.uint8, device=dev)      # (zero-filled once, see Transformer._workspace)
        # Sampling without the logits round trip (mm_fused_*): every row's k-th largest logit is bounded before its logits exist and the bound is
        # VERIFIED per row on the device; a row it cannot be proven for (heavy-tailed logits) raises `status` and the call is repeated on the
        # logits path -- the ids never depend on the estimate.  Reading the flag is the one host synchronisation of generate().  Under stream
        # capture (hipGraph) the read is not possible: by default capture takes the logits path; with fused_sampling='deferred' the fused path
        # is captured and the flag tensor is left in `self.fused_status` (zeroed inside the graph) for the caller to check after each replay --
        # non-zero means that replay's ids are invalid and the call has to be repeated with fused_sampling=False.
        capturing = torch.cuda.is_current_stream_capturing()
        deferred = fused_sampling == 'deferred'
        status = None
        if fused_sampling and (deferred or not capturing):
            status = torch.zeros(2, dtype=torch.int32, device=dev)      # [0] list overflow (repeat on the logits path), [1] rows finished by the on-device fallback
            p.status = L.ptr(status)
        else:
            p.flags |= L.MM_GEN_NO_FUSED_SAMPLING
        L.check(L.lib().mm_generate(h.ptr, L.stream(), C.byref(p), L.ptr(self._gen_ws), self._gen_ws.numel()), 'mm_generate')
        want_images = exists(self.vae) and (return_ids is False or return_ids == 'both')      # return_ids='both': (ids, images)
        if exists(loop_end_event):
            loop_end_event.record()                                            # (bench.py: the decode loop's end on the stream, before the VAE decode is enqueued)
        images = None
        if deferred:
            self.fused_status = status
            # a captured graph replays reads of every tensor this call handed to the library: they stay alive with the module, not with this frame
            self._deferred_keep = (keep, te, cond_ids, noise, ids, scores, trace, self._gen_ws, getattr(self, '_critic_ws', None), _seed_dev)
        elif status is not None:
            if want_images:      # the VAE decode is LAUNCHED before the flag is read: the one host synchronisation of generate() then waits behind it, not in front
                images = self.vae.decode_from_ids(ids.reshape(B, fmap, fmap))
            st = status.tolist()
            self.fused_row_fallbacks += st[1]
        probed = [x for x in (h, (ch if exists(critic) else None)) if x is not None and x.ln_probe is not None] if not capturing else []
        if sum(int(x.finish_ln_probe()) for x in probed):
            # first generate through a freshly packed model, LayerNorm(dim) fold probed and found unsafe for this checkpoint (Transformer.set_layernorm_fold):
            # the whole call once more on the LayerNorm kernels (the handles were re-created: same packed weights)
            images = None
            if isinstance(critic, Transformer):
                p.critic = ch.ptr
            if status is not None:
                status.zero_()
            L.check(L.lib().mm_generate(h.ptr, L.stream(), C.byref(p), L.ptr(self._gen_ws), self._gen_ws.numel()), 'mm_generate')
            if not deferred and status is not None:
                st = status.tolist()
        if (not deferred and status is not None and not capturing and tr.fused_bound == 'auto' and h.auto_bound == 'gaussian'
                and (st[0] != 0 or st[1] * 200 > B * sum(counts))):
            # 'auto': the Gaussian bound fails on this checkpoint (the whole call, or more than 0.5 % of its sampled rows needed the on-device fallback): this packed
            # model uses the sampled (distribution-free) bound from now on; a failed call is repeated with it before the logits path is considered
            h.auto_bound = 'quantile'
            h.ensure_logits_stats('quantile')
            self.fused_bound_switches += 1
            wsb = L.lib().mm_generate_workspace_bytes(h.ptr, B, seq_len, Lt, nc)      # (the sampled bound carves the rows' logits at the sampled columns)
            if self._gen_ws.numel() < wsb:
                self._gen_ws = torch.zeros(int(wsb), dtype=torch.uint8, device=dev)
            if st[0] != 0:
                images = None
                status.zero_()
                L.check(L.lib().mm_generate(h.ptr, L.stream(), C.byref(p), L.ptr(self._gen_ws), self._gen_ws.numel()), 'mm_generate')
                st = status.tolist()
                self.fused_row_fallbacks += st[1]
        if not deferred and status is not None and st[0] != 0:
            self.fused_sampling_fallbacks += 1
            images = None                                                      # (decoded from ids that are being replaced)
            p.flags, p.status = p.flags | L.MM_GEN_NO_FUSED_SAMPLING, None
            L.check(L.lib().mm_generate(h.ptr, L.stream(), C.byref(p), L.ptr(self._gen_ws), self._gen_ws.numel()), 'mm_generate')
        ids = ids.reshape(B, fmap, fmap)                                       # mmp.py:615
        if not want_images:
            return ids
        images = images if images is not None else self.vae.decode_from_ids(ids)  # mmp.py:620
        return (ids, images) if return_ids == 'both' else images

    def _generate_graphed(self, texts, cond_images, fmap_size, temperature, thres, can_remask, timesteps, cond_scale, text_embeds, seed, row_offset, return_ids,
                          force_not_use_token_critic=False, critic_noise_scale=1):
        """generate(graph=True).  Per call signature: the first call runs eagerly (it packs the weights, computes the vocabulary statistics, probes the
        LayerNorm fold, sizes the workspaces -- none of which may happen under capture), the second captures generate(fused_sampling='deferred') + the VAE
        decode on static input / output buffers, every later one copies its text embeddings (and condition images) into the static inputs, writes
        {seed, row_offset} into the key buffer and replays.  The fused-sampling status words are read after the replay exactly like the eager path reads
        them after its launches (the one host synchronisation); a replay whose status asks for the logits path is repeated eagerly with it.
        What a capture names: the packed weights of the transformer handle and of the VAE(s), the workspaces and the static inputs -- the entry holds
        references to all of them, and the key carries the pack GENERATION of every module (bumped by `invalidate_packed_weights()`, `set_layernorm_fold()`,
        `.to()`, `load_state_dict()`), so surgery the version counters cannot see never replays a stale capture."""
        tr = self.transformer
        dev = tr.token_emb.weight.device
        if not exists(text_embeds):
            text_embeds = tr.encode_text(texts)
        te = text_embeds.to(device=dev, dtype=torch.float32)
        if not exists(seed):
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())                # follows torch.manual_seed
        vkey = self.vae._pack_key() if exists(self.vae) else None
        ckey = self.cond_vae._pack_key() if (exists(cond_images) and self.cond_vae is not self.vae) else None
        hb = tr._model() if tr.precision == 'bf16' else None      # (packed here if need be: the key below names the bound this packed model currently uses)
        # (the C handle owns no device memory -- nothing in csrc/ allocates -- so re-creating it, as the LayerNorm-fold probe of the warm-up call does, leaves a
        #  capture's pointers valid; what a capture depends on are the packed tensors (held below) and the engine choices named in this key)
        gens = (getattr(tr, '_pack_gen', 0), getattr(self.vae, '_pack_gen', 0) if exists(self.vae) else None,
                getattr(self.cond_vae, '_pack_gen', 0) if exists(self.cond_vae) else None)
        key = (tuple(te.shape), None if not exists(cond_images) else tuple(cond_images.shape), fmap_size, float(temperature), float(thres), bool(can_remask), int(timesteps),
               float(cond_scale), return_ids, tr.precision, tr.fused_bound, hb.auto_bound if hb is not None else None, tr._pack_key(), vkey, ckey, gens,
               bool(force_not_use_token_critic), float(critic_noise_scale))
        eager = dict(cond_images=cond_images, fmap_size=fmap_size, temperature=temperature, topk_filter_thres=thres, can_remask_prev_masked=can_remask, timesteps=timesteps,
                     cond_scale=cond_scale, row_offset=row_offset, return_ids=return_ids, force_not_use_token_critic=force_not_use_token_critic,
                     critic_noise_scale=critic_noise_scale)
        entry = self._graphs.get(key)
        if entry is None:
            if len(self._graphs) >= 8:                                          # (weights changed / many shapes: drop the oldest captures)
                self._graphs.pop(next(iter(self._graphs)))
            self._graphs[key] = 'warm'
            return self.generate(texts, text_embeds=te, seed=seed, **eager)
        if entry == 'warm':
            st = dict(te=te.clone(), cond=cond_images.to(dev).clone() if exists(cond_images) else None,
                      keys=torch.zeros(2, dtype=torch.int64, device=dev), keys_host=torch.zeros(2, dtype=torch.int64).pin_memory())
            torch.cuda.synchronize()
            g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    out = self.generate(texts, text_embeds=st['te'], seed=0, _seed_dev=st['keys'], fused_sampling='deferred',
                                        **dict(eager, cond_images=st['cond'], row_offset=0))
            held = (tr._model(), self.vae._pack() if exists(self.vae) else None, getattr(self.vae, '_ws', None),
                    self.cond_vae._pack() if (exists(cond_images) and exists(self.cond_vae)) else None,
                    getattr(self.cond_vae, '_ws', None))      # what the captured launches read or write outside the graph's own pool: alive as long as the capture
            st.update(graph=g, out=out, status=self.fused_status, keep=self._deferred_keep, held=held)
            entry = self._graphs[key] = st
        st = entry
        st['te'].copy_(te, non_blocking=True)
        if exists(cond_images):
            st['cond'].copy_(cond_images, non_blocking=True)
        st['keys_host'][0], st['keys_host'][1] = seed, row_offset
        st['keys'].copy_(st['keys_host'], non_blocking=True)
        st['graph'].replay()
        stw = st['status'].tolist()                                            # the one host synchronisation, behind the VAE decode
        self.fused_row_fallbacks += stw[1]
        if stw[0] != 0:                                                        # more than 128 unverifiable rows in one step: this call on the logits path
            self.fused_sampling_fallbacks += 1
            return self.generate(texts, text_embeds=te, seed=seed, fused_sampling=False, **eager)
        out = st['out']
        return tuple(o.clone() for o in out) if isinstance(out, tuple) else out.clone()

    def _generate_stepwise(self, texts, cond_images, fmap_size, temperature, thres, can_remask, use_critic, timesteps, cond_scale,
                           critic_noise_scale, text_embeds, noise, noise_kind, seed, row_offset, return_ids, trace, critic_noise=None,
                           negative_texts=None, neg_text_embeds=None):
        """mmp.py:556-609 one step at a time through the public operators (general transformer forward over all positions, logits of every
        position materialised): the loop of the negative-prompt extension and of the fp8 / parity engines, and -- `generate(stepwise=True)` --
        the cross-check of mm_generate, which runs the same variants (critics, self-conditioning, can_remask_prev_masked, cond_scale == 1)
        inside one C call."""
        tr = self.transformer
        dev = tr.token_emb.weight.device
        fmap = fmap_size if exists(fmap_size) else self.vae.get_encoded_fmap_size(self.image_size)
        n = fmap ** 2
        if not exists(text_embeds):
            text_embeds = tr.encode_text(texts)
        te = text_embeds.to(device=dev, dtype=torch.float32).contiguous()
        B = te.shape[0]
        if exists(negative_texts) and not exists(neg_text_embeds):
            neg_text_embeds = tr.encode_text(negative_texts)                                         # mmp.py:543
        nte = neg_text_embeds.to(device=dev, dtype=torch.float32).contiguous() if exists(neg_text_embeds) else None
        cond_ids = None
        if self.resize_image_for_cond_image:
            assert exists(cond_images), 'conditioning image must be passed in to generate for super res maskgit'
            _, cond_ids, _ = self.cond_vae.encode(cond_images)
        if can_remask and not use_critic:
            assert self.no_mask_token_prob > 0., 'without training with some of the non-masked tokens forced to predict, not sure if the logits will be meaningful for these token'
        V = tr.dim_out
        k_keep = math.ceil((1 - thres) * V)
        counts = self._mask_counts(timesteps, n)
        temps = ops.step_temperatures(timesteps, temperature)
        kinds = dict(none=L.MM_NOISE_NONE, gumbel=L.MM_NOISE_GUMBEL, uniform=L.MM_NOISE_UNIFORM, philox=L.MM_NOISE_PHILOX)
        kind = kinds[noise_kind] if not exists(noise) or noise_kind != 'philox' else L.MM_NOISE_UNIFORM
        if exists(noise):
            noise = noise.to(device=dev, dtype=torch.float32).contiguous()
        if not exists(seed):
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        ids = torch.full((B, n), self.mask_id, dtype=torch.long, device=dev)
        scores = torch.zeros((B, n), dtype=torch.float32, device=dev)
        self_cond_embed = None
        for step in range(timesteps):
            ops.mask_step(scores, ids, counts[step], self.mask_id, want_rows=False)                  # mmp.py:558-563
            if trace is not None:
                trace.setdefault('masked_ids', []).append(ids.clone())
            is_mask = ids == self.mask_id
            if exists(nte):
                logits, embed = tr.forward_with_neg_prompt(ids, te, nte, cond_scale=cond_scale, return_embed=True, self_cond_embed=self_cond_embed,
                                                           conditioning_token_ids=cond_ids)
            else:
                logits, embed = tr.forward_with_cond_scale(ids, text_embeds=te, self_cond_embed=self_cond_embed,
                                                           conditioning_token_ids=cond_ids, cond_scale=cond_scale, return_embed=True)
            self_cond_embed = embed if self.self_cond else None                                      # mmp.py:574
            pred, conf = ops.sample_rows(logits.reshape(B * n, V), k_keep, temps[step], noise_kind=kind,
                                         noise=noise[step].reshape(B * n, V) if exists(noise) else None, seed=seed,
                                         row_offset=row_offset * n, step=step)                        # mmp.py:576-580, 603-606
            ids = torch.where(is_mask, pred.reshape(B, n), ids)                                      # mmp.py:582-588
            if use_critic:
                if exists(nte):
                    sc = self.token_critic.forward_with_neg_prompt(ids, te, nte, cond_scale=cond_scale, conditioning_token_ids=cond_ids)
                else:
                    sc = self.token_critic.forward_with_cond_scale(ids, text_embeds=te, conditioning_token_ids=cond_ids, cond_scale=cond_scale)
                sc = sc.reshape(B, n).float()
                steps_until_x0 = timesteps - 1 - step
                u = critic_noise[step].to(dev).reshape(sc.shape) if exists(critic_noise) else torch.rand(sc.shape, device=dev)
                scores = (sc + (u - 0.5) * critic_noise_scale * (steps_until_x0 / timesteps)).contiguous()
            else:
                scores = conf.reshape(B, n)
                if not can_remask:
                    scores = scores.masked_fill(~is_mask, -1e5)                                      # mmp.py:608-609
                scores = scores.contiguous()
            if trace is not None:
                trace.setdefault('ids', []).append(ids.clone())
                trace.setdefault('scores', []).append(scores.clone())
        ids = ids.reshape(B, fmap, fmap)
        if return_ids == 'both' and exists(self.vae):
            return ids, self.vae.decode_from_ids(ids)
        if return_ids or not exists(self.vae):
            return ids
        return self.vae.decode_from_ids(ids)

    def forward(self, images_or_ids: torch.Tensor, ignore_index=-1, cond_images: Optional[torch.Tensor] = None,
                cond_token_ids: Optional[torch.Tensor] = None, texts: Optional[List[str]] = None,
                text_embeds: Optional[torch.Tensor] = None, cond_drop_prob=None, train_only_generator=False,
                sample_temperature=None):
        """Training loss of mmp.py:623-741.  With autograd enabled the generator's cross-entropy is differentiable (training.py:
        hand-written MI355X backward; `loss.backward()` fills the transformer's .grad), under torch.no_grad() it is forward only.
        The random masking uses torch's device generator exactly like the reference does on a GPU."""
        with torch.no_grad():
            x, labels, text_embeds, cond_token_ids, cond_drop_prob, ids, mask = self._training_inputs(
                images_or_ids, ignore_index, cond_images, cond_token_ids, texts, text_embeds, cond_drop_prob)
        tr = self.transformer
        # self conditioning (mmp.py:694-707): with probability self_cond_prob the embed of a first, gradient-free pass is fed back
        self_cond_embed = None
        if tr.self_cond:
            import random as _random
            if _random.random() < self.self_cond_prob:
                with torch.no_grad():
                    _, self_cond_embed = tr(x, text_embeds=text_embeds, conditioning_token_ids=cond_token_ids, cond_drop_prob=0., return_embed=True)
        if not exists(self.token_critic) or train_only_generator:
            return tr(x, text_embeds=text_embeds, self_cond_embed=self_cond_embed, conditioning_token_ids=cond_token_ids, labels=labels,
                      cond_drop_prob=cond_drop_prob, ignore_index=ignore_index)
        # ---- generator loss + the logits of the labelled positions (the only ones the critic input can differ at: mask <= labels)
        trainable = torch.is_grad_enabled() and tr.to_logits.weight.requires_grad
        dev = tr.token_emb.weight.device
        if trainable:
            from .training import transformer_loss
            ce_loss, logits_rows, row_index = transformer_loss(tr, x, text_embeds, labels, ignore_index, cond_drop_prob,
                                                               grad_sync=tr.grad_sync, return_logits=True, self_cond_embed=self_cond_embed,
                                                               conditioning_token_ids=cond_token_ids)
        else:
            ce_loss, logits = tr(x, text_embeds=text_embeds, self_cond_embed=self_cond_embed, conditioning_token_ids=cond_token_ids,
                                 labels=labels, cond_drop_prob=cond_drop_prob, ignore_index=ignore_index, return_logits=True)
            row_index = torch.nonzero(labels.reshape(-1) != ignore_index).reshape(-1).to(torch.int32)
            logits_rows = logits.reshape(-1, logits.shape[-1])[row_index.long()].contiguous()
        with torch.no_grad():
            # token critic loss (mmp.py:726-741): sample ids from the generator's logits (plain Gumbel sampling, no top-k), the critic
            # learns to tell which positions differ from the real ids
            import random as _random
            T = default(sample_temperature, _random.random())
            V = logits_rows.shape[1]
            pred, _ = ops.sample_rows(logits_rows, V, max(T, 1e-10), noise_kind=L.MM_NOISE_PHILOX,
                                      seed=int(torch.randint(0, 2 ** 62, (1,)).item()))
            sampled = x.reshape(-1).clone()
            sampled[row_index.long()] = pred
            critic_input = torch.where(mask, sampled.reshape(x.shape), x)
            critic_labels = (ids != critic_input).float()
        bce_loss = self.token_critic(critic_input, text_embeds=text_embeds, conditioning_token_ids=cond_token_ids, labels=critic_labels,
                                     cond_drop_prob=cond_drop_prob)
        return ce_loss + self.critic_loss_weight * bce_loss

    def _training_inputs(self, images_or_ids, ignore_index, cond_images, cond_token_ids, texts, text_embeds, cond_drop_prob):
        dev = self.transformer.token_emb.weight.device
        if images_or_ids.dtype == torch.float:
            assert exists(self.vae), 'vqgan vae must be passed in if training from raw images'
            assert all(hw == self.image_size for hw in images_or_ids.shape[-2:]), 'the image you passed in is not of the correct dimensions'
            _, ids, _ = self.vae.encode(images_or_ids.to(dev))
        else:
            assert not self.resize_image_for_cond_image, 'you cannot pass in raw image token ids if you want the framework to autoresize image for conditioning super res transformer'
            ids = images_or_ids.to(dev)
        ids = ids.reshape(ids.shape[0], -1)
        batch, seq_len = ids.shape
        cond_drop_prob = default(cond_drop_prob, self.cond_drop_prob)
        assert not (exists(cond_images) and exists(cond_token_ids)), 'if conditioning on low resolution, cannot pass in both images and token ids'
        if exists(cond_images):
            assert exists(self.cond_vae), 'cond vqgan vae must be passed in'
            assert all(hw == self.cond_image_size for hw in cond_images.shape[-2:])
            _, cond_token_ids, _ = self.cond_vae.encode(cond_images.to(dev))
        # prepare mask (mmp.py:671-686); `uniform` ignores its bounds in the reference, plain U(0,1)
        rand_time = torch.zeros((batch,), device=dev).float().uniform_(0, 1)
        rand_mask_probs = self.noise_schedule(rand_time)
        num_token_masked = (seq_len * rand_mask_probs).round().clamp(min=1)
        batch_randperm = torch.rand((batch, seq_len), device=dev).argsort(dim=-1)
        mask = batch_randperm < num_token_masked[:, None]
        mask_id = self.transformer.mask_id
        labels = torch.where(mask, ids, torch.full_like(ids, ignore_index))
        if self.no_mask_token_prob > 0.:
            mask &= ~_get_mask_subset_prob(mask, self.no_mask_token_prob)
        x = torch.where(mask, torch.full_like(ids, mask_id), ids)
        if exists(texts):
            text_embeds = self.transformer.encode_text(texts)
        return x, labels, text_embeds, cond_token_ids, cond_drop_prob, ids, mask


class Muse(nn.Module):
    def __init__(self, base: MaskGit, superres: MaskGit):
        super().__init__()
        if not isinstance(base, MaskGit) or not isinstance(superres, MaskGit):
            raise TypeError('base and superres must be MaskGit instances')
        self.base_maskgit = base.eval()
        assert superres.resize_image_for_cond_image
        self.superres_maskgit = superres.eval()

    @torch.no_grad()
    def forward(self, texts: List[str], cond_scale=3., temperature=1., timesteps=18, superres_timesteps=None,
                return_lowres=False, return_pil_images=True):
        """mmp.py:758-791."""
        lowres_image = self.base_maskgit.generate(texts=texts, cond_scale=cond_scale, temperature=temperature, timesteps=timesteps)
        superres_image = self.superres_maskgit.generate(texts=texts, cond_scale=cond_scale, cond_images=lowres_image,
                                                        temperature=temperature, timesteps=default(superres_timesteps, timesteps))
        if return_pil_images:
            try:
                import torchvision.transforms as T
            except ImportError as e:
                raise RuntimeError('return_pil_images=True needs torchvision (absent here); pass return_pil_images=False') from e
            lowres_image = list(map(T.ToPILImage(), lowres_image))
            superres_image = list(map(T.ToPILImage(), superres_image))
        if not return_lowres:
            return superres_image
        return superres_image, lowres_image
