"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatements of the two third-party modules whose arithmetic sits on the
reference's hot path but whose source is NOT under /root/reference:

* ``vector-quantize-pytorch>=1.11.8`` (reference setup.py:33) -- ``LFQ``; call sites
  vqgan_vae.py:7, 331-335 (ctor), 424 (forward), 431 (indices_to_codes).
* ``memory-efficient-attention-pytorch>=0.1.4`` (reference setup.py:25) --
  ``FlashAttentionFunction``; call sites attend.py:9, 88, 105 with
  ``(q, k, v, mask, causal=False, q_bucket=512, k_bucket=512)``.

PARITY UNPINNED for these two: the packages are absent from the container, the reference
vendors neither and holds no test/golden vector for them.  What is restated here is their
published eval-mode algorithm; parity is anchored on the
reference's own call sites.  ``FlashAttentionFunction`` is additionally cross-checked against
the reference's in-tree math branch (attend.py:123-140), which is the arithmetic definition we
treat as authoritative.
"""
import math

import torch
from torch import nn


class LFQ(nn.Module):
    """Lookup-free quantizer, eval-mode forward + indices_to_codes (single codebook, scale 1).

    params : project_in  Linear(dim, log2 V, bias)   project_out Linear(log2 V, dim, bias)
             (nn.Identity when dim == log2 V)
    buffer : mask = 2 ** arange(c-1, -1, -1)          (MSB = channel 0)
    """

    def __init__(self, *, dim=None, codebook_size=None, entropy_loss_weight=0.1,
                 commitment_loss_weight=0.25, diversity_gamma=1., **_):
        super().__init__()
        assert codebook_size is not None and (codebook_size & (codebook_size - 1)) == 0
        codebook_dim = int(math.log2(codebook_size))
        dim = dim if dim is not None else codebook_dim
        has_proj = dim != codebook_dim
        self.project_in = nn.Linear(dim, codebook_dim) if has_proj else nn.Identity()
        self.project_out = nn.Linear(codebook_dim, dim) if has_proj else nn.Identity()
        self.dim = dim
        self.codebook_dim = codebook_dim
        self.codebook_size = codebook_size
        self.diversity_gamma = diversity_gamma
        self.register_buffer('mask', 2 ** torch.arange(codebook_dim - 1, -1, -1))
        self.register_buffer('zero', torch.tensor(0.), persistent=False)

    def indices_to_codes(self, indices, project_out=True):
        # reference passes 2-D ids (vqgan_vae.py:430-432): (B, N) -> (B, N, C) channel-last
        bits = ((indices[..., None].long() & self.mask) != 0).float()
        codes = bits * 2 - 1
        if project_out:
            codes = self.project_out(codes)
        return codes

    def forward(self, x):
        # x: (B, C, h, w) image feature map, eval mode (no entropy / commitment loss)
        b, c, h, w = x.shape
        t = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
        t = self.project_in(t)
        q = torch.where(t > 0, torch.ones_like(t), -torch.ones_like(t))
        idx = ((q > 0).long() * self.mask.long()).sum(dim=-1)
        out = self.project_out(q)
        out = out.reshape(b, h, w, c).permute(0, 3, 1, 2)
        return out, idx.reshape(b, h, w), self.zero


class VectorQuantize(nn.Module):
    """Name-only stub: the reference's non-LFQ branch is dead code (vqgan_vae.py:337-342 raises
    TypeError at construction; :434 reads an undefined attribute)."""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError('VectorQuantize branch is unreachable in the reference')


class FlashAttentionFunction:
    """Tiled online-softmax attention, forward only, as called at attend.py:88/105.

    scale = dh ** -0.5 internally (the reference pre-multiplies q and k by 8 each,
    attend.py:76-79, so the net logit scale is 8).  mask: True = keep, shape broadcastable to
    (b, h, i, j).  fp32 running max / sum; masked logits -> -finfo.max, masked weights zeroed;
    row-sum clamp 1e-10.
    """

    @staticmethod
    def apply(q, k, v, mask, causal, q_bucket_size, k_bucket_size):
        assert not causal
        scale = q.shape[-1] ** -0.5
        o = torch.zeros_like(q)
        n_q, n_k = q.shape[-2], k.shape[-2]
        all_row_sums = torch.zeros((*q.shape[:-1], 1), dtype=torch.float32)
        all_row_maxes = torch.full((*q.shape[:-1], 1), -torch.finfo(torch.float32).max)
        if mask is not None and mask.ndim == 2:
            mask = mask[:, None, None, :]
        for qs in range(0, n_q, q_bucket_size):
            qe = min(qs + q_bucket_size, n_q)
            qc = q[..., qs:qe, :]
            oc = o[..., qs:qe, :]
            row_sums = all_row_sums[..., qs:qe, :]
            row_maxes = all_row_maxes[..., qs:qe, :]
            for ks in range(0, n_k, k_bucket_size):
                ke = min(ks + k_bucket_size, n_k)
                kc, vc = k[..., ks:ke, :], v[..., ks:ke, :]
                s = torch.einsum('...id,...jd->...ij', qc, kc) * scale
                if mask is not None:
                    mc = mask[..., qs:qe, ks:ke] if mask.shape[-2] != 1 else mask[..., :, ks:ke]
                    s = s.masked_fill(~mc, -torch.finfo(s.dtype).max)
                block_max = s.amax(dim=-1, keepdim=True)
                new_max = torch.maximum(block_max, row_maxes)
                p = torch.exp(s - new_max)
                if mask is not None:
                    p = p.masked_fill(~mc, 0.)
                block_sum = p.sum(dim=-1, keepdim=True).clamp(min=1e-10)
                exp_diff = torch.exp(row_maxes - new_max)
                new_sums = exp_diff * row_sums + block_sum
                oc.mul_(exp_diff).add_(torch.einsum('...ij,...jd->...id', p, vc))
                row_maxes.copy_(new_max)
                row_sums.copy_(new_sums)
            oc.div_(row_sums)
        return o
