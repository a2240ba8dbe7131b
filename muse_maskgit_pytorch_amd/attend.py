"""The Attend(q, k, v, mask) operator seam (reference attend.py:34-140) on MI355X.

The reference's `flash=True` default never reaches a fused kernel (attend.py:93-94 raises on purpose and falls
back to a pure-PyTorch tiled softmax); both of its branches compute softmax(scale * q k^T, mask) v.  Here that
is one MFMA flash kernel behind `mm_attend` (csrc/attention.hip for dim_head 64, csrc/attention_f32.hip for 32 / 128).
"""
import torch
from torch import nn

from . import ops


class Attend(nn.Module):
    def __init__(self, scale=8, dropout=0., flash=False):
        super().__init__()
        self.scale = scale
        self.dropout = dropout
        self.flash = flash
        if dropout != 0.:
            raise NotImplementedError('attention dropout is never enabled by the reference (mmp.py:100 passes 0.)')

    @torch.no_grad()
    def forward(self, q, k, v, mask=None, force_non_flash=False):
        """q (b,h,i,d), k/v (b,h,j,d); mask bool broadcastable (b,h,i,j) that must be a key-padding mask, i.e.
        constant over h and i -- the only kind the reference builds (mmp.py:155-157)."""
        key_mask = None
        if mask is not None:
            if mask.dim() == 4:
                if not (mask.stride(1) == 0 and mask.stride(2) == 0) and not bool((mask == mask[:, :1, :1, :]).all()):
                    raise NotImplementedError('only key-padding masks (constant over heads and queries) are supported')
                key_mask = mask[:, 0, 0, :]
            else:
                key_mask = mask
        out = ops.attend(q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16), key_mask, scale=float(self.scale))
        return out.to(q.dtype)
