"""Randomised parity of Transformer.forward / forward_with_cond_scale against the CPU oracle (oracle/muse_oracle.py, pinned to the reference's
goldens) over seeded random shapes -- batch, length (not only multiples of the tile sizes), width, heads, depth, vocabulary, text length with
zero-padded rows, conditioning ids, self-conditioning, text projection or not (t5-small has the transformer's width at dim 512):
  * precision 'parity' (fp32 storage + fp32 MFMA) and the two term-product tiers 'bf16x3' / 'f16x2' (raw fp32 weights: all six / three term
    products) against the fp32 oracle: 1e-3 of the logit scale, as the north star states it;
  * the bf16 engine against the oracle run at the same rounding points: the bound it achieves on the fixtures (3 % of the logit scale)."""
import random

import pytest
import torch

import muse_oracle as O

import muse_maskgit_pytorch_amd as mm

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('seed', list(range(12)))
def test_forward_matches_the_oracle_on_random_shapes(seed):
    rng = random.Random(500 + seed)
    B, n = rng.randint(1, 4), rng.choice([9, 16, 50, 64, 100, 130])
    dim, heads, depth = rng.choice([128, 256, 512]), rng.choice([2, 4, 8]), rng.randint(1, 2)
    V, L = rng.choice([300, 512, 1000]), rng.randint(1, 9)
    self_cond, nc = rng.random() < 0.3, rng.choice([0, 0, 9])
    torch.manual_seed(seed)
    t = mm.MaskGitTransformer(num_tokens=V, seq_len=n, dim=dim, depth=depth, dim_head=64, heads=heads, t5_name='t5-small', self_cond=self_cond)
    with torch.no_grad():
        for p in t.parameters():                                       # de-trivialise the gains / scales the constructor sets to one
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    sd = {k: v.detach().float().clone() for k, v in t.state_dict().items()}
    cfg = dict(depth=depth, heads=heads, self_cond=self_cond)
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, V + 1, (B, n), generator=g)                 # includes the mask id
    te = torch.randn(B, L, 512, generator=g)
    if B > 1 and L > 2:
        te[1, L // 2:] = 0.
    cids = torch.randint(0, V, (B, nc), generator=g) if nc else None
    sce = torch.randn(B, n, dim, generator=g) if self_cond else None
    t = t.to(DEV)
    kw = dict(conditioning_token_ids=cids.to(DEV) if nc else None, self_cond_embed=sce.to(DEV) if self_cond else None)
    okw = dict(conditioning_token_ids=cids, self_cond_embed=sce)
    for precision, rp, rel in (('parity', None, 1e-3), ('bf16x3', None, 1e-3), ('f16x2', None, 1e-3), ('bf16', O.bf16_round, 3e-2)):      # (raw fp32 weights: the tiers take 6 / 3 term products)
        t.set_precision(precision)
        for drop in (0., 1.):
            got = t(ids.to(DEV), text_embeds=te.to(DEV), cond_drop_prob=drop, **kw).float().cpu()
            ref = O.transformer_forward(sd, cfg, ids, te, drop, rp=rp, **okw)
            scale = max(ref.abs().max().item(), 1.0)
            err = (got - ref).abs().max().item()
            assert err <= rel * scale, f'{precision} drop={drop}: max err {err:.3g} on scale {scale:.3g}; B={B} n={n} dim={dim} heads={heads} depth={depth} V={V} L={L} nc={nc} self_cond={self_cond}'
        got = t.forward_with_cond_scale(ids.to(DEV), text_embeds=te.to(DEV), cond_scale=3., **kw).float().cpu()
        ref = O.forward_with_cond_scale(sd, cfg, ids, te, 3., rp=rp, **okw)
        scale = max(ref.abs().max().item(), 1.0)
        assert (got - ref).abs().max().item() <= 3 * rel * scale, f'{precision} guidance'
