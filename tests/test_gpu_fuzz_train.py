"""Randomised gradient parity of the training path (training.py: forward with saved activations + hand-written MI355X backward) against torch
autograd of the CPU oracle over seeded random shapes -- width, heads, depth, vocabulary, text length with padded rows, labelled fraction.
Gradients pass through bf16 activations / operands: per-tensor relative error (max |diff| / max |ref|) below 6e-2, cosine similarity above
0.99; the loss within 2 %."""
import random

import pytest
import torch

import muse_oracle as O

import muse_maskgit_pytorch_amd as mm

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('seed', list(range(12)))
def test_gradients_match_oracle_autograd_on_random_shapes(seed):
    rng = random.Random(70 + seed)
    B, n = rng.choice([(2, 64), (4, 16), (1, 64), (2, 96), (8, 16), (2, 160), (4, 144), (1, 320)])      # batch * seq_len must be a multiple of 64 on the training path
    dim, heads, depth = rng.choice([128, 256]), rng.choice([2, 4, 8]), rng.randint(1, 2)
    V, L = rng.choice([300, 512, 1000]), rng.randint(2, 9)
    torch.manual_seed(seed)
    t = mm.MaskGitTransformer(num_tokens=V, seq_len=n, dim=dim, depth=depth, dim_head=64, heads=heads, t5_name='t5-small')
    with torch.no_grad():
        for p in t.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    sd0 = {k: v.detach().clone() for k, v in t.state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, V + 1, (B, n), generator=g)
    te = torch.randn(B, L, 512, generator=g)
    if B > 1:
        te[1, L // 2:] = 0.
    labels = torch.randint(0, V, (B, n), generator=g)
    labels[torch.rand(B, n, generator=g) < rng.choice([0.2, 0.5, 0.8])] = -1
    labels[0, 0] = 3                                                     # at least one labelled position
    t = t.to(DEV)
    loss = t(ids.to(DEV), text_embeds=te.to(DEV), labels=labels.to(DEV), ignore_index=-1)
    loss.backward()
    # the forward rounds the weights to bf16: give the oracle the same weights
    sd = {k: ((v.bfloat16().float() if v.dim() >= 2 else v.float()).clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd0.items()}
    ref = O.transformer_loss(sd, dict(depth=depth, heads=heads), ids, te, labels, ignore_index=-1, rp=O.bf16_round)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 2e-2 * abs(ref.item()), (loss.item(), ref.item())
    for name, p in t.named_parameters():
        if name.startswith('self_cond_to_init_embed') or name == 'norm.gamma':
            continue
        rg = sd[name].grad
        assert p.grad is not None and rg is not None, name
        gg = p.grad.float().cpu()
        rel = (gg - rg).abs().max().item() / (rg.abs().max().item() + 1e-20)
        cos = torch.nn.functional.cosine_similarity(gg.flatten(), rg.flatten(), dim=0).item()
        assert rel < 6e-2 and cos > 0.99, f'{name}: rel {rel:.3e} cos {cos:.5f}; B={B} n={n} dim={dim} heads={heads} depth={depth} V={V} L={L}'
