"""The 'bf16x3' precision tier (csrc/split.hip, attention_f32.hip; mm_transformer_desc.split_products): operator level against fp64 torch,
then the two drivers of its decode loop against each other on random shapes.  Its parity against the reference's recorded outputs is in
tests/test_gpu_parity_mode.py (tiny fixtures, general fp32 weights: 6 products) and tests/test_gpu_base_size.py (BASELINE configs[1] / [3] / [4]
at full size, bf16-representable checkpoints: 3 products)."""
import random

import pytest
import torch

import muse_maskgit_pytorch_amd as mm
from muse_maskgit_pytorch_amd import ops
from muse_maskgit_pytorch_amd import parity as P32

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('products', [3, 5, 6])
def test_split_rows_is_an_exact_three_term_split(products):
    g = torch.Generator().manual_seed(products)
    x = torch.randn(37, 192, generator=g) * torch.logspace(-6, 6, 192)[None]
    x[3, :8] = 0.
    xs = ops.split_rows(x.to(DEV), products).cpu().reshape(37, products, 192).float()
    h, m, l = (t.float() for t in ops.split_terms(x))
    order = [h, m, l, h, m, h][:products]
    for s in range(products):
        assert torch.equal(xs[:, s], order[s]), f'segment {s}'
    assert torch.equal((xs[:, 0] + xs[:, 1]) + xs[:, 2], x)          # h + m + l == x bit for bit
    assert torch.equal(ops.unsplit_rows(ops.split_rows(x.to(DEV), products), products, 192).cpu(), x)


@pytest.mark.parametrize('M,N,K', [(128, 256, 128), (300, 200, 64), (1, 65, 192), (513, 1408, 512)])
@pytest.mark.parametrize('wkind', ['bf16', 'two_term', 'fp32'])
def test_split_product_gemm_has_fp32_accuracy(M, N, K, wkind):
    """X'.W'^T over the kept term pairs on the bf16 MFMA GEMM against fp64: the error of an fp32 GEMM (the fp32-MFMA kernel is measured next to
    it), far below the bf16 engine's"""
    g = torch.Generator().manual_seed(M + N + K)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    if wkind == 'bf16':
        w = w.bfloat16().float()
    elif wkind == 'two_term':
        h, m, _ = ops.split_terms(w)
        w = h.float() + m.float()
    terms = ops.weight_terms(w)
    assert terms == dict(bf16=1, two_term=2, fp32=3)[wkind]
    Pn = ops.products_for_terms(terms)
    ref = x.double() @ w.double().t()
    got = ops.gemm(ops.split_rows(x.to(DEV), Pn), ops.split_pack_weight(w.to(DEV), Pn), out_f32=True).double().cpu()
    f32 = P32.gemm(x.to(DEV), w.to(DEV)).double().cpu()
    b16 = ops.gemm(x.to(DEV).bfloat16(), w.to(DEV).bfloat16(), out_f32=True).double().cpu()
    e, e32, e16 = ((t - ref).abs().max().item() for t in (got, f32, b16))
    print(f'[bf16x3] gemm {M}x{N}x{K} {wkind} weights ({Pn} products): max err {e:.3g}; fp32 MFMA {e32:.3g}; bf16 {e16:.3g}')
    assert e <= 4e-6 * max(1., ref.abs().max().item()) and e <= 8 * e32 + 1e-6


def _case(rng):
    return dict(B=rng.randint(1, 4), fmap=rng.choice([4, 6, 8]), dim=rng.choice([128, 256]), heads=rng.choice([2, 4]), depth=rng.randint(1, 2),
                V=rng.choice([512, 1000, 4096]), L=rng.randint(1, 9), T=rng.randint(2, 6), critic=rng.choice([None, None, 'token', 'self']),
                self_cond=rng.random() < 0.3, can_remask=rng.random() < 0.25, cond_scale=rng.choice([1, 3.0]), bf16_weights=rng.random() < 0.5)


@pytest.mark.parametrize('seed', list(range(12)))
def test_tier_mm_generate_equals_its_stepwise_loop_and_tracks_the_fp32_engine(seed):
    """one mm_generate call of the tier (compacted last layer, constant null cross-attention, fused sampling where the vocabulary allows)
    against the same tier run operator by operator from Python, and against the fp32-MFMA engine ('parity') on the same inputs"""
    rng = random.Random(7000 + seed)
    c = _case(rng)
    torch.manual_seed(seed)
    n = c['fmap'] ** 2
    kw = dict(num_tokens=c['V'], seq_len=n, dim=c['dim'], depth=c['depth'], dim_head=64, heads=c['heads'], t5_name='t5-small')
    t = mm.MaskGitTransformer(self_cond=c['self_cond'], **kw)
    with torch.no_grad():
        t.to_logits.weight.mul_(6.)
        if c['bf16_weights']:
            for p in t.parameters():
                p.copy_(p.bfloat16().float())
    extra, gkw = {}, {}
    if c['critic'] == 'token':
        extra['token_critic'] = mm.TokenCritic(**dict(kw, dim=128, heads=2))
    elif c['critic'] == 'self':
        extra['self_token_critic'] = True
    if c['can_remask']:
        extra['no_mask_token_prob'] = 0.25
        gkw['can_remask_prev_masked'] = True
    mg = mm.MaskGit(image_size=16 * c['fmap'], transformer=t, vae=None, **extra).to(DEV)
    te = torch.randn(c['B'], c['L'], 512, device=DEV)
    if c['L'] > 2 and c['B'] > 1:
        te[1, c['L'] // 2:] = 0.
    if c['critic'] is not None:
        gkw['critic_noise'] = torch.rand(c['T'], c['B'], n, device=DEV)
    common = dict(timesteps=c['T'], text_embeds=te, seed=seed, fmap_size=c['fmap'], cond_scale=c['cond_scale'], return_ids=True, **gkw)
    mg.set_precision('bf16x3')
    ta, tb = {}, {}
    a = mg.generate([''] * c['B'], trace=ta, **common)
    b = mg.generate([''] * c['B'], trace=tb, stepwise=True, **common)
    nofuse = mg.generate([''] * c['B'], fused_sampling=False, **common)
    assert isinstance(ta['ids'], torch.Tensor) and isinstance(tb['ids'], list)
    assert t._model().packed['P'] == (3 if c['bf16_weights'] and c['critic'] is None else 6)
    assert a.shape == (c['B'], c['fmap'], c['fmap']) and int(a.min()) >= 0 and int(a.max()) < c['V'], c
    mg.set_precision('parity')
    ref = mg.generate([''] * c['B'], **common)
    agree = lambda u, v: (u == v).float().mean().item()
    print(f'[bf16x3] {c}: C loop vs stepwise {agree(a, b):.4f}, logits path vs stepwise {agree(nofuse, b):.4f}, vs fp32 engine {agree(a, ref):.4f}')
    # Philox noise, random-init weights: near-ties are rare but possible, so the engines are held to >= 97 % of the ids; the first step (no
    # history) must be identical between the two drivers of the tier
    assert torch.equal(ta['ids'][0], torch.stack(tb['ids'])[0]), f'first step differs: {c}'
    assert agree(nofuse, b) >= 0.97 and agree(a, b) >= 0.97 and agree(a, ref) >= 0.95, c
