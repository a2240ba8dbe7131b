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


@pytest.mark.parametrize('products', [2, 3])
def test_split_rows_f16_terms(products):
    """'f16x2': segments [h | l | h][:P], h = fp16(x) and l = fp16(x - h) as torch's round-to-nearest-even conversions give them (subnormal terms kept)"""
    g = torch.Generator().manual_seed(products)
    x = torch.randn(37, 192, generator=g) * torch.logspace(-7, 4, 192)[None]
    x[3, :8] = 0.
    code = ops.MM_SPLIT_F16 | products
    xs = ops.split_rows(x.to(DEV), code).cpu().view(torch.float16).reshape(37, products, 192)
    h = x.half()
    l = (x - h.float()).half()
    order = [h, l, h][:products]
    for s in range(products):
        assert torch.equal(xs[:, s].view(torch.int16), order[s].view(torch.int16)), f'segment {s}'
    back = ops.unsplit_rows(ops.split_rows(x.to(DEV), code), code, 192).cpu()
    assert bool(((back - x).abs() <= x.abs() * 2.0 ** -21 + 2.0 ** -24).all())          # 22 bits, absolute floor of the subnormal low term
    # range guard: values beyond the fp16 range saturate term by term instead of becoming (inf, -inf) -> NaN products: exact-ish up to 2 x 65504,
    # clipped (finite) beyond; tiny values keep their relative accuracy down to the subnormal floor
    big = torch.tensor([[1.0e5, -1.2e5, 7.0e4, 3.0e5, -1.0e9, 65504., 1.0e-4, -3.1e-5] + [0.] * 184])
    bb = ops.unsplit_rows(ops.split_rows(big.to(DEV), code), code, 192).cpu()
    assert bool(torch.isfinite(bb).all())
    assert bool(((bb - big)[0, :3].abs() <= 16.).all()) and bb[0, 3].item() == 131008. and bb[0, 4].item() == -131008. and bb[0, 5].item() == 65504.
    assert bool(((bb - big)[0, 6:8].abs() <= big[0, 6:8].abs() * 2.0 ** -21 + 2.0 ** -24).all())


def test_f16_mfma_takes_subnormal_terms_unflushed():
    """the tier relies on v_mfma_f32_16x16x32_f16 multiplying fp16 SUBNORMAL operands exactly (the low terms of small values are subnormal):
    a product whose operands are all subnormal must come out exactly, not as zero"""
    g = torch.Generator().manual_seed(5)
    M, N, K = 128, 128, 64
    xi = torch.randint(-1023, 1024, (M, K), generator=g).float()          # subnormal fp16: integer multiples of 2^-24 below 2^-14
    wi = torch.randint(-1023, 1024, (N, K), generator=g).float()
    x16, w16 = (xi * 2.0 ** -24).half(), (wi * 2.0 ** -24).half()
    assert bool((x16.float() == xi * 2.0 ** -24).all()) and float(x16.float().abs().max()) < 2.0 ** -14
    got = ops.gemm_split(x16.view(torch.bfloat16).to(DEV), w16.view(torch.bfloat16).to(DEV), ops.MM_SPLIT_F16 | 2, 2.0 ** 48).double().cpu()
    ref = xi.double() @ wi.double().t()                                   # = the product x 2^48
    assert float(ref.abs().max()) > 1e5
    assert torch.equal(got, ref), f'max err {(got - ref).abs().max().item()} (a flushed operand would give 0)'


@pytest.mark.parametrize('M,N,K', [(128, 256, 128), (300, 200, 64), (1, 65, 192), (513, 1408, 512)])
@pytest.mark.parametrize('wkind', ['bf16', 'fp32', 'small'])
def test_f16_term_product_gemm_has_fp32_accuracy(M, N, K, wkind):
    """X'.W'^T over fp16 term pairs on the fp16 MFMA against fp64: the error class of an fp32 GEMM.  'small': weights 2^-12 of the others -- the
    power-of-two scale of the packed terms keeps their low terms out of the subnormal range"""
    g = torch.Generator().manual_seed(M + N + K)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    if wkind == 'bf16':
        w = w.bfloat16().float()
    elif wkind == 'small':
        w = w * 2.0 ** -12
    sc = ops.f16_weight_scale([w])
    terms = ops.weight_terms_f16(w, sc)
    assert terms == dict(bf16=1, fp32=2, small=2)[wkind]
    code = ops.MM_SPLIT_F16 | (1 + terms)
    ref = x.double() @ w.double().t()
    got = ops.gemm_split(ops.split_rows(x.to(DEV), code), ops.split_pack_weight(w.to(DEV), code, 64, sc), code, 1.0 / sc).double().cpu()
    f32 = P32.gemm(x.to(DEV), w.to(DEV)).double().cpu()
    e, e32 = ((t - ref).abs().max().item() for t in (got, f32))
    scale = max(1e-30, ref.abs().max().item())
    print(f'[f16x2] gemm {M}x{N}x{K} {wkind} weights ({1 + terms} products, scale 2^{int(torch.log2(torch.tensor(sc)).item())}): max err {e:.3g}; fp32 MFMA {e32:.3g}; |ref| {scale:.3g}')
    assert e <= 4e-6 * max(scale, 2.0 ** -12) and e <= 16 * e32 + 1e-6 * scale


@pytest.mark.parametrize('nk,masked_tail', [(256, 0), (192, 0), (128, 0), (256, 37)])
def test_term_product_attention_has_fp32_accuracy(nk, masked_tail):
    """csrc/attention_x2.hip (round 5): the 'f16x2' tier's self-attention as fp16 term products on the fp16 matrix pipe -- q^ / k^ / P / V each split into two fp16
    terms in registers, three v_mfma_f32_16x16x32_f16 per block, fp32 softmax, null key / value (mmp.py:137-162, attend.py:109-140) -- against fp64 torch, beside
    the fp32-MFMA kernel it replaces (attention_f32.hip): the same 1e-6-grade error.  `masked_tail`: queries beyond nq in the last block (nq no multiple of 32)."""
    from muse_maskgit_pytorch_amd import parity as P32
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(nk + masked_tail)
    b, h, nq = 2, 8, 256 - masked_tail
    I = h * 64
    qm = torch.randn(b, nq, I, generator=g) * 1.3
    kvm = torch.randn(b, nk, 2 * I, generator=g) * 1.3
    kvm[0, 3, :64] *= 30.                                     # a key row with a large norm (normalisation) ...
    kvm[1, 5, I: I + 64] *= 8.                                # ... and a large value row
    qs, ks = 1 + 0.2 * torch.randn(64, generator=g), 1 + 0.2 * torch.randn(64, generator=g)
    nkv, nvv = torch.randn(h, 64, generator=g), torch.randn(h, 64, generator=g)
    q = qm.double().reshape(b, nq, h, 64).permute(0, 2, 1, 3)
    k = kvm.double()[:, :, :I].reshape(b, nk, h, 64).permute(0, 2, 1, 3)
    v = kvm.double()[:, :, I:].reshape(b, nk, h, 64).permute(0, 2, 1, 3)
    k = torch.cat((nkv.double()[None, :, None, :].expand(b, -1, -1, -1), k), dim=2)
    v = torch.cat((nvv.double()[None, :, None, :].expand(b, -1, -1, -1), v), dim=2)
    qn, kn = F.normalize(q, dim=-1) * qs.double(), F.normalize(k, dim=-1) * ks.double()
    ref = (torch.softmax(qn @ kn.transpose(-1, -2) * 8., dim=-1) @ v).permute(0, 2, 1, 3).reshape(b * nq, I)
    qd, kvd = qm.to(DEV), kvm.to(DEV)
    args = (b, h, nq, nk, (nq * I, 64, I), (nk * 2 * I, 64, 2 * I), (nk * 2 * I, 64, 2 * I))
    kw = dict(q_scale=qs.to(DEV), k_scale=ks.to(DEV), null_k=nkv.to(DEV).contiguous(), null_v=nvv.to(DEV).contiguous())
    vview = P32._View(kvd, I)
    got = P32.attend_terms(qd, kvd, vview, *args, **kw).double().cpu()
    base = P32.attend(qd, kvd, vview, *args, **kw).double().cpu()
    scale = ref.abs().max().item()
    e_t, e_b = (got - ref).abs().max().item(), (base - ref).abs().max().item()
    print(f'[term-product attention] nk={nk} nq={nq}: max abs err vs fp64 {e_t:.3g} (fp32-MFMA kernel {e_b:.3g}) on scale {scale:.3g}')
    assert e_t <= 4e-6 * max(1.0, scale) and e_t <= 8 * e_b + 1e-6



@pytest.mark.parametrize('tier', ['bf16x3', 'f16x2'])
@pytest.mark.parametrize('seed', list(range(12)))
def test_tier_mm_generate_equals_its_stepwise_loop_and_tracks_the_fp32_engine(seed, tier):
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
    mg.set_precision(tier)
    ta, tb = {}, {}
    a = mg.generate([''] * c['B'], trace=ta, **common)
    b = mg.generate([''] * c['B'], trace=tb, stepwise=True, **common)
    nofuse = mg.generate([''] * c['B'], fused_sampling=False, **common)
    assert isinstance(ta['ids'], torch.Tensor) and isinstance(tb['ids'], list)
    mg.set_precision(tier)
    lo, hi = (3, 6) if tier == 'bf16x3' else (2, 3)
    assert t._model().packed['P'] == (lo if c['bf16_weights'] and c['critic'] is None else hi), (t._model().packed['P'], c)
    assert a.shape == (c['B'], c['fmap'], c['fmap']) and int(a.min()) >= 0 and int(a.max()) < c['V'], c
    mg.set_precision('parity')
    ref = mg.generate([''] * c['B'], **common)
    agree = lambda u, v: (u == v).float().mean().item()
    print(f'[{tier}] {c}: C loop vs stepwise {agree(a, b):.4f}, logits path vs stepwise {agree(nofuse, b):.4f}, vs fp32 engine {agree(a, ref):.4f}')
    # Philox noise, random-init weights: near-ties are rare but possible, so the engines are held to >= 97 % of the ids; the first step (no
    # history) must be identical between the two drivers of the tier
    assert torch.equal(ta['ids'][0], torch.stack(tb['ids'])[0]), f'first step differs: {c}'
    assert agree(nofuse, b) >= 0.97 and agree(a, b) >= 0.97 and agree(a, ref) >= 0.95, c
