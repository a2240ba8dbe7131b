"""The fp8 engine (BASELINE configs[4] "fp8 MFMA weights"): e4m3 weights AND activations on gfx950's K = 128 fp8 MFMA (csrc/gemm_fp8.hip), per-row
scales, quantising LayerNorms (csrc/fp8_act.hip), inside mm_transformer_forward / mm_generate (precision 'fp8').  The reference has no fp8, so the
numerics are self-defined (SURVEY 8c "L2"): the oracle is the fp32 restatement with the same per-row fake quantisation at the same places
(oracle/muse_oracle.py Fp8Rounding) on the de-quantised weights.  What IS exact: the quantisers (bit for bit against torch's float8_e4m3fn cast), and the
engine against itself (one mm_generate call == the stepwise loop == the oracle's sampling tail on the engine's own logits)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import golden_recipe as R
import muse_oracle as O
from conftest import sd_f32

import muse_maskgit_pytorch_amd as mm
from muse_maskgit_pytorch_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda'
bf16 = torch.bfloat16
C5_ROWS = [0, 77, 255, 256, 300, 411, 500, 511]      # the rows tests/golden/paper_c5.pt holds in full


def _dq(q, scale):
    return q.cpu().view(torch.float8_e4m3fn).float() * scale.cpu()[:, None]


@pytest.mark.parametrize('M,N,K', [(300, 512, 384), (1000, 2816, 1024), (64, 8192, 512), (257, 256, 1365)])
def test_fp8_quantisers_and_gemm(M, N, K):
    """weight / activation quantisers == torch's e4m3fn cast of row / scale, bit for bit; mm_gemm_fp8 (all three epilogues, both tile widths, ragged M,
    padded K) against fp64 on the de-quantised operands: fp32 epilogue within 2e-4 of the output scale (the fp8 MFMA accumulates a 128-deep block with
    less than fp32 precision: measured 6e-5), bf16 epilogues within one bf16 rounding of that"""
    g = torch.Generator().manual_seed(M + N)
    w = torch.randn(N, K, generator=g) * 0.05
    w[3] = 0
    x = torch.randn(M, K, generator=g)
    x[5] = 0
    wq, ws = ops.quantize_e4m3_rows(w.to(DEV))
    Kp = (K + 127) // 128 * 128
    assert wq.shape == (N, Kp) and wq.dtype == torch.uint8 and (wq[:, K:] == 0).all()
    for src, q, sc in ((w, wq, ws),) + tuple((x.to(dt).float(), *ops.quantize_act_e4m3(x.to(DEV, dt), Kp)) for dt in (torch.float32, bf16)):
        ref_scale = src.abs().amax(-1) / 448.
        ref_scale[ref_scale == 0] = 1.
        assert torch.equal(sc.cpu(), ref_scale)
        assert torch.equal(q.cpu()[:, :K].view(torch.float8_e4m3fn).float(), (src / ref_scale[:, None]).to(torch.float8_e4m3fn).float())
        assert (q[:, K:] == 0).all()
    xq, xs = ops.quantize_act_e4m3(x.to(DEV), Kp)
    exact = _dq(xq, xs).double() @ _dq(wq, ws).double().t()
    scale = exact.abs().max().item()
    res = torch.randn(M, N, generator=g)
    got = ops.gemm_fp8(xq, xs, wq, ws, epilogue=2, resid=res.to(DEV))
    assert (got.cpu().double() - (exact + res.double())).abs().max().item() <= 2e-4 * scale
    assert (ops.gemm_fp8(xq, xs, wq, ws, epilogue=2).cpu().double() - exact).abs().max().item() <= 2e-4 * scale
    gb = ops.gemm_fp8(xq, xs, wq, ws, epilogue=0)
    assert gb.dtype == bf16 and (gb.float().cpu().double() - exact).abs().max().item() <= 4.5e-3 * scale
    if N % 256 == 0:      # GEGLU over 64-row blocks of the weight: 32 value rows, then their 32 gate rows
        e = exact.reshape(M, N // 64, 2, 32)
        want = (e[:, :, 1] * torch.nn.functional.gelu(e[:, :, 0])).reshape(M, N // 2)
        gg = ops.gemm_fp8(xq, xs, wq, ws, epilogue=1)
        assert gg.shape == (M, N // 2) and (gg.float().cpu().double() - want).abs().max().item() <= 4.5e-3 * want.abs().max().item() + 1e-6


def _tiny(golden):
    g = golden('transformer_tiny.pt')
    t = mm.MaskGitTransformer(t5_name='t5-small', **g['cfg'])
    t.load_state_dict(sd_f32(g['sd']))
    return g, t.to(DEV).eval()


def _fake_quant_sd(t):
    return {k: (v.float().cpu() if v.is_floating_point() else v.cpu()) for k, v in t.fp8_dequantized_state_dict().items()}


def test_fp8_engine_forward_and_decode_tiny(golden):
    """tiny config (dim 128, 8 heads, inner width 341 -> padded to 384): logits against the fake-quant oracle (errors are re-quantisation flips of values
    that sit on an e4m3 boundary: a few percent of the logit scale at worst, a few 1e-3 on average); guidance = to_logits of the mixed embedding; the
    decode loop inside mm_generate == the stepwise loop == the oracle's sampling tail on the engine's own logits, bit for bit; the mode is really on."""
    g, t = _tiny(golden)
    te, ids = g['text_embeds'], g['ids']
    bf = t(ids.to(DEV), text_embeds=te.to(DEV))
    t.quantize_weights_fp8()
    try:
        assert t.precision == 'fp8'
        cfg = dict(depth=g['cfg']['depth'], heads=g['cfg']['heads'])
        sd = _fake_quant_sd(t)
        got = t(ids.to(DEV), text_embeds=te.to(DEV))
        ref = O.transformer_forward(sd, cfg, ids, te, 0., rp=O.Fp8Rounding())
        err = (got.cpu() - ref).abs()
        print(f'[fp8 engine] tiny logits vs fake-quant oracle: max {err.max().item():.4g}, mean {err.mean().item():.4g}, scale {ref.abs().max().item():.4g}')
        assert err.max() < 0.04 * ref.abs().max() and err.mean() < 0.004 * ref.abs().max()
        assert (got - bf).abs().max() > 1e-3 * bf.abs().max()
        null = t(ids.to(DEV), text_embeds=te.to(DEV), cond_drop_prob=1.)
        refn = O.transformer_forward(sd, cfg, ids, te, 1., rp=O.Fp8Rounding())
        assert (null.cpu() - refn).abs().max() < 0.04 * refn.abs().max()
        s3 = t.forward_with_cond_scale(ids.to(DEV), text_embeds=te.to(DEV), cond_scale=3.)
        assert (s3 - (null + (got - null) * 3.)).abs().max() < 2e-2 * s3.abs().max()      # (the mixed embedding is rounded to bf16 once)
        B, n, T = 2, 64, 4
        gumbel = O.gumbel_from_uniform(torch.rand(T, B, n, 512, generator=torch.Generator().manual_seed(21)))
        mg = mm.MaskGit(image_size=128, transformer=t, vae=None)
        kw = dict(timesteps=T, text_embeds=te, noise=gumbel, noise_kind='gumbel', fmap_size=8)
        a = mg.generate(['a', 'b'], **kw)
        b = mg.generate(['a', 'b'], stepwise=True, **kw)
        assert torch.equal(a, b)
        free = O.generate_ids(lambda i, s: t.forward_with_cond_scale(i.to(DEV), text_embeds=te.to(DEV), cond_scale=3.).cpu(), B, n, 512,
                              lambda s, shp: gumbel[s], timesteps=T)
        assert torch.equal(a.reshape(B, n).cpu(), free)
        c = mg.generate(['a', 'b'], timesteps=T, text_embeds=te, seed=5, fmap_size=8, return_ids=True)      # Philox noise, fused C loop
        assert torch.equal(c, mg.generate(['a', 'b'], timesteps=T, text_embeds=te, seed=5, fmap_size=8, return_ids=True, stepwise=True))
    finally:
        t.quantize_weights_fp8(False)
    assert torch.equal(t(ids.to(DEV), text_embeds=te.to(DEV)), bf)


@pytest.mark.parametrize('variant', ['self_cond', 'token_critic', 'single_pass'])
def test_fp8_engine_decode_variants_match_the_stepwise_loop(variant):
    """the decode variants on the fp8 engine: one mm_generate call against the loop run operator by operator (same kernels, so bit for bit)"""
    torch.manual_seed(3)
    kw = dict(num_tokens=1024, seq_len=64, dim=256, depth=2, dim_head=64, heads=4, t5_name='t5-small')
    t = mm.MaskGitTransformer(self_cond=variant == 'self_cond', **kw)
    extra = {}
    if variant == 'token_critic':
        extra['token_critic'] = mm.TokenCritic(**dict(kw, dim=128, heads=2))
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None, **extra).to(DEV)
    mg.set_precision('fp8')
    te = torch.randn(3, 6, 512, device=DEV)
    gkw = dict(timesteps=5, text_embeds=te, seed=9, fmap_size=8, cond_scale=1. if variant == 'single_pass' else 3., return_ids=True)
    if variant == 'token_critic':
        gkw['critic_noise'] = torch.rand(5, 3, 64, device=DEV)
    a = mg.generate([''] * 3, **gkw)
    b = mg.generate([''] * 3, stepwise=True, **gkw)
    assert torch.equal(a, b)


def test_fp8_engine_at_paper_scale(golden):
    """BASELINE configs[4] at full size (dim 1024, depth 24, 16 heads, inner width 2730 -> 2816, V = 8192, the 512 -> 1024 text projection; paper_c5.pt
    inputs and checkpoint).  Per-row e4m3 rounding is discontinuous, so over 24 layers the engine and its fake-quant oracle drift apart like two draws of
    the same quantisation noise (a last-bit difference of an fp32 accumulation flips an e4m3 rounding, 6 % of that element, and propagates).  Pinned in
    two ways: (a) LAYER PAIRS of the checkpoint (first two, last two) as depth-2 models, engine against oracle, where the flips are few: a few 1e-3 of the
    logit scale on average; (b) the whole stack statistically: the engine is no further from the reference's fp32 logits (golden) than the oracle's own
    quantisation noise is."""
    g = golden('paper_c5.pt')
    big = R.build_transformer(mm.MaskGitTransformer, peaky=False, cfg=R.C5_CFG, seed=R.C5_WEIGHT_SEED)
    assert R.state_checksum(big) == g['weight_checksum']
    inp = R.c5_inputs()
    te, ids = inp['text_embeds'], inp['ids']
    bsd = big.state_dict()
    for pair in ((0, 1), (22, 23)):
        cfg2 = dict(R.C5_CFG, depth=2)
        t2 = mm.MaskGitTransformer(**cfg2)
        sd2 = {}
        for k, v in bsd.items():
            if k.startswith('transformer_blocks.layers.'):
                li = int(k.split('.')[2])
                if li in pair:
                    sd2[k.replace(f'layers.{li}.', f'layers.{pair.index(li)}.', 1)] = v
            else:
                sd2[k] = v
        t2.load_state_dict(sd2)
        t2 = t2.to(DEV).eval().set_precision('fp8')
        got = t2(ids.to(DEV), text_embeds=te.to(DEV), cond_drop_prob=0.)
        ref = O.transformer_forward(_fake_quant_sd(t2), dict(depth=2, heads=R.C5_CFG['heads']), ids, te, 0., rp=O.Fp8Rounding())
        err, scale = (got.cpu() - ref).abs(), ref.abs().max().item()
        print(f'[fp8 engine] paper-scale layers {pair} vs fake-quant oracle: max {err.max().item():.4g}, mean {err.mean().item():.4g}, scale {scale:.4g}')
        assert err.max().item() < 0.06 * scale and err.mean().item() < 0.004 * scale
        del t2
    tr = big.to(DEV).eval()
    tr.set_precision('fp8')
    try:
        got = tr(ids.to(DEV), text_embeds=te.to(DEV), cond_drop_prob=0.)
        ref = O.transformer_forward(_fake_quant_sd(tr), dict(depth=R.C5_CFG['depth'], heads=R.C5_CFG['heads']), ids, te, 0., rp=O.Fp8Rounding())
        err, scale = (got.cpu() - ref).abs(), ref.abs().max().item()
        rows = g['forward']['logits_cond']['rows']
        q_noise = (ref.reshape(512, -1)[C5_ROWS] - rows).abs()
        d = (got.reshape(512, -1)[C5_ROWS].cpu() - rows).abs()
        print(f'[fp8 engine] paper-scale (24 layers) logits: engine vs fake-quant oracle max {err.max().item():.4g} mean {err.mean().item():.4g}; fake-quant oracle vs the '
              f'reference fp32 run max {q_noise.max().item():.4g} mean {q_noise.mean().item():.4g}; engine vs the reference fp32 run max {d.max().item():.4g} mean '
              f'{d.mean().item():.4g} (scale {scale:.4g}); arg-max agreement with the reference '
              f'{100 * (got.reshape(512, -1)[C5_ROWS].cpu().argmax(-1) == rows.argmax(-1)).float().mean().item():.1f} %')
        assert d.mean().item() <= 1.25 * q_noise.mean().item() and d.max().item() <= 1.6 * q_noise.max().item()
        assert err.mean().item() <= 1.5 * q_noise.mean().item()
    finally:
        tr.set_precision('bf16')
        torch.cuda.empty_cache()


@pytest.mark.parametrize('seed', list(range(8)))
def test_fp8_engine_matches_its_oracle_on_random_shapes(seed):
    """seeded random shapes (batch, ragged lengths, width, heads, depth <= 2, vocabulary, text length with zero-padded rows, conditioning ids,
    self-conditioning): the fp8 engine against the fake-quant oracle, both guidance passes and the guidance combine"""
    import random
    rng = random.Random(900 + seed)
    B, n = rng.randint(1, 4), rng.choice([9, 16, 50, 64, 100, 130])
    dim, heads, depth = rng.choice([128, 256, 384, 512]), rng.choice([2, 4, 8]), rng.randint(1, 2)
    V, L = rng.choice([300, 512, 1000]), rng.randint(1, 9)
    self_cond, nc = rng.random() < 0.3, rng.choice([0, 0, 9])
    torch.manual_seed(seed)
    t = mm.MaskGitTransformer(num_tokens=V, seq_len=n, dim=dim, depth=depth, dim_head=64, heads=heads, t5_name='t5-small', self_cond=self_cond)
    with torch.no_grad():
        for p in t.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, V + 1, (B, n), generator=g)
    te = torch.randn(B, L, 512, generator=g)
    if B > 1 and L > 2:
        te[1, L // 2:] = 0.
    cids = torch.randint(0, V, (B, nc), generator=g) if nc else None
    sce = torch.randn(B, n, dim, generator=g) if self_cond else None
    t = t.to(DEV).eval().set_precision('fp8')
    sd = _fake_quant_sd(t)
    cfg = dict(depth=depth, heads=heads, self_cond=self_cond)
    kw = dict(conditioning_token_ids=cids.to(DEV) if nc else None, self_cond_embed=sce.to(DEV) if self_cond else None)
    okw = dict(conditioning_token_ids=cids, self_cond_embed=sce)
    what = f'B={B} n={n} dim={dim} heads={heads} depth={depth} V={V} L={L} nc={nc} self_cond={self_cond}'
    for drop in (0., 1.):
        got = t(ids.to(DEV), text_embeds=te.to(DEV), cond_drop_prob=drop, **kw).float().cpu()
        ref = O.transformer_forward(sd, cfg, ids, te, drop, rp=O.Fp8Rounding(), **okw)
        scale = max(ref.abs().max().item(), 1.0)
        err = (got - ref).abs()
        assert err.max().item() <= 0.08 * scale and err.mean().item() <= 0.01 * scale, f'drop={drop}: max {err.max().item():.3g} mean {err.mean().item():.3g} on scale {scale:.3g}; {what}'
    got = t.forward_with_cond_scale(ids.to(DEV), text_embeds=te.to(DEV), cond_scale=3., **kw).float().cpu()
    ref = O.forward_with_cond_scale(sd, cfg, ids, te, 3., rp=O.Fp8Rounding(), **okw)
    assert (got - ref).abs().max().item() <= 0.2 * max(ref.abs().max().item(), 1.0), f'guidance; {what}'
