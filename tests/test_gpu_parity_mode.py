"""precision = 'parity' (fp32 storage + fp32 MFMA, csrc/parity.hip) and precision = 'bf16x3' (the tier inside mm_generate: exact bf16 term
splits on the bf16 matrix pipe, csrc/split.hip; all reference fixtures are bf16-representable checkpoints = its 3-product form, the 5- and
6-product forms for general fp32 weights are checked against fp64 and against the fp32 engine in tests/test_gpu_bf16x3.py) against the UNMODIFIED REFERENCE's recorded outputs -- not against a
rounding-point oracle: logits / pixels within 1e-3 absolute on unit scale (1e-3 x scale where the fixture's logits were made peaky), token
ids and LFQ ids 100 % equal, for every tiny fixture of round 1 (forward, guidance, T = 4 / 18 decode, all five decode variants, VAE).
The full-size counterpart is tests/test_gpu_base_size.py.  Operator-level checks against fp64 torch come first."""
import pytest
import torch
import torch.nn.functional as F

import muse_oracle as O
from conftest import sd_f32

import muse_maskgit_pytorch_amd as mm
from muse_maskgit_pytorch_amd import parity as P

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _close(name, got, ref, tol):
    err = (got.detach().double().cpu() - ref.double()).abs().max().item()
    print(f'[parity-mode] {name}: max abs err {err:.3g} (ref absmax {ref.abs().max().item():.3g}), bound {tol:g}')
    assert err <= tol, f'{name}: {err} > {tol}'


# ------------------------------------------------------------------------------------------------ operators vs fp64
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (300, 200, 72), (1, 65, 3), (513, 1365, 512), (17, 4, 1408)])
def test_f32_gemm_bias_act_resid(M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = x.double() @ w.double().t()
    _close('gemm', P.gemm(x.to(DEV), w.to(DEV)), ref, 2e-5 * K ** 0.5)
    ref2 = F.leaky_relu(ref + b.double(), 0.1) + r.double()
    _close('gemm + bias + leaky + resid', P.gemm(x.to(DEV), w.to(DEV), bias=b.to(DEV), act=True, resid=r.to(DEV)), ref2, 2e-5 * K ** 0.5)


def test_f32_layernorm_geglu_combine():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(37, 1365, generator=g) * 3 + 0.5
    gam, bet = torch.randn(1365, generator=g), torch.randn(1365, generator=g)
    _close('layernorm', P.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV)), F.layer_norm(x.double(), (1365,), gam.double(), bet.double()), 5e-6 * 8)
    h = torch.randn(19, 2 * 1365, generator=g) * 2
    a, gate = h.double().chunk(2, dim=-1)
    _close('geglu', P.geglu(h.to(DEV)), gate * F.gelu(a), 2e-6 * 10)
    c, n = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    assert torch.equal(P.cfg_combine(c.to(DEV), n.to(DEV), 3.).cpu(), n + (c - n) * 3.)


@pytest.mark.parametrize('nq,nk,masked', [(64, 64, False), (64, 7, True), (16, 300, True), (256, 256, False), (50, 1, True)])
def test_f32_attend_null_norm_mask(nq, nk, masked):
    g = torch.Generator().manual_seed(nq * 1000 + nk)
    b, h = 2, 3
    q, k, v = (torch.randn(b, h, n_, 64, generator=g) for n_ in (nq, nk, nk))
    qs, ks = 1 + 0.2 * torch.randn(64, generator=g), 1 + 0.2 * torch.randn(64, generator=g)
    nkv = torch.randn(2, h, 1, 64, generator=g)
    km = torch.rand(b, nk, generator=g) < 0.6 if masked else None
    # reference semantics in fp64 (mmp.py:145-157 + attend.py:123-140)
    kk = torch.cat((nkv[0][None].expand(b, -1, -1, -1), k), dim=-2).double()
    vv = torch.cat((nkv[1][None].expand(b, -1, -1, -1), v), dim=-2).double()
    qn = F.normalize(q.double(), dim=-1) * qs.double()
    kn = F.normalize(kk, dim=-1) * ks.double()
    m4 = F.pad(km[:, None, None, :].expand(b, h, nq, nk), (1, 0), value=True) if masked else None
    ref = O.attend(qn, kn, vv, mask=m4)
    qr = q.permute(0, 2, 1, 3).reshape(b * nq, h * 64).contiguous().to(DEV)
    kr = k.permute(0, 2, 1, 3).reshape(b * nk, h * 64).contiguous().to(DEV)
    vr = v.permute(0, 2, 1, 3).reshape(b * nk, h * 64).contiguous().to(DEV)
    I = h * 64
    out = P.attend(qr, kr, vr, b, h, nq, nk, (nq * I, 64, I), (nk * I, 64, I), (nk * I, 64, I), key_mask=km.to(torch.uint8).to(DEV) if masked else None,
                   q_scale=qs.to(DEV), k_scale=ks.to(DEV), null_k=nkv[0].reshape(h, 64).contiguous().to(DEV), null_v=nkv[1].reshape(h, 64).contiguous().to(DEV))
    _close('attend', out.reshape(b, nq, h, 64).permute(0, 2, 1, 3), ref, 2e-5)


def test_f32_conv_groupnorm_glu_vs_torch():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 24, 10, 12, generator=g)
    for (cout, k, stride) in ((40, 3, 1), (16, 1, 1), (8, 5, 1), (32, 4, 2)):
        c = torch.nn.Conv2d(24, cout, k, stride=stride, padding=1 if (k, stride) == (4, 2) else k // 2)
        ref = c.double()(x.double())
        xh = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        if stride == 2:
            out = P.conv(xh, P.pack_conv(c.weight.float()).to(DEV), cout, 4, 4, 2, (-1, -1), out_hw=(5, 6), bias=c.bias.float().to(DEV))
        else:
            out = P.conv(xh, P.pack_conv(c.weight.float()).to(DEV), cout, k, k, 1, (-(k // 2), -(k // 2)), bias=c.bias.float().to(DEV))
        _close(f'conv {k}x{k} stride {stride}', out.permute(0, 3, 1, 2), ref.detach(), 5e-5)
    ct = torch.nn.ConvTranspose2d(24, 20, 4, 2, 1)
    ref = ct.double()(x.double()).detach()
    out = torch.empty(2, 20, 24, 20, dtype=torch.float32, device=DEV)
    xh = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    for (py, px), wp in P.pack_convT(ct.weight.float()).items():
        P.conv(xh, wp.to(DEV), 20, 2, 2, 1, (py - 1, px - 1), out_hw=(10, 12), os_=2, parity=(py, px), full_hw=(20, 24), bias=ct.bias.float().to(DEV), out=out)
    _close('conv transpose 4/2/1', out.permute(0, 3, 1, 2), ref, 5e-5)
    gn = torch.nn.GroupNorm(4, 24)
    with torch.no_grad():
        gn.weight.normal_(); gn.bias.normal_()
    gnd = torch.nn.GroupNorm(4, 24).double()
    gnd.load_state_dict({k_: v_.double() for k_, v_ in gn.state_dict().items()})
    _close('groupnorm + leaky', P._groupnorm(xh, gn.to(DEV), act=True).permute(0, 3, 1, 2), F.leaky_relu(gnd(x.double()), 0.1).detach(), 2e-5)
    _close('glu', P._glu(xh).permute(0, 3, 1, 2), F.glu(x.double(), dim=1), 2e-6)


# ------------------------------------------------------------------------------------------------ tiny fixtures of the reference
TIERS = ['parity', 'bf16x3', 'f16x2']


def _tiny(golden, precision='parity'):
    g = golden('transformer_tiny.pt')
    t = mm.MaskGitTransformer(t5_name='t5-small', **g['cfg'])
    t.load_state_dict(sd_f32(g['sd']))
    return g, t.to(DEV).eval().set_precision(precision)


@pytest.mark.parametrize('precision', TIERS)
def test_parity_forward_and_guidance_vs_reference_golden(golden, precision):
    g, t = _tiny(golden, precision)
    if precision == 'bf16x3':
        assert t.split_products() == 3          # the fixture's weights are bf16-representable (the 5 / 6-product forms: tests/test_gpu_bf16x3.py)
    if precision == 'f16x2':
        assert t.split_products() == 2          # ... and one fp16 term holds each of them
    ids, te = g['ids'].to(DEV), g['text_embeds'].to(DEV)
    lc, emb = t(ids, text_embeds=te, return_embed=True)
    ln = t(ids, text_embeds=te, cond_drop_prob=1.)
    sc = t.forward_with_cond_scale(ids, text_embeds=te, cond_scale=3.)
    scale = g['logits_cond'].abs().max().item()          # the fixture's to_logits was scaled x8: logits are ~20x unit scale
    _close('tiny logits(cond) vs reference', lc, g['logits_cond'], 1e-3 * max(1., scale / 8))
    _close('tiny logits(null) vs reference', ln, g['logits_null'], 1e-3 * max(1., scale / 8))
    _close('tiny logits(guidance) vs reference', sc, g['logits_scaled'], 5e-3 * max(1., scale / 8))
    _close('tiny embed vs reference', emb, g['embed'], 1e-3)
    assert torch.equal(lc.cpu().argmax(-1), g['logits_cond'].argmax(-1))


@pytest.mark.parametrize('precision', TIERS)
@pytest.mark.parametrize('T', [4, 18])
def test_parity_generate_ids_equal_the_reference_run(golden, T, precision):
    g, t = _tiny(golden, precision)
    gen = golden(f'generate_tiny_T{T}.pt')
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None)
    uni = torch.stack(gen['uniform'])
    trace = {}
    ids = mg.generate(['a', 'b'], timesteps=T, text_embeds=g['text_embeds'], noise=uni, noise_kind='uniform', fmap_size=8, trace=trace)
    for s in range(T):
        assert torch.equal(trace['masked_ids'][s].cpu(), gen['step_ids'][s]), f'ids entering step {s} differ from the reference'
    assert torch.equal(ids.reshape(gen['final_ids'].shape).cpu(), gen['final_ids'])


@pytest.mark.parametrize('precision', TIERS)
@pytest.mark.parametrize('name', ['token_critic', 'self_critic', 'cond_scale_1', 'can_remask', 'self_cond'])
def test_parity_decode_variants_equal_the_reference_run(golden, name, precision):
    gv, gt = golden('generate_variants_tiny.pt')[name], golden('transformer_tiny.pt')
    te = gt['text_embeds']
    T, B, n = gv['timesteps'], 2, 64
    if name == 'self_cond':
        t = mm.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=1, dim_head=64, heads=8, t5_name='t5-small', self_cond=True)
        t.load_state_dict(sd_f32(gv['sd']))
        t = t.to(DEV)
    else:
        _, t = _tiny(golden, precision)
    kw = {}
    if name == 'token_critic':
        critic = mm.TokenCritic(num_tokens=512, seq_len=64, dim=128, depth=1, dim_head=64, heads=8, t5_name='t5-small')
        critic.load_state_dict(sd_f32(gv['critic_sd']))
        mg = mm.MaskGit(image_size=128, transformer=t, vae=None, token_critic=critic.to(DEV))
    elif name == 'self_critic':
        mg = mm.MaskGit(image_size=128, transformer=t, vae=None, self_token_critic=True)
        mg.token_critic.to_pred.load_state_dict(sd_f32(gv['to_pred']))
        mg = mg.to(DEV)
    elif name == 'can_remask':
        mg = mm.MaskGit(image_size=128, transformer=t, vae=None, no_mask_token_prob=0.25)
        kw['can_remask_prev_masked'] = True
    else:
        mg = mm.MaskGit(image_size=128, transformer=t, vae=None)
    mg.set_precision(precision)
    if name in ('token_critic', 'self_critic'):
        kw['critic_noise'] = torch.stack([u.reshape(B, n) for u in gv['critic_uniform']])
    trace = {}
    got = mg.generate(['a', 'b'], timesteps=T, text_embeds=te, noise=torch.stack(gv['uniform']), noise_kind='uniform', fmap_size=8,
                      cond_scale=1 if name == 'cond_scale_1' else 3, trace=trace, **kw)
    # (the recorder wrapped Transformer.forward_with_cond_scale, which the SelfCritic calls as well: two entries per step there --
    #  the generator's input and, after sampling, the critic's)
    stride = 2 if name == 'self_critic' else 1
    for s in range(T):
        assert torch.equal(trace['masked_ids'][s].cpu(), gv['step_ids'][stride * s]), f'{name}: ids entering step {s} differ from the reference'
        if name == 'self_critic':
            assert torch.equal(trace['ids'][s].cpu(), gv['step_ids'][2 * s + 1]), f'{name}: ids after step {s} (the critic input) differ from the reference'
    assert torch.equal(got.reshape(B, n).cpu(), gv['final_ids'].reshape(B, n))


def test_parity_vae_vs_reference_golden(golden):
    gv = golden('vae_tiny.pt')
    v = mm.VQGanVAE(**gv['cfg']).copy_for_eval()
    v.load_state_dict(sd_f32(gv['sd']))
    v = v.to(DEV).set_precision('parity')
    scale = gv['decoded'].abs().max().item()
    _close('tiny decoded pixels vs reference', v.decode_from_ids(gv['ids'].to(DEV)), gv['decoded'], 1e-3 * scale)
    fmap, ids, aux = v.encode(gv['image'].to(DEV))
    assert torch.equal(ids.cpu(), gv['enc_ids']), f'{(ids.cpu() != gv["enc_ids"]).sum().item()} LFQ ids differ from the reference'
    _close('tiny quantized fmap vs reference', fmap, gv['enc_fmap'], 1e-5)
    _close('tiny decode(fmap) vs reference path', v.decode(gv['enc_fmap'].to(DEV)), O.vae_decode(sd_f32(gv['sd']), gv['enc_fmap']), 1e-3 * scale)


# ------------------------------------------------------------------------------------------------ general fp32 weights (round 4)
@pytest.mark.parametrize('precision', TIERS)
def test_general_fp32_checkpoint_vs_reference_golden(golden, precision):
    """tiny_fp32.pt (oracle/make_golden_fp32.py): the reference run on parameters that were NOT rounded to bf16 -- what every checkpoint the
    reference initialises or trains holds.  Forward / guidance logits within the north star's bound, every step of a 4-step decode and the
    final ids equal to the reference's, VAE pixels and LFQ ids; 'bf16x3' needs all six term products here, 'f16x2' three.  (An engine that
    packed its weights through bf16 is 20x outside the logits bound on this fixture: test_oracle_vs_golden.py.)"""
    g = golden('tiny_fp32.pt')
    t = mm.MaskGitTransformer(t5_name='t5-small', **g['cfg'])
    t.load_state_dict(g['sd'])
    t = t.to(DEV).eval().set_precision(precision)
    if precision != 'parity':
        assert t.split_products() == dict(bf16x3=6, f16x2=3)[precision]
    ids, te = g['ids'].to(DEV), g['text_embeds'].to(DEV)
    lc, emb = t(ids, text_embeds=te, return_embed=True)
    ln = t(ids, text_embeds=te, cond_drop_prob=1.)
    sc = t.forward_with_cond_scale(ids, text_embeds=te, cond_scale=3.)
    scale = g['logits_cond'].abs().max().item()          # to_logits x8: logits are ~20x unit scale
    _close(f'{precision} fp32-checkpoint logits(cond) vs reference', lc, g['logits_cond'], 1e-3 * max(1., scale / 8))
    _close(f'{precision} fp32-checkpoint logits(null) vs reference', ln, g['logits_null'], 1e-3 * max(1., scale / 8))
    _close(f'{precision} fp32-checkpoint logits(guidance) vs reference', sc, g['logits_scaled'], 5e-3 * max(1., scale / 8))
    _close(f'{precision} fp32-checkpoint embed vs reference', emb, g['embed'], 1e-3)
    gen = g['generate']
    v = mm.VQGanVAE(**g['vae']['cfg'])
    v.load_state_dict(g['vae']['sd'], strict=False)
    mg = mm.MaskGit(image_size=128, transformer=t, vae=v).to(DEV).eval()
    mg.set_precision(precision)
    trace = {}
    out = mg.generate(['a', 'b'], timesteps=gen['timesteps'], text_embeds=g['text_embeds'], noise=torch.stack(gen['uniform']), noise_kind='uniform', trace=trace,
                      return_ids=True)
    for s_ in range(gen['timesteps']):
        assert torch.equal(trace['masked_ids'][s_].cpu(), gen['step_ids'][s_]), f'ids entering step {s_} differ from the reference'
    assert torch.equal(out.cpu(), gen['final_ids'])
    px = gen['images'].abs().max().item()
    _close(f'{precision} fp32-checkpoint decoded images vs reference', mg.vae.decode_from_ids(out), gen['images'], 1e-3 * px)
    gv = g['vae']
    _close(f'{precision} fp32-checkpoint decoded pixels vs reference', mg.vae.decode_from_ids(gv['ids'].to(DEV)), gv['decoded'], 1e-3 * gv['decoded'].abs().max().item())
    _, eids, _ = mg.vae.encode(gv['image'].to(DEV))
    safe = (gv['enc_pre_sign'].abs() > 2e-5 * gv['enc_pre_sign'].abs().max()).all(dim=-1).reshape(eids.shape)
    assert bool((eids.cpu() == gv['enc_ids'])[safe].all()) and safe.float().mean().item() > 0.9
