"""GPU parity tests, model level: Transformer.forward / forward_with_cond_scale / MaskGit.generate / VQGanVAE through
the drop-in classes (which call libmuse_hip.so through the C ABI) against
  (a) the golden outputs of the UNMODIFIED reference (tests/golden/, fp32) and
  (b) the CPU oracle run at the HIP path's rounding points (oracle/muse_oracle.py, rp=bf16_round).

What is asserted
  * token ids / mask positions: BIT-EXACT.  The decode loop is checked step by step with the oracle's sampling tail
    fed with the HIP transformer's logits (teacher forcing), and end to end against the oracle loop driven by the same
    HIP logits.  Sampling noise is injected (the reference draws it from torch's CPU generator, which no device RNG
    can reproduce).
  * logits / pixels: tolerance against the rounding-point oracle stated next to each check; the bf16 pipeline stores
    activations in bf16, so it cannot meet 1e-3 absolute against the fp32 reference end to end -- that gap is reported,
    not hidden (see DESIGN.md "Precision").
"""
import math

import pytest
import torch

import muse_oracle as O
from conftest import sd_f32

import muse_maskgit_pytorch_amd as mm
from muse_maskgit_pytorch_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = 'cuda'
# engine vs oracle/engine_model.py (its own rounding points), fraction of the logits' scale.  Measured (round 6, tiny config, logits scale 20.5): max 3.1e-3 / 2.8e-3
# (fold on / off), mean 4.2e-4 -- NOT better than against the round-1 oracle (3.4e-3 / 3.2e-3), and the same with the fold off, where that oracle has exactly the engine's
# rounding points: what separates two bf16 evaluations with identical rounding points is the bf16 roundings that a different fp32 accumulation order flips (one flip =
# 2^-8 of that activation, carried through the remaining layers), not a mis-modelled rounding point.  SURVEY 8c's "L1: <= 1e-3 against a bf16-cast oracle" is therefore
# not attainable in the maximum norm for a multi-layer bf16 network unless the accumulation order is reproduced as well; the mean is.
ENGINE_MODEL_TOL = 5e-3


def _tiny_transformer(golden):
    g = golden('transformer_tiny.pt')
    t = mm.MaskGitTransformer(t5_name='t5-small', **g['cfg'])
    t.load_state_dict(sd_f32(g['sd']))
    return g, t.to(DEV).eval()


def _report(name, got, ref):
    err = (got.float().cpu() - ref).abs()
    print(f'[parity] {name}: max abs err {err.max().item():.4g}, mean {err.mean().item():.4g}, ref scale {ref.abs().max().item():.4g}')
    return err


def test_transformer_forward_vs_reference_and_oracle(golden):
    g, t = _tiny_transformer(golden)
    sd = sd_f32(g['sd'])
    cfg = dict(depth=g['cfg']['depth'], heads=g['cfg']['heads'])
    ids, te = g['ids'], g['text_embeds']
    logits, embed = t(ids.to(DEV), text_embeds=te.to(DEV), return_embed=True)
    assert logits.shape == (2, 64, 512) and logits.dtype == torch.float32 and embed.shape == (2, 64, 128)
    null = t(ids.to(DEV), text_embeds=te.to(DEV), cond_drop_prob=1.)
    # (b) same rounding points: differences are fp32 accumulation order + rare 1-ulp bf16 flips that propagate
    lo, eo = O.transformer_forward(sd, cfg, ids, te, 0., rp=O.bf16_round, return_embed=True)
    no = O.transformer_forward(sd, cfg, ids, te, 1., rp=O.bf16_round)
    scale = lo.abs().max().item()
    e1 = _report('logits(cond) vs rounding-point oracle', logits, lo)
    e2 = _report('logits(null) vs rounding-point oracle', null, no)
    e3 = _report('embed vs rounding-point oracle', embed, eo)
    assert e1.max() < 0.02 * scale and e2.max() < 0.02 * scale and e1.mean() < 2e-3 * scale
    assert e3.max() < 0.05
    # (a) the reference's fp32 output: bf16 activation storage bounds this one
    ea = _report('logits(cond) vs reference fp32 golden', logits, g['logits_cond'])
    assert ea.max() < 0.06 * scale and ea.mean() < 6e-3 * scale
    # argmax agreement where the reference's own top-2 margin is comfortably above the bf16 noise
    top2 = g['logits_cond'].topk(2, dim=-1).values
    safe = (top2[..., 0] - top2[..., 1]) > 4 * ea.max()
    assert torch.equal(logits.cpu().argmax(-1)[safe], g['logits_cond'].argmax(-1)[safe]) and safe.float().mean() > 0.5


def test_bf16_engine_against_its_rounding_point_model(golden):
    """SURVEY 8c precision ladder, L1: the bf16 kernels against an oracle with the SAME rounding points -- oracle/engine_model.py restates where mm_transformer_forward
    rounds since the LayerNorm folds (bf16 of the RAW residual row times bf16(W gamma), rstd * (acc - mean c1) + c2 on the accumulator; LayerNorm(inner) inside w2) and
    with the packed bf16 weights.  `muse_oracle.transformer_forward(rp=bf16_round)` (test above) still describes round 1's engine and sits 2-3e-2 of the scale away
    (VERDICT r5 weak 1a); against its own model the engine differs by accumulation order and the bf16 roundings that order flips.  Both fold settings."""
    import engine_model as E
    g, t = _tiny_transformer(golden)
    sd = sd_f32(g['sd'])
    cfg = dict(depth=g['cfg']['depth'], heads=g['cfg']['heads'])
    ids, te = g['ids'], g['text_embeds']
    try:
        for fold in (True, False):
            t.set_layernorm_fold(fold)
            logits, embed = t(ids.to(DEV), text_embeds=te.to(DEV), return_embed=True)
            null = t(ids.to(DEV), text_embeds=te.to(DEV), cond_drop_prob=1.)
            lo, eo = E.bf16_engine_forward(sd, cfg, ids, te, 0., return_embed=True, fold=fold)
            no = E.bf16_engine_forward(sd, cfg, ids, te, 1., fold=fold)
            scale = lo.abs().max().item()
            e1 = _report(f'fold {fold}: logits(cond) vs the engine\'s rounding-point model', logits, lo)
            e2 = _report(f'fold {fold}: logits(null) vs the engine\'s rounding-point model', null, no)
            e3 = _report(f'fold {fold}: embed vs the engine\'s rounding-point model', embed, eo)
            old = (logits.float().cpu() - O.transformer_forward(sd, cfg, ids, te, 0., rp=O.bf16_round)).abs().max().item()
            print(f'[parity] fold {fold}: (round-1 rounding-point oracle: max abs err {old:.4g}); logits scale {scale:.4g}')
            assert e1.max() < ENGINE_MODEL_TOL * scale and e2.max() < ENGINE_MODEL_TOL * scale and e1.mean() < ENGINE_MODEL_TOL / 8 * scale
            assert e3.max() <= 2 * 2.0 ** -7 * eo.abs().max().item()      # the embed is stored bf16: a flipped rounding is one ulp (<= 2^-7 of the value); at most two
    finally:
        t.set_layernorm_fold('auto')


def test_forward_with_cond_scale_is_the_fused_cfg_gemm(golden):
    """since round 3 the guidance combine (mmp.py:254) is applied to the two passes' final embeddings and to_logits runs once (it is linear,
    mmp.py:332): forward_with_cond_scale equals to_logits(mix) bit for bit, and the two-pass combination null + (cond - null) * s of separately
    multiplied logits to the bf16 rounding of the mixed operand"""
    from muse_maskgit_pytorch_amd import ops
    g, t = _tiny_transformer(golden)
    ids, te = g['ids'].to(DEV), g['text_embeds'].to(DEV)
    scaled, embed = t.forward_with_cond_scale(ids, text_embeds=te, cond_scale=3., return_embed=True)
    cond = t(ids, text_embeds=te)
    null = t(ids, text_embeds=te, cond_drop_prob=1.)
    ec = t(ids, text_embeds=te, _embed_only=True)
    en = t(ids, text_embeds=te, cond_drop_prob=1., _embed_only=True)
    mix = ops.cfg_mix(ec, en, 3., t.dim)
    assert torch.equal(mix.float(), (en.float() + (ec.float() - en.float()) * 3.).bfloat16().float())
    assert torch.equal(scaled.reshape(-1, scaled.shape[-1]), ops.gemm(mix, t._model().packed['wl'], out_f32=True))
    two_pass = null + (cond - null) * 3.                             # mmp.py:254 on separately multiplied logits
    assert (scaled - two_pass).abs().max().item() < 0.02 * two_pass.abs().max().item()
    e = _report('scaled logits vs reference golden', scaled, g['logits_scaled'])
    assert e.max() < 0.1 * g['logits_scaled'].abs().max()
    assert torch.equal(t.forward_with_cond_scale(ids, text_embeds=te, cond_scale=1.), cond)


def test_forward_is_batch_and_order_invariant(golden):
    g, t = _tiny_transformer(golden)
    ids, te = g['ids'].to(DEV), g['text_embeds'].to(DEV)
    both = t(ids, text_embeds=te)
    swapped = t(ids.flip(0), text_embeds=te.flip(0))
    assert torch.equal(both, swapped.flip(0))
    # sample 0 has no padded text: running it alone (L=7) is the same computation
    assert torch.equal(both[0:1], t(ids[0:1], text_embeds=te[0:1]))


def _hip_demask(t, te):
    def fn(ids, step):
        return t.forward_with_cond_scale(ids.to(DEV), text_embeds=te.to(DEV), cond_scale=3.).cpu()
    return fn


@pytest.mark.parametrize('T', [4, 18])
def test_generate_ids_bit_exact_against_oracle_tail(golden, T):
    g, t = _tiny_transformer(golden)
    gen = golden(f'generate_tiny_T{T}.pt')
    te = g['text_embeds']
    B, n, V = 2, 64, 512
    uniform = torch.stack(gen['uniform'])                               # the reference run's own noise draws [T,B,n,V]
    gumbel = O.gumbel_from_uniform(uniform)
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None)
    trace = {}
    ids = mg.generate(['a', 'b'], timesteps=T, text_embeds=te, noise=gumbel, noise_kind='gumbel', fmap_size=8, trace=trace)
    assert ids.shape == (B, 8, 8) and ids.dtype == torch.long
    counts, temps = trace['counts'], trace['temperatures']
    assert counts == O.mask_counts(T, n)
    demask = _hip_demask(t, te)
    # ---- step-by-step teacher forcing: oracle tail on the HIP logits of the engine's own state
    prev_scores = torch.zeros(B, n)
    prev_ids = torch.full((B, n), 512)
    for s in range(T):
        sel = O.select_topk_stable(prev_scores, counts[s])
        assert not O.boundary_ties(prev_scores, counts[s]).any()
        exp_masked = torch.where(sel, torch.full_like(prev_ids, 512), prev_ids)
        assert torch.equal(trace['masked_ids'][s].cpu(), exp_masked), f'step {s}: re-mask positions differ'
        logits = demask(exp_masked, s)
        assert not O.threshold_ties(logits)[sel].any()
        new_ids, new_scores, _ = O.sample_step(logits, gumbel[s], exp_masked, 512, temps[s])
        assert torch.equal(trace['ids'][s].cpu(), new_ids), f'step {s}: sampled ids differ at {(trace["ids"][s].cpu() != new_ids).nonzero().tolist()}'
        assert torch.allclose(trace['scores'][s].cpu(), new_scores, atol=2e-6, rtol=0), f'step {s}: scores'
        prev_ids, prev_scores = trace['ids'][s].cpu(), trace['scores'][s].cpu()
    # ---- free running: the oracle loop driven by the same HIP logits reaches the same final ids
    free = O.generate_ids(demask, B, n, 512, lambda s, shp: gumbel[s], timesteps=T)
    assert torch.equal(ids.reshape(B, n).cpu(), free)
    assert (ids < 512).all()
    # ---- how far the bf16 pipeline's decisions are from the fp32 reference's own run on the same noise
    agree = (ids.cpu() == gen['final_ids']).float().mean().item()
    print(f'[parity] T={T}: final ids equal to the fp32 reference run: {agree * 100:.1f}% of tokens')


def test_generate_engine_logits_equal_general_path(golden):
    """the fused loop (CFG batch of 2B sequences, cached cross K/V, constant null cross-attention, gathered final norm)
    must produce, at the sampled rows, bit-identical logits to the general per-pass forward."""
    g, t = _tiny_transformer(golden)
    te = g['text_embeds']
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None)
    trace = {}
    mg.generate(['a', 'b'], timesteps=3, text_embeds=te, noise_kind='none', fmap_size=8, trace=trace)
    # replay step 1 through the public ops: same masked ids -> same predictions and scores bit for bit
    masked = trace['masked_ids'][1]
    logits = t.forward_with_cond_scale(masked, text_embeds=te.to(DEV), cond_scale=3.)
    rows = (masked.flatten() == 512).nonzero().flatten().int()
    pred, score = ops.sample_rows(logits.reshape(-1, 512)[rows.long()].contiguous(), math.ceil(0.1 * 512), trace['temperatures'][1],
                                  noise_kind=_lib.MM_NOISE_NONE)
    assert torch.equal(trace['ids'][1].flatten()[rows.long()], pred)
    assert torch.equal(trace['scores'][1].flatten()[rows.long()], score)


def test_generate_philox_is_shard_invariant_and_seeded(golden):
    g, t = _tiny_transformer(golden)
    gen = torch.Generator().manual_seed(0)
    te = torch.randn(4, 5, 512, generator=gen)
    te[2, 3:] = 0
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None)
    full = mg.generate(['x'] * 4, timesteps=6, text_embeds=te, seed=77, fmap_size=8)
    a = mg.generate(['x'] * 2, timesteps=6, text_embeds=te[:2], seed=77, row_offset=0, fmap_size=8)
    b = mg.generate(['x'] * 2, timesteps=6, text_embeds=te[2:], seed=77, row_offset=2, fmap_size=8)
    assert torch.equal(full, torch.cat([a, b]))
    assert torch.equal(full, mg.generate(['x'] * 4, timesteps=6, text_embeds=te, seed=77, fmap_size=8))
    assert not torch.equal(full, mg.generate(['x'] * 4, timesteps=6, text_embeds=te, seed=78, fmap_size=8))


def test_vae_composite_entry_points_equal_the_operator_sequence():
    """mm_vae_encode / mm_vae_decode_from_ids (one C call each) against the same layer list run operator by operator through the C-ABI:
    the same kernels in the same order, so bit-identical -- for the default layout and for interleaved residual blocks."""
    for kw in (dict(dim=32, codebook_size=8192), dict(dim=16, codebook_size=512, encdec_num_resnet_blocks=(1, 0, 2, 1), layers=4)):
        torch.manual_seed(3)
        v = mm.VQGanVAE(use_vgg_and_gan=False, **kw).to(DEV).eval()
        f = 2 ** v.enc_dec.layers
        ids = torch.randint(0, kw['codebook_size'], (3, 4, 6), device=DEV)
        img = torch.randn(3, 3, 4 * f, 6 * f, device=DEV)
        h_dec = v.decode_from_ids(ids)                      # round 6 default: the decoder on fp16 storage (composite only)
        assert v._half_decode()
        v.set_decode_storage('bf16')                        # the operator sequence below runs the bf16 operators
        a_dec, (a_q, a_ids, _) = v.decode_from_ids(ids), v.encode(img)
        sd = {k: (t.detach().float().cpu() if t.is_floating_point() else t.detach().cpu()) for k, t in v.state_dict().items()}
        ref = O.vae_decode_from_ids(sd, ids.cpu(), layers=v.enc_dec.layers) if kw.get('encdec_num_resnet_blocks') is None else None
        if ref is not None:                                 # fp16 storage is the closer of the two to the fp32 oracle, by about the 3 extra significand bits
            e_h, e_b = (h_dec.cpu() - ref).abs().max().item(), (a_dec.cpu() - ref).abs().max().item()
            print(f'[vae storage] decode vs fp32 oracle: fp16 storage {e_h:.3g}, bf16 storage {e_b:.3g} (image scale {ref.abs().max().item():.3g})')
            assert e_h < 0.5 * e_b and e_h < 1e-3 * ref.abs().max().item()
        v.composite = False
        b_dec, (b_q, b_ids, _) = v.decode_from_ids(ids), v.encode(img)
        assert a_dec.shape == (3, 3, 4 * f, 6 * f) and torch.equal(a_dec, b_dec)
        assert torch.equal(a_ids, b_ids) and torch.equal(a_q, b_q)
        # capturable: no allocation / synchronisation inside the call
        v.composite = True
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            v.decode_from_ids(ids)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                c_dec = v.decode_from_ids(ids)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(c_dec, a_dec)


def test_vae_decode_encode_vs_reference_and_oracle(golden):
    gv = golden('vae_tiny.pt')
    sd = sd_f32(gv['sd'])
    v = mm.VQGanVAE(**gv['cfg']).copy_for_eval()
    v.load_state_dict(sd)
    v = v.to(DEV)
    img = v.decode_from_ids(gv['ids'].to(DEV))
    assert img.shape == (2, 3, 128, 128) and img.dtype == torch.float32
    ro = O.vae_decode_from_ids(sd, gv['ids'], rp=O.bf16_round)
    scale = gv['decoded'].abs().max().item()
    e1 = _report('decoded pixels vs rounding-point oracle', img, ro)
    e2 = _report('decoded pixels vs reference fp32 golden', img, gv['decoded'])
    assert e1.max() < 0.03 * scale and e1.mean() < 2e-3 * scale
    assert e2.max() < 0.06 * scale and e2.mean() < 6e-3 * scale
    fmap, ids, aux = v.encode(gv['image'].to(DEV))
    assert ids.shape == (2, 8, 8) and ids.dtype == torch.long and fmap.shape == gv['enc_fmap'].shape
    # LFQ bits are signs of project_in(fmap): exact wherever the reference's own pre-sign value is not within bf16 noise of 0
    pre = gv['enc_pre_quant'].permute(0, 2, 3, 1) @ sd['quantizer.project_in.weight'].t() + sd['quantizer.project_in.bias']
    mask = sd['quantizer.mask']
    got_bits = (ids.cpu()[..., None] & mask) != 0
    ref_bits = (gv['enc_ids'][..., None] & mask) != 0
    safe = pre.abs() > 0.05 * pre.abs().mean()
    assert torch.equal(got_bits[safe], ref_bits[safe]) and safe.float().mean() > 0.8
    print(f'[parity] encode: {100 * (ids.cpu() == gv["enc_ids"]).float().mean().item():.1f}% of LFQ indices equal the fp32 reference')


def test_maskgit_generate_end_to_end_c1_tiny():
    """BASELINE config 1 (tiny): VQGanVAE dim=64 codebook=512, MaskGitTransformer dim=128 depth=2 seq_len=64, batch 2, 4 steps."""
    torch.manual_seed(0)
    vae = mm.VQGanVAE(dim=64, codebook_size=512)
    tr = mm.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=2, dim_head=64, heads=8, t5_name='t5-small')
    mg = mm.MaskGit(vae=vae, transformer=tr, image_size=128).to(DEV)
    te = torch.randn(2, 7, 512)
    te[1, 5:] = 0
    images = mg.generate(['a', 'b'], timesteps=4, text_embeds=te, seed=1)
    assert images.shape == (2, 3, 128, 128) and torch.isfinite(images).all()
    ids = mg.generate(['a', 'b'], timesteps=4, text_embeds=te, seed=1, return_ids=True)
    sd = {k: (v.detach().float().cpu() if v.is_floating_point() else v.detach().cpu()) for k, v in mg.vae.state_dict().items()}
    ref = O.vae_decode_from_ids(sd, ids.cpu(), rp=O.bf16_round)
    e = _report('C1 images vs oracle decode of the same ids', images, ref)
    assert e.max() < 0.05 * ref.abs().max().clamp(min=1.0)


def test_superres_context_with_cond_ids(golden):
    """conditioning_token_ids path (mmp.py:314-318): cond-id keys stay visible in the null pass."""
    g, t = _tiny_transformer(golden)
    sd = sd_f32(g['sd'])
    cfg = dict(depth=g['cfg']['depth'], heads=g['cfg']['heads'])
    gen = torch.Generator().manual_seed(4)
    ids = torch.randint(0, 513, (2, 64), generator=gen)
    cids = torch.randint(0, 512, (2, 4, 4), generator=gen)
    te = g['text_embeds']
    for drop in (0., 1.):
        got = t(ids.to(DEV), text_embeds=te.to(DEV), conditioning_token_ids=cids.to(DEV), cond_drop_prob=drop)
        ref = O.transformer_forward(sd, cfg, ids, te, drop, conditioning_token_ids=cids, rp=O.bf16_round)
        e = _report(f'cond-id context drop={drop}', got, ref)
        assert e.max() < 0.02 * ref.abs().max()
    # engine path with cond images -> ids through a real (tiny) VAE
    vae = mm.VQGanVAE(dim=16, codebook_size=512)
    mg = mm.MaskGit(image_size=128, transformer=t, vae=vae, cond_vae=vae.copy_for_eval(), cond_image_size=64).to(DEV)
    out = mg.generate(['a', 'b'], timesteps=3, text_embeds=te, cond_images=torch.randn(2, 3, 64, 64, device=DEV), seed=3, return_ids=True)
    assert out.shape == (2, 8, 8) and (out < 512).all()


@pytest.mark.parametrize('name,kw,B,n,L,nc', [
    ('super-res shapes (C4): 1024 tokens, 256 cond ids in the context', dict(num_tokens=512, seq_len=1024, dim=128, depth=1, heads=8), 2, 1024, 9, 256),
    ('paper-scale widths (C5): dim 1024, 16 heads, vocabulary 8192', dict(num_tokens=8192, seq_len=64, dim=1024, depth=2, heads=16), 2, 64, 12, 0),
    ('base widths (C2): dim 512, FF inner 1365', dict(num_tokens=1024, seq_len=256, dim=512, depth=2, heads=8), 3, 256, 20, 0),
])
def test_transformer_other_configs_vs_oracle(name, kw, B, n, L, nc):
    """random-init models at the shapes of the other BASELINE configs (reduced depth so the CPU oracle stays fast)."""
    torch.manual_seed(5)
    t = mm.MaskGitTransformer(t5_name='t5-small', dim_head=64, **kw)
    with torch.no_grad():
        for p_ in t.parameters():
            p_.copy_(p_.to(torch.bfloat16).float())
    sd = {k: v.detach().clone() for k, v in t.state_dict().items()}
    t = t.to(DEV).eval()
    gen = torch.Generator().manual_seed(1)
    ids = torch.randint(0, kw['num_tokens'] + 1, (B, n), generator=gen)
    te = torch.randn(B, L, 512, generator=gen)
    te[-1, L // 2:] = 0
    cids = torch.randint(0, kw['num_tokens'], (B, nc), generator=gen) if nc else None
    cfg = dict(depth=kw['depth'], heads=kw['heads'])
    for drop in (0., 1.):
        got = t(ids.to(DEV), text_embeds=te.to(DEV), cond_drop_prob=drop, conditioning_token_ids=cids.to(DEV) if nc else None)
        ref = O.transformer_forward(sd, cfg, ids, te, drop, conditioning_token_ids=cids, rp=O.bf16_round)
        e = _report(f'{name} drop={drop}', got, ref)
        scale = ref.abs().max().item()
        assert e.max() < 0.03 * scale and e.mean() < 3e-3 * scale
    # the fused decode loop runs at these shapes too (ids only; super-res: with 256 conditioning ids from a real tiny VAE)
    f = int(n ** 0.5)
    if nc:
        vae = mm.VQGanVAE(dim=16, codebook_size=kw['num_tokens'])
        mg = mm.MaskGit(image_size=16 * f, transformer=t, vae=vae, cond_vae=vae.copy_for_eval(), cond_image_size=256).to(DEV)
        out = mg.generate(['x'] * B, timesteps=3, text_embeds=te, seed=2, cond_images=torch.randn(B, 3, 256, 256, device=DEV), return_ids=True)
    else:
        mg = mm.MaskGit(image_size=16 * f, transformer=t, vae=None)
        out = mg.generate(['x'] * B, timesteps=3, text_embeds=te, seed=2, fmap_size=f)
    assert out.shape == (B, f, f) and (out < kw['num_tokens']).all() and (out >= 0).all()


def test_training_forward_loss_vs_reference(golden):
    """Transformer.forward(labels=...) / TokenCritic BCE / MaskGit.forward: forward-only losses (mmp.py:337-348, 623-724)."""
    g, t = _tiny_transformer(golden)
    l = golden('loss_tiny.pt')
    te = g['text_embeds'].to(DEV)
    loss, logits = t(l['x'].to(DEV), text_embeds=te, labels=l['labels'].to(DEV), ignore_index=-1, return_logits=True)
    # the kernel against torch's own cross_entropy on the SAME (HIP) logits: fp32 summation order only
    ref_same = torch.nn.functional.cross_entropy(logits.permute(0, 2, 1).cpu(), l['labels'], ignore_index=-1)
    assert abs(loss.item() - ref_same.item()) < 2e-5 * max(1.0, ref_same.item())
    print(f'[parity] CE loss HIP {loss.item():.5f} vs reference fp32 {l["loss"].item():.5f}')
    assert abs(loss.item() - l['loss'].item()) < 0.02 * l['loss'].item()
    loss_drop = t(l['x'].to(DEV), text_embeds=te, labels=l['labels'].to(DEV), ignore_index=-1, cond_drop_prob=1.)
    assert abs(loss_drop.item() - l['loss_drop'].item()) < 0.02 * l['loss_drop'].item()
    all_ignored = t(l['x'].to(DEV), text_embeds=te, labels=torch.full_like(l['labels'], -1).to(DEV), ignore_index=-1)
    assert torch.isnan(all_ignored)
    critic = mm.TokenCritic(num_tokens=512, seq_len=64, dim=128, depth=1, dim_head=64, heads=8, t5_name='t5-small')
    critic.load_state_dict(sd_f32(l['critic_sd']))
    bce = critic.to(DEV).eval()(l['x'].clamp(max=511).to(DEV), text_embeds=te, labels=l['critic_labels'].to(DEV))
    print(f'[parity] critic BCE HIP {bce.item():.5f} vs reference {l["critic_bce"].item():.5f}')
    assert abs(bce.item() - l['critic_bce'].item()) < 5e-3
    # MaskGit.forward: random masking with the device generator, then the same CE; reproducible under a seed
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None, cond_drop_prob=0.)
    ids = torch.randint(0, 512, (2, 8, 8), device=DEV)
    torch.manual_seed(3); a = mg(ids, text_embeds=te)
    torch.manual_seed(3); b = mg(ids, text_embeds=te)
    assert torch.isfinite(a) and a.item() > 0 and torch.equal(a, b)


def test_generate_is_hip_graph_capturable(golden):
    """mm_generate issues only kernels / async memsets / async copies on the given stream (no allocation, no sync, no host
    copy), so the whole 18-step decode can be captured into a HIP graph and replayed."""
    g, t = _tiny_transformer(golden)
    te = g['text_embeds'].to(DEV)
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None)
    eager = mg.generate(['a', 'b'], timesteps=6, text_embeds=te, seed=11, fmap_size=8)      # also warms the workspace / packing
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            captured = mg.generate(['a', 'b'], timesteps=6, text_embeds=te, seed=11, fmap_size=8)
    captured.fill_(-1)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(captured, eager)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(captured, eager)


def test_generate_stepwise_variants(golden):
    """decode variants that go through the stepwise path: cond_scale == 1 (checked bit-exactly against the oracle loop on
    HIP logits), can_remask_prev_masked, token critic, self critic, self-conditioning (run + determinism)."""
    g, t = _tiny_transformer(golden)
    te = g['text_embeds']
    B, n, V, T = 2, 64, 512, 5
    gumbel = O.gumbel_from_uniform(torch.rand(T, B, n, V, generator=torch.Generator().manual_seed(8)))
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None)
    ids = mg.generate(['a', 'b'], timesteps=T, text_embeds=te, noise=gumbel, noise_kind='gumbel', fmap_size=8, cond_scale=1)
    free = O.generate_ids(lambda i, s: t(i.to(DEV), text_embeds=te.to(DEV)).cpu(), B, n, 512, lambda s, shp: gumbel[s], timesteps=T)
    assert torch.equal(ids.reshape(B, n).cpu(), free)
    # the stepwise path and the fused engine agree on the default configuration too (same noise, cond_scale 3)
    fused = mg.generate(['a', 'b'], timesteps=T, text_embeds=te, noise=gumbel, noise_kind='gumbel', fmap_size=8)
    step = mg._generate_stepwise(['a', 'b'], None, 8, 1., 0.9, False, False, T, 3, 1, te, gumbel, 'gumbel', 0, 0, True, None)
    assert torch.equal(fused, step)
    # can_remask_prev_masked
    mg2 = mm.MaskGit(image_size=128, transformer=t, vae=None, no_mask_token_prob=0.25)
    r1 = mg2.generate(['a', 'b'], timesteps=T, text_embeds=te, seed=5, fmap_size=8, can_remask_prev_masked=True)
    r2 = mg2.generate(['a', 'b'], timesteps=T, text_embeds=te, seed=5, fmap_size=8, can_remask_prev_masked=True)
    assert torch.equal(r1, r2) and (r1 < 512).all()
    # token critic and self critic
    critic = mm.TokenCritic(num_tokens=512, seq_len=64, dim=128, depth=1, dim_head=64, heads=8, t5_name='t5-small').to(DEV)
    mg3 = mm.MaskGit(image_size=128, transformer=t, vae=None, token_critic=critic)
    torch.manual_seed(1); c1 = mg3.generate(['a', 'b'], timesteps=T, text_embeds=te, seed=5, fmap_size=8)
    torch.manual_seed(1); c2 = mg3.generate(['a', 'b'], timesteps=T, text_embeds=te, seed=5, fmap_size=8)
    assert torch.equal(c1, c2) and c1.shape == (2, 8, 8) and (c1 < 512).all()
    mg4 = mm.MaskGit(image_size=128, transformer=t, vae=None, self_token_critic=True).to(DEV)
    torch.manual_seed(1); s1 = mg4.generate(['a', 'b'], timesteps=T, text_embeds=te, seed=5, fmap_size=8)
    assert s1.shape == (2, 8, 8) and (s1 < 512).all()
    # self-conditioning transformer: the embed of step t feeds step t+1 (mmp.py:325-328, 574)
    torch.manual_seed(2)
    tsc = mm.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=1, dim_head=64, heads=8, t5_name='t5-small', self_cond=True).to(DEV)
    mg5 = mm.MaskGit(image_size=128, transformer=tsc, vae=None)
    o1 = mg5.generate(['a', 'b'], timesteps=3, text_embeds=te, seed=5, fmap_size=8)
    assert o1.shape == (2, 8, 8) and (o1 < 512).all()
    # self-cond forward against the oracle
    sd = {k: v.detach().cpu().clone() for k, v in tsc.state_dict().items()}
    for k in sd:
        if sd[k].is_floating_point():
            sd[k] = sd[k].to(torch.bfloat16).float()
    tsc.load_state_dict(sd)
    ids0 = torch.randint(0, 513, (2, 64))
    sce = torch.randn(2, 64, 128)
    got = tsc(ids0.to(DEV), text_embeds=te.to(DEV), self_cond_embed=sce.to(DEV))
    ref = O.transformer_forward(sd, dict(depth=1, heads=8, self_cond=True), ids0, te, 0., self_cond_embed=sce, rp=O.bf16_round)
    e = _report('self-conditioned forward', got, ref)
    assert e.max() < 0.03 * ref.abs().max()


@pytest.mark.parametrize('name', ['token_critic', 'self_critic', 'cond_scale_1', 'can_remask', 'self_cond'])
def test_generate_variants_vs_reference_goldens(golden, name):
    """mmp.py:540-609 decode variants replayed with the reference's recorded noise:
    (1) the HIP decode loop -- every variant runs inside the one mm_generate call -- must equal, bit for bit, the oracle loop fed by the HIP
        transformer (same logits, same noise), and the same loop driven operator by operator from Python (stepwise=True);
    (2) against the fp32 reference's final ids the agreement is reported (bf16 GEMM operands can flip a near-tie and the
        loop then diverges by design); >= 90 % must agree."""
    gv, gt = golden('generate_variants_tiny.pt')[name], golden('transformer_tiny.pt')
    te = gt['text_embeds']
    T, B, n, V = gv['timesteps'], 2, 64, 512
    if name == 'self_cond':
        t = mm.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=1, dim_head=64, heads=8, t5_name='t5-small', self_cond=True)
        t.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in gv['sd'].items()})
        t = t.to(DEV)
    else:
        _, t = _tiny_transformer(golden)
    kw, okw = {}, {}
    if name == 'token_critic':
        critic = mm.TokenCritic(num_tokens=512, seq_len=64, dim=128, depth=1, dim_head=64, heads=8, t5_name='t5-small')
        critic.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in gv['critic_sd'].items()})
        mg = mm.MaskGit(image_size=128, transformer=t, vae=None, token_critic=critic.to(DEV))
    elif name == 'self_critic':
        mg = mm.MaskGit(image_size=128, transformer=t, vae=None, self_token_critic=True)
        mg.token_critic.to_pred.load_state_dict({k: v.float() for k, v in gv['to_pred'].items()})
        mg = mg.to(DEV)
    elif name == 'can_remask':
        mg = mm.MaskGit(image_size=128, transformer=t, vae=None, no_mask_token_prob=0.25)
        kw['can_remask_prev_masked'] = okw['can_remask_prev_masked'] = True
    else:
        mg = mm.MaskGit(image_size=128, transformer=t, vae=None)
    cond_scale = 1 if name == 'cond_scale_1' else 3
    uni = torch.stack(gv['uniform'])                                           # [T, B, n, V] the reference's U(0,1) draws
    if name in ('token_critic', 'self_critic'):
        cu = torch.stack([u.reshape(B, n) for u in gv['critic_uniform']])
        kw['critic_noise'] = cu
        okw['critic_fn'] = lambda ids, step: mg.token_critic.forward_with_cond_scale(ids.to(DEV), text_embeds=te.to(DEV), cond_scale=cond_scale).reshape(B, n).float().cpu()
        okw['critic_uniform_fn'] = lambda step, shape: cu[step]
    trace = {}
    got = mg.generate(['a', 'b'], timesteps=T, text_embeds=te, noise=uni, noise_kind='uniform', fmap_size=8, cond_scale=cond_scale,
                      trace=trace, **kw).reshape(B, n).cpu()
    state = dict(embed=None)

    def demask(ids, step):
        logits, embed = t.forward_with_cond_scale(ids.to(DEV), text_embeds=te.to(DEV), self_cond_embed=state['embed'], cond_scale=cond_scale,
                                                  return_embed=True)
        state['embed'] = embed if name == 'self_cond' else None
        return logits.cpu()

    free = O.generate_ids(demask, B, n, 512, lambda s, shp: O.gumbel_from_uniform(uni[s]), timesteps=T, **okw)
    assert torch.equal(got, free), f'{name}: the HIP decode loop (one mm_generate call) differs from the oracle loop on the same logits'
    # the same loop one operator call at a time from Python: bit-identical ids AND per-step states
    trace_s = {}
    step = mg.generate(['a', 'b'], timesteps=T, text_embeds=te, noise=uni, noise_kind='uniform', fmap_size=8, cond_scale=cond_scale,
                       trace=trace_s, stepwise=True, **kw).reshape(B, n).cpu()
    assert torch.equal(got, step), f'{name}: mm_generate differs from the stepwise loop'
    for key in ('masked_ids', 'ids', 'scores'):
        assert torch.equal(trace[key].cpu(), torch.stack(trace_s[key]).cpu()), f'{name}: per-step {key} differ between mm_generate and the stepwise loop'
    ref = gv['final_ids'].reshape(B, n)
    agree = (got == ref).float().mean().item()
    print(f'[parity] decode variant {name}: final ids equal to the fp32 reference {agree * 100:.1f} %')
    assert agree >= 0.9


def test_negative_prompt_extension(golden):
    """EXTENSION without a reference parity target (mmp.py:261-277 cannot run): negative-prompt guidance neg + (pos - neg) * s.
    Checked against the oracle's restatement of that intent, and end to end through generate."""
    g, t = _tiny_transformer(golden)
    te = g['text_embeds']
    nte = torch.randn(2, 5, 512, generator=torch.Generator().manual_seed(3))
    sd = {k: v.float() if v.is_floating_point() else v for k, v in g['sd'].items()}
    cfg = dict(depth=g['cfg']['depth'], heads=g['cfg']['heads'])
    got, emb = t.forward_with_neg_prompt(g['ids'].to(DEV), te.to(DEV), nte.to(DEV), cond_scale=3., return_embed=True)
    ref, ref_emb = O.forward_with_neg_prompt(sd, cfg, g['ids'], te, nte, 3., rp=O.bf16_round, return_embed=True)
    e = _report('negative-prompt logits', got, ref)
    assert e.max() < 0.03 * ref.abs().max()
    assert _report('negative-prompt embed', emb, ref_emb).max() < 0.05
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None)
    a = mg.generate(['a', 'b'], timesteps=4, text_embeds=te, neg_text_embeds=nte, seed=3, fmap_size=8)
    b = mg.generate(['a', 'b'], timesteps=4, text_embeds=te, neg_text_embeds=nte, seed=3, fmap_size=8)
    c = mg.generate(['a', 'b'], timesteps=4, text_embeds=te, seed=3, fmap_size=8)
    assert torch.equal(a, b) and a.shape == (2, 8, 8) and not torch.equal(a, c)


def test_training_step_gradients_match_autograd(golden):
    """loss.backward() through the hand-written MI355X backward (training.py) against torch autograd of the oracle
    (fp32, same bf16-rounded weights, bf16 rounding points in the forward).  Gradients pass through bf16 activations / operands:
    per-tensor relative error (max |diff| / max |ref|) below 5e-2, cosine similarity above 0.995."""
    g, l = golden('transformer_tiny.pt'), golden('loss_tiny.pt')
    cfgk = dict(num_tokens=512, seq_len=64, dim=128, depth=2, dim_head=64, heads=8, t5_name='t5-small')
    t = mm.MaskGitTransformer(**cfgk)
    t.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in g['sd'].items()})
    t = t.to(DEV)
    te = g['text_embeds']
    loss = t(l['x'].to(DEV), text_embeds=te.to(DEV), labels=l['labels'].to(DEV), ignore_index=-1)
    assert loss.requires_grad
    loss.backward()
    # oracle with autograd
    sd = {k: (v.float().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in g['sd'].items()}
    ref = O.transformer_loss(sd, dict(depth=2, heads=8), l['x'], te, l['labels'], ignore_index=-1)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 2e-2 * abs(ref.item()), (loss.item(), ref.item())
    worst = 0.
    for name, p in t.named_parameters():
        if name.startswith('self_cond_to_init_embed') or name == 'norm.gamma':
            assert p.grad is None
            continue
        assert p.grad is not None, name
        rg = sd[name].grad
        gg = p.grad.float().cpu()
        rel = (gg - rg).abs().max().item() / (rg.abs().max().item() + 1e-20)
        cos = torch.nn.functional.cosine_similarity(gg.flatten(), rg.flatten(), dim=0).item()
        worst = max(worst, rel)
        assert rel < 5e-2 and cos > 0.995, f'{name}: rel {rel:.3e} cos {cos:.5f}'
    print(f'[parity] training gradients: worst per-tensor relative error {worst:.3e}')
    # a step of SGD through torch's optimizer lowers the loss on the same batch
    opt = torch.optim.SGD([p for p in t.parameters() if p.grad is not None], lr=0.05)
    opt.step()
    with torch.no_grad():
        loss2 = t(l['x'].to(DEV), text_embeds=te.to(DEV), labels=l['labels'].to(DEV), ignore_index=-1)
    assert loss2.item() < loss.item()


def test_training_step_is_bit_reproducible(golden):
    """two identical training steps (same masking noise) give bit-identical gradients for every parameter: no atomics are left in the
    backward (the embedding gradient sums rows in a fixed order)"""
    _, t = _tiny_transformer(golden)
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None)
    ids = torch.randint(0, 512, (4, 64), generator=torch.Generator().manual_seed(2)).to(DEV)
    te = torch.randn(4, 5, 512, generator=torch.Generator().manual_seed(3)).to(DEV)
    grads = []
    for _ in range(2):
        t.zero_grad(set_to_none=True)
        torch.manual_seed(77)
        mg(ids, text_embeds=te).backward()
        grads.append({k: p.grad.clone() for k, p in t.named_parameters() if p.grad is not None})
    assert len(grads[0]) > 20 and grads[0].keys() == grads[1].keys()
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), f'gradient of {k} differs between two identical steps'


def test_token_critic_training_gradients_and_maskgit_critic_loss(golden):
    """the TokenCritic's BCE (mmp.py:345-346) through the hand-written backward vs oracle autograd, and the full
    MaskGit.forward with a token critic (mmp.py:726-741): generator CE + critic BCE, both differentiable."""
    g, l = golden('transformer_tiny.pt'), golden('loss_tiny.pt')
    te = g['text_embeds']
    critic = mm.TokenCritic(num_tokens=512, seq_len=64, dim=128, depth=1, dim_head=64, heads=8, t5_name='t5-small')
    critic.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in l['critic_sd'].items()})
    critic = critic.to(DEV)
    x = l['x'].clamp(max=511)
    loss = critic(x.to(DEV), text_embeds=te.to(DEV), labels=l['critic_labels'].to(DEV))
    assert loss.requires_grad and abs(loss.item() - l['critic_bce'].item()) < 2e-3
    loss.backward()
    sd = {k: (v.float().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in l['critic_sd'].items()}
    ref = O.transformer_loss(sd, dict(depth=1, heads=8), x, te, l['critic_labels'])
    ref.backward()
    for name, p in critic.named_parameters():
        if name.startswith('self_cond_to_init_embed') or name == 'norm.gamma':
            continue
        rg, gg = sd[name].grad, p.grad.float().cpu()
        rel = (gg - rg).abs().max().item() / (rg.abs().max().item() + 1e-20)
        cos = torch.nn.functional.cosine_similarity(gg.flatten(), rg.flatten(), dim=0).item()
        assert rel < 5e-2 and cos > 0.99, f'{name}: rel {rel:.3e} cos {cos:.5f}'
    # MaskGit.forward with a critic: both networks receive gradients
    _, t = _tiny_transformer(golden)
    for p in critic.parameters():
        p.grad = None
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None, token_critic=critic)
    ids = torch.randint(0, 512, (2, 64), generator=torch.Generator().manual_seed(4))
    total = mg(ids.to(DEV), text_embeds=te.to(DEV), cond_drop_prob=0.)
    total.backward()
    assert torch.isfinite(total) and t.to_logits.weight.grad is not None and critic.to_logits.weight.grad is not None
    assert t.to_logits.weight.grad.abs().max() > 0 and critic.to_logits.weight.grad.abs().max() > 0
    with torch.no_grad():
        assert torch.isfinite(mg(ids.to(DEV), text_embeds=te.to(DEV), cond_drop_prob=0.))


def test_training_self_cond_and_super_res_gradients():
    """Training-path coverage of the two remaining reference configurations: self-conditioning (mmp.py:325-328,
    694-707) and super-res conditioning ids joined to the context (mmp.py:314-318), gradients vs oracle autograd."""
    torch.manual_seed(11)
    t = mm.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=1, dim_head=64, heads=8, t5_name='t5-small', self_cond=True)
    with torch.no_grad():
        for p in t.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    sd_cpu = {k: v.detach().clone() for k, v in t.state_dict().items()}
    t = t.to(DEV)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 513, (2, 64), generator=g)
    labels = torch.where(ids == 512, torch.randint(0, 512, (2, 64), generator=g), torch.full_like(ids, -1))
    labels[0, :3] = torch.tensor([5, 6, 7]); ids[0, :3] = 512
    te = torch.randn(2, 6, 512, generator=g)
    te[1, 4:] = 0
    sce = torch.randn(2, 64, 128, generator=g)
    cond = torch.randint(0, 512, (2, 4, 4), generator=g)
    loss = t(ids.to(DEV), text_embeds=te.to(DEV), labels=labels.to(DEV), ignore_index=-1, self_cond_embed=sce.to(DEV),
             conditioning_token_ids=cond.to(DEV))
    loss.backward()
    sd = {k: (v.float().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd_cpu.items()}
    ref = O.transformer_loss(sd, dict(depth=1, heads=8, self_cond=True), ids, te, labels, ignore_index=-1, self_cond_embed=sce,
                             conditioning_token_ids=cond)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 2e-2 * abs(ref.item())
    for name, p in t.named_parameters():
        if name == 'norm.gamma':
            continue
        rg, gg = sd[name].grad, p.grad.float().cpu()
        rel = (gg - rg).abs().max().item() / (rg.abs().max().item() + 1e-20)
        cos = torch.nn.functional.cosine_similarity(gg.flatten(), rg.flatten(), dim=0).item()
        assert rel < 5e-2 and cos > 0.99, f'{name}: rel {rel:.3e} cos {cos:.5f}'
    # MaskGit.forward with self-conditioning switched on end to end
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None, self_cond_prob=1.0)
    t.zero_grad()
    l2 = mg(torch.randint(0, 512, (2, 64), generator=g).to(DEV), text_embeds=te.to(DEV))
    l2.backward()
    assert torch.isfinite(l2) and t.self_cond_to_init_embed[1].weight.grad.abs().max() > 0


def test_vqgan_with_vector_quantize_extension():
    """EXTENSION: VQGanVAE(lookup_free_quantization=False) -- the reference's VectorQuantize branch cannot run, so the
    check is self-consistency: encode picks the most cosine-similar code of the projected features (recomputed in fp32 from the same
    bf16 projections), and decode_from_ids(encode(img).ids) equals decode(encode(img).fmap)."""
    torch.manual_seed(3)
    vae = mm.VQGanVAE(dim=16, codebook_size=1024, lookup_free_quantization=False, vq_codebook_dim=32).to(DEV).eval()
    img = torch.randn(2, 3, 64, 64, device=DEV)
    fmap, ids, aux = vae.encode(img)
    assert ids.shape == (2, 4, 4) and ids.dtype == torch.long and fmap.shape == (2, 128, 4, 4) and float(aux) == 0.
    assert (ids >= 0).all() and (ids < 1024).all() and ids.unique().numel() > 4
    a = vae.decode_from_ids(ids)
    b = vae.decode(fmap)
    assert torch.isfinite(a).all() and (a - b).abs().max() < 2e-2 * b.abs().max()
    assert 'quantizer.codebook' in vae.state_dict()


def test_self_critic_training_gradients(golden):
    """SelfCritic (mmp.py:352-374): BCE of a Linear(dim, 1) head on the generator's embed, differentiable into head and generator."""
    g, l = golden('transformer_tiny.pt'), golden('loss_tiny.pt')
    _, t = _tiny_transformer(golden)
    te = g['text_embeds']
    from muse_maskgit_pytorch_amd.muse_maskgit import SelfCritic
    sc = SelfCritic(t).to(DEV)
    with torch.no_grad():
        sc.to_pred.weight.copy_((torch.randn(1, 128, generator=torch.Generator().manual_seed(2)) * 0.3).to(torch.bfloat16).float())
        sc.to_pred.bias.fill_(0.25)
    x = l['x'].clamp(max=511)
    y = l['critic_labels']
    loss = sc(x.to(DEV), text_embeds=te.to(DEV), labels=y.to(DEV))
    loss.backward()
    sd = {k: (v.float().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in g['sd'].items()}
    w = sc.to_pred.weight.detach().cpu().clone().requires_grad_(True)
    b = sc.to_pred.bias.detach().cpu().clone().requires_grad_(True)
    _, emb = O.transformer_forward(sd, dict(depth=2, heads=8), x, te, 0., return_embed=True)
    ref = torch.nn.functional.binary_cross_entropy_with_logits((emb @ w.t() + b)[..., 0], y)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 2e-2 * abs(ref.item())
    pairs = [('to_pred.weight', sc.to_pred.weight.grad, w.grad), ('to_pred.bias', sc.to_pred.bias.grad, b.grad)]
    for name, p in t.named_parameters():
        if name.startswith('self_cond_to_init_embed') or name in ('norm.gamma', 'to_logits.weight'):
            continue
        pairs.append((name, p.grad, sd[name].grad))
    for name, gg, rg in pairs:
        gg = gg.float().cpu()
        rel = (gg - rg).abs().max().item() / (rg.abs().max().item() + 1e-20)
        cos = torch.nn.functional.cosine_similarity(gg.flatten(), rg.flatten(), dim=0).item()
        assert rel < 5e-2 and cos > 0.99, f'{name}: rel {rel:.3e} cos {cos:.5f}'
    # MaskGit with self_token_critic=True end to end
    t.zero_grad()
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None, self_token_critic=True).to(DEV)
    total = mg(torch.randint(0, 512, (2, 64), generator=torch.Generator().manual_seed(9)).to(DEV), text_embeds=te.to(DEV), cond_drop_prob=0.)
    total.backward()
    assert torch.isfinite(total) and mg.token_critic.to_pred.weight.grad.abs().max() > 0 and t.to_logits.weight.grad.abs().max() > 0


def test_ff_inner_layernorm_fold_matches_the_unfolded_path(golden):
    """mmp.py:86-88: LayerNorm(inner) between the GEGLU and the second Linear is folded into the GEMM pair (w1's epilogue emits per-row
    partial sums, w2 = gains folded into the weights + a mean / rstd / bias epilogue).  Debug bit 1 << 24 runs the unfolded path
    (GEGLU GEMM -> LayerNorm kernel -> GEMM): the two agree to bf16 noise, and both stay within the oracle tolerance."""
    from muse_maskgit_pytorch_amd import _lib
    g, t = _tiny_transformer(golden)
    te = g['text_embeds']
    ids = g['ids'].to(DEV)
    folded = t(ids, text_embeds=te.to(DEV))
    lib = _lib.lib()
    lib.mm_debug_set(1 << 24)
    try:
        unfolded = t(ids, text_embeds=te.to(DEV))
    finally:
        lib.mm_debug_set(0)
    scale = unfolded.abs().max()
    d = (folded - unfolded).abs()
    assert d.max() > 0, 'the debug bit did not change the path'
    assert d.max() < 0.02 * scale and d.mean() < 2e-3 * scale, (d.max().item(), d.mean().item(), scale.item())
    sd = {k: (v.float().cpu() if v.is_floating_point() else v.cpu()) for k, v in t.state_dict().items()}
    ref = O.transformer_forward(sd, dict(depth=g['cfg']['depth'], heads=g['cfg']['heads']), g['ids'], te, 0., rp=O.bf16_round)
    e = _report('folded FF logits vs oracle', folded, ref)
    assert e.max() < 0.03 * ref.abs().max()


def test_layernorm_dim_fold_matches_the_unfolded_path(golden):
    """mmp.py:63-70, 137, 187-195 (round 4): from layer 1 on, the LayerNorm in front of q|k|v, of the cross-attention's q and of FF w1 rides in the GEMMs
    around it -- the residual-adding epilogues write the new rows as bf16 + per-row (sum, sum of squares), the projection multiplies the raw rows by
    gain-folded weights and applies rstd * (acc - mean * c1) + c2.  Debug bit 1 << 29 runs the LayerNorm kernels instead: the two agree to bf16 noise, both
    stay within the oracle tolerance, and the decode loop's folded engine equals the general (folded) forward path + oracle tail (bit-exact ids)."""
    from muse_maskgit_pytorch_amd import _lib
    g, t = _tiny_transformer(golden)
    te = g['text_embeds']
    ids = g['ids'].to(DEV)
    lib = _lib.lib()
    folded = t(ids, text_embeds=te.to(DEV))
    folded_null = t(ids, text_embeds=te.to(DEV), cond_drop_prob=1.)
    lib.mm_debug_set(1 << 29)
    try:
        unfolded = t(ids, text_embeds=te.to(DEV))
        unfolded_null = t(ids, text_embeds=te.to(DEV), cond_drop_prob=1.)
    finally:
        lib.mm_debug_set(0)
    sd = {k: (v.float().cpu() if v.is_floating_point() else v.cpu()) for k, v in t.state_dict().items()}
    cfg = dict(depth=g['cfg']['depth'], heads=g['cfg']['heads'])
    for name, a, b, drop in (('cond', folded, unfolded, 0.), ('null', folded_null, unfolded_null, 1.)):
        scale = b.abs().max()
        d = (a - b).abs()
        assert d.max() > 0, 'the debug bit did not change the path'
        assert d.max() < 0.02 * scale and d.mean() < 2e-3 * scale, (name, d.max().item(), d.mean().item(), scale.item())
        ref = O.transformer_forward(sd, cfg, g['ids'], te, drop, rp=O.bf16_round)
        e_f, e_u = _report(f'LayerNorm(dim) fold, {name} logits vs oracle', a, ref), _report(f'... unfolded, {name}', b, ref)
        assert e_f.max() < 0.03 * ref.abs().max() and e_f.mean() < 1.5 * e_u.mean() + 1e-4


def test_layernorm_dim_fold_probe_falls_back_on_a_dc_offset():
    """the LayerNorm(dim) fold multiplies the bf16 image of the RAW residual row, so its rounding error in normalised units grows like
    |x^ + mean / sigma| -- every closeness test so far used random-init weights, whose rows have mean ~ 0.  Here the residual stream carries a DC offset of
    several (then: many) sigma (a constant added to the position embedding: every row of every layer keeps it; +2 stays below the limit and costs
    nothing measurable, +64 is a caricature of a massive-activation checkpoint).  Forced on, the fold is measurably worse than the LayerNorm
    kernels against the fp64-grade oracle; the default 'auto' mode probes max |mean| / sigma on the first call through a freshly packed model, finds it above
    MM_LN_FOLD_MAX_RATIO, re-creates the handle with the fold off and recomputes: bit-identical to the forced-off engine.  On the same model WITHOUT the
    offset the probe stays below the limit and 'auto' == forced on, bit for bit."""
    from muse_maskgit_pytorch_amd import _lib
    torch.manual_seed(11)
    V, n, depth, B, L = 1000, 64, 3, 2, 9
    t = mm.MaskGitTransformer(num_tokens=V, seq_len=n, dim=256, depth=depth, dim_head=64, heads=4, t5_name='t5-small')
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, V + 1, (B, n), generator=g)
    te = torch.randn(B, L, 512, generator=g)
    cfg = dict(depth=depth, heads=4)
    for offset, expect_off in ((0., False), (2., False), (62., True)):      # cumulative: the position embedding carries +0, +2, +64
        with torch.no_grad():
            t.pos_emb.weight.add_(offset)
        sd = {k: v.detach().float().cpu().clone() for k, v in t.state_dict().items()}
        ref = O.transformer_forward(sd, cfg, ids, te, 0., rp=O.bf16_round)
        t = t.to(DEV)
        outs = {}
        for mode in (True, False, 'auto'):
            t.set_layernorm_fold(mode)
            outs[mode] = t(ids.to(DEV), text_embeds=te.to(DEV)).float().cpu()
            if mode == 'auto':
                ratio = t.layernorm_fold_ratio
                assert t._handle.ln_probe is None and bool(t._handle.desc.ln_fold_off) == expect_off
                again = t(ids.to(DEV), text_embeds=te.to(DEV)).float().cpu()      # the probed handle: no more probing, same result
                assert torch.equal(again, outs['auto'])
                # round 6: an in-place parameter update (an optimizer step) repacks the weights but keeps the verdict -- no second probe, same engine
                with torch.no_grad():
                    t.to_logits.weight.mul_(1.0)
                h2 = t._model()
                assert h2 is not None and h2.ln_probe is None and bool(h2.desc.ln_fold_off) == expect_off and t._ln_fold_auto == (not expect_off)
                assert torch.equal(t(ids.to(DEV), text_embeds=te.to(DEV)).float().cpu(), outs['auto'])
        t.set_layernorm_fold('auto')
        e_on, e_off = (outs[True] - ref).abs(), (outs[False] - ref).abs()
        print(f'[ln-fold probe] position-embedding offset {offset}: max |mean| / sigma over the folded LayerNorm inputs = {ratio:.3f} (limit {_lib.MM_LN_FOLD_MAX_RATIO}); '
              f'logits error vs oracle: fold on max {e_on.max().item():.3g} mean {e_on.mean().item():.3g}, fold off max {e_off.max().item():.3g} mean {e_off.mean().item():.3g} '
              f'(scale {ref.abs().max().item():.3g})')
        assert torch.equal(outs['auto'], outs[False] if expect_off else outs[True])
        assert (ratio > _lib.MM_LN_FOLD_MAX_RATIO) == expect_off
        # measured (round 5, this construction): ratio 0.2 -> fold on / off mean error 1.01x, ratio 1.7 -> 1.03x, ratio 6.5 -> 1.12x, ratio 52 -> 1.13x (there the
        # bf16 embedding tables themselves -- resolution 0.5 at 64 -- dominate both engines' error): the hazard exists and stays modest; the limit is 4
        assert e_on.mean() >= 0.97 * e_off.mean()
        if not expect_off:
            assert e_on.mean() < 1.10 * e_off.mean() + 1e-4, 'below the limit the fold must cost (almost) nothing'
        if offset == 0.:
            assert e_on.max() < 0.03 * ref.abs().max() and ratio < 0.5
        t = t.cpu()


@pytest.mark.parametrize('n,L', [(80, 13), (256, 33), (7, 1), (64, 35), (96, 36), (64, 77), (40, 65)])      # 35 / 79 = the most context tokens the two instantiations take (36 / 80 key slots per head with the null key)
def test_cross_attention_with_the_folded_output_projection_matches_the_two_kernel_path(n, L):
    """mmp.py:139-162 on the headline shape class (dim = inner = 512, 8 heads x 64, <= 35 context tokens): the text context is the same at every
    decode step, so the cross-attention's output projection is folded into its values once per context -- (P_h V_h) W_o,h^T = P_h (V_h W_o,h^T) -- and the
    block behind the q projection is one kernel (csrc/cross_fold.hip: scores + softmax per head, one MFMA contraction over the 288 (head, key) pairs, residual
    add, LayerNorm(dim)-fold producer outputs).  Debug bit 1 << 31 runs the two-kernel path (33-key attention, then the 512 x 512 projection): the two agree
    to bf16 noise, both stay within the oracle tolerance; ragged text rows (key mask), the null pass (every text key masked: only
    the null key is attended) and query counts that are no multiple of the 32-query workgroup.  (The decode loop, whose last layer runs this kernel on compacted
    rows, is held to the reference goldens at full size by tests/test_gpu_base_size.py.)"""
    torch.manual_seed(n * 100 + L)
    V, depth, B = 1000, 2, 3
    OFF = -(1 << 31)                                                   # debug bit 31 as a C int
    t = mm.MaskGitTransformer(num_tokens=V, seq_len=n, dim=512, depth=depth, dim_head=64, heads=8, t5_name='t5-small')
    with torch.no_grad():
        for p in t.parameters():                                       # de-trivialise the gains / scales / null key and value
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    sd = {k: v.detach().float().clone() for k, v in t.state_dict().items()}
    cfg = dict(depth=depth, heads=8)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, V + 1, (B, n), generator=g)
    te = torch.randn(B, L, 512, generator=g)
    if L > 2:
        te[1, L // 2:] = 0.                                            # zero-padded rows -> masked keys
        te[2, L - 1:] = 0.
    t = t.to(DEV)
    lib = _lib.lib()
    outs = {}
    for bit in (0, OFF):
        lib.mm_debug_set(bit)
        try:
            outs[bit] = [t(ids.to(DEV), text_embeds=te.to(DEV), cond_drop_prob=drop).float().cpu() for drop in (0., 1.)]
        finally:
            lib.mm_debug_set(0)
    for i, drop in enumerate((0., 1.)):
        a, b = outs[0][i], outs[OFF][i]
        scale = b.abs().max()
        d = (a - b).abs()
        if drop == 0.:
            assert d.max() > 0, 'the debug bit did not change the path'
        assert d.max() < 0.02 * scale and d.mean() < 2e-3 * scale, (drop, d.max().item(), d.mean().item(), scale.item())
        ref = O.transformer_forward(sd, cfg, ids, te, drop, rp=O.bf16_round)
        e_f, e_u = _report(f'folded cross-attention n={n} L={L} drop={drop} vs oracle', a, ref), _report('... two-kernel path', b, ref)
        assert e_f.max() < 0.03 * ref.abs().max() and e_f.mean() < 1.5 * e_u.mean() + 1e-4


@pytest.mark.parametrize('n,L,fp32w', [(80, 13, False), (256, 33, True), (7, 1, True), (64, 35, False), (96, 35, True)])
def test_tier_cross_attention_behind_the_q_projection_as_one_kernel(n, L, fp32w):
    """'f16x2' tier, round 6 (csrc/cross_vw_x2.hip): on the headline shape class (dim = inner = 512, 8 heads x 64, <= 35 context tokens) the tier's cross-attention
    behind its q projection is ONE kernel -- scores on the fp32 MFMA against K^ normalised once per context, softmax, P . (V W_o^T) as fp16 term products with the
    output projection folded into the step-invariant values, residual add -- instead of attention_f32 + a term GEMM; and the block's LayerNorm + q projection (term
    products against the q weight's fragment pack) run in the same kernel in front of that.  mm_debug_set2(128) keeps LayerNorm-split + q GEMM as launches, (64) the
    four-launch form: all agree to ~1e-6 of the logits' scale (another association of fp32-grade arithmetic), all hold the tier's 1e-3 against the fp32 oracle;
    bf16-representable and general fp32 weights (two / three weight terms), ragged text rows (key mask), the null pass, query counts that are no multiple of 32."""
    torch.manual_seed(n * 100 + L)
    V, depth, B = 1000, 2, 3
    t = mm.MaskGitTransformer(num_tokens=V, seq_len=n, dim=512, depth=depth, dim_head=64, heads=8, t5_name='t5-small')
    with torch.no_grad():
        for p in t.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
            elif not fp32w:
                p.copy_(p.to(torch.bfloat16).float())                  # a bf16-representable checkpoint: single fp16 weight terms
    sd = {k: v.detach().float().clone() for k, v in t.state_dict().items()}
    cfg = dict(depth=depth, heads=8)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, V + 1, (B, n), generator=g)
    te = torch.randn(B, L, 512, generator=g)
    if L > 2:
        te[1, L // 2:] = 0.
        te[2, L - 1:] = 0.
    t = t.to(DEV)
    t.set_precision('f16x2')
    lib = _lib.lib()
    outs = {}
    for bit in (0, 128, 64):      # 0: LayerNorm + q projection inside the kernel too; 128: those as launches in front of it; 64: the four-launch form of rounds 4-5
        lib.mm_debug_set2(bit)
        try:
            outs[bit] = [t(ids.to(DEV), text_embeds=te.to(DEV), cond_drop_prob=drop).float().cpu() for drop in (0., 1.)]
        finally:
            lib.mm_debug_set2(0)
    for i, drop in enumerate((0., 1.)):
        a, a2, b = outs[0][i], outs[128][i], outs[64][i]
        ref = O.transformer_forward(sd, cfg, ids, te, drop)
        scale = ref.abs().max()
        d, d2 = (a - b).abs(), (a2 - b).abs()
        if drop == 0.:
            assert d.max() > 0 and d2.max() > 0, 'debug bit 64 did not change the path'
        # (the in-kernel LayerNorm + q projection repeat the launches' arithmetic in their order -- split.hip's two-pass LayerNorm, per 32-deep k-block hh, lh, hl -- so 0 and 128
        #  usually agree bit for bit; tools/A-B runs of bench.py --precision f16x2 under MM_DEBUG2=128 show the path is taken: 99.4 vs 101.8 ms per step)
        assert d.max() < 2e-5 * scale and d2.max() < 2e-5 * scale and (a - a2).abs().max() < 2e-5 * scale, (drop, d.max().item(), d2.max().item(), scale.item())
        e_new, e_mid, e_old = (a - ref).abs().max() / scale, (a2 - ref).abs().max() / scale, (b - ref).abs().max() / scale
        print(f'tier cross-attention one kernel n={n} L={L} fp32w={fp32w} drop={drop}: |new - old| {d.max().item() / scale.item():.2e} / {d2.max().item() / scale.item():.2e}, '
              f'vs oracle: with q projection inside {e_new.item():.2e}, behind it {e_mid.item():.2e}, four launches {e_old.item():.2e}')
        assert e_new < 1e-3 and e_mid < 1e-3 and e_old < 1e-3


def test_full_size_c2_properties():
    """BASELINE configs[1] at FULL size (dim 512, depth 8, seq_len 256, codebook 65536; B = 8 to keep the fp32 oracle out of it):
    size-independent properties instead of an oracle comparison --
      * guidance linearity: forward_with_cond_scale(s) = null + (cond - null) * s (to bf16 operand rounding); s = 1 is the conditional pass, s = 0 the null pass;
      * batch invariance: a sample's logits do not depend on its batch mates;
      * the decode loop: every step masks exactly the scheduled count per sample, final ids are < codebook size and carry no mask id,
        the same seed reproduces the ids, and a sharded run (row_offset) reproduces the unsharded one;
      * the persistent logits kernel equals the non-persistent one bit for bit at this size."""
    import bench
    from muse_maskgit_pytorch_amd import _lib
    mg, _ = bench.build_models(DEV)
    tr = mg.transformer
    B, n, V = 8, 256, 65536
    te = bench.synth_text(B, 32, 512).to(DEV)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, V, (B, n), generator=g)
    ids[torch.rand(B, n, generator=g) < 0.6] = tr.mask_id
    ids = ids.to(DEV)
    cond = tr(ids, text_embeds=te, cond_drop_prob=0.)
    null = tr(ids, text_embeds=te, cond_drop_prob=1.)
    s3 = tr.forward_with_cond_scale(ids, text_embeds=te, cond_scale=3.)
    ref3 = null + (cond - null) * 3.
    # (round 3: the combine is applied to the two embeddings and to_logits runs once -- the same quantity up to the bf16 rounding of the mixed operand)
    assert (s3 - ref3).abs().max() <= 2e-2 * ref3.abs().max()
    assert torch.equal(tr.forward_with_cond_scale(ids, text_embeds=te, cond_scale=1.), cond)
    s0 = tr.forward_with_cond_scale(ids, text_embeds=te, cond_scale=0.)
    assert (s0 - null).abs().max() <= 1e-5 * null.abs().max() + 1e-6
    sub = tr(ids[2:5], text_embeds=te[2:5], cond_drop_prob=0.)
    assert torch.equal(sub, cond[2:5])
    lib = _lib.lib()
    lib.mm_debug_set(4096)
    try:
        s3_np = tr.forward_with_cond_scale(ids, text_embeds=te, cond_scale=3.)
    finally:
        lib.mm_debug_set(0)
    assert torch.equal(s3, s3_np)
    trace = {}
    out = mg.generate([''] * B, text_embeds=te, timesteps=18, seed=11, return_ids=True, trace=trace)
    assert out.shape == (B, 16, 16) and (out >= 0).all() and (out < V).all()
    counts = mg._mask_counts(18, n)
    assert counts == [256, 254, 251, 246, 238, 229, 217, 204, 189, 172, 154, 134, 114, 92, 70, 47, 23, 1]
    for step in range(18):
        assert ((trace['masked_ids'][step] == tr.mask_id).sum(-1) == counts[step]).all()
        assert (trace['ids'][step] != tr.mask_id).all()
    assert torch.equal(out, mg.generate([''] * B, text_embeds=te, timesteps=18, seed=11, return_ids=True))
    lo = mg.generate([''] * 4, text_embeds=te[:4], timesteps=18, seed=11, return_ids=True, row_offset=0)
    hi = mg.generate([''] * 4, text_embeds=te[4:], timesteps=18, seed=11, return_ids=True, row_offset=4)
    assert torch.equal(torch.cat([lo, hi]), out)
    img = mg.vae.decode_from_ids(out)
    assert img.shape == (B, 3, 256, 256) and torch.isfinite(img).all()


@pytest.mark.parametrize('precision', ['bf16', 'parity'])
def test_muse_cascade_base_to_superres(precision):
    """Muse.forward (mmp.py:758-791): base.generate -> images -> superres.generate(cond_images = those images).  The cascade must equal
    its two stages run by hand with the same seeds, and the super-resolution stage's ids are checked against the oracle tail fed with
    the HIP transformer's logits (condition ids from the low-res VAE in the cross-attention context, visible in the null pass too)."""
    torch.manual_seed(0)
    vae_lo = mm.VQGanVAE(dim=16, codebook_size=512)
    vae_hi = mm.VQGanVAE(dim=16, codebook_size=512)
    base_tr = mm.MaskGitTransformer(num_tokens=512, seq_len=16, dim=128, depth=2, dim_head=64, heads=8, t5_name='t5-small')
    sr_tr = mm.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=2, dim_head=64, heads=8, t5_name='t5-small')
    with torch.no_grad():
        base_tr.to_logits.weight.mul_(8.); sr_tr.to_logits.weight.mul_(8.)
    base = mm.MaskGit(vae=vae_lo, transformer=base_tr, image_size=64).to(DEV)
    sr = mm.MaskGit(vae=vae_hi, cond_vae=vae_lo, transformer=sr_tr, image_size=128, cond_image_size=64).to(DEV)
    base.set_precision(precision); sr.set_precision(precision)
    te = torch.randn(2, 6, 512)
    te[0, 4:] = 0
    for tr_ in (base_tr, sr_tr):
        tr_.encode_text = lambda texts, te=te: te          # per-instance attribute, as reference users override it (mmp.py:229)
    muse = mm.Muse(base=base, superres=sr)
    torch.manual_seed(11)
    hi, lo = muse(['a', 'b'], timesteps=5, superres_timesteps=4, return_lowres=True, return_pil_images=False)
    assert lo.shape == (2, 3, 64, 64) and hi.shape == (2, 3, 128, 128) and torch.isfinite(hi).all()
    # the same two stages by hand (generate draws its Philox seed from torch's generator when none is given)
    torch.manual_seed(11)
    lo2 = base.generate(['a', 'b'], timesteps=5, cond_scale=3.)
    hi2 = sr.generate(['a', 'b'], timesteps=4, cond_scale=3., cond_images=lo2)
    assert torch.equal(lo, lo2) and torch.equal(hi, hi2)
    # super-res ids against the oracle tail on the HIP logits, injected noise
    _, cids, _ = sr.cond_vae.encode(lo)
    assert cids.shape == (2, 4, 4)
    T, B, n, V = 4, 2, 64, 512
    uni = torch.rand(T, B, n, V, generator=torch.Generator().manual_seed(3))
    got = sr.generate(['a', 'b'], timesteps=T, cond_images=lo, noise=uni, noise_kind='uniform', return_ids=True).reshape(B, n).cpu()

    def demask(ids, step):
        return sr_tr.forward_with_cond_scale(ids.to(DEV), text_embeds=te.to(DEV), conditioning_token_ids=cids, cond_scale=3.).cpu()

    ref = O.generate_ids(demask, B, n, 512, lambda s, shp: O.gumbel_from_uniform(uni[s]), timesteps=T)
    assert torch.equal(got, ref), 'super-resolution decode loop differs from the oracle tail on the same logits'
