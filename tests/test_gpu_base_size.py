"""BASELINE configs[1] AT FULL SIZE against the reference: dim 512, depth 8, 65536-entry codebook, 256 tokens, VQGanVAE(dim=256), B = 2,
18 decode steps.  tests/golden/base_c2.pt was recorded from the UNMODIFIED reference (oracle/make_golden_base.py); checkpoint and noise are
rebuilt from the seeded recipe (oracle/golden_recipe.py) and their checksums asserted first.

Engines compared with the same fixtures (bf16-representable AND general fp32 checkpoints):
  precision 'f16x2'  -- THE tolerance-meeting tier (rounds 4-5; bench.py's `parity_tier`) INSIDE mm_generate / mm_transformer_forward: every GEMM operand as two
                        fp16 terms, products xh wh + xl wh (+ xh wl) on the fp16 matrix pipe with term sharing (csrc/gemm_terms.hip; the 'f16x2-terms' entries
                        force those kernels onto the fixtures' B = 2), self-attention as fp16 term products: the north star's bar -- logits within 1e-3
                        absolute, every step's ids 100 % -- through the C loop.
  precision 'bf16x3' -- round 3's tier (three bf16 terms, 3 / 6 products): same bar, kept as a second independent implementation of it.
  precision 'parity' -- fp32 storage + fp32 MFMA (csrc/parity.hip), operator by operator from Python: same bar, the tiers' verification baseline.
  precision 'bf16'   -- the production engine (the bench line's `value`): logit bounds are what bf16 operands achieve at this size (stated per assert); ids are
                        compared through the oracle tail on the engine's own logits (bit-exact) and -- round 6 -- held to a TIE-AWARE CONTRACT against the
                        reference's run on all five full-size fixtures: every id / re-masked position that differs must lie inside a tie band of the REFERENCE's
                        own margins whose width is the engine's stated logit bound (test_bf16_engine_differs_from_the_reference_only_inside_tie_bands_*).
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import golden_recipe as R
import muse_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'
FULL_ROWS = [0, 77, 255, 256, 300, 301, 448, 511]
class _TermSharing(str):
    """'f16x2' with mm_debug_set2(1): the tier's q|k|v and feed-forward on csrc/gemm_terms.hip (round 5) whatever the tile count -- at B = 32 the bench's rows
    take that kernel by themselves, the fixtures' B = 2 would not (the plain 'f16x2' entry keeps covering the concatenated-depth kernels small batches run on)"""


PRECISIONS = ['parity', 'bf16x3', 'f16x2', pytest.param(_TermSharing('f16x2'), id='f16x2-terms'), 'bf16']
EXACT = ('parity', 'bf16x3', 'f16x2')      # the engines held to the north star's bar


@pytest.fixture(autouse=True)
def _term_sharing_kernel(request):
    from muse_maskgit_pytorch_amd import _lib as L
    cs = getattr(request.node, 'callspec', None)
    forced = cs is not None and isinstance(cs.params.get('precision'), _TermSharing)
    if forced:
        L.lib().mm_debug_set2(1)
    yield
    if forced:
        L.lib().mm_debug_set2(0)
# the bf16 engine on a GENERAL fp32 checkpoint (weights rounded to bf16 at pack time on top of the bf16 activations).  Measured in round 5 (printed by the
# tests): logits 1.95e-2 max / 2.9e-3 mean (1.33e-2 / 2.2e-3 on the bf16-representable checkpoint), guidance-combined 6.5e-2, embed 3.7e-2,
# final ids 95.9 % / worst step 94.3 % equal to the reference run -- the weight rounding costs about half again of the activation rounding's error
BF16_FP32W_LOGITS, BF16_FP32W_EMBED, BF16_FP32W_IDS = 3.5e-2, 6e-2, 0.90
# Round 6: the bf16 engine's stated bound G on the GUIDANCE-COMBINED logits null + 3 (cond - null) (what the sampler consumes), per fixture, at the fixtures' unit logit
# scale: measured 0.045 (base, bf16-representable weights) / 0.065 (base, general fp32 weights) / 0.040 (super-res) / 0.070 (paper scale), stated 1.3 x that
# (rounds 1-5 asserted 5 x the per-pass bound = 0.125 / 0.175: no contract at all).  The SAME number is the width of the tie band the engine's ids are held to:
# a decision of the reference run may differ only where the reference's own margin is below PEAK x G logit units (_bf16_tie_contract).
BF16_GUIDANCE = dict(base_bf16w=0.06, base_fp32w=0.085, c4=0.055, c5=0.09)


@pytest.fixture(scope='module', params=['base_c2.pt', 'base_c2_fp32.pt'], ids=['bf16w', 'fp32w'])
def base(golden, request):
    """base_c2.pt: the checkpoint rounded to bf16-representable values before the reference ran (exactly loadable by the bf16 engine; the tier
    needs 3 term products).  base_c2_fp32.pt (round 4): the SAME recipe without that rounding -- general fp32 parameters, what every checkpoint
    the reference initialises / trains holds (mmp.py:85,88,118-124,233); the tier needs all six term products and an engine that packed
    its weights through bf16 would miss the bound by an order of magnitude (tests/test_oracle_vs_golden.py shows the fixture discriminates)."""
    import muse_maskgit_pytorch_amd as mm
    g = golden(request.param)
    bf16w = g['recipe'].get('bf16_weights', True)
    tr = R.build_transformer(mm.MaskGitTransformer, peaky=False, bf16_weights=bf16w)
    assert R.state_checksum(tr) == g['weight_checksum'], 'the seeded recipe did not reproduce the reference checkpoint'
    vae = R.build_vae(mm.VQGanVAE, bf16_weights=bf16w)
    mg = mm.MaskGit(vae=vae, transformer=tr, image_size=256).to(DEV).eval()
    assert R.state_checksum(mg.vae) == g['vae_weight_checksum']
    inp = R.inputs(g['recipe'].get('input_seed'))
    assert {k: R.checksum(v.float()) for k, v in inp.items()} == g['input_checksum']
    yield g, mg, inp
    del mg, tr, vae
    torch.cuda.empty_cache()


def _fp32w(g):
    """the fixture's checkpoint holds general fp32 parameters (what the reference's constructors / training produce): the bf16 engine then ALSO rounds
    every weight to bf16 at pack time -- the case a drop-in user hits with `maskgit.load('maskgit.pt')`.  Its bounds on that fixture are stated
    separately below (round 5; rounds 1-4 skipped the combination)."""
    return not g['recipe'].get('bf16_weights', True)


@pytest.fixture(scope='module')
def noise(golden):
    g = golden('base_c2.pt')
    assert g['generate']['noise_checksum'] == golden('base_c2_fp32.pt')['generate']['noise_checksum']      # one noise recipe for both fixtures
    us = []
    for s, u in enumerate(R.noise_stream()):
        assert R.checksum(u) == g['generate']['noise_checksum'][s], f'noise recipe does not reproduce step {s}'
        us.append(u)
    return torch.stack(us)          # [T, B, n, V] uniforms, 2.4 GB


def _err(name, got, ref, abs_tol):
    d = (got.float().cpu() - ref).abs()
    print(f'[base-size parity] {name}: max abs err {d.max().item():.3g}, mean {d.mean().item():.3g} (reference absmax {ref.abs().max().item():.3g}); bound {abs_tol:g}')
    assert d.max().item() <= abs_tol, f'{name}: {d.max().item()} > {abs_tol}'


def _samples(logits):
    f = logits.reshape(R.B * R.N, -1)
    return f[FULL_ROWS], f[:, ::128]


@pytest.mark.parametrize('precision', PRECISIONS)
def test_transformer_forward_logits_at_base_size(base, precision):
    """Transformer.forward (mmp.py:279-335) and forward_with_cond_scale (:240-259): logits of 8 full rows + every 128th vocabulary column of
    all 512 rows, and the final-LayerNorm embed, against the reference's fp32 run.  Random-init logits are unit scale (std 0.59, |max| 3.2),
    so the absolute bound IS the north star's 1e-3 for the parity engine; the bf16 engine's bound is what 8 layers of bf16 operands give."""
    g, mg, inp = base
    tr = mg.transformer.set_precision(precision)
    if precision in ('bf16x3', 'f16x2'):      # term products per GEMM: bf16 terms 3 / 6, fp16 terms 2 / 3 (bf16-representable / general fp32 checkpoint)
        assert tr.split_products() == dict(bf16x3=(6, 3), f16x2=(3, 2))[precision][int(g['recipe'].get('bf16_weights', True))]
    try:
        ids, te = inp['ids'].to(DEV), inp['text_embeds'].to(DEV)
        lc, emb = tr(ids, text_embeds=te, cond_drop_prob=0., return_embed=True)
        ln = tr(ids, text_embeds=te, cond_drop_prob=1.)
        sc = tr.forward_with_cond_scale(ids, text_embeds=te, cond_scale=3.)
    finally:
        tr.set_precision('bf16')
    fw = g['forward']
    # bf16 engine: 2.5e-2 on the bf16-representable checkpoint (activation rounding only), BF16_FP32W_LOGITS on the general fp32 one (+ weight rounding)
    tol = 1e-3 if precision in EXACT else (BF16_FP32W_LOGITS if _fp32w(g) else 2.5e-2)
    for name, got, rec in (('logits(cond)', lc, fw['logits_cond']), ('logits(null)', ln, fw['logits_null'])):
        rows, cols = _samples(got)
        _err(f'{precision} {name} full rows', rows, rec['rows'], tol)
        _err(f'{precision} {name} strided columns', cols, rec['cols'], tol)
    rows, cols = _samples(sc)
    gtol = tol * 5 if precision in EXACT else BF16_GUIDANCE['base_fp32w' if _fp32w(g) else 'base_bf16w']      # null + 3 (cond - null): errors add up to 5x
    _err(f'{precision} logits(guidance 3.0) full rows', rows, fw['logits_scaled']['rows'], gtol)
    _err(f'{precision} logits(guidance 3.0) strided columns', cols, fw['logits_scaled']['cols'], gtol)
    _err(f'{precision} embed', emb.reshape(R.B, R.N, -1), fw['embed'], 1e-3 if precision in EXACT else (BF16_FP32W_EMBED if _fp32w(g) else 4e-2))


@pytest.mark.parametrize('precision', PRECISIONS)
def test_generate_at_base_size_against_the_reference_run(base, noise, precision):
    """MaskGit.generate (mmp.py:491-621), 18 steps, the reference's own noise, peaky logits.  parity: every step's re-masked ids and the final
    ids equal the reference's.  bf16: the fused engine (logits only at masked rows, constant null cross-attention, compacted last layer) is
    checked bit-exactly at V = 65536 through the oracle tail fed with the GENERAL path's logits on the engine's own states (three steps),
    and its agreement with the fp32 reference trajectory is reported (bf16 operands may flip a near-tie; the run then follows another path)."""
    g, mg, inp = base
    gen = g['generate']
    tr = mg.transformer
    te = inp['text_embeds'].to(DEV)
    with torch.no_grad():
        tr.to_logits.weight.mul_(R.PEAK)
    mg.set_precision(precision)
    try:
        assert R.state_checksum(tr) == g['weight_checksum_peaky']
        trace = {}
        u = noise.to(DEV)
        ids = mg.generate(['a', 'b'], timesteps=R.T, cond_scale=3., text_embeds=te, noise=u, noise_kind='uniform', return_ids=True, trace=trace)
        masked = torch.stack(list(trace['masked_ids'])).cpu() if isinstance(trace['masked_ids'], list) else trace['masked_ids'].cpu()
        ref_in = gen['step_in_ids'].long()
        agree_steps = [(masked[s] == ref_in[s]).float().mean().item() for s in range(R.T)]
        final_agree = (ids.cpu().reshape(gen['final_ids'].shape) == gen['final_ids']).float().mean().item()
        print(f'[base-size parity] {precision} generate: final ids equal to the reference run: {100 * final_agree:.2f} %; per-step state agreement min '
              f'{100 * min(agree_steps):.2f} % (step {agree_steps.index(min(agree_steps))})')
        if precision in ('bf16x3', 'f16x2'):      # the tiers run inside the one mm_generate call (the stepwise loop returns lists)
            assert isinstance(trace['masked_ids'], torch.Tensor)
            assert tr.split_products() == dict(bf16x3=(6, 3), f16x2=(3, 2))[precision][int(g['recipe'].get('bf16_weights', True))]
        if precision in EXACT:
            assert min(agree_steps) == 1.0 and final_agree == 1.0
        else:
            assert final_agree >= (BF16_FP32W_IDS if _fp32w(g) else 0.90)
            # the fused engine against the general path + oracle tail at full vocabulary, on its own states
            counts, temps = O.mask_counts(R.T, R.N), O.step_temperatures(R.T, 1.)
            st_ids = trace['ids'].cpu()
            for s in (0, 7, R.T - 1):
                ids_in = masked[s]
                logits = tr.forward_with_cond_scale(ids_in.to(DEV), text_embeds=te, cond_scale=3.).cpu()
                new_ids, scores, _ = O.sample_step(logits, O.gumbel_from_uniform(noise[s]), ids_in, 65536, temps[s])
                assert torch.equal(new_ids, st_ids[s]), f'bf16 fused engine vs general path + oracle tail: ids after step {s} differ in {(new_ids != st_ids[s]).sum().item()} places'
            # sampling without the logits round trip (default) against the logits path: same ids, no fallback needed
            assert mg.fused_sampling_fallbacks == 0
            ids2 = mg.generate(['a', 'b'], timesteps=R.T, cond_scale=3., text_embeds=te, noise=u, noise_kind='uniform', return_ids=True, fused_sampling=False)
            assert torch.equal(ids, ids2), f'fused sampling vs logits path: {(ids != ids2).sum().item()} ids differ'
        del u
    finally:
        mg.set_precision('bf16')
        with torch.no_grad():
            tr.to_logits.weight.div_(R.PEAK)
        torch.cuda.empty_cache()


@pytest.mark.parametrize('precision', ['parity', 'bf16x3', 'f16x2', pytest.param('f16x2', id='f16x2-terms-decode'), 'bf16', pytest.param('bf16', id='bf16-bf16-decode')])
def test_vqgan_vae_dim_256_against_the_reference(base, precision, request):
    """VQGanVAE(dim=256) decode_from_ids / encode (vqgan_vae.py:422-441), the VAE the bench decodes with.  Pixels: 1e-3 of the image scale
    (|max| 0.063 at random init) for the parity engine.  LFQ ids: a bit is the SIGN of a projection, so ids are compared where the reference's
    own pre-sign value clears the engine's error band (all 16 bits of the position), and the fraction of positions covered is reported.
    Round 6: the fast engines ('bf16' -- the bench line -- and 'f16x2') decode on fp16 STORAGE with single fp16 terms (VQGanVAE.decode_storage = 'f16': fp16 MFMA
    at the bf16 rate, 11 significand bits): the decoded pixels of BOTH now meet the north star's 1e-3 (measured ~2e-4 of the scale; rounds 1-5: the bf16 engine
    1.6e-3 with a bound of 8e-3, the tier 6e-8 at three times the time).  The earlier forms stay selectable and tested: 'bf16' storage, and the tier's three-product
    term split ('terms')."""
    g, mg, inp = base
    v = g['vae']
    vae = mg.vae.set_precision(precision)
    cid = request.node.callspec.id
    storage = 'terms' if 'terms-decode' in cid else ('bf16' if 'bf16-decode' in cid else 'f16')
    vae.set_decode_storage(storage)
    if precision in ('bf16x3', 'f16x2'):
        assert vae.x3_products() == dict(bf16x3=(6, 3), f16x2=(3, 2))[precision][int(g['recipe'].get('bf16_weights', True))]
    try:
        assert vae._half_decode() == (precision in ('bf16', 'f16x2') and storage == 'f16')
        dec = vae.decode_from_ids(inp['vae_ids'].to(DEV))
        fmap, ids, _ = vae.encode(inp['image'].to(DEV))
    finally:
        vae.set_precision('bf16')
        vae.set_decode_storage('f16')
    scale = v['decoded_absmax']
    exact_decode = precision in EXACT or storage == 'f16'      # (fp16 storage: single fp16 terms meet the pixel bar)
    tol = (1e-3 if exact_decode else 8e-3) * scale      # 'bf16x3': the convolutions as exact bf16 term products on the bf16 MFMA (decode); encode = the fp32 engine
    _err(f'{cid} decoded pixels (strided)', dec[:, :, ::4, ::4], v['decoded_strided'], tol)
    _err(f'{cid} decoded pixels (64x64 crop)', dec[:, :, 96:160, 96:160], v['decoded_crop'], tol)
    pre = v['enc_pre_sign']                                   # (B, 256, 16): the reference's values whose signs are the id bits
    band = (2e-5 if precision in EXACT else 2e-2) * pre.abs().max().item()
    safe = (pre.abs() > band).all(dim=-1)                     # positions whose 16 bits are all outside the band
    same = ids.cpu().reshape(R.B, -1) == v['enc_ids'].reshape(R.B, -1)
    print(f'[base-size parity] {cid} LFQ encode: {100 * same.float().mean().item():.2f} % of ids equal the reference; '
          f'{100 * safe.float().mean().item():.1f} % of positions have every pre-sign value outside +-{band:.3g}')
    assert bool(same[safe].all()), f'{(~same[safe]).sum().item()} ids differ at positions whose bits are all outside the error band'
    if precision in EXACT:
        assert safe.float().mean().item() > 0.95 and same.float().mean().item() > 0.99
    d = (fmap[:, ::16].float().cpu() - v['enc_fmap_strided']).abs().amax(dim=1)          # (B, 16, 16): project_out(+-1 codes) of equal ids is the same sum
    assert d[same.reshape(R.B, 16, 16)].max().item() <= (1e-5 if precision in EXACT else 2e-2)


@pytest.mark.parametrize('storage', ['f16', 'bf16'])
def test_vae_head_fused_into_the_last_upsampling_convolution(base, storage):
    """Round 6 (vqgan_vae.py:230-232, 246-249): at dim = 256 the decoder's last ConvTranspose2d has 256 output channels = one 256 x 256 tile of gemm_wide_conv.hip,
    so the 1 x 1 head rides in its epilogue and the (B, 256, 256, 256) activation (1 GiB at B = 32) is never written.  Same rounding points as the separate sequence
    (the activation is rounded to its 16-bit storage type before the head multiplies it, the head's weights are its own 16-bit pack); what differs is the fp32
    summation order of the 256-term head product: the images agree to ~1e-7 of the scale.  mm_debug_set2(16) keeps the two convolutions apart."""
    from muse_maskgit_pytorch_amd import _lib as L
    g, mg, inp = base
    vae = mg.vae.set_decode_storage(storage)
    ids = torch.randint(0, 65536, (4, 16, 16), generator=torch.Generator().manual_seed(5)).to(DEV)
    try:
        fused = vae.decode_from_ids(ids)
        L.lib().mm_debug_set2(16)
        try:
            apart = vae.decode_from_ids(ids)
        finally:
            L.lib().mm_debug_set2(0)
    finally:
        vae.set_decode_storage('f16')
    scale = apart.abs().max().item()
    d = (fused - apart).abs().max().item()
    print(f'[vae head fusion] {storage}: fused vs separate head, max abs diff {d:.3g} on image scale {scale:.3g}')
    assert scale > 1e-3 and d <= 2e-6 * scale and not torch.equal(fused, torch.zeros_like(fused))


# ------------------------------------------------------------------------------------------------ the un-scanned fp32 fixture, tie-aware
TIE_EPS = 5e-4      # logit units (tests/tie_aware.py): 12x the fp32-grade engines' largest logit error at the fixtures' x8 logit scale


@pytest.fixture(scope='module')
def base_s77(golden):
    """tests/golden/base_c2_fp32_s77.pt (round 5): the general-fp32 checkpoint on the ORIGINAL input seed 77 -- NOT chosen by tools/find_golden_input_seed.py.
    The reference's own run has two confidences 2.3e-6 apart at the re-masking boundary entering step 8; no implementation that is not bit-identical to
    the reference's BLAS can be expected to order them, so the run is held to "equal outside eps-ties" instead of plain equality."""
    import muse_maskgit_pytorch_amd as mm
    g = golden('base_c2_fp32_s77.pt')
    assert g['recipe']['input_seed'] == R.INPUT_SEED and not g['recipe']['bf16_weights']
    tr = R.build_transformer(mm.MaskGitTransformer, peaky=True, bf16_weights=False)
    assert R.state_checksum(tr) == g['weight_checksum_peaky'], 'the seeded recipe did not reproduce the reference checkpoint'
    mg = mm.MaskGit(vae=None, transformer=tr, image_size=256).to(DEV).eval()
    inp = R.inputs(R.INPUT_SEED)
    assert {k: R.checksum(v.float()) for k, v in inp.items()} == g['input_checksum']
    yield g, mg, inp
    del mg, tr
    torch.cuda.empty_cache()


@pytest.mark.parametrize('precision', ['parity', 'bf16x3', 'f16x2', pytest.param(_TermSharing('f16x2'), id='f16x2-terms')])
def test_generate_on_the_unscanned_fp32_fixture_tie_aware(base_s77, noise, precision):
    """free run: every sample equals the reference's trajectory step by step, or its FIRST difference lies inside a tie band of the reference's own
    scores (then the trajectories legitimately part).  Teacher-forced on the reference's states (every step, so also what lies behind the tie):
    sampled ids equal, re-masked sets equal outside the band."""
    import tie_aware as TA
    g, mg, inp = base_s77
    gen = g['generate']
    assert gen['noise_checksum'] == [R.checksum(u) for u in noise]
    tr = mg.transformer
    te = inp['text_embeds'].to(DEV)
    counts, temps = O.mask_counts(R.T, R.N), O.step_temperatures(R.T, 1.)
    mg.set_precision(precision)
    try:
        trace = {}
        ids = mg.generate(['a', 'b'], fmap_size=16, timesteps=R.T, cond_scale=3., text_embeds=te, noise=noise.to(DEV), noise_kind='uniform', return_ids=True, trace=trace)
        masked = torch.stack(list(trace['masked_ids'])).cpu() if isinstance(trace['masked_ids'], list) else trace['masked_ids'].cpu()
        rep = TA.compare_free_run(gen, masked, ids.cpu(), counts, 65536, TIE_EPS)
        print(f'[base-size parity] {precision} un-scanned fp32 fixture, free run vs the reference (tie-aware, eps {TIE_EPS:g}): '
              + '; '.join(f"sample {b}: {r['status']}" + (f" at step {r['step']} ({r['kind']}, positions {r['positions']})" if r['status'] == 'tie' else '') for b, r in enumerate(rep)))
        assert rep[1]['status'] == 'equal'                                    # sample 1 has no tie anywhere: plain equality
        assert rep[0]['status'] == 'equal' or rep[0]['step'] in (7, 8)       # sample 0: the known pairs of steps 7 / 8
        skipped = [0, 0]
        steps = range(R.T) if precision != 'parity' else (0, 7, 8, R.T - 1)   # (the fp32-MFMA engine is the slow yardstick)
        for s in steps:
            ids_in = gen['step_in_ids'][s].long()
            logits = tr.forward_with_cond_scale(ids_in.to(DEV), text_embeds=te, cond_scale=3.).cpu()
            new_ids, scores, _ = O.sample_step(logits, O.gumbel_from_uniform(noise[s]), ids_in, 65536, temps[s])
            near, band = TA.compare_forced_step(gen, s, new_ids, scores, counts, 65536, TIE_EPS, O.select_topk_stable)
            skipped[0] += near
            skipped[1] += band
        print(f'[base-size parity] {precision} un-scanned fp32 fixture, teacher-forced over {len(list(steps))} steps: all sampled ids and re-masked sets equal the '
              f"reference's outside {skipped[0]} sampling near-ties and {skipped[1]} boundary-band positions")
    finally:
        mg.set_precision('bf16')
        torch.cuda.empty_cache()



# ------------------------------------------------------------------------------------------------ the bf16 engine's id contract (round 6, VERDICT r5 missing 2)
def _bf16_tie_contract(name, gen, mg, te, noise, V, E, fwd_kw=None, gen_kw=None, texts=('a', 'b'), free=True):
    """The headline engine cannot be bit-equal to an fp32 run: its logits are off by up to its stated bound.  What CAN be demanded of it -- and falsified -- is
    that it never takes a decision the reference's own numbers call clear: with eps* = the smallest eps (logit units) at which tests/tie_aware.py explains
    every difference by a tie band of the REFERENCE's recorded margins,
      * teacher-forced on the reference's states with the reference's noise, every one of the T steps (so also what lies behind a divergence): every
        sampled id equals the reference's outside its arg-max near-ties, every re-masked set outside the boundary band;
      * free-running through mm_generate: each sample equals the reference's trajectory or first parts from it inside a band;
    eps* must not exceed E = PEAK x G, G = the engine's asserted bound on the guidance-combined logits (BF16_GUIDANCE; the forward tests hold the engine to the same
    number).  Printed beside it: the share of the reference's decisions a band of that width covers, i.e. how much the contract excuses.
    Measured (round 6, gpurun_out/r6b): eps* = 0.21 - 0.30 logit units on logits of std 4.7 (x 8 peaky): a band that covers 7 - 20 % of the reference's decisions;
    the rounds 1-5 floor "ids >= 0.90 equal" said nothing about WHERE the other 4 - 9 % may lie."""
    import tie_aware as TA
    tr = mg.transformer
    T, B, n = gen['step_in_ids'].shape
    counts, temps = O.mask_counts(T, n), O.step_temperatures(T, 1.)
    forced, agree = [], []
    for s in range(T):
        ids_in = gen['step_in_ids'][s].long()
        logits = tr.forward_with_cond_scale(ids_in.to(DEV), text_embeds=te, cond_scale=3., **(fwd_kw or {})).cpu()
        new_ids, scores, _ = O.sample_step(logits, O.gumbel_from_uniform(noise[s].cpu()), ids_in, V, temps[s])
        del logits
        masked = ids_in == V
        agree.append(((new_ids == gen['pred_ids'][s].long()) | ~masked).float().mean().item())
        forced.append((new_ids, scores))

    def check_forced(eps):
        for s, (ni, sc) in enumerate(forced):
            TA.compare_forced_step(gen, s, ni, sc, counts, V, eps, O.select_topk_stable)

    eps_forced = TA.min_eps(check_forced)
    if not free:      # (super-resolution when the engine's own condition ids differ from the reference's: a free run then starts from another condition)
        assert eps_forced is not None and eps_forced <= E, (name, eps_forced, E)
        near, band = TA.band_population(gen, counts, V, max(eps_forced, 1e-9))
        print(f'[bf16 id contract] {name}: teacher-forced over {T} steps sampled ids equal to the reference {100 * min(agree):.2f} % (worst step); smallest eps that explains every '
              f'difference: {eps_forced:.3g} logit units (contract E = PEAK x G = {E:.3g}); a band of that width covers {100 * near:.2f} % of the sampling decisions and '
              f'{100 * band:.2f} % of the re-masking candidates; free run not compared (the condition ids differ)')
        return eps_forced, None
    trace = {}
    ids = mg.generate(list(texts[:B]), timesteps=T, cond_scale=3., text_embeds=te, noise=noise.to(DEV), noise_kind='uniform', return_ids=True, trace=trace, **(gen_kw or {}))
    masked_ids = torch.stack(list(trace['masked_ids'])).cpu() if isinstance(trace['masked_ids'], list) else trace['masked_ids'].cpu()
    eps_free = TA.min_eps(lambda e: TA.compare_free_run(gen, masked_ids, ids.cpu(), counts, V, e))
    assert eps_forced is not None and eps_free is not None, f'{name}: a difference from the reference run lies outside every tie band'
    eps_star = max(eps_forced, eps_free)
    near, band = TA.band_population(gen, counts, V, max(eps_star, 1e-9))
    near_b, band_b = TA.band_population(gen, counts, V, E)
    rep = TA.compare_free_run(gen, masked_ids, ids.cpu(), counts, V, max(eps_free, 1e-9))
    print(f'[bf16 id contract] {name}: teacher-forced over {T} steps sampled ids equal to the reference {100 * min(agree):.2f} % (worst step); smallest eps that explains EVERY '
          f'difference by a tie band of the reference: teacher-forced {eps_forced:.3g}, free run {eps_free:.3g} logit units (contract E = PEAK x G = {E:.3g}); a band of eps* covers '
          f'{100 * near:.2f} % of the sampling decisions and {100 * band:.2f} % of the re-masking candidates (at E: {100 * near_b:.2f} % / {100 * band_b:.2f} %); free run: '
          + '; '.join(f"sample {b}: {r['status']}" + (f" at step {r['step']} ({r['kind']})" if r['status'] == 'tie' else '') for b, r in enumerate(rep)))
    assert eps_star <= E, f'{name}: eps* = {eps_star:.3g} > E = {E:.3g}: the bf16 engine took a decision the reference calls clear at its stated logit bound'
    return eps_forced, eps_free


def test_bf16_engine_differs_from_the_reference_only_inside_tie_bands_base(base, noise):
    """configs[1], both checkpoints (scanned input seeds).  E = PEAK x the asserted bound on the guidance-combined logits (test_transformer_forward_logits_at_base_size
    holds the engine to exactly that number)."""
    g, mg, inp = base
    tr = mg.transformer
    G = BF16_GUIDANCE['base_fp32w' if _fp32w(g) else 'base_bf16w']
    with torch.no_grad():
        tr.to_logits.weight.mul_(R.PEAK)
    try:
        assert R.state_checksum(tr) == g['weight_checksum_peaky']
        _bf16_tie_contract('base_c2_fp32' if _fp32w(g) else 'base_c2', g['generate'], mg, inp['text_embeds'].to(DEV), noise, 65536, R.PEAK * G)
    finally:
        with torch.no_grad():
            tr.to_logits.weight.div_(R.PEAK)
        torch.cuda.empty_cache()


def test_bf16_engine_differs_from_the_reference_only_inside_tie_bands_unscanned(base_s77, noise):
    """the general fp32 checkpoint on the UN-SCANNED input seed (the reference's own run has a 2.3e-6 near-tie there)"""
    g, mg, inp = base_s77
    try:
        _bf16_tie_contract('base_c2_fp32_s77', g['generate'], mg, inp['text_embeds'].to(DEV), noise, 65536, R.PEAK * BF16_GUIDANCE['base_fp32w'], gen_kw=dict(fmap_size=16))
    finally:
        torch.cuda.empty_cache()

# ------------------------------------------------------------------------------------------------ BASELINE configs[3]: super-resolution at full size
C4_ROWS = [0, 77, 255, 256, 511, 700, 1000, 1023]


@pytest.fixture(scope='module')
def superres(golden):
    import muse_maskgit_pytorch_amd as mm
    g = golden('superres_c4.pt')
    tr = R.build_transformer(mm.MaskGitTransformer, peaky=False, cfg=R.C4_CFG, seed=R.C4_WEIGHT_SEED)
    assert R.state_checksum(tr) == g['weight_checksum']
    mg = mm.MaskGit(vae=R.build_vae(mm.VQGanVAE), transformer=tr, image_size=512, cond_image_size=256).to(DEV).eval()
    assert R.state_checksum(mg.vae) == g['vae_weight_checksum']
    inp = R.c4_inputs()
    assert {k: R.checksum(v.float()) for k, v in inp.items()} == g['input_checksum']
    return g, mg, inp


@pytest.mark.parametrize('precision', PRECISIONS)
def test_superres_forward_and_generate_at_full_size(superres, precision):
    """configs[3] (1024 tokens, 256 low-resolution condition ids + text in the cross-attention context, V = 65536), batch 1, against the
    reference's fp32 run: condition ids from the low-resolution VAE, logits of the conditioned / null / guidance passes, and a 6-step
    generate with the reference's noise.  parity engine: ids 100 %, logits within 1e-3; bf16 engine: its own bounds, ids reported."""
    g, mg, inp = superres
    tr = mg.transformer
    te = inp['text_embeds'].to(DEV)
    mg.set_precision(precision)
    try:
        _, cids, _ = mg.cond_vae.encode(inp['cond_image'].to(DEV))
        agree = (cids.cpu() == g['cond_ids']).float().mean().item()
        print(f'[super-res parity] {precision} condition ids equal to the reference: {100 * agree:.2f} %')
        if precision in EXACT:
            assert agree == 1.0
        cids = g['cond_ids'].to(DEV)                         # the forward is compared on the reference's condition ids
        ids = inp['ids'].to(DEV)
        lc, emb = tr(ids, text_embeds=te, cond_drop_prob=0., conditioning_token_ids=cids, return_embed=True)
        ln = tr(ids, text_embeds=te, cond_drop_prob=1., conditioning_token_ids=cids)
        sc = tr.forward_with_cond_scale(ids, text_embeds=te, conditioning_token_ids=cids, cond_scale=3.)
        fw = g['forward']
        tol = 1e-3 if precision in EXACT else 2.5e-2
        for name, got, rec, k in (('logits(cond)', lc, fw['logits_cond'], 1), ('logits(null)', ln, fw['logits_null'], 1), ('logits(guidance)', sc, fw['logits_scaled'], 5)):
            f = got.reshape(1024, -1)
            bound = BF16_GUIDANCE['c4'] if (k == 5 and precision not in EXACT) else tol * k
            _err(f'{precision} super-res {name} full rows', f[C4_ROWS], rec['rows'], bound)
            _err(f'{precision} super-res {name} strided columns', f[:, ::128], rec['cols'], bound)
        _err(f'{precision} super-res embed', emb.reshape(1, 1024, -1), fw['embed'], 1e-3 if precision in EXACT else 4e-2)
        # ---- generate, peaky logits, the reference's noise
        gen = g['generate']
        us = []
        for s, u in enumerate(R.noise_stream(R.C4_T, R.C4_NOISE_SEED, (1, 1024, 65536))):
            assert R.checksum(u) == gen['noise_checksum'][s]
            us.append(u)
        noise = torch.stack(us).to(DEV)
        with torch.no_grad():
            tr.to_logits.weight.mul_(R.PEAK)
        try:
            assert R.state_checksum(tr) == g['weight_checksum_peaky']
            trace = {}
            out = mg.generate(['a'], cond_images=inp['cond_image'].to(DEV), timesteps=R.C4_T, cond_scale=3., text_embeds=te, noise=noise, noise_kind='uniform',
                              return_ids=True, trace=trace)
            masked = torch.stack(list(trace['masked_ids'])).cpu() if isinstance(trace['masked_ids'], list) else trace['masked_ids'].cpu()
            steps = [(masked[s] == gen['step_in_ids'][s].long()).float().mean().item() for s in range(R.C4_T)]
            final = (out.cpu().reshape(gen['final_ids'].shape) == gen['final_ids']).float().mean().item()
            print(f'[super-res parity] {precision} generate: final ids equal to the reference run {100 * final:.2f} %, per-step states min {100 * min(steps):.2f} %')
            if precision in ('bf16x3', 'f16x2'):
                assert isinstance(trace['masked_ids'], torch.Tensor)
            if precision in EXACT:
                assert final == 1.0 and min(steps) == 1.0
            else:
                assert final >= 0.85
                assert mg.fused_sampling_fallbacks == 0
                out2 = mg.generate(['a'], cond_images=inp['cond_image'].to(DEV), timesteps=R.C4_T, cond_scale=3., text_embeds=te, noise=noise, noise_kind='uniform',
                                   return_ids=True, fused_sampling=False)
                assert torch.equal(out, out2)      # both sampling paths give bit-identical confidences (common.h tile_softmax_stats)
                # round 6: the tie-aware id contract (the engine's own condition ids differ from the reference's in a few LFQ sign near-ties: the free run uses them,
                # the teacher-forced steps the reference's)
                _bf16_tie_contract('superres_c4', gen, mg, te, noise, 65536, R.PEAK * BF16_GUIDANCE['c4'], fwd_kw=dict(conditioning_token_ids=g['cond_ids'].to(DEV)),
                                   gen_kw=dict(cond_images=inp['cond_image'].to(DEV)), texts=('a',), free=(agree == 1.0))
        finally:
            with torch.no_grad():
                tr.to_logits.weight.div_(R.PEAK)
    finally:
        mg.set_precision('bf16')
        torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------ BASELINE configs[4] shape: paper-scale transformer at full size
C5_ROWS = [0, 77, 255, 256, 300, 411, 500, 511]


@pytest.fixture(scope='module')
def paper(golden):
    import muse_maskgit_pytorch_amd as mm
    g = golden('paper_c5.pt')
    tr = R.build_transformer(mm.MaskGitTransformer, peaky=False, cfg=R.C5_CFG, seed=R.C5_WEIGHT_SEED)
    assert R.state_checksum(tr) == g['weight_checksum']
    mg = mm.MaskGit(vae=None, transformer=tr, image_size=256).to(DEV).eval()
    inp = R.c5_inputs()
    assert {k: R.checksum(v.float()) for k, v in inp.items()} == g['input_checksum']
    return g, mg, inp


@pytest.mark.parametrize('precision', PRECISIONS)
def test_paper_scale_forward_and_generate_at_full_size(paper, precision):
    """configs[4] shape (dim 1024, depth 24, 16 heads, V = 8192, the 512 -> 1024 text projection), batch 2, against the reference's fp32 run:
    logits of the conditioned / null / guidance passes, the embed, and a 5-step generate with the reference's noise.
    parity engine: ids 100 %, logits within 1e-3; bf16 engine: its own bounds (24 layers of bf16 activations), ids reported."""
    g, mg, inp = paper
    tr = mg.transformer
    te, ids = inp['text_embeds'].to(DEV), inp['ids'].to(DEV)
    mg.set_precision(precision)
    try:
        lc, emb = tr(ids, text_embeds=te, cond_drop_prob=0., return_embed=True)
        ln = tr(ids, text_embeds=te, cond_drop_prob=1.)
        sc = tr.forward_with_cond_scale(ids, text_embeds=te, cond_scale=3.)
        fw = g['forward']
        tol = 1e-3 if precision in EXACT else 3.5e-2
        for name, got, rec, k in (('logits(cond)', lc, fw['logits_cond'], 1), ('logits(null)', ln, fw['logits_null'], 1), ('logits(guidance)', sc, fw['logits_scaled'], 3)):
            f = got.reshape(512, -1)
            bound = BF16_GUIDANCE['c5'] if (k == 3 and precision not in EXACT) else tol * k
            _err(f'{precision} paper-scale {name} full rows', f[C5_ROWS], rec['rows'], bound)
            _err(f'{precision} paper-scale {name} strided columns', f[:, ::16], rec['cols'], bound)
        _err(f'{precision} paper-scale embed', emb[:, ::4], fw['embed'], 1e-3 if precision in EXACT else 0.06)
        gen = g['generate']
        us = []
        for s, u in enumerate(R.noise_stream(R.C5_T, R.C5_NOISE_SEED, (2, 256, 8192))):
            assert R.checksum(u) == gen['noise_checksum'][s]
            us.append(u)
        noise = torch.stack(us).to(DEV)
        with torch.no_grad():
            tr.to_logits.weight.mul_(R.PEAK)
        try:
            assert R.state_checksum(tr) == g['weight_checksum_peaky']
            trace = {}
            out = mg.generate(['a', 'b'], fmap_size=16, timesteps=R.C5_T, cond_scale=3., text_embeds=te, noise=noise, noise_kind='uniform', return_ids=True,
                              trace=trace)
            masked = torch.stack(list(trace['masked_ids'])).cpu() if isinstance(trace['masked_ids'], list) else trace['masked_ids'].cpu()
            steps = [(masked[s] == gen['step_in_ids'][s].long()).float().mean().item() for s in range(R.C5_T)]
            final = (out.cpu().reshape(gen['final_ids'].shape) == gen['final_ids']).float().mean().item()
            print(f'[paper-scale parity] {precision} generate: final ids equal to the reference run {100 * final:.2f} %, per-step states min {100 * min(steps):.2f} %')
            if precision in ('bf16x3', 'f16x2'):
                assert isinstance(trace['masked_ids'], torch.Tensor)
            if precision in EXACT:
                assert final == 1.0 and min(steps) == 1.0
            else:
                assert final >= 0.80
                out2 = mg.generate(['a', 'b'], fmap_size=16, timesteps=R.C5_T, cond_scale=3., text_embeds=te, noise=noise, noise_kind='uniform', return_ids=True,
                                   fused_sampling=False)
                assert torch.equal(out, out2)      # both sampling paths give bit-identical confidences (common.h tile_softmax_stats)
                _bf16_tie_contract('paper_c5', gen, mg, te, noise, 8192, R.PEAK * BF16_GUIDANCE['c5'], gen_kw=dict(fmap_size=16))
        finally:
            with torch.no_grad():
                tr.to_logits.weight.div_(R.PEAK)
    finally:
        mg.set_precision('bf16')
        torch.cuda.empty_cache()


def test_training_gradients_at_base_size():
    """The training path at BASELINE configs[1] size (dim 512, depth 8, V = 65536, B = 2 x 256 tokens): loss and every parameter gradient of
    the hand-written MI355X backward against torch autograd of the oracle on the same (bf16-representable) weights -- the fixed-shape
    gradient tests run at the tiny widths only."""
    import muse_oracle as O
    import muse_maskgit_pytorch_amd as mm
    tr = R.build_transformer(mm.MaskGitTransformer, peaky=False).train()
    sd0 = {k: v.detach().clone() for k, v in tr.state_dict().items()}
    inp = R.inputs()
    ids, te = inp['ids'], inp['text_embeds']
    g = torch.Generator().manual_seed(11)
    labels = torch.randint(0, 65536, ids.shape, generator=g)
    labels[ids != 65536] = -1                                      # loss on the masked positions, like MaskGit.forward (mmp.py:700-712)
    tr = tr.to(DEV)
    loss = tr(ids.to(DEV), text_embeds=te.to(DEV), labels=labels.to(DEV), ignore_index=-1)
    loss.backward()
    sd = {k: (v.float().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd0.items()}
    ref = O.transformer_loss(sd, dict(depth=8, heads=8), ids, te, labels, ignore_index=-1, rp=O.bf16_round)
    ref.backward()
    print(f'[base-size parity] training loss {loss.item():.5f} vs oracle {ref.item():.5f}')
    assert abs(loss.item() - ref.item()) < 1e-2 * abs(ref.item())
    worst = (0., '')
    for name, p in tr.named_parameters():
        if name.startswith('self_cond_to_init_embed') or name == 'norm.gamma':
            continue
        rg, gg = sd[name].grad, p.grad.float().cpu()
        rel = (gg - rg).abs().max().item() / (rg.abs().max().item() + 1e-20)
        cos = torch.nn.functional.cosine_similarity(gg.flatten(), rg.flatten(), dim=0).item()
        worst = max(worst, (rel, name))
        assert rel < 4e-2 and cos > 0.995, f'{name}: rel {rel:.3e} cos {cos:.5f}'
    print(f'[base-size parity] training gradients: worst per-tensor relative error {worst[0]:.3e} ({worst[1]})')
    del tr
    torch.cuda.empty_cache()
