"""Randomised cross-check of the two drivers of the decode loop: one mm_generate call (every variant inside, fused sampling where the vocabulary
allows it) against the same loop run operator by operator from Python (stepwise=True, logits materialised), over seeded random shapes --
batch, grid, width, heads, depth, vocabulary, text length, conditioning ids, decode variant, timesteps, guidance scale.  With a critic, or with
fused sampling off, the two must agree bit for bit; so they must with fused sampling on: both sampling paths combine the same per-tile softmax
statistics in the same order (common.h tile_softmax_stats), the confidences are bit-identical."""
import random

import pytest
import torch

import muse_maskgit_pytorch_amd as mm

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _case(rng):
    heads = rng.choice([2, 4, 8])
    cfg = dict(B=rng.randint(1, 5), fmap=rng.choice([4, 6, 8, 10]), dim=rng.choice([128, 256, 512]), heads=heads, depth=rng.randint(1, 2),
               V=rng.choice([512, 1000, 4096, 8192]), L=rng.randint(1, 9), nc=rng.choice([0, 0, 16]), T=rng.randint(2, 7),
               # the variants combine freely (mmp.py:540-609): a critic (token / self), self-conditioning, re-masking, single pass
               critic=rng.choice([None, None, 'token', 'self']), self_cond=rng.random() < 0.3, can_remask=rng.random() < 0.25,
               cond_scale=rng.choice([1, 1.5, 3.0, 4.0]))
    return cfg


@pytest.mark.parametrize('seed', list(range(40)))
def test_mm_generate_equals_the_stepwise_loop_on_random_shapes(seed):
    rng = random.Random(1000 + seed)
    c = _case(rng)
    torch.manual_seed(seed)
    n = c['fmap'] ** 2
    kw = dict(num_tokens=c['V'], seq_len=n, dim=c['dim'], depth=c['depth'], dim_head=64, heads=c['heads'], t5_name='t5-small')
    t = mm.MaskGitTransformer(self_cond=c['self_cond'], **kw)
    with torch.no_grad():
        t.to_logits.weight.mul_(6.)
    extra, gkw = {}, {}
    if c['critic'] == 'token':
        extra['token_critic'] = mm.TokenCritic(**dict(kw, dim=rng.choice([128, 256]), heads=rng.choice([2, 4])))
    elif c['critic'] == 'self':
        extra['self_token_critic'] = True
    if c['can_remask']:
        extra['no_mask_token_prob'] = 0.25
        gkw['can_remask_prev_masked'] = True
    cond_scale = c['cond_scale']
    vae = cond = None
    image_size = 16 * c['fmap']
    if c['nc']:
        vae = mm.VQGanVAE(dim=16, codebook_size=c['V'] if c['V'] in (512, 4096, 8192) else 512)
        if vae.codebook_size != c['V']:
            c['nc'] = 0
    if c['nc']:
        mg = mm.MaskGit(image_size=image_size, transformer=t, vae=vae, cond_vae=vae.copy_for_eval(), cond_image_size=64, **extra).to(DEV)
        cond = torch.randn(c['B'], 3, 64, 64, device=DEV)
    else:
        mg = mm.MaskGit(image_size=image_size, transformer=t, vae=None, **extra).to(DEV)
    te = torch.randn(c['B'], c['L'], 512, device=DEV)
    if c['L'] > 2 and c['B'] > 1:
        te[1, c['L'] // 2:] = 0.                                       # zero-padded text rows = masked keys
    critic = c['critic'] is not None
    if critic:
        gkw['critic_noise'] = torch.rand(c['T'], c['B'], n, device=DEV)
    common = dict(timesteps=c['T'], text_embeds=te, seed=seed, fmap_size=c['fmap'], cond_scale=cond_scale, cond_images=cond, return_ids=True, **gkw)
    ta, tb = {}, {}
    a = mg.generate([''] * c['B'], trace=ta, **common)
    b = mg.generate([''] * c['B'], trace=tb, stepwise=True, **common)
    nofuse = mg.generate([''] * c['B'], fused_sampling=False, **common)
    assert a.shape == (c['B'], c['fmap'], c['fmap']) and int(a.min()) >= 0 and int(a.max()) < c['V'], c
    assert torch.equal(nofuse, b), f'logits path of mm_generate != stepwise loop: {c}'
    assert torch.equal(ta['ids'][0], torch.stack(tb['ids'])[0]), f'first step differs: {c}'
    fused_possible = t._model().fused_ready and cond_scale != 1
    assert torch.equal(a, b), f'mm_generate != stepwise loop: {c}'      # fused or not: the same per-tile softmax statistics on both sampling paths
    assert mg.fused_sampling_fallbacks == 0 or fused_possible
