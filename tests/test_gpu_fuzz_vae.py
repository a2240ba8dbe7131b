"""Randomised parity of VQGanVAE.decode_from_ids / encode (one C call each: mm_vae_decode_from_ids / mm_vae_encode) against the CPU oracle over
seeded random shapes -- width, number of down/up-sampling layers, codebook size, batch, non-square images.  Parity engine: pixels within 1e-3
of the image scale and every LFQ bit equal wherever the fp32 pre-sign value is not itself within 1e-5 of zero; bf16 engine: the bound it
achieves on the fixtures (pixels within 3 % of the image scale against the rounding-point oracle)."""
import random

import pytest
import torch

import muse_oracle as O

import muse_maskgit_pytorch_amd as mm

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('seed', list(range(8)))
def test_vae_matches_the_oracle_on_random_shapes(seed):
    rng = random.Random(900 + seed)
    dim, layers = rng.choice([16, 32]), rng.choice([2, 3, 4])
    V = rng.choice([256, 512, 4096, 65536])
    B, h, w = rng.randint(1, 3), rng.randint(1, 5), rng.randint(1, 5)
    torch.manual_seed(seed)
    vae = mm.VQGanVAE(dim=dim, layers=layers, codebook_size=V, use_vgg_and_gan=False).eval()
    sd = {k: (v.detach().float() if v.is_floating_point() else v.detach()).clone() for k, v in vae.state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, V, (B, h, w), generator=g)
    f = 2 ** layers
    img = torch.randn(B, 3, h * f, w * f, generator=g)
    vae = vae.to(DEV)
    for precision, rp, rel in (('parity', None, 1e-3), ('bf16', O.bf16_round, 3e-2)):
        vae.set_precision(precision)
        got = vae.decode_from_ids(ids.to(DEV)).float().cpu()
        ref = O.vae_decode_from_ids(sd, ids, layers=layers, rp=rp)
        assert got.shape == ref.shape == (B, 3, h * f, w * f)
        scale = max(ref.abs().max().item(), 1e-3)
        err = (got - ref).abs().max().item()
        assert err <= rel * scale, f'{precision} decode: {err:.3g} on scale {scale:.3g}; dim={dim} layers={layers} V={V} B={B} h={h} w={w}'
        fmap, eids, aux = vae.encode(img.to(DEV))
        rf, rids = O.vae_encode(sd, img, layers=layers, rp=rp)[:2]
        assert eids.shape == rids.shape == (B, h, w) and fmap.shape == rf.shape
        if precision == 'parity':
            # the ids are signs of project_in(features): equal wherever the oracle's own pre-sign value is not numerically zero
            agree = (eids.cpu() == rids).float().mean().item()
            assert agree >= 0.98, f'parity encode ids: {agree}'
            same = eids.cpu() == rids
            if same.all():
                assert (fmap.float().cpu() - rf).abs().max().item() <= 1e-3 * max(rf.abs().max().item(), 1e-3)
