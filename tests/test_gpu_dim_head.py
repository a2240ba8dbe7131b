"""dim_head 32 and 128 (muse_maskgit_pytorch.py:165-174 accepts any head width; VERDICT r2 missing #6): the Attend seam, the transformer forward in
all three precisions against the oracle, and the decode loop (one mm_generate call against the stepwise loop).  dim_head 64 keeps the tuned bf16
kernels of csrc/attention.hip; other widths run on the templated fp32-MFMA kernel (csrc/attention_f32.hip), also inside the bf16 engine."""
import pytest
import torch
import torch.nn.functional as F

import muse_oracle as O
from conftest import sd_f32

import muse_maskgit_pytorch_amd as mm
from muse_maskgit_pytorch_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('dh', [32, 128])
@pytest.mark.parametrize('nq,nk,masked', [(64, 64, False), (70, 33, True), (256, 257, True)])
def test_attend_seam_other_head_widths(dh, nq, nk, masked):
    g = torch.Generator().manual_seed(dh + nq + nk)
    b, h = 2, 3
    q, k, v = (torch.randn(b, h, n_, dh, generator=g) for n_ in (nq, nk, nk))
    q, k = F.normalize(q, dim=-1).bfloat16().float(), F.normalize(k, dim=-1).bfloat16().float()      # scores in [-8, 8] like the l2-normalised q / k of the model
    v = v.bfloat16().float()
    km = (torch.rand(b, nk, generator=g) < 0.6) if masked else None
    if masked:
        km[:, 0] = True
    mask4 = km[:, None, None, :].expand(b, h, nq, nk) if masked else None
    ref = O.attend(q.double(), k.double(), v.double(), mask=mask4, scale=8.0)
    got = mm.attend.Attend(scale=8)(q.to(DEV), k.to(DEV), v.to(DEV), mask=mask4.to(DEV) if masked else None)
    err = (got.double().cpu() - ref).abs().max().item()
    print(f'[dim_head] attend dh={dh} nq={nq} nk={nk}: max err {err:.3g}')
    assert err < 0.02 * ref.abs().max().item() + 1e-3          # bf16 output rounding


def _model(dh, heads, self_cond=False):
    torch.manual_seed(dh)
    t = mm.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=2, dim_head=dh, heads=heads, t5_name='t5-small', self_cond=self_cond)
    with torch.no_grad():
        t.to_logits.weight.mul_(6.)
        for p in t.parameters():
            p.copy_(p.bfloat16().float())
    return t.to(DEV).eval()


@pytest.mark.parametrize('dh,heads', [(32, 4), (128, 2), (32, 2)])
@pytest.mark.parametrize('precision', ['bf16', 'bf16x3', 'parity'])
def test_forward_other_head_widths_vs_oracle(dh, heads, precision):
    t = _model(dh, heads).set_precision(precision)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 513, (3, 64), generator=g)
    te = torch.randn(3, 6, 512, generator=g)
    te[2, 4:] = 0.
    sd = sd_f32({k: v.cpu() for k, v in t.state_dict().items()})
    ref = O.transformer_forward(sd, dict(depth=2, heads=heads), ids, te, 0.)
    got = t(ids.to(DEV), text_embeds=te.to(DEV)).cpu()
    err, scale = (got - ref).abs().max().item(), ref.abs().max().item()
    print(f'[dim_head] forward dh={dh} heads={heads} {precision}: max err {err:.3g} (scale {scale:.3g})')
    assert err < (0.05 if precision == 'bf16' else 1e-4) * scale
    sc = t.forward_with_cond_scale(ids.to(DEV), text_embeds=te.to(DEV), cond_scale=3.).cpu()
    ref_sc = O.forward_with_cond_scale(sd, dict(depth=2, heads=heads), ids, te, 3.)
    assert (sc - ref_sc).abs().max().item() < (0.08 if precision == 'bf16' else 3e-4) * ref_sc.abs().max().item()


@pytest.mark.parametrize('dh,heads', [(32, 4), (128, 2)])
@pytest.mark.parametrize('precision', ['bf16', 'bf16x3'])
def test_generate_other_head_widths(dh, heads, precision):
    t = _model(dh, heads)
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None).to(DEV).set_precision(precision)
    te = torch.randn(3, 5, 512, device=DEV)
    kw = dict(timesteps=5, text_embeds=te, seed=3, fmap_size=8, cond_scale=3., return_ids=True)
    a = mg.generate([''] * 3, **kw)
    b = mg.generate([''] * 3, stepwise=True, **kw)
    assert a.shape == (3, 8, 8) and int(a.min()) >= 0 and int(a.max()) < 512
    assert torch.equal(a, b), f'{(a != b).sum().item()} ids differ between mm_generate and the stepwise loop'
    if precision == 'bf16x3':
        mg.set_precision('parity')
        ref = mg.generate([''] * 3, **kw)
        assert (a == ref).float().mean().item() >= 0.97


def test_training_rejects_other_head_widths_loudly():
    t = _model(32, 4).train()
    with pytest.raises(NotImplementedError, match='dim_head 64'):
        t(torch.randint(0, 512, (2, 64), device=DEV), text_embeds=torch.randn(2, 4, 512, device=DEV), labels=torch.randint(0, 512, (2, 64), device=DEV))
