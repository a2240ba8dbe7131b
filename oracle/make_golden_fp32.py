"""TEST INFRASTRUCTURE ONLY.  tests/golden/tiny_fp32.pt: the C1-shape run of oracle/make_golden.py repeated on GENERAL fp32 weights.

Every other tiny fixture rounds the parameters to bf16-representable values before the UNMODIFIED reference runs (make_golden.py:21-25), so
an engine that silently multiplied by bf16-rounded weights would pass all of them.  Here nothing is rounded: the module constructors'
fp32 values (mmp.py:85,88,118-124,233 -- what every checkpoint the reference initialises or trains holds), non-trivial scales / gains, peaky
logits.  Recorded from the reference on CPU in fp32: logits of the conditioned / null / guidance passes and the embed, a 4-step
`MaskGit.generate` with the reference's own noise draws (ids entering every step, final ids, images), VQGanVAE decode / encode.
The state dicts are stored as fp32 (that is the point).  Run in the build container only:   python oracle/make_golden_fp32.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from reference_harness import reference_modules, NoiseTape  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'tiny_fp32.pt')


def main():
    pkg, mmp, vaemod, att = reference_modules()
    gen = torch.Generator().manual_seed(4321)
    torch.manual_seed(10)
    tcfg = dict(num_tokens=512, seq_len=64, dim=128, depth=2, dim_head=64, heads=8)
    tr = pkg.MaskGitTransformer(t5_name='t5-small', **tcfg)
    with torch.no_grad():
        for name, p in tr.named_parameters():
            if name.endswith('q_scale') or name.endswith('k_scale') or name.endswith('gamma'):
                p.mul_(1 + 0.2 * torch.randn(p.shape, generator=gen))
        tr.to_logits.weight.mul_(8.)      # peaky logits: well-separated confidences (SURVEY 8c determinism control 3)
    tr.eval()
    n_general = sum(int((p != p.to(torch.bfloat16).float()).sum()) for p in tr.parameters())
    assert n_general > 0.9 * sum(p.numel() for p in tr.parameters() if p.dim() > 1), 'the point of this fixture: weights that are NOT bf16-representable'
    b, n, L = 2, 64, 7
    ids = torch.randint(0, 512, (b, n), generator=gen)
    ids[torch.rand(b, n, generator=gen) < 0.5] = tr.mask_id
    te = torch.randn(b, L, 512, generator=gen)
    te[1, L - 2:] = 0
    with torch.no_grad():
        logits_c, embed = tr(ids, text_embeds=te, cond_drop_prob=0., return_embed=True)
        logits_n = tr(ids, text_embeds=te, cond_drop_prob=1.)
        scaled = tr.forward_with_cond_scale(ids, text_embeds=te, cond_scale=3.)
    out = dict(cfg=tcfg, sd={k: v.clone() for k, v in tr.state_dict().items()}, ids=ids, text_embeds=te, mask_id=tr.mask_id,
               logits_cond=logits_c, logits_null=logits_n, logits_scaled=scaled, embed=embed)

    torch.manual_seed(11)
    vcfg = dict(dim=16, codebook_size=512)
    vae = pkg.VQGanVAE(**vcfg).eval()
    vae_eval = vae.copy_for_eval()
    vids = torch.randint(0, 512, (2, 8, 8), generator=gen)
    img_in = torch.randn(2, 3, 128, 128, generator=gen)
    with torch.no_grad():
        dec = vae_eval.decode_from_ids(vids)
        fmap, enc_ids, aux = vae_eval.encode(img_in)
        pre = vae_eval.enc_dec.encode(img_in)
        pre_sign = vae_eval.quantizer.project_in(pre.permute(0, 2, 3, 1).reshape(2, 64, -1))
    out['vae'] = dict(cfg=vcfg, sd={k: v.clone() for k, v in vae_eval.state_dict().items()}, ids=vids, decoded=dec, image=img_in, enc_ids=enc_ids,
                      enc_pre_sign=pre_sign)

    T = 4
    mg = pkg.MaskGit(vae=vae, transformer=tr, image_size=128)
    tr.encode_text = lambda texts, te=te: te
    rec = dict(step_ids=[])
    orig = tr.forward_with_cond_scale

    def fwcs(ids_, *a, **kw):
        rec['step_ids'].append(ids_.clone())
        return orig(ids_, *a, **kw)

    tr.forward_with_cond_scale = fwcs
    final = {}
    orig_dec = mg.vae.decode_from_ids

    def dec_rec(i):
        final['ids'] = i.clone()
        return orig_dec(i)

    mg.vae.decode_from_ids = dec_rec
    torch.manual_seed(104)
    with NoiseTape(mmp) as tape, torch.no_grad():
        images = mg.generate(['a', 'b'], timesteps=T)
    tr.forward_with_cond_scale = orig
    out['generate'] = dict(timesteps=T, uniform=tape.uniform_draws, step_ids=rec['step_ids'], final_ids=final['ids'], images=images)
    torch.save(out, OUT)
    print(f'wrote {OUT} ({os.path.getsize(OUT) / 1e6:.1f} MB); logits absmax {logits_c.abs().max().item():.2f}, {n_general} parameters are not bf16-representable')


if __name__ == '__main__':
    main()
