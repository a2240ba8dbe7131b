"""TEST INFRASTRUCTURE ONLY.  BASELINE configs[3] (super-resolution 512 x 512: 1024 tokens, 256 condition ids from the low-resolution VAE in
the cross-attention context) AT FULL SIZE from the UNMODIFIED reference, batch 1, fp32 CPU.  Same conventions as make_golden_base.py: the
checkpoint and the noise are rebuilt from seeds (oracle/golden_recipe.py), exact checksums stored.   python oracle/make_golden_c4.py  (~3 min)
Stored: logits at 8 full rows + every 128th column of the 1024 rows (cond / null / guidance) and the embed of one forward with the
reference-encoded condition ids; the condition ids; per-step ids and final ids of a 6-step generate with peaky logits."""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_recipe as R  # noqa: E402
from reference_harness import DecisionRecorder, reference_modules  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'superres_c4.pt')
FULL_ROWS = [0, 77, 255, 256, 511, 700, 1000, 1023]


def main():
    t0 = time.time()
    pkg, mmp, vaemod, att = reference_modules()
    inp = R.c4_inputs()
    ids, te = inp['ids'], inp['text_embeds']
    out = dict(full_rows=FULL_ROWS, col_stride=128, input_checksum={k: R.checksum(v.float()) for k, v in inp.items()})
    tr = R.build_transformer(pkg.MaskGitTransformer, peaky=False, cfg=R.C4_CFG, seed=R.C4_WEIGHT_SEED)
    out['weight_checksum'] = R.state_checksum(tr)
    vae = R.build_vae(pkg.VQGanVAE)
    mg = pkg.MaskGit(vae=vae, transformer=tr, image_size=512, cond_image_size=256)
    out['vae_weight_checksum'] = R.state_checksum(mg.vae)
    with torch.no_grad():
        _, cond_ids, _ = mg.cond_vae.encode(inp['cond_image'])
        lc, emb = tr(ids, text_embeds=te, cond_drop_prob=0., conditioning_token_ids=cond_ids, return_embed=True)
        ln = tr(ids, text_embeds=te, cond_drop_prob=1., conditioning_token_ids=cond_ids)
        sc = tr.forward_with_cond_scale(ids, text_embeds=te, conditioning_token_ids=cond_ids, cond_scale=3.)

    def sample(lg):
        f = lg.reshape(1024, -1)
        return dict(rows=f[FULL_ROWS].clone(), cols=f[:, ::128].clone())

    out['cond_ids'] = cond_ids.clone()
    out['forward'] = dict(logits_cond=sample(lc), logits_null=sample(ln), logits_scaled=sample(sc), embed=emb.clone())
    print(f'forward done {time.time() - t0:.0f}s: logits std {lc.std().item():.3f}')
    del lc, ln, sc
    with torch.no_grad():
        tr.to_logits.weight.mul_(R.PEAK)
    out['weight_checksum_peaky'] = R.state_checksum(tr)
    tr.encode_text = lambda texts, te=te: te
    rec = dict(step_in_ids=[], noise_checksum=[])
    orig_fw = tr.forward_with_cond_scale

    def fw(ids_, *a, **kw):
        rec['step_in_ids'].append(ids_.clone().to(torch.int32))
        return orig_fw(ids_, *a, **kw)

    tr.forward_with_cond_scale = fw
    dec = DecisionRecorder(mmp, (1, 1024), on_noise=lambda u: rec['noise_checksum'].append(R.checksum(u)))      # round 6: + what tests/tie_aware.py needs
    final = {}
    orig_dec = mg.vae.decode_from_ids

    def dec_rec(i):
        final['ids'] = i.clone()
        return orig_dec(i)

    mg.vae.decode_from_ids = dec_rec
    torch.manual_seed(R.C4_NOISE_SEED)
    with torch.no_grad(), dec:
        images = mg.generate(['a'], cond_images=inp['cond_image'], timesteps=R.C4_T, cond_scale=3.)
    tr.forward_with_cond_scale = orig_fw
    for s, u in enumerate(R.noise_stream(R.C4_T, R.C4_NOISE_SEED, (1, 1024, 65536))):
        assert R.checksum(u) == rec['noise_checksum'][s], f'noise recipe does not reproduce step {s}'
    out['generate'] = dict(step_in_ids=torch.stack(rec['step_in_ids']), final_ids=final['ids'].clone(), noise_checksum=rec['noise_checksum'], **dec.stacked(),
                           images_strided=images[:, :, ::8, ::8].clone(), images_absmax=images.abs().max().item())
    torch.save(out, OUT)
    print(f'wrote {OUT} ({os.path.getsize(OUT) / 1e6:.1f} MB) in {time.time() - t0:.0f}s')


if __name__ == '__main__':
    main()
