"""TEST INFRASTRUCTURE ONLY.  BASELINE configs[4] SHAPE (paper-scale base transformer: dim 1024, depth 24, 16 heads, codebook 8192, text projection
512 -> 1024) AT FULL SIZE from the UNMODIFIED reference, batch 2, fp32 CPU.  Same conventions as make_golden_base.py: the 0.43 G-parameter
checkpoint and the noise are rebuilt from seeds (oracle/golden_recipe.py), exact checksums stored.   python oracle/make_golden_c5.py  (~3 min)
Stored: logits at 8 full rows + every 16th column of the 512 rows (cond / null / guidance) and the embed of one forward; per-step ids and final
ids of a 5-step generate (no VAE) with peaky logits."""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_recipe as R  # noqa: E402
from reference_harness import DecisionRecorder, reference_modules  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'paper_c5.pt')
FULL_ROWS = [0, 77, 255, 256, 300, 411, 500, 511]


def main():
    t0 = time.time()
    pkg, mmp, vaemod, att = reference_modules()
    inp = R.c5_inputs()
    ids, te = inp['ids'], inp['text_embeds']
    out = dict(full_rows=FULL_ROWS, col_stride=16, input_checksum={k: R.checksum(v.float()) for k, v in inp.items()})
    tr = R.build_transformer(pkg.MaskGitTransformer, peaky=False, cfg=R.C5_CFG, seed=R.C5_WEIGHT_SEED)
    out['weight_checksum'] = R.state_checksum(tr)
    with torch.no_grad():
        lc, emb = tr(ids, text_embeds=te, cond_drop_prob=0., return_embed=True)
        ln = tr(ids, text_embeds=te, cond_drop_prob=1.)
        sc = tr.forward_with_cond_scale(ids, text_embeds=te, cond_scale=3.)

    def sample(lg):
        f = lg.reshape(512, -1)
        return dict(rows=f[FULL_ROWS].clone(), cols=f[:, ::16].clone())

    out['forward'] = dict(logits_cond=sample(lc), logits_null=sample(ln), logits_scaled=sample(sc), embed=emb[:, ::4].clone())
    print(f'forward done {time.time() - t0:.0f}s: logits std {lc.std().item():.3f}')
    del lc, ln, sc
    with torch.no_grad():
        tr.to_logits.weight.mul_(R.PEAK)
    out['weight_checksum_peaky'] = R.state_checksum(tr)
    tr.encode_text = lambda texts, te=te: te
    torch.manual_seed(R.VAE_SEED)
    mg = pkg.MaskGit(vae=pkg.VQGanVAE(dim=16, codebook_size=8192), transformer=tr, image_size=256)      # (the reference's constructor insists on a VAE; only its ids hook is used)
    final = {}

    def dec_rec(i):
        final['ids'] = i.clone()
        return torch.zeros(i.shape[0], 3, 8, 8)

    mg.vae.decode_from_ids = dec_rec
    rec = dict(step_in_ids=[], noise_checksum=[])
    orig_fw = tr.forward_with_cond_scale

    def fw(ids_, *a, **kw):
        rec['step_in_ids'].append(ids_.clone().to(torch.int32))
        return orig_fw(ids_, *a, **kw)

    tr.forward_with_cond_scale = fw
    dec = DecisionRecorder(mmp, (2, 256), on_noise=lambda u: rec['noise_checksum'].append(R.checksum(u)))      # round 6: + what tests/tie_aware.py needs
    torch.manual_seed(R.C5_NOISE_SEED)
    with torch.no_grad(), dec:
        mg.generate(['a', 'b'], fmap_size=16, timesteps=R.C5_T, cond_scale=3.)
    tr.forward_with_cond_scale = orig_fw
    for s, u in enumerate(R.noise_stream(R.C5_T, R.C5_NOISE_SEED, (2, 256, 8192))):
        assert R.checksum(u) == rec['noise_checksum'][s], f'noise recipe does not reproduce step {s}'
    out['generate'] = dict(step_in_ids=torch.stack(rec['step_in_ids']), final_ids=final['ids'].clone(), noise_checksum=rec['noise_checksum'], **dec.stacked())
    torch.save(out, OUT)
    print(f'wrote {OUT} ({os.path.getsize(OUT) / 1e6:.1f} MB) in {time.time() - t0:.0f}s')


if __name__ == '__main__':
    main()
