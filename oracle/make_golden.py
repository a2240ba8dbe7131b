"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.pt by running the UNMODIFIED reference
(/root/reference, imported through oracle/reference_harness.py) on CPU in fp32.

Run in the build container only:   python oracle/make_golden.py
The reference holds no tests / golden vectors of its own, so these files are
the pin for oracle/muse_oracle.py.  All weights are rounded to bf16-representable values BEFORE
the reference runs, so the same checkpoint is exactly loadable by the bf16 HIP path.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from reference_harness import reference_modules, NoiseTape  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def round_module_to_bf16_(m):
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    return m


def sd_bf16(sd):
    """store float tensors as bf16 (exact: they were rounded already), keep int buffers."""
    return {k: (v.to(torch.bfloat16) if v.is_floating_point() else v.clone()) for k, v in sd.items()}


def make_text_embeds(b, L, d, gen):
    te = torch.randn(b, L, d, generator=gen)
    te[1, L - 2:] = 0          # row 1 has a padded tail (t5.py:93 zero-fills padding)
    return te


def main():
    os.makedirs(OUT, exist_ok=True)
    pkg, mmp, vaemod, att = reference_modules()
    gen = torch.Generator().manual_seed(1234)

    # ------------------------------------------------------------------ tiny transformer (C1 shapes)
    torch.manual_seed(0)
    tcfg = dict(num_tokens=512, seq_len=64, dim=128, depth=2, dim_head=64, heads=8)
    tr = pkg.MaskGitTransformer(t5_name='t5-small', **tcfg)
    with torch.no_grad():
        # make the learned scales / norms non-trivial so a dropped multiply is caught
        for name, p in tr.named_parameters():
            if name.endswith('q_scale') or name.endswith('k_scale') or name.endswith('gamma'):
                p.mul_(1 + 0.2 * torch.randn(p.shape, generator=gen))
        # "peaky" logits so confidence scores are well separated (SURVEY 8c determinism control 3)
        tr.to_logits.weight.mul_(8.)
    round_module_to_bf16_(tr).eval()
    assert isinstance(tr.text_embed_proj, torch.nn.Linear)     # t5-small d_model 512 != dim 128
    b, n, L = 2, 64, 7
    ids = torch.randint(0, 512, (b, n), generator=gen)
    ids[torch.rand(b, n, generator=gen) < 0.5] = tr.mask_id
    te = make_text_embeds(b, L, 512, gen)
    with torch.no_grad():
        logits_c, embed = tr(ids, text_embeds=te, cond_drop_prob=0., return_embed=True)
        logits_n = tr(ids, text_embeds=te, cond_drop_prob=1.)
        scaled = tr.forward_with_cond_scale(ids, text_embeds=te, cond_scale=3.)
        # per-op known answers from the reference's own submodules
        x = torch.randn(b, n, 128, generator=gen)
        layer0 = tr.transformer_blocks.layers[0]
        ctx = tr.text_embed_proj(te)
        cmask = (te != 0).any(dim=-1)
        op = dict(
            x=x, ctx=ctx, cmask=cmask,
            ln=layer0[0].norm(x),
            self_attn=layer0[0](x),
            cross_attn=layer0[1](x, context=ctx, context_mask=cmask),
            ff=layer0[2](x),
        )
        # Attend: math branch vs default flash branch on the same q,k,v,mask
        q = torch.randn(b, 8, n, 64, generator=gen)
        k = torch.randn(b, 8, L + 1, 64, generator=gen)
        v = torch.randn(b, 8, L + 1, 64, generator=gen)
        m4 = torch.nn.functional.pad(cmask[:, None, None, :].expand(b, 8, n, L), (1, 0), value=True)
        att_math = att.Attend(flash=False)(q, k, v, mask=m4)
        att_flash = att.Attend(flash=True)(q, k, v, mask=m4)
        op.update(q=q, k=k, v=v, m4=m4, attend_math=att_math, attend_flash=att_flash)
    torch.save(dict(cfg=tcfg, sd=sd_bf16(tr.state_dict()), ids=ids, text_embeds=te, mask_id=tr.mask_id,
                    logits_cond=logits_c, logits_null=logits_n, logits_scaled=scaled, embed=embed, op=op),
               os.path.join(OUT, 'transformer_tiny.pt'))

    # ------------------------------------------------------------------ tiny VAE (dim=16 keeps the fixture small)
    torch.manual_seed(1)
    vcfg = dict(dim=16, codebook_size=512)
    vae = pkg.VQGanVAE(**vcfg)
    round_module_to_bf16_(vae).eval()
    vae_eval = vae.copy_for_eval()
    vids = torch.randint(0, 512, (2, 8, 8), generator=gen)
    img_in = torch.randn(2, 3, 128, 128, generator=gen)
    with torch.no_grad():
        dec = vae_eval.decode_from_ids(vids)
        fmap, enc_ids, aux = vae_eval.encode(img_in)
        enc_pre = vae_eval.enc_dec.encode(img_in)
    torch.save(dict(cfg=vcfg, sd=sd_bf16(vae_eval.state_dict()), ids=vids, decoded=dec, image=img_in,
                    enc_fmap=fmap, enc_ids=enc_ids, enc_pre_quant=enc_pre, fmap_size=vae_eval.get_encoded_fmap_size(128)),
               os.path.join(OUT, 'vae_tiny.pt'))

    # ------------------------------------------------------------------ full generate trace (tiny, T=4 and T=18)
    for T in (4, 18):
        mg = pkg.MaskGit(vae=vae, transformer=tr, image_size=128)
        tr.encode_text = lambda texts, te=te: te
        rec = dict(step_ids=[], step_logits=[])
        orig_fwcs = tr.forward_with_cond_scale

        def fwcs(ids_, *a, **kw):
            out = orig_fwcs(ids_, *a, **kw)
            rec['step_ids'].append(ids_.clone())
            rec['step_logits'].append(out[0].clone())
            return out

        tr.forward_with_cond_scale = fwcs
        final = {}
        orig_dec = mg.vae.decode_from_ids

        def dec_rec(i):
            final['ids'] = i.clone()
            return orig_dec(i)

        mg.vae.decode_from_ids = dec_rec
        torch.manual_seed(100 + T)
        with NoiseTape(mmp) as tape:
            images = mg.generate(['a', 'b'], timesteps=T)
        tr.forward_with_cond_scale = orig_fwcs
        torch.save(dict(timesteps=T, uniform=tape.uniform_draws, step_ids=rec['step_ids'],
                        step_logits=rec['step_logits'] if T == 4 else None,
                        final_ids=final['ids'], images=images if T == 4 else None),
                   os.path.join(OUT, f'generate_tiny_T{T}.pt'))

    # ------------------------------------------------------------------ sampling helpers at a larger vocabulary
    V = 8192
    lg = torch.randn(2, 8, V, generator=gen) * 1.5
    torch.manual_seed(7)
    with NoiseTape(mmp) as tape:
        filt = mmp.top_k(lg, 0.9)
        pred = mmp.gumbel_sample(filt, temperature=0.5)
        pred0 = mmp.gumbel_sample(filt, temperature=0.)
    torch.save(dict(logits=lg, filtered_isinf=torch.isinf(filt), pred_T05=pred, pred_T0=pred0,
                    uniform=tape.uniform_draws), os.path.join(OUT, 'sampling_v8192.pt'))

    # ------------------------------------------------------------------ mask schedule
    sched = {}
    for T, nseq in ((18, 256), (18, 1024), (4, 64), (18, 64), (8, 256)):
        cnt = []
        for timestep in torch.linspace(0, 1, T):
            cnt.append(max(int((mmp.cosine_schedule(timestep) * nseq).item()), 1))
        sched[(T, nseq)] = cnt
    torch.save(sched, os.path.join(OUT, 'schedule.pt'))
    # ------------------------------------------------------------------ training-forward losses (mmp.py:337-348, 383-386)
    gen2 = torch.Generator().manual_seed(99)
    lids = torch.randint(0, 512, (2, 64), generator=gen2)
    lmask = torch.rand(2, 64, generator=gen2) < 0.6
    labels = torch.where(lmask, lids, torch.full_like(lids, -1))
    lx = torch.where(lmask, torch.full_like(lids, tr.mask_id), lids)
    with torch.no_grad():
        loss, llogits = tr(lx, text_embeds=te, labels=labels, ignore_index=-1, return_logits=True)
        loss_drop = tr(lx, text_embeds=te, labels=labels, ignore_index=-1, cond_drop_prob=1.)
        critic = pkg.TokenCritic(num_tokens=512, seq_len=64, dim=128, depth=1, dim_head=64, heads=8, t5_name='t5-small')
        round_module_to_bf16_(critic).eval()
        y = (torch.rand(2, 64, generator=gen2) < 0.5).float()
        bce = critic(lx.clamp(max=511), text_embeds=te, labels=y)
    torch.save(dict(x=lx, labels=labels, loss=loss, logits=llogits, loss_drop=loss_drop, critic_sd=sd_bf16(critic.state_dict()),
                    critic_labels=y, critic_bce=bce), os.path.join(OUT, 'loss_tiny.pt'))
    # ------------------------------------------------------------------ decode variants (mmp.py:540-609): token critic,
    # self critic, self-conditioning, can_remask_prev_masked, cond_scale == 1
    variants = {}

    def run_variant(name, mg_, T, **gen_kw):
        tr_ = mg_.transformer
        tr_.encode_text = lambda texts, te=te: te
        rec = dict(step_ids=[])
        orig = tr_.forward_with_cond_scale

        def fw(ids_, *a, **kw):
            rec['step_ids'].append(ids_.clone())
            return orig(ids_, *a, **kw)

        tr_.forward_with_cond_scale = fw
        critic_u = []
        orig_uniform = mmp.uniform

        def uniform_rec(shape, min=0, max=1, device=None):
            u = orig_uniform(shape, min, max, device)
            critic_u.append(u.clone())
            return u

        mmp.uniform = uniform_rec
        final = {}
        orig_dec = mg_.vae.decode_from_ids

        def dec_rec(i):
            final['ids'] = i.clone()
            return orig_dec(i)

        mg_.vae.decode_from_ids = dec_rec
        torch.manual_seed(500 + len(variants))
        with NoiseTape(mmp) as tape:
            mg_.generate(['a', 'b'], timesteps=T, **gen_kw)
        out = final['ids']
        mmp.uniform = orig_uniform
        tr_.forward_with_cond_scale = orig
        variants[name] = dict(timesteps=T, uniform=tape.uniform_draws, critic_uniform=critic_u, step_ids=rec['step_ids'], final_ids=out)

    torch.manual_seed(3)
    critic2 = pkg.TokenCritic(num_tokens=512, seq_len=64, dim=128, depth=1, dim_head=64, heads=8, t5_name='t5-small')
    with torch.no_grad():
        critic2.to_logits.weight.mul_(8.)
    round_module_to_bf16_(critic2).eval()
    critic2.encode_text = lambda texts, te=te: te
    run_variant('token_critic', pkg.MaskGit(vae=vae, transformer=tr, image_size=128, token_critic=critic2), 5)
    variants['token_critic']['critic_sd'] = sd_bf16(critic2.state_dict())

    torch.manual_seed(4)
    mg_sc = pkg.MaskGit(vae=vae, transformer=tr, image_size=128, self_token_critic=True)
    with torch.no_grad():
        mg_sc.token_critic.to_pred.weight.mul_(8.)
    round_module_to_bf16_(mg_sc.token_critic.to_pred)
    run_variant('self_critic', mg_sc, 5)
    variants['self_critic']['to_pred'] = sd_bf16(mg_sc.token_critic.to_pred.state_dict())

    run_variant('cond_scale_1', pkg.MaskGit(vae=vae, transformer=tr, image_size=128), 5, cond_scale=1)
    run_variant('can_remask', pkg.MaskGit(vae=vae, transformer=tr, image_size=128, no_mask_token_prob=0.25), 5,
                can_remask_prev_masked=True)

    torch.manual_seed(5)
    trsc = pkg.MaskGitTransformer(t5_name='t5-small', self_cond=True, **dict(tcfg, depth=1))
    with torch.no_grad():
        trsc.to_logits.weight.mul_(8.)
    round_module_to_bf16_(trsc).eval()
    run_variant('self_cond', pkg.MaskGit(vae=vae, transformer=trsc, image_size=128), 4)
    variants['self_cond']['sd'] = sd_bf16(trsc.state_dict())
    torch.save(variants, os.path.join(OUT, 'generate_variants_tiny.pt'))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
