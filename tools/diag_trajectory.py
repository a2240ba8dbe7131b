"""diagnostic (round 4): where does a precision tier leave the reference trajectory on base_c2_fp32.pt?  Writes per-step agreement, the first
deviating step's positions, and the tier's ids / scores there to gpurun_out/diag/diag_<tier>.pt"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import golden_recipe as R
import muse_maskgit_pytorch_amd as mm
g = torch.load(os.path.join(ROOT, 'tests', 'golden', 'base_c2_fp32.pt'), weights_only=False)
tr = R.build_transformer(mm.MaskGitTransformer, peaky=True, bf16_weights=False)
mg = mm.MaskGit(vae=None, transformer=tr, image_size=256).to('cuda').eval()
inp = R.inputs(g['recipe'].get('input_seed'))
te = inp['text_embeds'].cuda()
u = torch.stack(list(R.noise_stream())).cuda()
gen = g['generate']
ref_in = gen['step_in_ids'].long()
os.makedirs(os.path.join(ROOT, 'gpurun_out', 'diag'), exist_ok=True)
for tier, kw in (('f16x2', {}), ('f16x2', dict(fused_sampling=False)), ('bf16x3', {}), ('parity', {})):
    mg.set_precision(tier)
    trace = {}
    ids = mg.generate(['a', 'b'], timesteps=R.T, cond_scale=3., text_embeds=te, noise=u, noise_kind='uniform', return_ids=True, trace=trace, fmap_size=16, **kw)
    masked = torch.stack(list(trace['masked_ids'])).cpu() if isinstance(trace['masked_ids'], list) else trace['masked_ids'].cpu()
    after = torch.stack(list(trace['ids'])).cpu() if isinstance(trace['ids'], list) else trace['ids'].cpu()
    scores = torch.stack(list(trace['scores'])).cpu() if isinstance(trace['scores'], list) else trace['scores'].cpu()
    agree = [(masked[s] == ref_in[s]).float().mean().item() for s in range(R.T)]
    first = next((s for s in range(R.T) if agree[s] < 1.0), None)
    print(tier, kw, 'agree per step', [f'{a:.4f}' for a in agree], 'first deviating input state: step', first, 'fallback rows', mg.fused_row_fallbacks, flush=True)
    torch.save(dict(agree=agree, first=first, masked=masked.to(torch.int32), after=after.to(torch.int32), scores=scores), os.path.join(ROOT, 'gpurun_out', 'diag', f'diag_{tier}_{int(bool(kw))}.pt'))
