"""TEST-FIXTURE TOOLING (build container, CPU).  Chooses the INPUT seed (text embeddings) of tests/golden/base_c2_fp32.pt.

An 18-step decode makes 2 x 17 top-k selections over 256 confidence scores and ~5800 Gumbel arg-max decisions; with random draws some of them are
near-ties far below what ANY fp32 implementation that is not bit-identical to the reference's BLAS can resolve.  The first seed tried (20260924, the
bf16-weight fixture's) has, on the general-fp32 checkpoint, two confidences 4.2e-6 apart (0.1015) at the re-masking boundary of step 7: the fp32
'parity' engine happened to order them like the reference, both term-product tiers (logits 1e-5 away at this x8 logit scale) swapped them -- a coin
flip, not an accuracy statement (SURVEY 8c determinism control 4 prescribes a tie-aware comparison for exactly this).  Other NOISE seeds do not help: with
the x8 logits the trajectory barely depends on the Gumbel draws and 8 of 9 seeds hit the same pair.  So the fixture's text embeddings are drawn from another
seed instead: this tool free-runs the CPU oracle over candidate input seeds and reports, per seed, the smallest decision margins of the whole trajectory:
  boundary: min over steps / samples of (s_k - s_{k+1}) / (s_k (1 - s_k))      -- the logit perturbation that would swap the re-masking boundary
  argmax  : min over sampled rows of T x the top-1 - top-2 gap of (logit / T + gumbel) among the kept candidates -- the same, for the sampled id
Both are in LOGIT units.  A seed is accepted when both exceed 5e-4: 12x the largest logit error any of the fp32-grade engines shows at this x8 logit
scale (4e-5 on the guidance-combined logits), 100x the fp32 summation-order noise.  python tools/find_golden_input_seed.py 78 8"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import golden_recipe as R  # noqa: E402
import muse_oracle as O  # noqa: E402


def margins(sd, te, verbose=False):
    cfg = dict(depth=8, heads=8)
    counts, temps = O.mask_counts(R.T, R.N), O.step_temperatures(R.T, 1.)
    mask_id = 65536
    ids = torch.full((R.B, R.N), mask_id, dtype=torch.long)
    scores = torch.zeros(R.B, R.N)
    worst_b, worst_a = 1e9, 1e9
    for s, u in enumerate(R.noise_stream()):
        sel = O.select_topk_stable(scores, counts[s])
        ids = torch.where(sel, torch.full_like(ids, mask_id), ids)
        with torch.no_grad():
            logits = O.forward_with_cond_scale(sd, cfg, ids, te, 3.)
        gum = O.gumbel_from_uniform(u)
        filt = O.top_k_filter(logits, 0.9)
        pert = filt / max(temps[s], 1e-10) + gum
        top2 = pert.topk(2, dim=-1).values
        masked = ids == mask_id
        gap_a = (top2[..., 0] - top2[..., 1])[masked]
        if temps[s] > 0:
            worst_a = min(worst_a, gap_a.min().item() * temps[s])      # in LOGIT units: an error d of a logit moves the perturbed value by d / T
        else:                       # last step: pure arg-max of the logits (division by 1e-10: compare the logits themselves)
            t2 = filt.topk(2, dim=-1).values
            worst_a = min(worst_a, (t2[..., 0] - t2[..., 1])[masked].min().item())
        ids, scores, _ = O.sample_step(logits, gum, ids, mask_id, temps[s])
        if s < R.T - 1:
            k = counts[s + 1]
            srt = torch.sort(scores, dim=-1, descending=True).values
            a, b = srt[:, k - 1], srt[:, k]
            rel = ((a - b) / (a * (1 - a)).clamp_min(1e-30)).min().item()
            worst_b = min(worst_b, rel)
            if verbose:
                print(f'  step {s}: boundary margin {rel:.3g}, arg-max margin so far {worst_a:.3g}', flush=True)
    return worst_b, worst_a


def main():
    import muse_maskgit_pytorch_amd as mm
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 78
    tries = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    tr = R.build_transformer(mm.MaskGitTransformer, peaky=True, bf16_weights=False)
    sd = {k: v.detach() for k, v in tr.state_dict().items()}
    for seed in range(first, first + tries):
        t0 = time.time()
        b, a = margins(sd, R.inputs(seed)['text_embeds'])
        ok = b >= 5e-4 and a >= 5e-4
        print(f'input seed {seed}: boundary margin {b:.3g}, arg-max margin {a:.3g} -> {"ACCEPT" if ok else "reject"}  ({time.time() - t0:.0f}s)', flush=True)
        if ok:
            break


if __name__ == '__main__':
    main()
