"""TEST INFRASTRUCTURE.  The tie-aware comparison of a decode run with the reference's run (determinism control 4:
"mask-index sets must match after removing positions whose score equals the k-th score"), generalised to both kinds of discrete
decision MaskGit.generate makes (muse_maskgit_pytorch.py:556-609):

  * re-masking (:561): `scores.topk(k)` -- position i is AT THE BOUNDARY of step s when a logit perturbation of at most `eps` could move its
    confidence score across the k-th / (k+1)-th largest one.  A score is 1 - softmax probability p of the sampled id, so a logit error d moves
    it by at most d * p (1 - p): the band is  |score_i - boundary| <= eps * s (1 - s)  around both boundary scores (the margin convention of
    tools/find_golden_input_seed.py);
  * sampling (:576-580): `argmax(filtered_logits / T + gumbel)` -- position i is a NEAR-TIE of step s when the reference's own top-1 / top-2
    gap of the perturbed logits, in logit units, is below `eps` (recorded by oracle/make_golden_base.py as `argmax_margin`).

Everything is judged on the REFERENCE's recorded values (scores entering every top-k, sampled ids, arg-max margins: the fixture's
`generate.scores_in / pred_ids / argmax_margin`), never on the candidate's: a candidate cannot excuse itself.

Two comparisons:
  compare_free_run   -- a free-running trace (the ids entering every step + the final ids).  Per sample: equal to the reference step by step, or
                        the FIRST difference is confined to positions inside a tie band of the step that produced it; behind such a divergence the
                        two runs legitimately follow different trajectories and the sample is reported as 'diverged at a tie' (not compared further).
  compare_forced_step -- one teacher-forced step (the candidate sampled from the reference's own state with the reference's noise): sampled ids
                        equal outside the sampling near-ties, next mask set equal outside the boundary band.  Run over every step it covers the
                        whole trajectory, including what lies behind a tie.
"""
import torch

MASK_FILL = -1e5      # mmp.py:609: the score of a position that is not masked (never re-masked)


def boundary_band(scores, k, eps):
    """scores (n,) entering `scores.topk(k)`; returns a bool (n,) of the positions a logit perturbation <= eps could move across the top-k boundary."""
    n = scores.numel()
    if k >= n or k <= 0:
        return torch.zeros(n, dtype=torch.bool)
    srt = torch.sort(scores, descending=True).values
    a, b = srt[k - 1].item(), srt[k].item()                 # the last score selected, the first one left out
    if b <= MASK_FILL / 2:                                  # everything that can be re-masked is selected: no boundary among live scores
        return torch.zeros(n, dtype=torch.bool)

    def width(s):
        s = min(max(s, 0.), 1.)
        return eps * s * (1. - s)

    # a selected position (score >= a) is in the band when a perturbation could bring it down to the best score left out (b, itself perturbed upwards);
    # a position left out (score <= b) when it could reach the last selected score (a, perturbed downwards)
    sc = scores.double()
    reach = eps * sc.clamp(0, 1) * (1 - sc.clamp(0, 1))
    down = (sc >= a) & (sc - reach <= b + width(b))
    up = (sc <= b) & (sc + reach >= a - width(a))
    return (down | up) & (scores > MASK_FILL / 2)


def compare_free_run(ref, got_step_in, got_final, counts, mask_id, eps):
    """ref: the fixture's `generate` dict; got_step_in [T,B,n] ids entering every step (after the re-mask scatter), got_final [B,n].
    Returns a list of per-sample dicts {status: 'equal' | 'tie', step, kind, positions}; raises AssertionError on an unexplained difference."""
    T, B, n = ref['step_in_ids'].shape
    ref_in = ref['step_in_ids'].long()
    ref_final = ref['final_ids'].reshape(B, n).long()
    got_step_in = got_step_in.long().cpu()
    got_final = got_final.reshape(B, n).long().cpu()
    out = []
    for b in range(B):
        status = dict(status='equal', step=None, kind=None, positions=[])
        for s in range(T + 1):
            r = ref_in[s, b] if s < T else ref_final[b]
            g = got_step_in[s, b] if s < T else got_final[b]
            diff = (r != g).nonzero().flatten()
            if diff.numel() == 0:
                continue
            # what decided the state entering step s: the sampling of step s - 1 (ids) and the top-k of step s (which of them are masked again)
            assert s > 0, f'sample {b}: the initial state differs from the reference'
            near = ref['argmax_margin'][s - 1, b] < eps                              # sampling near-ties of the step that produced these ids
            was_masked = ref_in[s - 1, b] == mask_id
            near = near & was_masked
            band = boundary_band(ref['scores_in'][s, b], counts[s], eps) if s < T else torch.zeros(n, dtype=torch.bool)
            ok = near | band
            bad = [int(i) for i in diff if not ok[i]]
            assert not bad, (f'sample {b}: state entering step {s} differs from the reference at positions {bad[:8]} that are neither sampling '
                             f'near-ties nor at the re-masking boundary (eps = {eps:g} logit units)')
            status = dict(status='tie', step=s, kind='boundary' if band[diff].any() else 'sampling', positions=[int(i) for i in diff])
            break
        out.append(status)
    return out


def compare_forced_step(ref, s, new_ids, new_scores, counts, mask_id, eps, select_topk):
    """One teacher-forced step: the candidate ran step s on the reference's state `step_in_ids[s]` with the reference's noise and returned
    `new_ids` [B,n] (ids after torch.where, :584-588) and `new_scores` [B,n] (:603-609).  `select_topk(scores, k)` -> bool mask of the k
    re-masked positions (the oracle's stable top-k).  Returns (number of sampling near-ties skipped, number of boundary positions skipped)."""
    T, B, n = ref['step_in_ids'].shape
    ref_in = ref['step_in_ids'][s].long()
    masked = ref_in == mask_id
    ref_new = torch.where(masked, ref['pred_ids'][s].long(), ref_in)
    near = (ref['argmax_margin'][s] < eps) & masked
    new_ids = new_ids.long().cpu()
    wrong = (new_ids != ref_new) & ~near
    assert not wrong.any(), f'step {s}: {int(wrong.sum())} sampled ids differ from the reference outside its near-ties (eps = {eps:g})'
    skipped_band = 0
    if s + 1 < T:
        ref_next_masked = ref['step_in_ids'][s + 1].long() == mask_id
        sel = select_topk(new_scores.float().cpu(), counts[s + 1])
        for b in range(B):
            band = boundary_band(ref['scores_in'][s + 1, b], counts[s + 1], eps)
            # a sampling near-tie changes that position's own confidence: it may legitimately move across the boundary as well
            free = band | near[b]
            d = (sel[b] != ref_next_masked[b]) & ~free
            assert not d.any(), f'step {s}, sample {b}: re-masked set differs from the reference at {d.nonzero().flatten().tolist()[:8]} outside the boundary band'
            # outside the excused positions the two sets have the same size on both sides
            skipped_band += int(band.sum())
    return int(near.sum()), skipped_band


# ------------------------------------------------------------------------------------------------ the smallest eps that explains a run (round 6)
def _passes(fn, eps):
    try:
        fn(eps)
        return True
    except AssertionError:
        return False


def min_eps(fn, lo=1e-7, hi=64., rel=0.02):
    """`fn(eps)` raises AssertionError when eps does not explain the run; bands only grow with eps, so the set of passing eps is an interval [eps*, inf).
    Returns eps* to `rel` relative accuracy by geometric bisection (0. when even `lo` passes, None when `hi` does not)."""
    if _passes(fn, lo):
        return 0.
    if not _passes(fn, hi):
        return None
    while hi / lo > 1. + rel:
        mid = (lo * hi) ** 0.5
        if _passes(fn, mid):
            hi = mid
        else:
            lo = mid
    return hi


def band_population(ref, counts, mask_id, eps):
    """how much of the reference's run lies inside a tie band at `eps` -- what a contract at that eps EXCUSES: (sampling near-ties / sampled positions,
    boundary-band positions / live positions at the re-masking steps)"""
    T, B, n = ref['step_in_ids'].shape
    near = sampled = band = live = 0
    for s in range(T):
        masked = ref['step_in_ids'][s].long() == mask_id
        near += int(((ref['argmax_margin'][s] < eps) & masked).sum())
        sampled += int(masked.sum())
        if s + 1 < T:
            for b in range(B):
                band += int(boundary_band(ref['scores_in'][s + 1, b], counts[s + 1], eps).sum())
                live += int((ref['scores_in'][s + 1, b] > MASK_FILL / 2).sum())
    return near / max(sampled, 1), band / max(live, 1)
