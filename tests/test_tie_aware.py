"""The tie-aware comparator (tests/tie_aware.py, SURVEY 8c determinism control 4) on synthetic cases and on the fixture it exists for:
tests/golden/base_c2_fp32_s77.pt -- the general-fp32 checkpoint on the UN-SCANNED input seed, whose reference run has two confidences 2.3e-6 apart
at the re-masking boundary entering step 8 (oracle/make_golden_base.py --fp32 --unscanned)."""
import pytest
import torch

import golden_recipe as R
import muse_oracle as O
import tie_aware as TA

EPS = 5e-4      # logit units: 12x the largest logit error of the fp32-grade engines at the fixtures' x8 logit scale (tools/find_golden_input_seed.py)
MASK_ID = 65536


def test_boundary_band_synthetic():
    s = torch.tensor([0.9, 0.5, 0.500001, 0.2, 0.1, TA.MASK_FILL])
    # k = 2: selected {0.9, 0.500001}, first left out 0.5: the pair 1e-6 apart is the band, nothing else
    assert TA.boundary_band(s, 2, 1e-3).tolist() == [False, True, True, False, False, False]
    # a clear gap (0.5 vs 0.2): no band even though both are "the boundary scores"
    assert not TA.boundary_band(s, 3, 1e-3).any()
    # exact duplicates straddling the boundary are always in the band
    d = torch.tensor([0.7, 0.3, 0.3, 0.3, 0.1])
    assert TA.boundary_band(d, 2, 0.).tolist() == [False, True, True, True, False]
    # the k-th score is a live one and everything behind it is the "not masked" fill: no boundary
    f = torch.tensor([0.7, 0.3, TA.MASK_FILL, TA.MASK_FILL])
    assert not TA.boundary_band(f, 2, 1e-3).any()
    # the band scales with p (1 - p): the same 1e-4 gap is a tie at s = 0.5 and not at s = 0.999 (a logit error barely moves a saturated probability)
    assert TA.boundary_band(torch.tensor([0.5001, 0.5, 0.0]), 1, 1e-3).tolist() == [True, True, False]
    assert not TA.boundary_band(torch.tensor([0.9991, 0.999, 0.0]), 1, 1e-3).any()


def _gen(golden):
    return golden('base_c2_fp32_s77.pt')['generate']


def test_unscanned_fixture_has_the_known_near_tie(golden):
    """the fixture is the hard case on purpose: its reference run has a boundary pair inside the band at the step-7 -> 8 re-masking, and nowhere else"""
    gen = _gen(golden)
    counts = O.mask_counts(R.T, R.N)
    hits = {(s, b): int(TA.boundary_band(gen['scores_in'][s, b], counts[s], EPS).sum()) for s in range(R.T) for b in range(R.B)}
    hits = {k: v for k, v in hits.items() if v}
    print(f'[tie-aware] boundary-band positions of the un-scanned fp32 fixture at eps {EPS:g}: {hits}')
    assert (8, 0) in hits and hits[(8, 0)] == 2
    masked = gen['step_in_ids'].long() == MASK_ID
    assert int(((gen['argmax_margin'] < EPS) & masked).sum()) == 0      # its sampling decisions are all clear of eps


def test_free_run_comparator_on_the_fixture(golden):
    gen = _gen(golden)
    counts = O.mask_counts(R.T, R.N)
    ref_in, final = gen['step_in_ids'].long(), gen['final_ids'].reshape(R.B, R.N).long()
    # the reference against itself
    assert [r['status'] for r in TA.compare_free_run(gen, ref_in, final, counts, MASK_ID, EPS)] == ['equal', 'equal']
    # the tie resolved the other way: the two boundary positions of (step 8, sample 0) swap roles, the run then goes elsewhere
    band = TA.boundary_band(gen['scores_in'][8, 0], counts[8], EPS).nonzero().flatten().tolist()
    alt = ref_in.clone()
    i, j = band
    if alt[8, 0, i] != MASK_ID:
        i, j = j, i
    prev = torch.where(ref_in[7, 0] == MASK_ID, gen['pred_ids'][7, 0].long(), ref_in[7, 0])      # ids after step 7
    alt[8, 0, i], alt[8, 0, j] = prev[i], MASK_ID
    alt[9:, 0] = torch.randint(0, 65536, alt[9:, 0].shape)                                      # behind a tie the trajectories are unrelated
    rep = TA.compare_free_run(gen, alt, final, counts, MASK_ID, EPS)
    assert rep[0]['status'] == 'tie' and rep[0]['step'] == 8 and rep[0]['kind'] == 'boundary' and sorted(rep[0]['positions']) == sorted(band)
    assert rep[1]['status'] == 'equal'
    # the same swap at a step whose boundary is clear is NOT excused
    bad = ref_in.clone()
    s = 12
    m = (bad[s, 1] == MASK_ID).nonzero().flatten()
    u = (bad[s, 1] != MASK_ID).nonzero().flatten()
    prev = torch.where(ref_in[s - 1, 1] == MASK_ID, gen['pred_ids'][s - 1, 1].long(), ref_in[s - 1, 1])
    bad[s, 1, m[0]], bad[s, 1, u[0]] = prev[m[0]], MASK_ID
    with pytest.raises(AssertionError):
        TA.compare_free_run(gen, bad, final, counts, MASK_ID, EPS)
    # a wrong sampled id is not excused either
    bad2 = ref_in.clone()
    p = (bad2[5, 0] != MASK_ID).nonzero().flatten()[0]
    bad2[5:, 0, p] = (bad2[5, 0, p] + 1) % 65536
    with pytest.raises(AssertionError):
        TA.compare_free_run(gen, bad2, final, counts, MASK_ID, EPS)


def test_forced_step_comparator_with_the_oracle_at_the_tie(golden):
    """teacher-forced step 7 of the un-scanned fixture through the CPU oracle (the step whose confidences carry the near-tie): sampled ids bit-equal,
    next mask set equal outside the 2-position band -- whichever way the oracle's own summation order happens to resolve the pair"""
    import muse_maskgit_pytorch_amd as mm
    g = golden('base_c2_fp32_s77.pt')
    gen = g['generate']
    tr = R.build_transformer(mm.MaskGitTransformer, peaky=True, bf16_weights=False)
    assert R.state_checksum(tr) == g['weight_checksum_peaky']
    sd = {k: v.detach().clone() for k, v in tr.state_dict().items()}
    inp = R.inputs(g['recipe']['input_seed'])
    assert g['recipe']['input_seed'] == R.INPUT_SEED
    counts, temps = O.mask_counts(R.T, R.N), O.step_temperatures(R.T, 1.)
    for s, u in enumerate(R.noise_stream()):
        if s < 7:
            continue
        ids_in = gen['step_in_ids'][s].long()
        with torch.no_grad():
            logits = O.forward_with_cond_scale(sd, dict(depth=8, heads=8), ids_in, inp['text_embeds'], 3.)
        new_ids, scores, _ = O.sample_step(logits, O.gumbel_from_uniform(u), ids_in, MASK_ID, temps[s])
        near, band = TA.compare_forced_step(gen, s, new_ids, scores, counts, MASK_ID, EPS, O.select_topk_stable)
        assert near == 0 and band == 2
        break


def test_min_eps_finds_the_smallest_explaining_band(golden):
    """tie_aware.min_eps (round 6: the bf16 engine's id contract): bands only grow with eps, so the smallest eps that explains a run is found by bisection.
    On the un-scanned fixture: swapping the two boundary positions of (step 8, sample 0) is explained from eps ~ 2.3e-6 / (s (1 - s)) on -- far below 5e-4 --
    the reference against itself needs eps 0, and a wrong id at a position whose reference margin is m needs eps just above m (never less)."""
    gen = _gen(golden)
    counts = O.mask_counts(R.T, R.N)
    ref_in, final = gen['step_in_ids'].long(), gen['final_ids'].reshape(R.B, R.N).long()
    assert TA.min_eps(lambda e: TA.compare_free_run(gen, ref_in, final, counts, MASK_ID, e)) == 0.
    band = TA.boundary_band(gen['scores_in'][8, 0], counts[8], EPS).nonzero().flatten().tolist()
    alt = ref_in.clone()
    i, j = band
    if alt[8, 0, i] != MASK_ID:
        i, j = j, i
    prev = torch.where(ref_in[7, 0] == MASK_ID, gen['pred_ids'][7, 0].long(), ref_in[7, 0])
    alt[8, 0, i], alt[8, 0, j] = prev[i], MASK_ID
    e_swap = TA.min_eps(lambda e: TA.compare_free_run(gen, alt, final, counts, MASK_ID, e))
    assert 0. < e_swap < EPS, e_swap
    # a wrong sampled id at a masked position of step 3: explained only by a band at least as wide as that position's own arg-max margin
    s = 3
    masked = (ref_in[s, 1] == MASK_ID).nonzero().flatten()
    pos = int(masked[gen['argmax_margin'][s, 1][masked].argmin()])
    m = float(gen['argmax_margin'][s, 1, pos])
    new_ids = torch.where(ref_in[s] == MASK_ID, gen['pred_ids'][s].long(), ref_in[s])
    new_ids[1, pos] = (new_ids[1, pos] + 1) % 65536
    scores = gen['scores_in'][s + 1].clone()

    def forced(e):
        TA.compare_forced_step(gen, s, new_ids, scores, counts, MASK_ID, e, O.select_topk_stable)

    e_id = TA.min_eps(forced)
    assert e_id is not None and m < e_id <= m * 1.03 + 1e-7, (m, e_id)
    near, bandfrac = TA.band_population(gen, counts, MASK_ID, e_id)
    assert 0. < near <= 1. and 0. <= bandfrac <= 1.
    assert TA.band_population(gen, counts, MASK_ID, 1e-9)[0] == 0.      # no sampling decision of this fixture is an exact tie
