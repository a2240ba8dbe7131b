"""Sampling without the logits round trip (csrc/sampling_fused.hip + the emission in gemm_cfg.hip) against the logits path
(mm_gemm_cfg_logits + mm_sample_rows, itself bit-exact against the oracle: tests/test_gpu_ops.py): same predicted ids AND bit-identical
confidences on the same logits, for every noise mode; the GEMM's emission equals the emission computed from its materialised logits; rows
whose candidate set cannot be proven complete raise the flag instead of returning a wrong id."""
import math

import pytest
import torch

import muse_maskgit_pytorch_amd as mm
from muse_maskgit_pytorch_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _both(logits, k_keep, T, **kw):
    R, V = logits.shape
    ref_pred, ref_score = ops.sample_rows(logits, k_keep, T, **kw)
    fb = ops.fused_buffers(R, V, DEV)
    z = ops.fused_z(k_keep, V)
    thr = (logits.mean(dim=1) + z * logits.std(dim=1)).contiguous()
    ops.fused_emit(logits, thr, fb)
    pred, score = ops.fused_sample(fb, thr, R, V, k_keep, T, **kw)
    return ref_pred, ref_score, pred, score, fb


@pytest.mark.parametrize('V,R', [(65536, 37), (8192, 64), (1024, 5), (256, 3)])
@pytest.mark.parametrize('mode', ['philox', 'uniform', 'none'])
def test_fused_sample_equals_sample_rows(V, R, mode):
    g = torch.Generator().manual_seed(V + R)
    logits = (torch.randn(R, V, generator=g) * 2.5 + torch.randn(R, 1, generator=g)).to(DEV)
    k_keep = math.ceil(0.1 * V)
    kw = {}
    if mode == 'philox':
        kw = dict(noise_kind=_lib.MM_NOISE_PHILOX, seed=1234, row_offset=77, step=5)
    elif mode == 'uniform':
        kw = dict(noise_kind=_lib.MM_NOISE_UNIFORM, noise=torch.rand(R, V, generator=g).to(DEV))
    for T in (1.0, 0.5, 1e-10):
        ref_pred, ref_score, pred, score, fb = _both(logits, k_keep, T, **kw)
        assert int(fb['fail'].item()) == 0
        assert torch.equal(pred, ref_pred), f'T={T}: {(pred != ref_pred).sum().item()} of {R} ids differ'
        # the softmax denominator is combined from the same per-tile statistics in the same order on both paths (common.h tile_softmax_stats)
        assert torch.equal(score, ref_score), f'T={T}: confidences differ by {(score - ref_score).abs().max().item():.3g}'


def test_fused_sample_hot_tile_and_failure_paths():
    """a tile whose 256 entries all pass the bound fills its slot completely (it cannot overflow), still exact; a heavy-tailed row whose
    bound estimate keeps fewer than k entries raises the flag (the caller repeats on the logits path)"""
    g = torch.Generator().manual_seed(9)
    R, V = 6, 65536
    logits = torch.randn(R, V, generator=g)
    logits[:, 512:768] += 6.0                                        # one hot tile: 256 candidates in a 64-entry slot
    logits = logits.to(DEV)
    k_keep = math.ceil(0.1 * V)
    ref_pred, ref_score, pred, score, fb = _both(logits, k_keep, 1.0, noise_kind=_lib.MM_NOISE_PHILOX, seed=5)
    assert int(fb['fail'].item()) == 0
    assert torch.equal(pred, ref_pred) and torch.equal(score, ref_score)
    # heavy tail: a few huge outliers inflate sigma, the Gaussian bound lands far above the true 90th percentile
    lt = torch.randn(4, V, generator=g)
    lt[:, :40] = 4000.
    lt = lt.to(DEV)
    fb = ops.fused_buffers(4, V, DEV)
    thr = (lt.mean(dim=1) + ops.fused_z(k_keep, V) * lt.std(dim=1)).contiguous()
    ops.fused_emit(lt, thr, fb)
    ops.fused_sample(fb, thr, 4, V, k_keep, 1.0)
    assert int(fb['fail'].item()) == 1


@pytest.mark.parametrize('pattern', ['two_of_three', 'upper_half', 'three_of_four', 'all_but_first_quarter'])
def test_fused_sample_crowded_pieces_keep_their_columns(pattern):
    """pieces with 65 .. 128 kept granules (the finisher reads a slot densely, 64 entries per round, and recovers every entry's granule number
    from the piece's keep mask: the second round takes its granule numbers from the wrapped half of the second lane permutation).  With
    T -> 0 the predicted id IS the row's largest logit, which is planted in the crowded piece behind its 64th kept granule; with T = 1 / 0.5
    the Gumbel winner lies in the crowded piece for most rows"""
    g = torch.Generator().manual_seed(21)
    R, V = 9, 65536
    base = torch.randn(R, V, generator=g)
    cols = torch.arange(256)
    hot = {'two_of_three': cols % 3 != 0, 'upper_half': cols >= 100, 'three_of_four': cols % 4 != 1, 'all_but_first_quarter': cols >= 64}[pattern]
    hot_cols = cols[hot]
    planted_logits = None
    planted = []
    for r in range(R):
        piece = 3 + 7 * r
        base[r, piece * 256:(piece + 1) * 256][hot] += 7.0
    planted_logits = base.clone()
    for r in range(R):
        piece = 3 + 7 * r
        c = int(hot_cols[len(hot_cols) - 1 - 3 * r]) + piece * 256      # behind the 64th kept granule of the piece
        planted_logits[r, c] = 40.0 + r
        planted.append(c)
    k_keep = math.ceil(0.1 * V)
    ref_pred, ref_score, pred, score, fb = _both(planted_logits.to(DEV), k_keep, 1e-10)
    assert int(fb['fail'].item()) == 0
    assert torch.equal(pred, ref_pred) and torch.equal(score, ref_score)
    assert pred.flatten().tolist() == planted
    in_piece = 0
    for T, kw in ((1.0, dict(noise_kind=_lib.MM_NOISE_PHILOX, seed=11, step=2)), (0.5, dict(noise_kind=_lib.MM_NOISE_PHILOX, seed=12, row_offset=5))):
        ref_pred, ref_score, pred, score, fb = _both(base.to(DEV), k_keep, T, **kw)
        assert int(fb['fail'].item()) == 0
        assert torch.equal(pred, ref_pred) and torch.equal(score, ref_score)
        in_piece += sum(int(pc) // 256 == 3 + 7 * r for r, pc in enumerate(pred.flatten().tolist()))
    assert in_piece >= R            # the crowded pieces do carry the winners (otherwise this test shows nothing)


def _valid_slots(stats):
    """which of the 128 float2 entries of every (row, piece) slot hold a candidate, as a mask over the slot viewed as [.., 64, 4] floats: the slot is one
    compacted list, as long as the record's 128-bit mask has bits set (common.h fs_pos)"""
    m = stats[..., 4:].contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF           # [R, NT, 4]: the four quarters' mask words
    cnt = sum(((m >> b) & 1) for b in range(32)).sum(-1)                                      # kept granules of the piece
    pos = torch.arange(2 * ops.FUSED_SLOT, device=stats.device)                               # float2 entry index
    ok2 = pos[None, None, :] < cnt[..., None]                                                 # [R, NT, 128]
    return ok2.reshape(*ok2.shape[:2], ops.FUSED_SLOT, 2).repeat_interleave(2, dim=-1)        # [R, NT, 64, 4]: two floats per entry


@pytest.mark.parametrize("M", [4608, 5140, 129, 135, 391, 3])      # 135 = B 1 x k 135 (edge tiles with 1..7 rows; waves without a piece must not count a statistics store)
def test_gemm_emission_equals_emission_from_its_logits(M):
    """the guidance-logits GEMM with the fused epilogue emits exactly what fused_emit computes from the logits the plain GEMM writes;
    and the threshold estimated from the embeddings + vocabulary statistics keeps >= k candidates per row"""
    torch.manual_seed(M)
    V, D = 65536, 512
    W = (torch.randn(V, D) * (D ** -0.5)).to(torch.bfloat16).to(DEV)
    ec = torch.randn(M, D).to(torch.bfloat16).to(DEV)
    en = torch.randn(M, D).to(torch.bfloat16).to(DEV)
    s = 3.0
    Wf = W.float()
    wmean = Wf.mean(dim=0).contiguous()
    wcov = ((Wf.t() @ Wf) / V - torch.outer(wmean, wmean)).to(torch.bfloat16).contiguous()
    k_keep = math.ceil(0.1 * V)
    z = ops.fused_z(k_keep, V)
    thr = ops.fused_threshold(ec, en, s, wmean, wcov, z)
    logits = ops.gemm_cfg_logits(ec, en, W, s)
    direct = logits.mean(dim=1) + z * logits.std(dim=1)
    assert (thr - direct).abs().max().item() < 0.02 * logits.std().item()
    fa, fb = ops.fused_buffers(M, V, DEV), ops.fused_buffers(M, V, DEV)
    ops.fused_emit(logits, thr, fa)
    ops.gemm_cfg_logits_fused(ec, en, W, s, thr, fb)
    assert torch.equal(fa['stats'].view(torch.int32), fb['stats'].view(torch.int32))
    valid = _valid_slots(fa['stats'])
    assert torch.equal(fa['cand'][valid].view(torch.int32), fb['cand'][valid].view(torch.int32))
    total = (logits >= thr[:, None]).sum(dim=1)
    assert int(total.min().item()) >= k_keep and int(total.max().item()) < 11264
    # the single-pass form mm_generate runs (round 3): the embeddings are mixed first (mm_cfg_mix) and multiplied once on the 128 x 256 kernel's
    # third instantiation; its logits (any dense kernel of the family) and its emission must agree with each other the same way
    em = ops.cfg_mix(ec, en, s, D)
    lm = ops.gemm(em, W, out_f32=True)
    assert (lm - logits).abs().max().item() < 0.08 * logits.std().item()          # same quantity, bf16 rounding of the mix instead of the two operands
    thr_m = ops.fused_threshold(em, em, 1.0, wmean, wcov, z)
    fm, fn = ops.fused_buffers(M, V, DEV), ops.fused_buffers(M, V, DEV)
    ops.fused_emit(lm.contiguous(), thr_m, fm)
    ops.gemm_cfg_logits_fused(em, None, W, 1.0, thr_m, fn)
    assert torch.equal(fm['stats'].view(torch.int32), fn['stats'].view(torch.int32))
    valid_m = _valid_slots(fm['stats'])
    assert torch.equal(fm['cand'][valid_m].view(torch.int32), fn['cand'][valid_m].view(torch.int32))
    pm, sm = ops.fused_sample(fn, thr_m, M, V, k_keep, 1.0, noise_kind=_lib.MM_NOISE_PHILOX, seed=3)
    rp_, rs_ = ops.sample_rows(lm.contiguous(), k_keep, 1.0, noise_kind=_lib.MM_NOISE_PHILOX, seed=3)
    assert int(fn['fail'].item()) == 0 and torch.equal(pm, rp_) and torch.equal(sm, rs_)
    pa, sa = ops.fused_sample(fa, thr, M, V, k_keep, 1.0, noise_kind=_lib.MM_NOISE_PHILOX, seed=3)
    pb, sb = ops.fused_sample(fb, thr, M, V, k_keep, 1.0, noise_kind=_lib.MM_NOISE_PHILOX, seed=3)
    ref_pred, ref_score = ops.sample_rows(logits, k_keep, 1.0, noise_kind=_lib.MM_NOISE_PHILOX, seed=3)
    assert int(fa['fail'].item()) == 0 and int(fb['fail'].item()) == 0
    assert torch.equal(pa, ref_pred) and torch.equal(pb, ref_pred) and torch.equal(sa, sb)


@pytest.mark.parametrize('M,V,D', [(2500, 8192, 1024), (1100, 8192, 512), (1024, 65536, 576), (3000, 8192, 640)])
def test_single_pass_emission_on_other_shapes(M, V, D):
    """the single-pass logits GEMM with the accumulator emission (gemm_wide_fused_kernel where K % 128 == 0 and M >= 1024, gemm_cfg2 otherwise) at the
    paper-scale shape (V = 8192, D = 1024: 16 k-steps, 32 column tiles), with fewer 256-row tiles than CUs (one tile per workgroup), with K % 128 != 0
    (D = 576 -> the older kernel) and with ten k-steps (D = 640): statistics and candidates equal the emission computed from the materialised logits, the sampled ids / scores equal sample_rows"""
    torch.manual_seed(M + V)
    W = (torch.randn(V, D) * (D ** -0.5)).to(torch.bfloat16).to(DEV)
    em = torch.randn(M, D).to(torch.bfloat16).to(DEV)
    Wf = W.float()
    wmean = Wf.mean(dim=0).contiguous()
    wcov = ((Wf.t() @ Wf) / V - torch.outer(wmean, wmean)).to(torch.bfloat16).contiguous()
    k_keep = math.ceil(0.1 * V)
    thr = ops.fused_threshold(em, em, 1.0, wmean, wcov, ops.fused_z(k_keep, V))
    lm = ops.gemm(em, W, out_f32=True).contiguous()
    fm, fn = ops.fused_buffers(M, V, DEV), ops.fused_buffers(M, V, DEV)
    ops.fused_emit(lm, thr, fm)
    for rep in range(2):
        ops.gemm_cfg_logits_fused(em, None, W, 1.0, thr, fn)
        assert torch.equal(fm['stats'].view(torch.int32), fn['stats'].view(torch.int32))
        valid = _valid_slots(fm['stats'])
        assert torch.equal(fm['cand'][valid].view(torch.int32), fn['cand'][valid].view(torch.int32))
    pm, sm = ops.fused_sample(fn, thr, M, V, k_keep, 1.0, noise_kind=_lib.MM_NOISE_PHILOX, seed=3)
    rp_, rs_ = ops.sample_rows(lm, k_keep, 1.0, noise_kind=_lib.MM_NOISE_PHILOX, seed=3)
    assert int(fn['fail'].item()) == 0 and torch.equal(pm, rp_) and torch.equal(sm, rs_)


def test_generate_with_fused_sampling_at_bench_size_and_its_fallback():
    """mm_generate at B = 32: the fused path is taken (no fallback on Gaussian-like logits), is repeatable, and a model whose logits defeat
    the bound (a few enormous to_logits rows) falls back to the logits path and returns that path's ids"""
    import bench
    mg, _ = bench.build_models(DEV)
    tr = mg.transformer
    with torch.no_grad():
        tr.to_logits.weight.mul_(8.)          # well-separated confidences: the two paths' last-bit score differences cannot reorder the re-masking
    te = bench.synth_text(32, 32, 512).to(DEV)
    ta, tb = {}, {}
    a = mg.generate([''] * 32, timesteps=18, cond_scale=3, text_embeds=te, seed=7, return_ids=True, trace=ta)
    assert mg.fused_sampling_fallbacks == 0 and tr._model().fused_ready
    b = mg.generate([''] * 32, timesteps=18, cond_scale=3, text_embeds=te, seed=7, return_ids=True, fused_sampling=False, trace=tb)
    # Same logits, same noise, and -- since round 3 -- the same per-tile softmax statistics combined in the same order on both paths: ids AND
    # confidences are bit-identical at every step, so the trajectories cannot part (the ids no longer depend on which path ran).
    for s_ in range(18):
        assert torch.equal(ta['masked_ids'][s_], tb['masked_ids'][s_]) and torch.equal(ta['ids'][s_], tb['ids'][s_]), f'step {s_}'
        assert torch.equal(ta['scores'][s_], tb['scores'][s_]), f'step {s_}: confidences differ by {(ta["scores"][s_] - tb["scores"][s_]).abs().max().item():.3g}'
    assert torch.equal(a, b)
    assert torch.equal(a, mg.generate([''] * 32, timesteps=18, cond_scale=3, text_embeds=te, seed=7, return_ids=True))
    # heavy-tailed vocabulary: 30 huge rows inflate sigma, the Gaussian bound keeps fewer than k entries -> flag.  fused_bound 'gaussian' (rounds 2-4): the call
    # is repeated on the logits path.  'auto' (default): the packed model moves to the sampled bound and the call is repeated with THAT -- no logits path, and the
    # next calls start there.  The ids are the logits path's either way.
    with torch.no_grad():
        tr.to_logits.weight[:30].mul_(400.)
    d = mg.generate([''] * 32, timesteps=6, cond_scale=3, text_embeds=te, seed=7, return_ids=True, fused_sampling=False)
    tr.fused_bound = 'gaussian'
    c = mg.generate([''] * 32, timesteps=6, cond_scale=3, text_embeds=te, seed=7, return_ids=True)
    assert mg.fused_sampling_fallbacks == 1 and torch.equal(c, d)
    tr.fused_bound = 'auto'
    assert tr._model().auto_bound == 'gaussian' and mg.fused_bound_switches == 0
    c = mg.generate([''] * 32, timesteps=6, cond_scale=3, text_embeds=te, seed=7, return_ids=True)
    assert mg.fused_sampling_fallbacks == 1 and mg.fused_bound_switches == 1 and tr._model().auto_bound == 'quantile' and torch.equal(c, d)
    c = mg.generate([''] * 32, timesteps=6, cond_scale=3, text_embeds=te, seed=7, return_ids=True)
    assert mg.fused_sampling_fallbacks == 1 and mg.fused_bound_switches == 1 and torch.equal(c, d)


@pytest.mark.parametrize('name', ['self_critic', 'token_critic', 'self_cond', 'can_remask'])
def test_decode_variants_inside_mm_generate_with_fused_sampling(name):
    """The decode variants at a vocabulary where the fused sampler applies (V = 8192): one mm_generate call (fused sampling on) against the
    stepwise loop over the public operators (logits materialised): ids and scores bit for bit (with a critic the scores depend on the sampled
    ids only; without one both sampling paths combine the same per-tile softmax statistics in the same order)."""
    torch.manual_seed(11)
    kw = dict(num_tokens=8192, seq_len=64, dim=256, depth=2, dim_head=64, heads=4, t5_name='t5-small')
    t = mm.MaskGitTransformer(self_cond=name == 'self_cond', **kw)
    with torch.no_grad():
        t.to_logits.weight.mul_(6.)
    extra, gkw = {}, {}
    if name == 'token_critic':
        extra['token_critic'] = mm.TokenCritic(**kw)
    elif name == 'self_critic':
        extra['self_token_critic'] = True
    elif name == 'can_remask':
        extra['no_mask_token_prob'] = 0.25
        gkw['can_remask_prev_masked'] = True
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None, **extra).to(DEV)
    B, T = 4, 6
    te = torch.randn(B, 7, 512, device=DEV)
    if name in ('token_critic', 'self_critic'):
        gkw['critic_noise'] = torch.rand(T, B, 64, device=DEV)
    ta, tb = {}, {}
    a = mg.generate([''] * B, timesteps=T, text_embeds=te, seed=5, fmap_size=8, trace=ta, **gkw)
    assert mg.fused_sampling_fallbacks == 0 and t._model().fused_ready
    b = mg.generate([''] * B, timesteps=T, text_embeds=te, seed=5, fmap_size=8, trace=tb, stepwise=True, **gkw)
    assert torch.equal(ta['ids'][0], torch.stack(tb['ids'])[0])
    assert torch.equal(a, b)
    assert torch.equal(ta['scores'], torch.stack(tb['scores']))
    # the logits path of the same call is the stepwise loop bit for bit
    c = mg.generate([''] * B, timesteps=T, text_embeds=te, seed=5, fmap_size=8, fused_sampling=False, **gkw)
    assert torch.equal(c, b)


def test_fused_sampling_inside_a_hip_graph_with_the_deferred_status_flag():
    """generate(fused_sampling='deferred') under stream capture: the fused path (no logits) is captured, the device flag is left for the caller;
    replays reproduce the eager fused ids and leave the flag at 0"""
    torch.manual_seed(2)
    t = mm.MaskGitTransformer(num_tokens=8192, seq_len=64, dim=256, depth=2, dim_head=64, heads=4, t5_name='t5-small')
    with torch.no_grad():
        t.to_logits.weight.mul_(6.)
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None).to(DEV)
    te = torch.randn(4, 7, 512, device=DEV)
    eager = mg.generate([''] * 4, timesteps=6, text_embeds=te, seed=9, fmap_size=8)
    assert mg.fused_sampling_fallbacks == 0
    torch.cuda.synchronize()
    graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            captured = mg.generate([''] * 4, timesteps=6, text_embeds=te, seed=9, fmap_size=8, fused_sampling='deferred')
    for _ in range(2):
        captured.fill_(-1)
        graph.replay()
        torch.cuda.synchronize()
        assert int(mg.fused_status[0].item()) == 0 and torch.equal(captured, eager)


def test_quantile_bound_operator_matches_kthvalue():
    """mm_fused_quantile (round 5): thr[r] = the rank-th largest of the S sampled logits of row r, exactly (radix descent on order-preserving keys) -- against
    torch.topk, incl. duplicated values, negative rows, S that is no multiple of the block"""
    g = torch.Generator().manual_seed(8)
    for R, S, k_keep, V in ((300, 2048, 6554, 65536), (17, 1000, 820, 8192), (5, 4096, 410, 4096), (64, 2048, 1, 65536)):
        sub = torch.randn(R, S, generator=g) * 3 - 1
        sub[0, : S // 2] = sub[0, 0]                                   # a row that is half one value
        sub[1] = -sub[1].abs() - 5                                     # all negative
        thr, rank = ops.fused_quantile(sub.to(DEV), k_keep, V)
        p = k_keep / V
        assert rank == min(S, max(1, math.ceil(S * p + 4.5 * math.sqrt(S * p * (1 - p)) + 1)))
        ref = sub.topk(rank, dim=-1).values[:, -1]
        assert torch.equal(thr.cpu(), ref), (R, S, rank)


@pytest.mark.parametrize('shape', ['peaky_heavy_block', 'bimodal_rows'])
def test_quantile_bound_keeps_the_fused_path_on_non_gaussian_checkpoints(shape):
    """the Gaussian bound of rounds 2-4 assumes a row's logits are normal over the vocabulary -- correctness never depended on it (verified
    per row, on-device fallback), speed did, and every measurement used random-init weights, the ideal case.  A to_logits that is x 8 with a heavy block of
    rows x 4 on top (bench.py's `non_gaussian_logits` leg) defeats it for EVERY row: more than 128 failing rows per step, the whole call repeated on the logits
    path.  The distribution-free bound (sampled vocabulary columns, Transformer.fused_bound = 'quantile', the default) keeps such a checkpoint on the fused
    path: no whole-call fallback, at most a handful of row fallbacks, ids identical to the logits path either way."""
    torch.manual_seed(6)
    V = 16384
    t = mm.MaskGitTransformer(num_tokens=V, seq_len=64, dim=256, depth=2, dim_head=64, heads=4, t5_name='t5-small')
    with torch.no_grad():
        if shape == 'peaky_heavy_block':
            t.to_logits.weight.mul_(8.)
            t.to_logits.weight[1024:2048].mul_(4.)
        else:                                                          # two populations of vocabulary rows with opposite offsets along one embedding direction
            t.to_logits.weight.mul_(4.)
            t.to_logits.weight[: V // 3, 0] += 3.
            t.to_logits.weight[V // 3:, 0] -= 1.5
    mg = mm.MaskGit(image_size=128, transformer=t, vae=None).to(DEV)
    B, T = 8, 6
    te = torch.randn(B, 7, 512, device=DEV)
    ref = mg.generate([''] * B, timesteps=T, text_embeds=te, seed=3, fmap_size=8, fused_sampling=False)
    res = {}
    for mode in ('quantile', 'gaussian'):
        t.fused_bound = mode
        f0, r0 = mg.fused_sampling_fallbacks, mg.fused_row_fallbacks
        out = mg.generate([''] * B, timesteps=T, text_embeds=te, seed=3, fmap_size=8)
        assert torch.equal(out, ref), f'{mode}: ids differ from the logits path'
        res[mode] = (mg.fused_sampling_fallbacks - f0, mg.fused_row_fallbacks - r0)
    t.fused_bound = 'auto'
    rows = B * sum(mg._mask_counts(T, 64))
    print(f'[fused bound] {shape}: whole-call fallbacks / rows finished by the on-device fallback (of {rows} sampled rows): quantile {res["quantile"]}, gaussian {res["gaussian"]}')
    assert res['quantile'][0] == 0 and res['quantile'][1] <= max(2, rows // 200)
    # (at this size the Gaussian estimate survives both shapes; at BASELINE configs[1] size the same recipe fails every row of every step with it -- bench.py's
    #  `non_gaussian_logits` leg, profiles/r05_*: 263 images/s with the call repeated on the logits path vs 547 with the sampled bound)


def test_generate_graph_mode_replays_with_fresh_seeds_bit_equal_to_eager():
    """generate(graph=True) (round 5; mmp.py:556-559 has a host synchronisation per step, this path has one per generate): the first call per signature runs
    eagerly, the second captures decode loop + VAE decode in a hipGraph, later ones replay.  The Philox keys are read from a device buffer at execution
    time (mm_generate_params.seed_dev), so a replay with seed s and row offset r is bit-identical -- ids and pixels -- to the eager call with the same keys,
    for keys the capture never saw; new text embeddings go through the static input; a different signature gets its own graph."""
    torch.manual_seed(4)
    t = mm.MaskGitTransformer(num_tokens=8192, seq_len=64, dim=256, depth=2, dim_head=64, heads=4, t5_name='t5-small')
    with torch.no_grad():
        t.to_logits.weight.mul_(6.)
    vae = mm.VQGanVAE(dim=32, codebook_size=8192)
    mg = mm.MaskGit(image_size=128, transformer=t, vae=vae).to(DEV)
    B, T = 4, 6
    tes = [torch.randn(B, 7, 512, device=DEV) for _ in range(2)]
    tes[1][2, 4:] = 0.
    calls = [(11, 0, 0), (12, 0, 0), (13, 4, 1), (14, 0, 0), (11, 0, 1)]      # (seed, row_offset, which text embeddings): call 0 warms, call 1 captures + replays, the rest replay
    for i, (seed, roff, k) in enumerate(calls):
        ids_g, img_g = mg.generate([''] * B, timesteps=T, text_embeds=tes[k], seed=seed, row_offset=roff, return_ids='both', graph=True)
        ids_e, img_e = mg.generate([''] * B, timesteps=T, text_embeds=tes[k], seed=seed, row_offset=roff, return_ids='both')
        assert torch.equal(ids_g, ids_e), f'call {i}: graph-mode ids differ from the eager call with the same keys'
        assert torch.equal(img_g, img_e), f'call {i}: pixels differ'
    assert len(mg._graphs) == 1 and isinstance(next(iter(mg._graphs.values())), dict) and mg.fused_sampling_fallbacks == 0
    a = mg.generate([''] * B, timesteps=T, text_embeds=tes[0], seed=11, return_ids='both', graph=True)[1]
    b = mg.generate([''] * B, timesteps=T, text_embeds=tes[0], seed=12, return_ids='both', graph=True)[1]
    assert not torch.equal(a, b), 'two replays with different seeds produced the same images'
    # another signature (timesteps): its own warm-up / capture, same equality
    for seed in (21, 22, 23):
        x = mg.generate([''] * B, timesteps=4, text_embeds=tes[0], seed=seed, return_ids=True, graph=True)
        assert torch.equal(x, mg.generate([''] * B, timesteps=4, text_embeds=tes[0], seed=seed, return_ids=True))
    assert len(mg._graphs) == 2


def test_generate_graph_mode_never_replays_a_stale_capture():
    """ADVICE r5 (medium, twice).  (1) `param.data` surgery + `invalidate_packed_weights()` (the documented EMA recipe) and `set_layernorm_fold()` drop the packed
    weights; `_pack_key()` cannot see either, so round 5's graph cache replayed the old capture on freed / stale packs.  The key now carries every module's pack
    generation and the entry holds what the capture reads: after the surgery the graph call must equal the EAGER call on the new weights (and differ from the
    old result).  (2) a MaskGit with a token critic and force_not_use_token_critic=True is admitted to the graph path: the flag must reach the warm-up and the
    captured call (round 5 dropped it: the critic ran, with a captured torch.rand)."""
    torch.manual_seed(9)
    t = mm.MaskGitTransformer(num_tokens=8192, seq_len=64, dim=256, depth=2, dim_head=64, heads=4, t5_name='t5-small')
    with torch.no_grad():
        t.to_logits.weight.mul_(6.)
    vae = mm.VQGanVAE(dim=32, codebook_size=8192)
    mg = mm.MaskGit(image_size=128, transformer=t, vae=vae).to(DEV)
    B, T = 4, 5
    te = torch.randn(B, 7, 512, device=DEV)
    kw = dict(timesteps=T, text_embeds=te, return_ids='both')
    for seed in (1, 2, 3):                                                        # warm, capture, replay
        old_ids, old_img = mg.generate([''] * B, seed=seed, graph=True, **kw)
    tr = mg.transformer
    with torch.no_grad():                                                         # surgery the version counters cannot see
        tr.to_logits.weight.data.copy_(tr.to_logits.weight.data.roll(1, 0))
        mg.vae.enc_dec.decoders[-1].weight.data.mul_(2.)
    tr.invalidate_packed_weights()
    mg.vae.invalidate_packed_weights()
    for seed in (3, 4, 5, 3):
        ids_g, img_g = mg.generate([''] * B, seed=seed, graph=True, **kw)
        ids_e, img_e = mg.generate([''] * B, seed=seed, **kw)
        assert torch.equal(ids_g, ids_e) and torch.equal(img_g, img_e), 'graph mode replayed a capture of the old weights'
    assert not torch.equal(ids_g, old_ids) and not torch.equal(img_g, old_img)
    tr.set_layernorm_fold(False)                                                  # another engine for the same parameters: its own capture
    for seed in (3, 6, 7):
        ids_g, img_g = mg.generate([''] * B, seed=seed, graph=True, **kw)
        ids_e, img_e = mg.generate([''] * B, seed=seed, **kw)
        assert torch.equal(ids_g, ids_e) and torch.equal(img_g, img_e)
    tr.set_layernorm_fold('auto')
    # (2) the critic flag
    torch.manual_seed(10)
    critic = mm.TokenCritic(num_tokens=8192, seq_len=64, dim=256, depth=1, dim_head=64, heads=4, t5_name='t5-small')
    mgc = mm.MaskGit(image_size=128, transformer=t, vae=vae, token_critic=critic).to(DEV)
    ref = mgc.generate([''] * B, seed=5, force_not_use_token_critic=True, **kw)
    with_critic = mgc.generate([''] * B, seed=5, **kw)
    assert not torch.equal(ref[0], with_critic[0]), 'the test cannot tell the two decode algorithms apart'
    for _ in range(3):
        got = mgc.generate([''] * B, seed=5, force_not_use_token_critic=True, graph=True, **kw)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), 'graph mode ran the token critic although force_not_use_token_critic=True'
    got = mgc.generate([''] * B, seed=5, graph=True, **kw)                        # with the critic: not a graph signature, runs eagerly
    assert got[0].shape == ref[0].shape and len(mgc._graphs) == 1


@pytest.mark.parametrize('family', ['zipf_bias', 'hot_tokens', 'row_temperature', 'student_t', 'bimodal'])
def test_non_gaussian_logits_are_finished_row_by_row(family):
    """Trained checkpoints do not have Gaussian logits.  Families that defeat the Gaussian bound for SOME rows: the finishing kernel must list
    exactly the rows it cannot prove, the listed rows finished on the logits path must complete a result that equals sample_rows on every row
    (ids and confidences), and nothing else may be wrong.  The fraction of rows that needed the fallback is reported per family."""
    g = torch.Generator().manual_seed(hash(family) % 1000)
    R, V = 96, 65536
    base = torch.randn(R, V, generator=g)
    if family == 'zipf_bias':          # every row has its own popularity ranking: a log-rank bias, long left tail, a few dominant tokens
        ranks = torch.stack([torch.randperm(V, generator=g) for _ in range(R)]).float()
        logits = base - 1.2 * torch.log1p(ranks) * torch.linspace(0.2, 1.5, R)[:, None]
    elif family == 'hot_tokens':       # 8 hot tokens per row, 12 sigma above the rest
        logits = base.clone()
        idx = torch.stack([torch.randperm(V, generator=g)[:8] for _ in range(R)])
        logits.scatter_add_(1, idx, torch.full((R, 8), 12.))
    elif family == 'row_temperature':  # per-row temperature 0.3 .. 3 (still bell-shaped: the bound must hold everywhere)
        logits = base * torch.logspace(-0.52, 0.48, R)[:, None]
    elif family == 'student_t':        # heavy tails on both sides (t, 3 degrees of freedom)
        logits = base / torch.sqrt(torch.distributions.Chi2(3.).sample((R, V)) / 3.)
    else:                              # two populations: 5 % of the vocabulary shifted up by 4 sigma in half the rows
        logits = base.clone()
        logits[::2, : V // 20] += 4.
    logits = logits.contiguous().to(DEV)
    k_keep = math.ceil(0.1 * V)
    kw = dict(noise_kind=_lib.MM_NOISE_PHILOX, seed=99, row_offset=3, step=2)
    ref_pred, ref_score = ops.sample_rows(logits, k_keep, 1.0, **kw)
    fb = ops.fused_buffers(R, V, DEV)
    thr = (logits.mean(dim=1) + ops.fused_z(k_keep, V) * logits.std(dim=1)).contiguous()
    ops.fused_emit(logits, thr, fb)
    fail_rows, fail_count = torch.full((128,), -1, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    pred, score = ops.fused_sample(fb, thr, R, V, k_keep, 1.0, fail_list=(fail_rows, fail_count), **kw)
    nf = int(fail_count.item())
    assert int(fb['fail'].item()) == 0 and nf <= 128
    failed = fail_rows[:nf].long().sort().values
    # exactly the rows with fewer than k values above their bound are listed
    short = ((logits >= thr[:, None]).sum(dim=1) < k_keep).nonzero().flatten()
    assert torch.equal(failed, short), f'{family}: listed {failed.tolist()} but rows short of candidates are {short.tolist()}'
    if nf:
        p2, s2 = ops.sample_rows(logits[failed].contiguous(), k_keep, 1.0, rows=failed.to(torch.int32), **kw)      # rows: the positions that key the noise stream
        pred[failed], score[failed] = p2, s2
    print(f'[fused sampling] {family}: {nf} of {R} rows ({100 * nf / R:.0f} %) needed the per-row fallback')
    assert torch.equal(pred, ref_pred) and torch.equal(score, ref_score)
    if family == 'row_temperature':
        assert nf == 0


def test_generate_finishes_unverifiable_rows_on_the_device():
    """mm_generate with the test hook that declares every 97th row unverifiable: those rows take the on-device fallback (gather -> 128 x 128 dense
    GEMM with a device-side row count -> sample_kernel) and the ids, the per-step states and the confidences equal the run without the hook bit
    for bit; the counter reports the rows, no whole-call fallback happens."""
    import bench
    mg, _ = bench.build_models(DEV)
    te = bench.synth_text(32, 32, 512).to(DEV)
    ta, tb = {}, {}
    a = mg.generate([''] * 32, timesteps=8, cond_scale=3, text_embeds=te, seed=11, return_ids=True, trace=ta)
    assert mg.fused_row_fallbacks == 0 and mg.fused_sampling_fallbacks == 0
    _lib.lib().mm_debug_set(1 << 27)
    try:
        b = mg.generate([''] * 32, timesteps=8, cond_scale=3, text_embeds=te, seed=11, return_ids=True, trace=tb)
    finally:
        _lib.lib().mm_debug_set(0)
    counts = mg._mask_counts(8, 256)
    expect = sum(len(range(5, 32 * k, 97)) for k in counts)
    assert mg.fused_sampling_fallbacks == 0 and mg.fused_row_fallbacks == expect, (mg.fused_row_fallbacks, expect)
    assert torch.equal(a, b) and torch.equal(ta['ids'], tb['ids']) and torch.equal(ta['scores'], tb['scores'])
