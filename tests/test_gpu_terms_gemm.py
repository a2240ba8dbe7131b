"""csrc/gemm_terms.hip (round 5): the 'f16x2' tier's wide projections with every operand term staged once -- the products xh.wh + xl.wh (+ xh.wl) of a
k-block run from ONE staging of the term planes instead of a depth-3K GEMM over the duplicated segment packs.  Operator level against fp64 torch and
against the concatenated-depth kernels of rounds 4-5 (same terms, another summation order: equal to fp32 rounding, not bit for bit); the GEGLU epilogue
(term-split output + LayerNorm(inner) partial sums) against the fp64 evaluation of mmp.py:72-88.  Model level: tests/test_gpu_base_size.py (the tier's
full-size parity runs pass through this kernel with mm_debug_set2(1))."""
import pytest
import torch

from muse_maskgit_pytorch_amd import _lib as L
from muse_maskgit_pytorch_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture
def force_terms():
    L.lib().mm_debug_set2(1)      # the term-sharing kernel whatever the tile count
    yield
    L.lib().mm_debug_set2(0)


def _operands(M, N, K, wkind, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g) * (1. + torch.rand(M, 1, generator=g))
    w = torch.randn(N, K, generator=g) / K ** 0.5
    if wkind == 'bf16':
        w = w.bfloat16().float()
    sc = ops.f16_weight_scale([w])
    terms = ops.weight_terms_f16(w, sc)
    assert terms == dict(bf16=1, fp32=2)[wkind]
    code = ops.MM_SPLIT_F16 | (1 + terms)
    return x, w, sc, code


# (M, N, K): the base config's q|k|v (192-row weight tiles) and FF w1 (256-row tiles) at the bench's row count, a ragged row count, the paper-scale width
@pytest.mark.parametrize('M,N,K', [(16384, 1536, 512), (16384, 2816, 512), (16500, 1536, 512), (8192, 3072, 1024)])
@pytest.mark.parametrize('wkind', ['fp32', 'bf16'])
def test_term_sharing_gemm_against_fp64_and_the_concatenated_kernel(M, N, K, wkind):
    x, w, sc, code = _operands(M, N, K, wkind, M + N + K)
    xs, ws = ops.split_rows(x.to(DEV), code), ops.split_pack_weight(w.to(DEV), code, 64, sc)
    ref = (x.double().to(DEV) @ w.double().to(DEV).t())
    got = ops.gemm_split(xs, ws, code, 1.0 / sc, shared=True).double()
    L.lib().mm_debug_set2(2)
    try:
        old = ops.gemm_split(xs, ws, code, 1.0 / sc, shared=True).double()
    finally:
        L.lib().mm_debug_set2(0)
    scale = ref.abs().max().item()
    e, eo, d = (got - ref).abs().max().item(), (old - ref).abs().max().item(), (got - old).abs().max().item()
    print(f'[terms] gemm {M}x{N}x{K} {wkind} ({ops.split_count(code)} products): max err {e:.3g} (concatenated kernel {eo:.3g}), between the two {d:.3g}; |ref| {scale:.3g}')
    assert d > 0. or M * N == 0, 'the term-sharing kernel was not dispatched (identical bits to the concatenated order)'
    assert e <= 4e-6 * scale and e <= 2 * eo + 1e-7 * scale


@pytest.mark.parametrize('M,N,K,wkind', [(700, 512, 128, 'fp32'), (256, 768, 256, 'bf16'), (1000, 192, 64, 'fp32'), (257, 256, 384, 'bf16'), (33, 576, 128, 'fp32')])
def test_term_sharing_gemm_small_and_ragged_shapes(force_terms, M, N, K, wkind):
    """mm_debug_set2(1): few tiles, ragged last row tile, one workgroup walking several tiles or none; K = 64 and 384 (fp32 weights only multiples of 64,
    bf16-representable ones multiples of 128: others fall back to the concatenated kernel and still have to be right)"""
    x, w, sc, code = _operands(M, N, K, wkind, 7 * M + N + K)
    ref = x.double() @ w.double().t()
    got = ops.gemm_split(ops.split_rows(x.to(DEV), code), ops.split_pack_weight(w.to(DEV), code, 64, sc), code, 1.0 / sc, shared=True).double().cpu()
    scale = ref.abs().max().item()
    e = (got - ref).abs().max().item()
    print(f'[terms] forced gemm {M}x{N}x{K} {wkind}: max err {e:.3g}; |ref| {scale:.3g}')
    assert e <= 4e-6 * scale


@pytest.mark.parametrize('M,F,D,wkind', [(16384, 1365, 512, 'fp32'), (16384, 1365, 512, 'bf16'), (16390, 1365, 512, 'fp32')])
def test_w1_geglu_term_split_epilogue(M, F, D, wkind):
    """FF w1 of the tier: GEGLU + the term split of its output + LayerNorm(inner) partial sums in the GEMM's epilogue, against fp64 of mmp.py:72-77, 85"""
    g = torch.Generator().manual_seed(M + F)
    Fp = (F + 63) // 64 * 64
    x = torch.randn(M, D, generator=g)
    w1 = torch.randn(2 * F, D, generator=g) / D ** 0.5
    if wkind == 'bf16':
        w1 = w1.bfloat16().float()
    sc = ops.f16_weight_scale([w1])
    code = ops.MM_SPLIT_F16 | (1 + ops.weight_terms_f16(w1, sc))
    w1g = ops.split_pack_weight(ops.pack_w1_geglu(w1.to(DEV), Fp, dtype=torch.float32), code, 1, sc)
    out, part = ops.gemm_split_geglu(ops.split_rows(x.to(DEV), code), w1g, code, 1.0 / sc)
    h = ops.unsplit_rows(out, code, Fp).double()
    y = x.double().to(DEV) @ w1.double().to(DEV).t()
    xv, gate = y[:, :F], y[:, F:]
    ref = gate * (xv * 0.5 * (1. + torch.erf(xv / 2 ** 0.5)))
    scale = ref.abs().max().item()
    e = (h[:, :F] - ref).abs().max().item()
    assert bool((h[:, F:] == 0).all()), 'padding columns'
    s1, s2 = part[..., 0].double().sum(dim=1), part[..., 1].double().sum(dim=1)
    e1 = (s1 - ref.sum(dim=1)).abs().max().item() / ref.abs().sum(dim=1).max().item()
    e2 = (s2 - (ref * ref).sum(dim=1)).abs().max().item() / (ref * ref).sum(dim=1).max().item()
    P = ops.split_count(code)
    seg = out.view(torch.float16).reshape(M, P, Fp)
    assert P == 2 or torch.equal(seg[:, 0], seg[:, 2]), 'segment 2 repeats the high term'
    print(f'[terms] w1 + GEGLU {M}x{2 * Fp}x{D} {wkind} ({P} products): max err {e:.3g} on |ref| {scale:.3g}; row sums {e1:.3g}, sums of squares {e2:.3g} (relative)')
    assert e <= 4e-6 * scale and e1 <= 1e-5 and e2 <= 1e-5


@pytest.mark.parametrize('M,N,K', [(16384, 512, 512), (16384, 512, 1408), (8192, 512, 512), (8200, 512, 512), (300, 200, 64)])
@pytest.mark.parametrize('wkind', ['fp32', 'bf16'])
def test_term_sharing_in_the_residual_and_narrow_kernels(M, N, K, wkind):
    """the fp32-residual projections (attention out, FF w2: 256 x 128 kernel) and the narrow ones (cross-attention q / out: 128 x 128 and 64-token tiles) with the
    same term sharing in their k-loops (gemm_big.hip / gemm.hip NP = 2, 3), against fp64 and the concatenated-depth form (mm_debug_set2(2))"""
    x, w, sc, code = _operands(M, N, K, wkind, 3 * M + N + K)
    g = torch.Generator().manual_seed(M + 1)
    resid = torch.randn(M, N, generator=g).to(DEV)
    xs, ws = ops.split_rows(x.to(DEV), code), ops.split_pack_weight(w.to(DEV), code, 64, sc)
    ref = x.double().to(DEV) @ w.double().to(DEV).t() + resid.double()
    got = ops.gemm_split(xs, ws, code, 1.0 / sc, resid=resid, shared=True).double()
    L.lib().mm_debug_set2(2)
    try:
        old = ops.gemm_split(xs, ws, code, 1.0 / sc, resid=resid, shared=True).double()
    finally:
        L.lib().mm_debug_set2(0)
    scale = ref.abs().max().item()
    e, eo, d = (got - ref).abs().max().item(), (old - ref).abs().max().item(), (got - old).abs().max().item()
    print(f'[terms] gemm + residual {M}x{N}x{K} {wkind}: max err {e:.3g} (concatenated {eo:.3g}), between the two {d:.3g}; |ref| {scale:.3g}')
    assert d > 0., 'term sharing was not dispatched'
    assert e <= 4e-6 * scale and e <= 2 * eo + 1e-7 * scale


@pytest.mark.parametrize('bf16_weights', [False, True], ids=['fp32w', 'bf16w'])
def test_tier_generate_at_bench_size_fused_logits_kernel_against_the_logits_path(bf16_weights):
    """'f16x2' mm_generate at B = 32 (the rows of the bench: every GEMM of the loop on its term-sharing form by itself).  The fused logits kernel
    (gemm_wide_fused_kernel<F16, NP>: emission from the accumulators) and the logits path (gemm_terms_kernel + sample_kernel) run the same MFMA sequence per
    logit and combine the same tile statistics in the same order: ids AND confidences bit for bit at every step -- which ties the B = 32 fused path to the
    path tests/test_gpu_base_size.py holds to the reference at B = 2."""
    import bench
    mg, _ = bench.build_models(DEV)
    tr = mg.transformer
    with torch.no_grad():
        if bf16_weights:
            for q in mg.parameters():
                q.copy_(q.bfloat16().float())
        tr.to_logits.weight.mul_(8.)
    mg.set_precision('f16x2')
    try:
        assert tr.split_products() == (2 if bf16_weights else 3)
        te = bench.synth_text(32, 32, 512).to(DEV)
        ta, tb = {}, {}
        a = mg.generate([''] * 32, timesteps=18, cond_scale=3, text_embeds=te, seed=11, return_ids=True, trace=ta)
        assert mg.fused_sampling_fallbacks == 0 and tr._model().fused_ready
        b = mg.generate([''] * 32, timesteps=18, cond_scale=3, text_embeds=te, seed=11, return_ids=True, fused_sampling=False, trace=tb)
        for s_ in range(18):
            assert torch.equal(ta['masked_ids'][s_], tb['masked_ids'][s_]) and torch.equal(ta['ids'][s_], tb['ids'][s_]), f'step {s_}'
            assert torch.equal(ta['scores'][s_], tb['scores'][s_]), f'step {s_}: confidences differ by {(ta["scores"][s_] - tb["scores"][s_]).abs().max().item():.3g}'
        assert torch.equal(a, b)
        L.lib().mm_debug_set2(2)      # the concatenated-depth kernels of rounds 4-5: same terms, another summation order -> the same ids on these well-separated logits
        try:
            c = mg.generate([''] * 32, timesteps=18, cond_scale=3, text_embeds=te, seed=11, return_ids=True)
        finally:
            L.lib().mm_debug_set2(0)
        agree = (a == c).float().mean().item()
        print(f'[terms] f16x2 generate B = 32 ({tr.split_products()} products): term-sharing vs concatenated kernels: {100 * agree:.3f} % of the final ids equal')
        assert agree >= 0.95      # (a near-tie flipped by the last-bit difference sends one image down another trajectory)
    finally:
        mg.set_precision('bf16')


@pytest.mark.parametrize('B,H,C,Cout,wkind', [(2, 128, 256, 256, 'fp32'), (2, 128, 256, 256, 'bf16'), (1, 256, 128, 128, 'fp32'), (8, 64, 512, 256, 'bf16')])
def test_term_sharing_convolution(B, H, C, Cout, wkind):
    """3 x 3 convolution on fp16 term segments per pixel (the tier's VAE decoder, parity.conv_x3): the 256 x 128 implicit-GEMM kernel with every term plane of a
    32-channel step staged once (mm_conv2d_nhwc_terms + MM_SPLIT_SHARED) against the concatenated-depth form and torch's fp64 convolution"""
    from muse_maskgit_pytorch_amd import parity as P32
    g = torch.Generator().manual_seed(B + H + C)
    x = torch.randn(B, H, H, C, generator=g).to(DEV)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (9 * C) ** 0.5)
    if wkind == 'bf16':
        w = w.bfloat16().float()
    w = w.to(DEV)
    sc = ops.f16_weight_scale([w])
    code = ops.MM_SPLIT_F16 | (1 + ops.weight_terms_f16(w, sc))
    P = ops.split_count(code)
    xs = ops.split_rows(x.reshape(-1, C), code).reshape(B, H, H, P * C)
    wp = P32._pack_x3(P32._taps_conv(w), code, sc)
    outs = []
    for shared in (ops.MM_SPLIT_SHARED, 0):
        out = torch.empty(B, H, H, Cout, dtype=torch.float32, device=DEV)
        L.check(L.lib().mm_conv2d_nhwc_terms(L.stream(), L.ptr(xs), B, H, H, P * C, L.ptr(wp), Cout, 3, 3, 1, -1, -1, H, H, 1, 0, 0, H, H, None, 0, None, L.ptr(out), 2,
                                             1.0 / sc, code | shared), 'mm_conv2d_nhwc_terms')
        outs.append(out.double())
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    scale = ref.abs().max().item()
    e, eo, d = (outs[0] - ref).abs().max().item(), (outs[1] - ref).abs().max().item(), (outs[0] - outs[1]).abs().max().item()
    print(f'[terms] conv 3x3 {B}x{H}x{H}x{C} -> {Cout} {wkind} ({P} products): max err {e:.3g} (concatenated {eo:.3g}), between the two {d:.3g}; |ref| {scale:.3g}')
    assert (d > 0.) == (P == 3), 'three products: term sharing; two: the concatenated form stays (measured faster, gemm_big.hip mm_gemm_big_launch)'
    assert e <= 4e-6 * scale and e <= 2 * eo + 1e-7 * scale


def test_tier_feed_forward_fused_form_and_its_refusal(force_terms):
    """model.hip ff_block of the tier: w1 + GEGLU + term split + LayerNorm(inner) partials (gemm_terms.hip EPI 1) -> w2 with the LayerNorm folded in, against the fp32
    engine ('parity': fp32 MFMA, operator by operator).  Then a checkpoint whose LayerNorm(inner) gains push the gain-folded w2 out of fp16 range at the model's
    term scale: the pack refuses the fold for those layers (mm_ff_weights.w1_terms_geglu = NULL) and the three-kernel form must give the same accuracy."""
    import muse_maskgit_pytorch_amd as mm
    torch.manual_seed(3)
    tr = mm.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=2, dim_head=64, heads=2, t5_name='t5-small').to(DEV).eval()
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(0, 513, (3, 64), generator=g).to(DEV)
    te = torch.randn(3, 6, 512, generator=g).to(DEV)

    def both():
        with torch.no_grad():
            ref = tr.set_precision('parity')(ids, text_embeds=te, cond_drop_prob=0.).float()
            got = tr.set_precision('f16x2')(ids, text_embeds=te, cond_drop_prob=0.).float()
        fused = [t_.get('w1g') is not None for t_ in tr._model().keep if isinstance(t_, dict) and 'w1g' in t_]
        tr.set_precision('bf16')
        return (got - ref).abs().max().item(), ref.abs().max().item(), fused

    e, sc, fused = both()
    print(f'[terms] tier feed-forward, fused form on {sum(fused)} of {len(fused)} blocks: logits max err {e:.3g} on |ref| {sc:.3g}')
    assert all(fused) and e <= 2e-5 * max(1., sc)
    with torch.no_grad():
        for _, _, ff in tr.transformer_blocks.layers:
            ff[3].gamma.mul_(20.)            # |w2 . gamma| x the model's term scale (the largest |w| at 2^13 .. 2^14) leaves the fp16 range: no fold for these blocks
    tr.invalidate_packed_weights()
    e2, sc2, fused2 = both()
    print(f'[terms] ... with the fold refused on {len(fused2) - sum(fused2)} of {len(fused2)} blocks: logits max err {e2:.3g} on |ref| {sc2:.3g}')
    assert not any(fused2[:2]) and e2 <= 2e-5 * max(1., sc2)
