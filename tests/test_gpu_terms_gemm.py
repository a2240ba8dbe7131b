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
    got = ops.gemm_split(xs, ws, code, 1.0 / sc).double()
    L.lib().mm_debug_set2(2)
    try:
        old = ops.gemm_split(xs, ws, code, 1.0 / sc).double()
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
    got = ops.gemm_split(ops.split_rows(x.to(DEV), code), ops.split_pack_weight(w.to(DEV), code, 64, sc), code, 1.0 / sc).double().cpu()
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
