"""GPU parity tests, operator level: every C-ABI operator against the CPU oracle (oracle/muse_oracle.py) or a
plain fp64/fp32 torch restatement, on the same seeded inputs.

Tolerances (stated once):
  * integer outputs (token ids, mask positions, LFQ indices): bit-exact.
  * fp32 outputs computed from bf16 inputs with fp32 accumulation (GEMM fp32 out, logits, scores): <= 1e-3 absolute
    on unit-scale data -- in practice ~1e-5; the bound is the north star's.
  * bf16 outputs: the same 1e-3 bound PLUS one bf16 ulp of the reference value (2^-7 relative), because the result
    is stored in bf16 and a rounding flip is one ulp.

`MUSE_TEST_DRYRUN=1` runs these tests on CPU with the operators replaced by torch emulations: that only checks
the TEST code (shapes, comparisons) in the GPU-less build container and is never a parity claim.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

import emu
import muse_oracle as O

from muse_maskgit_pytorch_amd import _lib, ops

DRY = os.environ.get('MUSE_TEST_DRYRUN') == '1'
DEV = 'cpu' if DRY else 'cuda'
pytestmark = [] if DRY else [pytest.mark.gpu]
bf16 = torch.bfloat16
ULP = 2.0 ** -7


def rnd(*shape, gen, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


def r16(t):
    return t.to(bf16).float()


def check_close(got, ref, atol=1e-3, rtol=0.0, what=''):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, f'{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}'
    assert torch.isfinite(got).all(), f'{what}: non-finite output'
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = err > bound
    if bad.any():
        i = torch.argmax((err - bound)).item()
        idx = list(torch.unravel_index(torch.tensor(i), got.shape))
        raise AssertionError(f'{what}: {int(bad.sum())}/{bad.numel()} out of tolerance; worst at {[int(j) for j in idx]} '
                             f'got {got.flatten()[i].item():.6g} ref {ref.flatten()[i].item():.6g} (max err {err.max().item():.3g})')


@pytest.fixture(autouse=True)
def _dry(monkeypatch):
    if not DRY:
        return
    monkeypatch.setattr(ops, '_chk_cuda', lambda *a: None)

    def gemm(x, w, out_f32=False, resid=None, out=None):
        y = x.float() @ w.float().t()
        if resid is not None:
            y = y + resid
        return y if out_f32 else y.to(bf16)
    monkeypatch.setattr(ops, 'gemm', gemm)
    monkeypatch.setattr(ops, 'gemm_cfg_logits', lambda xc, xn, w, s, out=None: (lambda c, n: n + (c - n) * s)(xc.float() @ w.float().t(), xn.float() @ w.float().t()))
    monkeypatch.setattr(ops, 'embed', lambda ids, tok, pos: (tok.float()[ids] + pos.float()[torch.arange(ids.shape[1])]).reshape(-1, tok.shape[1]))
    monkeypatch.setattr(ops, 'layernorm', lambda x, g, b=None, row_index=None: O.layer_norm(x if row_index is None else x[row_index.long()], g, b if b is not None else torch.zeros_like(g)).to(bf16))

    def geglu_ln(h, F_, g, b=None):
        Fp = h.shape[1] // 2
        a = h.float()[:, Fp:Fp + F_] * F.gelu(h.float()[:, :F_])
        out = torch.zeros(h.shape[0], Fp)
        out[:, :F_] = O.layer_norm(a, g, b if b is not None else torch.zeros_like(g))
        return out.to(bf16)
    monkeypatch.setattr(ops, 'geglu_ln', geglu_ln)

    def attend(q, k, v, key_mask=None, scale=8.0, normalize=False, q_scale=None, k_scale=None, null_k=None, null_v=None):
        q, k, v = q.float(), k.float(), v.float()
        b, h = q.shape[:2]
        if normalize:
            q = r16(F.normalize(q, dim=-1) * q_scale)
            k = F.normalize(k, dim=-1) * k_scale
        if null_k is not None:
            nk = null_k[None, :, None, :].expand(b, -1, -1, -1)
            if normalize:
                nk = F.normalize(nk, dim=-1) * k_scale
            k = torch.cat((nk, k), dim=2)
            v = torch.cat((r16(null_v)[None, :, None, :].expand(b, -1, -1, -1), v), dim=2)
            if key_mask is not None:
                key_mask = F.pad(key_mask.bool(), (1, 0), value=True)
        k = r16(k) if normalize else k
        m = key_mask.bool()[:, None, None, :] if key_mask is not None else None
        return O.attend(q, k, v, mask=m, scale=scale, rp=O.bf16_round).to(bf16)
    monkeypatch.setattr(ops, 'attend', attend)

    def mask_step(scores, ids, k, mask_id, want_rows=True):
        sel = O.select_topk_stable(scores, k)
        ids[sel] = mask_id
        scores[~sel] = O.MASK_FILL
        return sel.flatten().nonzero().flatten().int()
    monkeypatch.setattr(ops, 'mask_step', mask_step)

    def sample_rows(logits, k_keep, temperature, rows=None, noise_kind=0, noise=None, seed=0, row_offset=0, step=0, ids=None, scores=None):
        R, V = logits.shape
        pos = rows.long() if rows is not None else torch.arange(R)
        g = torch.zeros(R, V)
        if noise_kind == _lib.MM_NOISE_GUMBEL:
            g = noise.reshape(-1, V)[pos]
        elif noise_kind == _lib.MM_NOISE_UNIFORM:
            g = O.gumbel_from_uniform(noise.reshape(-1, V)[pos])
        kth = logits.topk(k_keep, dim=-1).values[:, -1:]
        filt = torch.where(logits >= kth, logits, torch.full_like(logits, float('-inf')))
        pred = (filt / temperature + g).argmax(-1)
        sc = 1 - logits.softmax(-1).gather(1, pred[:, None])[:, 0]
        if ids is not None:
            ids.view(-1)[pos] = pred
        if scores is not None:
            scores.view(-1)[pos] = sc
        return pred, sc
    monkeypatch.setattr(ops, 'sample_rows', sample_rows)
    monkeypatch.setattr(ops, 'philox_uniform', lambda seed, ro, step, rows, V, device: torch.rand(rows, V))
    emu.install(monkeypatch, ops)


# ------------------------------------------------------------------------------------------------ GEMM family
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (256, 1536, 512), (200, 300, 128), (37, 2816, 512), (8192, 512, 1408),
                                   (1, 512, 512), (129, 1, 128), (515, 516, 192)])
def test_gemm_dense(M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    x, w = r16(rnd(M, K, gen=g)), r16(rnd(N, K, gen=g, scale=0.1))
    ref = x.double() @ w.double().t()
    xd, wd = x.to(DEV, bf16), w.to(DEV, bf16)
    tol = 2e-6 * math.sqrt(K) * 8          # fp32 accumulation of K products of |x||w| ~ 0.1
    if N % 4 == 0:
        check_close(ops.gemm(xd, wd, out_f32=True), ref, atol=tol, what=f'gemm f32 {M}x{N}x{K}')
        resid = rnd(M, N, gen=g)
        out = torch.empty(M, N, dtype=torch.float32, device=DEV)
        check_close(ops.gemm(xd, wd, out_f32=True, resid=resid.to(DEV), out=out), ref + resid, atol=tol + 1e-6, what='gemm f32 + resid')
        check_close(ops.gemm(xd, wd), ref, atol=1e-3, rtol=ULP, what='gemm bf16 out')
    else:
        out = torch.empty(M, 4, dtype=torch.float32, device=DEV)     # N=1: ldc padded to 4
        if not DRY:
            _lib.check(_lib.lib().mm_gemm_bf16(_lib.stream(), _lib.ptr(xd), K, _lib.ptr(wd), K, M, N, K, _lib.ptr(out), 4, 1, None))
            check_close(out[:, :N], ref, atol=tol, what='gemm N=1')


def test_gemm_is_transpose_sensitive_and_row_independent():
    """asymmetric data; and the result of a row must not depend on how many rows are in the call (the CFG batching
    and the masked-row gather rely on it being bitwise identical)."""
    g = torch.Generator().manual_seed(3)
    x, w = r16(rnd(300, 512, gen=g)).to(DEV, bf16), r16(rnd(640, 512, gen=g, scale=0.1)).to(DEV, bf16)
    full = ops.gemm(x, w, out_f32=True)
    part = ops.gemm(x[100:177].contiguous(), w, out_f32=True)
    assert torch.equal(full[100:177], part)


@pytest.mark.parametrize('M,N,K', [(2048, 1536, 512), (700, 2816, 512), (1030, 512, 1408), (512, 128, 64)])
def test_gemm_large_tile_kernel_is_bit_identical_to_small(M, N, K):
    """M >= 512 takes the 256x128 / 3-stage kernel (gemm_big.hip); debug bit 8 forces the 128x128 kernel.  Same MFMA
    sequence per output element -> identical bits, repeated to shake out pipeline races."""
    if DRY:
        pytest.skip('kernel-structure test')
    g = torch.Generator().manual_seed(M + N)
    x, w = r16(rnd(M, K, gen=g)).to(DEV, bf16), r16(rnd(N, K, gen=g, scale=0.1)).to(DEV, bf16)
    resid = rnd(M, N, gen=g).to(DEV)
    lib = _lib.lib()
    lib.mm_debug_set(8)
    small_f32 = ops.gemm(x, w, out_f32=True, resid=resid, out=torch.empty(M, N, device=DEV))
    small_bf = ops.gemm(x, w)
    xc, xn = x[: M // 2].contiguous(), x[M // 2: 2 * (M // 2)].contiguous()
    small_cfg = ops.gemm_cfg_logits(xc, xn, w, 3.0)
    lib.mm_debug_set(0)
    ref = x.double().cpu() @ w.double().cpu().t()
    check_close(small_bf, ref, atol=1e-3, rtol=ULP, what='small kernel')
    for rep in range(5):
        assert torch.equal(ops.gemm(x, w, out_f32=True, resid=resid, out=torch.empty(M, N, device=DEV)), small_f32), f'f32 rep {rep}'
        assert torch.equal(ops.gemm(x, w), small_bf), f'bf16 rep {rep}'
        assert torch.equal(ops.gemm_cfg_logits(xc, xn, w, 3.0), small_cfg), f'cfg rep {rep}'


@pytest.mark.parametrize('kind,M,N,K', [('dense', 16384, 1536, 512), ('dense', 8192, 2048, 1024), ('dense', 9000, 4096, 512),
                                         ('geglu', 16384, 2816, 512), ('geglu', 16300, 4096, 1024), ('geglu', 8192, 8192, 1024),
                                         ('geglu_pers', 16384, 4096, 1024), ('cfg', 8128, 8192, 512), ('cfg', 4096, 65536, 512),
                                         ('cfg', 4000, 8192, 1024), ('cfg', 1000, 16384, 512), ('cfg', 130, 65536, 512),
                                         ('cfg_pers', 8128, 8192, 512), ('cfg_pers', 4096, 65536, 512)])
def test_gemm_persistent_kernel_is_bit_identical(kind, M, N, K):
    """>= 256 tiles of 256x128 with a pure-store epilogue take the persistent kernel (gemm_pers.hip: DMA cursor running ahead
    across tile boundaries, double-buffered fragments, stores overlapped with the next tile, counted vmcnt); debug bit 4096
    disables it; the guidance logits take gemm_cfg.hip (128 tokens x 256 columns, K in steps of 32, output in two halves; debug
    bit 8192 falls back to gemm_pers), and so does FF w1 with its GEGLU epilogue in the 256 x 256 variant of that kernel.  Identical MFMA sequence -> identical bits; repeated runs to shake out pipeline races
    (ragged M included)."""
    if DRY:
        pytest.skip('kernel-structure test')
    g = torch.Generator().manual_seed(M + N + K)
    lib = _lib.lib()
    w = r16(rnd(N, K, gen=g, scale=0.1)).to(DEV, bf16)
    if kind in ('cfg', 'cfg_pers'):
        xc, xn = r16(rnd(M, K, gen=g)).to(DEV, bf16), r16(rnd(M, K, gen=g)).to(DEV, bf16)
        run = lambda: ops.gemm_cfg_logits(xc, xn, w, 3.0)
    elif kind in ('geglu', 'geglu_pers'):
        x = r16(rnd(M, K, gen=g)).to(DEV, bf16)
        run = lambda: ops.gemm_geglu(x, w)
    else:
        x = r16(rnd(M, K, gen=g)).to(DEV, bf16)
        run = lambda: ops.gemm(x, w)
    lib.mm_debug_set(4096)
    ref = run()
    lib.mm_debug_set(8192 if kind in ('cfg_pers', 'geglu_pers') else 0)
    try:
        for rep in range(6):
            got = run()
            assert torch.equal(got, ref), f'{kind} rep {rep}: {(got != ref).sum().item()} of {got.numel()} elements differ'
    finally:
        lib.mm_debug_set(0)
    if kind == 'dense':
        sub = slice(0, 512)
        exact = x[sub].double().cpu() @ w.double().cpu().t()
        check_close(ref[sub], exact, atol=1e-3, rtol=ULP, what='non-persistent reference itself')


@pytest.mark.parametrize('M,N,K', [(64, 512, 128), (100, 8192, 512), (2, 65536, 512), (257, 640, 64)])
def test_gemm_cfg_logits(M, N, K):
    g = torch.Generator().manual_seed(11 + M)
    xc, xn, w = r16(rnd(M, K, gen=g)), r16(rnd(M, K, gen=g)), r16(rnd(N, K, gen=g, scale=0.1))
    c, n = xc.double() @ w.double().t(), xn.double() @ w.double().t()
    ref = n + (c - n) * 3.0
    got = ops.gemm_cfg_logits(xc.to(DEV, bf16), xn.to(DEV, bf16), w.to(DEV, bf16), 3.0)
    check_close(got, ref, atol=2e-4, what='cfg logits')
    # equals two dense GEMMs combined in fp32 (same accumulation order) bit for bit
    if not DRY:
        cd = ops.gemm(xc.to(DEV, bf16), w.to(DEV, bf16), out_f32=True)
        nd = ops.gemm(xn.to(DEV, bf16), w.to(DEV, bf16), out_f32=True)
        assert torch.equal(got, nd + (cd - nd) * 3.0)


# ------------------------------------------------------------------------------------------------ row kernels
def test_embed():
    g = torch.Generator().manual_seed(5)
    tok, pos = r16(rnd(513, 128, gen=g)), r16(rnd(64, 128, gen=g))
    ids = torch.randint(0, 513, (3, 64), generator=g)
    ref = (tok[ids] + pos[torch.arange(64)]).reshape(-1, 128)
    got = ops.embed(ids.to(DEV), tok.to(DEV, bf16), pos.to(DEV, bf16))
    check_close(got, ref, atol=0, what='embed')


@pytest.mark.parametrize('D', [128, 512, 1024, 2048])
def test_layernorm(D):
    g = torch.Generator().manual_seed(D)
    x = rnd(77, D, gen=g, scale=3.0) + 0.5
    gamma, beta = 1 + 0.2 * rnd(D, gen=g), 0.1 * rnd(D, gen=g)
    ref = O.layer_norm(x, gamma, beta)
    check_close(ops.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV)), ref, atol=1e-3, rtol=ULP, what='layernorm')
    check_close(ops.layernorm(x.to(DEV), gamma.to(DEV), None), O.layer_norm(x, gamma, torch.zeros(D)), atol=1e-3, rtol=ULP, what='layernorm beta=None')
    idx = torch.tensor([5, 0, 76, 5, 33], dtype=torch.int32)
    check_close(ops.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV), row_index=idx.to(DEV)), ref[idx.long()], atol=1e-3, rtol=ULP, what='layernorm gather')


@pytest.mark.parametrize('F_', [341, 1365, 2730])
def test_geglu_ln(F_):
    Fp = (F_ + 63) // 64 * 64
    g = torch.Generator().manual_seed(F_)
    h = torch.zeros(50, 2 * Fp)
    h[:, :F_] = rnd(50, F_, gen=g)
    h[:, Fp:Fp + F_] = rnd(50, F_, gen=g)
    h[:, F_:Fp] = 7.0          # garbage in the padding must not leak into the statistics
    h = r16(h)
    gamma, beta = 1 + 0.2 * rnd(F_, gen=g), 0.1 * rnd(F_, gen=g)
    a = h[:, Fp:Fp + F_] * F.gelu(h[:, :F_])
    ref = torch.zeros(50, Fp)
    ref[:, :F_] = O.layer_norm(a, gamma, beta)
    got = ops.geglu_ln(h.to(DEV, bf16), F_, gamma.to(DEV), beta.to(DEV))
    check_close(got, ref, atol=1e-3, rtol=ULP, what='geglu_ln')
    assert (got[:, F_:].float() == 0).all()


@pytest.mark.parametrize('M,D,F_', [(256, 128, 341), (1000, 512, 1365), (64, 512, 1365)])
def test_gemm_geglu_fused_and_layernorm_inner(M, D, F_):
    if DRY:
        pytest.skip('kernel-structure test')
    Fp = (F_ + 63) // 64 * 64
    g = torch.Generator().manual_seed(M + F_)
    x = r16(rnd(M, D, gen=g))
    w1 = r16(rnd(2 * F_, D, gen=g, scale=0.1))
    h = x.double() @ w1.double().t()
    ref = (h[:, F_:] * F.gelu(h[:, :F_])).float()
    a = ops.gemm_geglu(x.to(DEV, bf16), ops.pack_w1_geglu(w1.to(DEV), Fp))
    assert a.shape == (M, Fp) and (a[:, F_:].float() == 0).all()
    check_close(a[:, :F_], ref, atol=1e-3, rtol=ULP, what='fused GEGLU GEMM')
    gamma, beta = 1 + 0.2 * rnd(F_, gen=g), 0.1 * rnd(F_, gen=g)
    got = ops.layernorm_inner(a, F_, gamma.to(DEV), beta.to(DEV))
    exp = O.layer_norm(a[:, :F_].float().cpu(), gamma, beta)
    check_close(got[:, :F_], exp, atol=1e-3, rtol=ULP, what='layernorm_inner')
    assert (got[:, F_:].float() == 0).all()


def test_geglu_gelu_is_exact_over_the_whole_range():
    """common.h gelu_phi evaluates the erfc tail directly (rcp + exp2 + 8 FMAs) instead of libm's erff: the exact (erf) GELU of
    mmp.py:72-77 to within the bf16 rounding of the result (+ 1e-4 relative) for every input, including the far negative tail where the 1 + erf form
    cancels.  x values are produced exactly by a one-hot GEMM (x = 1 * w), the gate is exactly 1."""
    if DRY:
        pytest.skip('kernel-structure test')
    F_, D = 64, 64
    xs = torch.cat((torch.linspace(-12, 12, 4001), torch.tensor([0.0, -0.0, 1e-4, -1e-4, -5.0, -6.5, -8.0, 30.0, -30.0])))
    xs = r16(xs)
    M = (xs.numel() + F_ - 1) // F_
    xs = F.pad(xs, (0, M * F_ - xs.numel()))
    # row m of X holds the F_ values x[m*F_ .. ] in its first F_ features and a 1 in feature F_ - 1 + ... : use two one-hot blocks
    X = torch.zeros(M, 2 * D)
    X[:, :F_] = xs.view(M, F_)
    X[:, D] = 1.0
    w1 = torch.zeros(2 * F_, 2 * D)
    w1[torch.arange(F_), torch.arange(F_)] = 1.0        # gelu half: h[:, j] = x_j
    w1[F_ + torch.arange(F_), D] = 1.0                  # gate half: exactly 1
    a = ops.gemm_geglu(X.to(DEV, bf16), ops.pack_w1_geglu(w1.to(DEV), F_))
    got = a[:, :F_].float().cpu().reshape(-1).double()
    x64 = xs.double()
    ref = x64 * 0.5 * torch.special.erfc(-x64 / 2 ** 0.5)
    err = (got - ref).abs()
    tol = ref.abs() * (2.0 ** -8 + 1e-4) + (ref.abs() < 1e-37) * 1e-37          # half a bf16 ulp + 1e-4 relative (+ the flush of sub-normal results)
    bad = err > tol
    assert not bad.any(), (xs[bad][:5], got[bad][:5], ref[bad][:5])


# ------------------------------------------------------------------------------------------------ attention
def test_attend_seam_against_reference_golden(golden):
    """bare Attend(q,k,v,mask) on the reference's own tensors (golden from attend.py's math branch)."""
    op = golden('transformer_tiny.pt')['op']
    q, k, v = r16(op['q']), r16(op['k']), r16(op['v'])
    km = op['m4'][:, 0, 0, :]
    ref = O.attend(q, k, v, mask=op['m4'], rp=O.bf16_round)
    got = ops.attend(q.to(DEV, bf16), k.to(DEV, bf16), v.to(DEV, bf16), key_mask=km.to(DEV), scale=8.0)
    check_close(got, ref, atol=2e-3, rtol=2 * ULP, what='attend vs rounding-point oracle')
    # against the reference's own fp32 output on the UNROUNDED inputs: the golden q,k are unnormalised randn, so
    # logits reach |8 q.k| ~ 200 and rounding the inputs to bf16 moves a few near-tied softmax rows; compare in the mean
    err = (got.float().cpu() - op['attend_math']).abs()
    assert err.mean() < 5e-3 and (err > 0.05).float().mean() < 0.01, (err.mean().item(), err.max().item())


@pytest.mark.parametrize('n,j,masked', [(64, 64, False), (256, 256, False), (256, 37, True), (100, 300, True), (16, 1, False),
                                        (256, 192, False), (200, 128, False), (512, 256, False), (130, 129, False),
                                        (1024, 1024, False), (300, 384, False), (256, 512, False), (700, 640, False), (1024, 384, True)])
def test_attention_fused_norm_null(n, j, masked):
    """mmp.py:143-159 fused: l2norm + scales + null kv + key mask, strided q/k/v straight from projection buffers.
    j in {128, 192, 256} without a key mask and n >= 128 take the all-keys-resident kernel (attention_full_kernel); everything else -- the 1024-token
    self-attention of the super-resolution config included (round 6 added the long cases) -- the 64-key tile kernel with its online softmax."""
    g = torch.Generator().manual_seed(n * 3 + j)
    b, h = 2, 8
    q = r16(rnd(b, n, h * 64, gen=g))
    kv = r16(rnd(b, j, 2 * h * 64, gen=g))
    qs, ks = 1 + 0.2 * rnd(64, gen=g), 1 + 0.2 * rnd(64, gen=g)
    nk, nv = rnd(h, 64, gen=g), rnd(h, 64, gen=g)
    km = None
    if masked:
        km = torch.rand(b, j, generator=g) > 0.3
        km[1, :] = False              # a fully masked context: only the null key is left (the CFG null pass)
    # oracle on the same rounding points
    qh = q.reshape(b, n, h, 64).permute(0, 2, 1, 3)
    kh = kv[..., :h * 64].reshape(b, j, h, 64).permute(0, 2, 1, 3)
    vh = kv[..., h * 64:].reshape(b, j, h, 64).permute(0, 2, 1, 3)
    kk = torch.cat((nk[None, :, None, :].expand(b, -1, -1, -1), kh), dim=2)
    vv = torch.cat((r16(nv)[None, :, None, :].expand(b, -1, -1, -1), vh), dim=2)
    qn, kn = r16(F.normalize(qh, dim=-1) * qs), r16(F.normalize(kk, dim=-1) * ks)
    m4 = F.pad(km[:, None, None, :].expand(b, h, n, j), (1, 0), value=True) if km is not None else None
    ref = O.attend(qn, kn, vv, mask=m4, rp=O.bf16_round)
    qd, kvd = q.to(DEV, bf16), kv.to(DEV, bf16)
    q4 = qd.view(b, n, h, 64).permute(0, 2, 1, 3)
    k4 = kvd[..., :h * 64].reshape(b, j, h, 64) if False else kvd.view(b, j, 2 * h, 64)[:, :, :h].permute(0, 2, 1, 3)
    v4 = kvd.view(b, j, 2 * h, 64)[:, :, h:].permute(0, 2, 1, 3)
    got = ops.attend(q4, k4, v4, key_mask=km.to(DEV) if km is not None else None, scale=8.0, normalize=True,
                     q_scale=qs.to(DEV), k_scale=ks.to(DEV), null_k=nk.to(DEV).contiguous(), null_v=nv.to(DEV).contiguous())
    check_close(got, ref, atol=2e-3, rtol=2 * ULP, what=f'fused attention n={n} j={j}')
    if masked:   # fully masked context -> exactly the (bf16) null value for every query
        exp = r16(nv)[None, :, None, :].expand(1, h, n, 64)
        check_close(got[1:2], exp, atol=0, what='null-only attention')


@pytest.mark.parametrize('n,L', [(80, 13), (256, 33), (7, 1), (64, 35), (96, 36), (40, 77), (256, 64), (33, 79), (48, 80)])
def test_cross_attention_block_as_an_operator(n, L):
    """mmp.py:139-162, 191 at OPERATOR level: x + CrossAttention(LayerNorm(x), context) of one layer against the oracle's attention() on the
    same rounding points, for the one-kernel form (csrc/cross_fold.hip: L <= 35 on 36 key slots per head, L <= 79 on 80 -- 64 / 65 tokens straddle the two
    64-bit halves of its key-validity mask) AND the q-projection + attention + output-projection form (L = 80 and the debug bit), with ragged contexts (key masks), a fully masked context (the classifier-free null pass: only the null key), and query counts that are no
    multiple of the 32-query workgroup.  A wrong key slot or head offset shows up here at full size instead of hiding in a model's logits.  Bound: the bf16
    operator bound of this file (2e-3 absolute + relative ULPs) on the block's OUTPUT CONTRIBUTION (out - x: the fp32 residual add itself is exact)."""
    import muse_maskgit_pytorch_amd as mm
    from muse_maskgit_pytorch_amd import _lib
    torch.manual_seed(n * 131 + L)
    t = mm.MaskGitTransformer(num_tokens=512, seq_len=max(n, 8), dim=512, depth=2, dim_head=64, heads=8, t5_name='t5-small')
    with torch.no_grad():
        for p in t.parameters():                                       # de-trivialise gains / scales / null key and value
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
        for p in t.parameters():                                       # bf16-representable parameters: the engine's weight packing is exact, the oracle sees the same values
            p.copy_(p.to(torch.bfloat16).float())
    layer = 1
    sd = {k: v.detach().float().clone() for k, v in t.state_dict().items()}
    B = 3
    g = torch.Generator().manual_seed(L)
    x = torch.randn(B, n, 512, generator=g) * 1.5 + 0.1
    ctx = r16(torch.randn(B, L, 512, generator=g))
    mask = torch.ones(B, L, dtype=torch.bool)
    if L > 2:
        mask[1, L // 2:] = False                                       # ragged
    mask[2, :] = False                                                 # the null pass: every context key masked
    prefix = f'transformer_blocks.layers.{layer}.1.'
    ref = O.attention(x, sd, prefix, 8, context=ctx, context_mask=mask, rp=O.bf16_round)      # the block's contribution (no residual)
    t = t.to(DEV)
    lib = _lib.lib()
    outs = {}
    for bit in (0, -(1 << 31)):
        lib.mm_debug_set(bit)
        try:
            outs[bit] = (t.cross_attention_block(layer, x, ctx, mask).cpu() - x)
        finally:
            lib.mm_debug_set(0)
    one_kernel = L <= 79
    if one_kernel:
        assert not torch.equal(outs[0], outs[-(1 << 31)]), 'the debug bit did not change the path'
    for name, got in (('default path' + (' (one kernel)' if one_kernel else ''), outs[0]), ('three-kernel path', outs[-(1 << 31)])):
        check_close(got, ref, atol=3e-3 * max(1.0, ref.abs().max().item()), rtol=2 * ULP, what=f'cross-attention block n={n} L={L} {name}')
        print(f'[cross-attention operator] n={n} L={L} {name}: max abs err {(got - ref).abs().max().item():.3g} on scale {ref.abs().max().item():.3g}')
    # per head: the contribution of head h lives in all 512 output features, but a swapped key slot changes P: compare per sequence as well (a fully masked one = a constant row)
    d = (outs[0][2] - outs[0][2][0:1]).abs().max().item()
    assert d <= 1e-6 * max(1.0, ref.abs().max().item()), 'null pass: every query must receive the same row'


# ------------------------------------------------------------------------------------------------ sampling tail
def test_mask_step_matches_stable_topk():
    g = torch.Generator().manual_seed(9)
    for n, k in [(64, 64), (64, 55), (256, 23), (256, 1), (1024, 617), (1024, 1024), (1023, 300), (450, 0), (2500, 1999), (4096, 5)]:      # (round 6: up to 1024 threads, chunked beyond; lengths that are no multiple of 4 / 64)
        scores = torch.rand(3, n, generator=g)
        scores[0, ::3] = 0.5                      # ties, including across the k boundary
        scores[1, :] = 0.0                        # the all-zero first step (mmp.py:520)
        scores[2, n // 2:] = O.MASK_FILL
        ids = torch.randint(0, 500, (3, n), generator=g)
        sel = O.select_topk_stable(scores, k)
        sd, idd = scores.clone().to(DEV), ids.clone().to(DEV)
        rows = ops.mask_step(sd, idd, k, 512)
        exp_ids = torch.where(sel, torch.full_like(ids, 512), ids)
        assert torch.equal(idd.cpu(), exp_ids), f'mask ids n={n} k={k}'
        exp_rows = sel.flatten().nonzero().flatten().int()
        assert torch.equal(rows.cpu(), exp_rows), f'mask rows n={n} k={k}'
        exp_scores = torch.where(sel, scores, torch.full_like(scores, O.MASK_FILL))
        assert torch.equal(sd.cpu(), exp_scores)


@pytest.mark.parametrize('V,R', [(512, 40), (8192, 24), (65536, 6)])
@pytest.mark.parametrize('temperature', [1.0, 0.5, 1e-10])
def test_sample_rows_bit_exact_ids(V, R, temperature):
    g = torch.Generator().manual_seed(V + R)
    logits = rnd(R, V, gen=g, scale=1.5)
    logits[0] = r16(logits[0])                    # many exact duplicates (bf16-valued logits)
    if R > 3:
        logits[1, : V // 2] = logits[1, V // 2:]  # every value twice: threshold ties
        logits[2] = 0.25                          # constant row
        logits[3, 7] = 40.0                       # one dominant token: confidence -> 1, score -> 0
    u = torch.rand(R, V, generator=g)
    gum = O.gumbel_from_uniform(u)
    k_keep = math.ceil(0.1 * V)
    # oracle: keep every entry >= k-th largest (identical to the reference's topk+scatter unless the threshold ties)
    kth = logits.topk(k_keep, dim=-1).values[:, -1:]
    filt = torch.where(logits >= kth, logits, torch.full_like(logits, float('-inf')))
    pred_ref = O.gumbel_sample(filt, gum, temperature)
    score_ref = 1 - logits.softmax(-1).gather(1, pred_ref[:, None])[:, 0]
    pred, score = ops.sample_rows(logits.to(DEV), k_keep, temperature, noise_kind=_lib.MM_NOISE_GUMBEL, noise=gum.to(DEV))
    assert torch.equal(pred.cpu(), pred_ref), f'pred ids differ: {(pred.cpu() != pred_ref).nonzero().flatten().tolist()}'
    check_close(score, score_ref, atol=2e-6, what='confidence score')
    # uniform-noise mode applies mmp.py:403-408 on device: same ids unless two candidates are within an ulp
    pred_u, _ = ops.sample_rows(logits.to(DEV), k_keep, temperature, noise_kind=_lib.MM_NOISE_UNIFORM, noise=u.to(DEV))
    assert (pred_u.cpu() == pred_ref).float().mean() > 0.95
    # scatter form
    rows = torch.randperm(R * 2, generator=g)[:R].int()
    ids = torch.full((R * 2,), -1, dtype=torch.long, device=DEV)
    sc = torch.full((R * 2,), -7.0, device=DEV)
    noise_full = torch.zeros(R * 2, V)
    noise_full[rows.long()] = gum
    ops.sample_rows(logits.to(DEV), k_keep, temperature, rows=rows.to(DEV), noise_kind=_lib.MM_NOISE_GUMBEL, noise=noise_full.to(DEV),
                    ids=ids, scores=sc)
    assert torch.equal(ids.cpu()[rows.long()], pred_ref)
    untouched = torch.ones(R * 2, dtype=torch.bool)
    untouched[rows.long()] = False
    assert (ids.cpu()[untouched] == -1).all() and (sc.cpu()[untouched] == -7.0).all()


def test_sample_rows_argmax_mode_and_philox():
    g = torch.Generator().manual_seed(1)
    V, R = 8192, 16
    logits = rnd(R, V, gen=g)
    k_keep = math.ceil(0.1 * V)
    pred, score = ops.sample_rows(logits.to(DEV), k_keep, 1.0, noise_kind=_lib.MM_NOISE_NONE)
    assert torch.equal(pred.cpu(), logits.argmax(-1))
    if DRY:
        return
    # Philox mode == UNIFORM mode fed with the uniforms mm_philox_uniform reports for the same (seed, rows, step)
    u = ops.philox_uniform(1234, 100, 3, R, V, DEV)
    assert 0.0 <= u.min().item() and u.max().item() < 1.0 and abs(u.mean().item() - 0.5) < 0.01
    p1, s1 = ops.sample_rows(logits.to(DEV), k_keep, 0.7, noise_kind=_lib.MM_NOISE_PHILOX, seed=1234, row_offset=100, step=3)
    p2, s2 = ops.sample_rows(logits.to(DEV), k_keep, 0.7, noise_kind=_lib.MM_NOISE_UNIFORM, noise=u)
    assert torch.equal(p1, p2) and torch.equal(s1, s2)
    p3, _ = ops.sample_rows(logits.to(DEV), k_keep, 0.7, noise_kind=_lib.MM_NOISE_PHILOX, seed=1235, row_offset=100, step=3)
    assert not torch.equal(p1, p3)


# ------------------------------------------------------------------------------------------------ VAE operators
@pytest.mark.parametrize('cin,cout,hw,kind', [(64, 128, 8, 'c3'), (16, 32, 8, 'c3'), (128, 64, 6, 'c1'), (64, 64, 8, 'c4s2'),
                                              (64, 32, 5, 'ct'), (8, 16, 12, 'stem'), (64, 3, 16, 'head'), (200, 72, 4, 'c3')])
def test_conv_variants_match_emulation(cin, cout, hw, kind):
    g = torch.Generator().manual_seed(cin + cout + hw)
    B = 2
    x = r16(rnd(B, hw, hw, cin, gen=g))
    bias = 0.1 * rnd(cout, gen=g)
    xd = x.to(DEV, bf16)

    def both(fn):
        return fn(ops, xd, True), fn(emu, x, False)

    if kind in ('c3', 'c1'):
        k = 3 if kind == 'c3' else 1
        w = ops.pack_conv_weight(rnd(cout, cin, k, k, gen=g, scale=0.05))
        resid = r16(rnd(B, hw, hw, cout, gen=g))
        got, ref = both(lambda m, xx, dev: m.conv2d_nhwc(xx, w.to(DEV) if dev else w.float(), cout, k, k, 1, (-(k // 2), -(k // 2)),
                                                        bias=bias.to(DEV) if dev else bias, act=(k == 3),
                                                        resid=(resid.to(DEV, bf16) if dev else resid) if k == 1 else None))
    elif kind == 'c4s2':
        w = ops.pack_conv_weight(rnd(cout, cin, 4, 4, gen=g, scale=0.05))
        got, ref = both(lambda m, xx, dev: m.conv2d_nhwc(xx, w.to(DEV) if dev else w.float(), cout, 4, 4, 2, (-1, -1), out_hw=(hw // 2, hw // 2),
                                                        bias=bias.to(DEV) if dev else bias, act=True))
    elif kind == 'ct':
        packs = ops.pack_convT_weight(rnd(cin, cout, 4, 4, gen=g, scale=0.05))

        def run(m, xx, dev):
            out = torch.zeros(B, 2 * hw, 2 * hw, cout, dtype=bf16 if dev else torch.float32, device=DEV if dev else 'cpu')
            for (py, px), wp in packs.items():
                m.conv2d_nhwc(xx, wp.to(DEV) if dev else wp.float(), cout, 2, 2, 1, (py - 1, px - 1), out_hw=(hw, hw), os_=2, parity=(py, px),
                              full_hw=(2 * hw, 2 * hw), bias=bias.to(DEV) if dev else bias, act=True, out=out)
            return out
        got, ref = both(run)
    elif kind == 'stem':
        w = ops.pack_conv_weight_cin8(rnd(cout, 3, 5, 5, gen=g, scale=0.1))
        x[..., 3:] = 0
        xd = x.to(DEV, bf16)
        got, ref = both(lambda m, xx, dev: m.conv2d_nhwc(xx, w.to(DEV) if dev else w.float(), cout, 5, 5, 1, (-2, -2), bias=bias.to(DEV) if dev else bias))
    else:  # head: 1x1 to 3 channels, NCHW fp32 out
        w = ops.pack_conv_weight(rnd(cout, cin, 1, 1, gen=g, scale=0.1))
        got, ref = both(lambda m, xx, dev: m.conv2d_nhwc(xx, w.to(DEV) if dev else w.float(), cout, 1, 1, 1, (0, 0), bias=bias.to(DEV) if dev else bias,
                                                        out_nchw_f32=True))
        check_close(got, ref, atol=1e-3, what='conv head fp32')
        return
    check_close(got, ref, atol=2e-3, rtol=ULP, what=f'conv {kind}')


@pytest.mark.parametrize('shape', [(2, 16, 64, 128, 3), (2, 16, 72, 40, 1), (8, 64, 64, 256, 3), (2, 12, 256, 3, 1)])
def test_conv_half_storage_against_fp64(shape):
    """mm_conv2d_nhwc_half (round 6; vqgan_vae.py:224-232 on the fp16-storage decoder): fp16 NHWC activations x fp16 weights (packed x 2^j, undone by alpha) on
    the fp16 MFMA with fp32 accumulation, + bias, LeakyReLU, fp16 residual -> fp16 NHWC or NCHW fp32, against torch's fp64 convolution of the SAME fp16 values:
    what is left is fp32 accumulation order and the final fp16 rounding.  Shapes cover the 128x128 kernel, ragged channel counts, the 256x128 kernel (>= 256
    tiles, Cin % 64 == 0) and the narrow NCHW head."""
    B, hw, cin, cout, k = shape
    g = torch.Generator().manual_seed(cin * 7 + cout)
    f16 = torch.float16
    x = rnd(B, hw, hw, cin, gen=g).to(f16)
    w = rnd(cout, cin, k, k, gen=g, scale=0.05)
    scale = ops.f16_weight_scale([w])
    wp = ops.pack_conv_weight(w, f16, scale)
    wq = (w * scale).to(f16).double() / scale               # the weights the kernel multiplies by
    bias = 0.1 * rnd(cout, gen=g)
    head = cout < 8
    resid = None if (k == 3 or head) else rnd(B, hw, hw, cout, gen=g).to(f16)
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wq, bias.double(), padding=k // 2)
    if k == 3:
        ref = torch.nn.functional.leaky_relu(ref, 0.1)
    if resid is not None:
        ref = ref + resid.double().permute(0, 3, 1, 2)
    got = ops.conv2d_nhwc_half(x.to(DEV), wp.to(DEV), cout, k, k, 1, (-(k // 2), -(k // 2)), bias=bias.to(DEV), act=(k == 3),
                               resid=resid.to(DEV) if resid is not None else None, out_nchw_f32=head, alpha=1.0 / scale)
    got = got.double().cpu() if head else got.double().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    sc = ref.abs().max().item()
    print(f'[conv half] {shape}: max abs err {err:.3g} on scale {sc:.3g}')
    assert err <= (2e-5 if head else 6e-4) * sc               # fp16 output rounding: 2^-11 relative


@pytest.mark.parametrize('half', [False, True], ids=['bf16', 'f16'])
@pytest.mark.parametrize('kind', ['c3', 'c1', 'c4s2', 'ct'])
def test_conv_wide_tile_kernel_is_bit_identical_to_the_256x128_kernel(kind, half):
    """Round 6: NHWC convolutions with Cin % 64 == 0, Cout % 256 == 0, an even number of 64-deep k-steps and >= 256 tiles of 256 pixels x 256 channels run on
    gemm_wide_conv.hip (the persistent 256 x 256 x 64 tile of gemm_wide.hip with an implicit-GEMM loader: taps in the zero padding are requested beyond the buffer
    extent and arrive as zeros).  Same MFMA order per output element as gemm_big_kernel<MODE_CONV>, which test_conv_variants_match_emulation pins through the
    128 x 128 kernel -> bit-identical outputs, for bf16 and for the fp16-storage form; mm_debug_set2(8) selects the older kernel."""
    lib = _lib.lib()
    g = torch.Generator().manual_seed(23)
    B, hw, cin, cout = 8, 64, 128, 512
    dt = torch.float16 if half else bf16
    scale = 64.0 if half else 1.0
    conv = (lambda *a, **k: ops.conv2d_nhwc_half(*a, alpha=1.0 / scale, **k)) if half else ops.conv2d_nhwc
    pk = (lambda w: ops.pack_conv_weight(w, torch.float16, scale)) if half else ops.pack_conv_weight
    pkT = (lambda w: ops.pack_convT_weight(w, torch.float16, scale)) if half else ops.pack_convT_weight
    x = rnd(B, hw, hw, cin, gen=g).to(DEV, dt)
    bias = (0.1 * rnd(cout, gen=g)).to(DEV)
    if kind in ('c3', 'c1'):
        k = 3 if kind == 'c3' else 1
        w = pk(rnd(cout, cin, k, k, gen=g, scale=0.05)).to(DEV)
        run = lambda: conv(x, w, cout, k, k, 1, (-(k // 2), -(k // 2)), bias=bias, act=(k == 3))
    elif kind == 'c4s2':
        x = rnd(B, 2 * hw, 2 * hw, cin, gen=g).to(DEV, dt)
        w = pk(rnd(cout, cin, 4, 4, gen=g, scale=0.05)).to(DEV)
        run = lambda: conv(x, w, cout, 4, 4, 2, (-1, -1), out_hw=(hw, hw), bias=bias, act=True)
    else:
        packs = {k_: v.to(DEV) for k_, v in pkT(rnd(cin, cout, 4, 4, gen=g, scale=0.05)).items()}

        def run():
            out = torch.zeros(B, 2 * hw, 2 * hw, cout, dtype=dt, device=DEV)
            for (py, px), wp in packs.items():
                conv(x, wp, cout, 2, 2, 1, (py - 1, px - 1), out_hw=(hw, hw), os_=2, parity=(py, px), full_hw=(2 * hw, 2 * hw), bias=bias, act=True, out=out)
            return out
    wide = run()
    lib.mm_debug_set2(8)
    try:
        big = run()
    finally:
        lib.mm_debug_set2(0)
    assert torch.isfinite(wide.float()).all() and wide.float().abs().max() > 0.1
    assert torch.equal(wide, big), f'{(wide != big).sum().item()} of {wide.numel()} outputs differ (max {(wide.float() - big.float()).abs().max().item():.3g})'
    # a ragged pixel count (the last tile's rows beyond M are requested out of bounds and never stored)
    if kind == 'c3':
        xr = x[:, :61].contiguous()
        a_ = conv(xr, w, cout, 3, 3, 1, (-1, -1), bias=bias, act=True)
        lib.mm_debug_set2(8)
        try:
            b_ = conv(xr, w, cout, 3, 3, 1, (-1, -1), bias=bias, act=True)
        finally:
            lib.mm_debug_set2(0)
        assert torch.equal(a_, b_)


@pytest.mark.parametrize('kind', ['c3', 'c1', 'c4s2', 'ct'])
def test_conv_large_tile_kernel_is_bit_identical_to_small(kind):
    """Cin % 64 == 0 convolutions with >= 256 tiles run on the 256x128 three-stage kernel (gemm_big.hip: wave-uniform tap walk);
    same MFMA order as the 128x128 kernel that test_conv_variants_match_emulation pins -> bit-identical outputs."""
    lib = _lib.lib()
    g = torch.Generator().manual_seed(11)
    B, hw, cin, cout = 8, 64, 64, 256
    x = rnd(B, hw, hw, cin, gen=g).to(DEV, bf16)
    bias = (0.1 * rnd(cout, gen=g)).to(DEV)
    if kind in ('c3', 'c1'):
        k = 3 if kind == 'c3' else 1
        w = ops.pack_conv_weight(rnd(cout, cin, k, k, gen=g, scale=0.05)).to(DEV)
        resid = rnd(B, hw, hw, cout, gen=g).to(DEV, bf16) if k == 1 else None
        run = lambda: ops.conv2d_nhwc(x, w, cout, k, k, 1, (-(k // 2), -(k // 2)), bias=bias, act=(k == 3), resid=resid)
    elif kind == 'c4s2':
        x = rnd(B, 2 * hw, 2 * hw, cin, gen=g).to(DEV, bf16)
        w = ops.pack_conv_weight(rnd(cout, cin, 4, 4, gen=g, scale=0.05)).to(DEV)
        run = lambda: ops.conv2d_nhwc(x, w, cout, 4, 4, 2, (-1, -1), out_hw=(hw, hw), bias=bias, act=True)
    else:
        packs = {k_: v.to(DEV) for k_, v in ops.pack_convT_weight(rnd(cin, cout, 4, 4, gen=g, scale=0.05)).items()}

        def run():
            out = torch.zeros(B, 2 * hw, 2 * hw, cout, dtype=bf16, device=DEV)
            for (py, px), wp in packs.items():
                ops.conv2d_nhwc(x, wp, cout, 2, 2, 1, (py - 1, px - 1), out_hw=(hw, hw), os_=2, parity=(py, px), full_hw=(2 * hw, 2 * hw),
                                bias=bias, act=True, out=out)
            return out
    big = run()
    lib.mm_debug_set(8)
    try:
        small = run()
    finally:
        lib.mm_debug_set(0)
    assert torch.equal(big, small)
    assert big.float().abs().max() > 0.05


def test_glu_groupnorm_lfq_layout():
    g = torch.Generator().manual_seed(2)
    x = r16(rnd(2, 4, 4, 256, gen=g))
    check_close(ops.glu_nhwc(x.to(DEV, bf16)), emu.glu_nhwc(x), atol=1e-3, rtol=ULP, what='glu')
    xg = r16(rnd(2, 8, 8, 128, gen=g, scale=2.0) + 1.0)
    gamma, beta = 1 + 0.2 * rnd(128, gen=g), 0.1 * rnd(128, gen=g)
    for act in (False, True):
        check_close(ops.groupnorm_nhwc(xg.to(DEV, bf16), 16, gamma.to(DEV), beta.to(DEV), act=act), emu.groupnorm_nhwc(xg, 16, gamma, beta, act=act),
                    atol=2e-3, rtol=ULP, what='groupnorm')
    ids = torch.randint(0, 512, (2, 8, 8), generator=g)
    wo, bo = rnd(128, 9, gen=g), rnd(128, gen=g)
    check_close(ops.lfq_decode(ids.to(DEV), 9, 128, wo.to(DEV), bo.to(DEV)), emu.lfq_decode(ids, 9, 128, wo, bo), atol=1e-3, rtol=ULP, what='lfq decode')
    # the wide-projection kernel (16-bit / 13-bit codebooks, register-resident weights, pixel strips)
    for bits, C in ((16, 2048), (13, 512)):
        ids_w = torch.randint(0, 2 ** bits, (3, 7, 5), generator=g)
        wo_w, bo_w = rnd(C, bits, gen=g), rnd(C, gen=g)
        check_close(ops.lfq_decode(ids_w.to(DEV), bits, C, wo_w.to(DEV), bo_w.to(DEV)), emu.lfq_decode(ids_w, bits, C, wo_w, bo_w),
                    atol=1e-3, rtol=ULP, what=f'lfq decode {bits} bit')
    wi, bi = rnd(9, 128, gen=g), rnd(9, gen=g)
    xe = r16(rnd(2, 8, 8, 128, gen=g))
    ids_ref, q_ref = emu.lfq_encode(xe, 9, wi, bi, wo, bo)
    ids_got, q_got = ops.lfq_encode(xe.to(DEV, bf16), 9, wi.to(DEV), bi.to(DEV), wo.to(DEV), bo.to(DEV))
    margin = (xe @ wi.t() + bi).abs().min(-1).values > 1e-3          # ids are only defined away from the sign boundary
    assert torch.equal(ids_got.cpu()[margin], ids_ref[margin])
    check_close(q_got.cpu()[margin], q_ref[margin], atol=1e-3, rtol=ULP, what='lfq encode out')
    img = rnd(2, 3, 8, 8, gen=g)
    check_close(r16(ops.nchw_to_nhwc8(img.to(DEV)).float().cpu()), r16(emu.nchw_to_nhwc8(img)), atol=0, what='nchw->nhwc8')
    check_close(ops.nhwc_to_nchw_f32(x.to(DEV, bf16)), emu.nhwc_to_nchw_f32(x), atol=0, what='nhwc->nchw')


@pytest.mark.parametrize('N,K,C,cosine', [(100, 512, 32, False), (1000, 8192, 256, True), (257, 1000, 64, False), (33, 65536, 256, True)])
def test_vq_nearest_lookup(N, K, C, cosine):
    """EXTENSION (self-defined oracle: the reference's VectorQuantize branch cannot run): nearest-codebook lookup,
    Euclidean and cosine.  fp32 MFMA vs torch fp32 differ in summation order, so rows whose top-2 margin is below 1e-4 are only
    required to pick a code within 1e-4 of the optimum; all others must match exactly.  Exact ties -> the lower index."""
    if DRY:
        pytest.skip('no CPU emulation of this operator')
    g = torch.Generator().manual_seed(N + K)
    x = rnd(N, C, gen=g)
    cb = rnd(K, C, gen=g)
    cb[7] = cb[3]                              # an exact duplicate: the lower index must win wherever code 3 is nearest
    x[0] = cb[3] * 1.5 if cosine else cb[3]
    if cosine:
        sc = torch.nn.functional.normalize(x, dim=-1) @ torch.nn.functional.normalize(cb, dim=-1).t()
    else:
        sc = x @ cb.t() - 0.5 * (cb * cb).sum(-1)[None]
    top2 = sc.topk(2, dim=-1)
    ref = sc.argmax(-1)
    got = ops.vq_nearest(x.to(DEV), cb.to(DEV), cosine=cosine).cpu()
    assert got[0] == 3
    clear = (top2.values[:, 0] - top2.values[:, 1]) > 1e-4
    clear[0] = False
    assert torch.equal(got[clear], ref[clear])
    assert (sc.gather(1, got[:, None])[:, 0] >= top2.values[:, 0] - 1e-4).all()
    q = ops.vq_gather(got.to(DEV).reshape(N), cb.to(DEV)).cpu()
    assert torch.equal(q, cb[got])


@pytest.mark.parametrize('M,N,K', [(16384, 1536, 512), (16384, 1536, 192), (16300, 1024, 320), (5000, 4096, 448), (16384, 3072, 1024), (9000, 2304, 640)])
def test_gemm_wide_kernels_bit_identical_to_the_128x128_kernel(M, N, K):
    """gemm_wide.hip (256 x 256 | 192 x 64 tile, persistent with an even k-step count, one tile per workgroup with an odd one): the plain bf16 projections
    and the GEGLU + LayerNorm-partial-sum form against the 128x128 kernel (debug bit 8), ragged row counts included, three repetitions (race screen)"""
    if DRY:
        pytest.skip('kernel-structure test')
    lib = _lib.lib()
    g = torch.Generator().manual_seed(M + N + K)
    x = r16(rnd(M, K, gen=g)).to(DEV, bf16)
    w = r16(rnd(N, K, gen=g, scale=0.1)).to(DEV, bf16)
    runs = [lambda: ops.gemm(x, w)]
    if N % 256 == 0:
        runs.append(lambda: ops.gemm_geglu(x, w))
    lib.mm_debug_set(8)
    try:
        refs = [f() for f in runs]
    finally:
        lib.mm_debug_set(0)
    for rep in range(3):
        for f, ref in zip(runs, refs):
            got = f()
            assert torch.equal(got, ref), f'shape {(M, N, K)} rep {rep}: {(got != ref).sum().item()} elements differ'


def test_gemm_kernel_family_random_shapes_bit_identical():
    """Race screen for the counted-vmcnt / LDS-DMA pipelines: random (ragged) shapes through whatever kernel the dispatcher picks
    (persistent, 256x128 three-stage, guidance 128x256) against the 128x128 kernel (debug bit 8), several repetitions each."""
    if DRY:
        pytest.skip('kernel-structure test')
    lib = _lib.lib()
    g = torch.Generator().manual_seed(77)
    shapes = []
    for _ in range(10):
        M = int(torch.randint(300, 20000, (1,), generator=g))
        N = int(torch.randint(2, 40, (1,), generator=g)) * 128
        K = int(torch.randint(1, 24, (1,), generator=g)) * 64
        shapes.append((M, N, K))
    for M, N, K in shapes:
        x = r16(rnd(M, K, gen=g)).to(DEV, bf16)
        x2 = r16(rnd(M, K, gen=g)).to(DEV, bf16)
        w = r16(rnd(N, K, gen=g, scale=0.1)).to(DEV, bf16)
        res = rnd(M, N, gen=g).to(DEV)
        runs = [lambda: ops.gemm(x, w), lambda: ops.gemm(x, w, out_f32=True, resid=res), lambda: ops.gemm_cfg_logits(x, x2, w, 2.5)]
        lib.mm_debug_set(8)
        try:
            refs = [f() for f in runs]
        finally:
            lib.mm_debug_set(0)
        for rep in range(3):
            for f, ref in zip(runs, refs):
                got = f()
                assert torch.equal(got, ref), f'shape {(M, N, K)} rep {rep}: {(got != ref).sum().item()} elements differ'
