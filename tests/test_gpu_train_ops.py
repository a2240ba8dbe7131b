"""GPU parity of the backward operators (csrc/train.hip, csrc/attention_bwd.hip) against torch autograd (fp32, CPU) of the same
forward definitions the oracle restates (oracle/muse_oracle.py cites mmp.py for each).  Tolerances: gradients pass through
bf16 operands (2^-8 relative per rounding), so 2e-2 relative to the gradient's scale unless stated."""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import muse_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda'
bf16 = torch.bfloat16


@pytest.fixture(scope='module')
def ops():
    from muse_maskgit_pytorch_amd import ops
    return ops


def r16(t):
    return t.to(bf16).float()


def rel_err(got, ref):
    return (got.float().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)


def test_transpose_and_small_ops(ops):
    g = torch.Generator().manual_seed(0)
    for R, C in ((64, 64), (200, 72), (1000, 512), (8, 1408)):
        x = torch.randn(R, C, generator=g).to(bf16)
        assert torch.equal(ops.transpose(x.to(DEV)).cpu(), x.t())
    x = torch.randn(3, 1000, generator=g)
    assert torch.equal(ops.to_bf16(x.to(DEV)).cpu(), x.to(bf16))
    part = torch.randn(37, 300, generator=g)
    assert (ops.colsum(part.to(DEV)).cpu() - part.sum(0)).abs().max() < 1e-4
    src = torch.randn(5, 64, generator=g).to(bf16)
    idx = torch.tensor([7, 0, 3, 9, 4], dtype=torch.int32)
    dst = ops.scatter_rows(src.to(DEV), idx.to(DEV), 10).cpu()
    ref = torch.zeros(10, 64, dtype=bf16)
    ref[idx.long()] = src
    assert torch.equal(dst, ref)


@pytest.mark.parametrize('rows,D', [(100, 128), (1000, 512), (70, 1024)])
def test_layernorm_bwd(ops, rows, D):
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(D, generator=g)).requires_grad_(True)
    dy = r16(torch.randn(rows, D, generator=g))
    y = O.layer_norm(x, gamma, torch.zeros(D))
    y.backward(dy)
    base = torch.randn(rows, D, generator=g)
    dx = base.clone().to(DEV)
    dg = ops.layernorm_bwd(x.detach().to(DEV), dy.to(DEV, bf16), gamma.detach().to(DEV), dx, accumulate=True)
    assert rel_err(dx.cpu() - base, x.grad) < 1e-4
    assert rel_err(dg, gamma.grad) < 1e-4
    # gathered form: dy rows belong to a subset of x rows; only those rows of dx are written
    idx = torch.randperm(rows, generator=g)[:rows // 3].to(torch.int32)
    x2 = x.detach().clone().requires_grad_(True)
    y2 = O.layer_norm(x2[idx.long()], gamma.detach(), torch.zeros(D))
    y2.backward(dy[:idx.numel()])
    dx2 = torch.zeros(rows, D, device=DEV)
    ops.layernorm_bwd(x.detach().to(DEV), dy[:idx.numel()].to(DEV, bf16).contiguous(), gamma.detach().to(DEV), dx2, accumulate=False, row_index=idx.to(DEV))
    assert rel_err(dx2, x2.grad) < 1e-4


@pytest.mark.parametrize('rows,F', [(100, 341), (300, 1365)])
def test_geglu_ln_bwd(ops, rows, F):
    g = torch.Generator().manual_seed(F)
    Fp = (F + 63) // 64 * 64
    h = torch.zeros(rows, 2 * Fp)
    h[:, :F] = torch.randn(rows, F, generator=g)
    h[:, Fp:Fp + F] = torch.randn(rows, F, generator=g)
    h = r16(h)
    gamma = (1 + 0.2 * torch.randn(F, generator=g)).requires_grad_(True)
    dz = torch.zeros(rows, Fp)
    dz[:, :F] = torch.randn(rows, F, generator=g)
    dz = r16(dz)
    xh = h[:, :F].clone().requires_grad_(True)
    gt = h[:, Fp:Fp + F].clone().requires_grad_(True)
    a = gt * torch.nn.functional.gelu(xh)                       # mmp.py:72-77
    z = O.layer_norm(a, gamma, torch.zeros(F))                  # mmp.py:86
    z.backward(dz[:, :F])
    dh, dg = ops.geglu_ln_bwd(h.to(DEV, bf16), dz.to(DEV, bf16), F, gamma.detach().to(DEV))
    dh = dh.float().cpu()
    assert rel_err(dh[:, :F], xh.grad) < 1e-2 and rel_err(dh[:, Fp:Fp + F], gt.grad) < 1e-2
    assert dh[:, F:Fp].abs().max() == 0 and dh[:, Fp + F:].abs().max() == 0
    assert rel_err(dg, gamma.grad) < 1e-3


def test_ce_bwd_and_embed_bwd(ops):
    g = torch.Generator().manual_seed(3)
    R, V = 37, 8192
    logits = (torch.randn(R, V, generator=g) * 2).requires_grad_(True)
    labels = torch.randint(0, V, (R,), generator=g)
    torch.nn.functional.cross_entropy(logits, labels).backward()
    dl = ops.ce_bwd(logits.detach().to(DEV), labels.to(DEV), 1.0 / R)
    assert (dl.float().cpu() - logits.grad).abs().max() < 2e-3 * logits.grad.abs().max()
    B, n, D, T = 3, 16, 128, 50
    ids = torch.randint(0, T, (B, n), generator=g)
    ids[0, :5] = 7
    dx = torch.randn(B * n, D, generator=g)
    tok = torch.zeros(T, D, requires_grad=True)
    pos = torch.zeros(n, D, requires_grad=True)
    ((tok[ids] + pos[torch.arange(n)]).reshape(B * n, D) * dx).sum().backward()
    dtok, dpos = ops.embed_bwd(ids.to(DEV), dx.to(DEV), T)
    assert rel_err(dtok, tok.grad) < 1e-5 and rel_err(dpos, pos.grad) < 1e-5
    # many rows share one id (the mask id in training): the sum runs in ascending row order -> equal to a sequential fp32 sum, bit for bit,
    # and repeatable; more rows than one 256-row scan chunk, ids straddling the chunk boundaries
    B, n, D, T = 9, 128, 192, 40
    ids = torch.randint(0, T, (B, n), generator=g)
    ids[ids % 3 == 0] = 39
    dx = torch.randn(B * n, D, generator=g)
    a_tok, a_pos = ops.embed_bwd(ids.to(DEV), dx.to(DEV), T, two_level=False)
    b_tok, b_pos = ops.embed_bwd(ids.to(DEV), dx.to(DEV), T, two_level=False)
    assert torch.equal(a_tok, b_tok) and torch.equal(a_pos, b_pos)
    seq = torch.zeros(T, D)
    flat = ids.reshape(-1)
    for j in range(B * n):
        seq[flat[j]] += dx[j]
    assert torch.equal(a_tok.cpu(), seq)
    # the two-level form (round 6, what the training step uses): per 256-row block in ascending row order, then the blocks in ascending order
    c_tok, c_pos = ops.embed_bwd(ids.to(DEV), dx.to(DEV), T)
    d_tok, _ = ops.embed_bwd(ids.to(DEV), dx.to(DEV), T)
    assert torch.equal(c_tok, d_tok) and torch.equal(c_pos, a_pos)
    seq2 = torch.zeros(T, D)
    for b0 in range(0, B * n, 256):
        part = torch.zeros(T, D)
        seen = set()
        for j in range(b0, min(b0 + 256, B * n)):
            part[flat[j]] += dx[j]
            seen.add(int(flat[j]))
        for t in sorted(seen):
            seq2[t] += part[t]
    assert torch.equal(c_tok.cpu(), seq2)
    assert rel_err(c_tok, a_tok.cpu()) < 1e-6


def _attn_ref(q, k, v, qs, ks, nk, nv, mask):
    """mmp.py:143-162 + attend.py:123-140 on (b,h,n,64) tensors, fp32."""
    b, h, n, _ = q.shape
    kk = torch.cat((nk[None, :, None, :].expand(b, -1, -1, -1), k), dim=2)
    vv = torch.cat((nv[None, :, None, :].expand(b, -1, -1, -1), v), dim=2)
    qn = torch.nn.functional.normalize(q, dim=-1) * qs
    kn = torch.nn.functional.normalize(kk, dim=-1) * ks
    sim = torch.einsum('bhid,bhjd->bhij', qn, kn) * 8
    if mask is not None:
        m = torch.nn.functional.pad(mask, (1, 0), value=True)[:, None, None, :]
        sim = sim.masked_fill(~m, -torch.finfo(sim.dtype).max)
    return torch.einsum('bhij,bhjd->bhid', sim.softmax(-1), vv), qn, kn


@pytest.mark.parametrize('n,j,masked', [(64, 64, False), (256, 256, False), (64, 7, True), (128, 40, True), (256, 33, True), (512, 300, True), (1024, 1024, False)])
def test_attention_bwd_and_qk_norm_bwd(ops, n, j, masked):
    g = torch.Generator().manual_seed(n + j)
    b, h = 2, 3
    q = r16(torch.randn(b, h, n, 64, generator=g)).requires_grad_(True)
    k = r16(torch.randn(b, h, j, 64, generator=g)).requires_grad_(True)
    v = r16(torch.randn(b, h, j, 64, generator=g)).requires_grad_(True)
    qs = (1 + 0.2 * torch.randn(64, generator=g)).requires_grad_(True)
    ks = (1 + 0.2 * torch.randn(64, generator=g)).requires_grad_(True)
    nk = torch.randn(h, 64, generator=g).requires_grad_(True)
    nv = torch.randn(h, 64, generator=g).requires_grad_(True)
    mask = None
    if masked:
        mask = torch.rand(b, j, generator=g) < 0.7
    dout = r16(torch.randn(b, h, n, 64, generator=g))
    out, qn, kn = _attn_ref(q, k, v, qs, ks, nk, nv, mask)
    qn.retain_grad(); kn.retain_grad()
    out.backward(dout)
    d = lambda t: t.detach().to(DEV, bf16)
    o_dev = ops.attend(d(q), d(k), d(v), key_mask=mask.to(DEV) if masked else None, normalize=True, q_scale=qs.detach().to(DEV),
                       k_scale=ks.detach().to(DEV), null_k=nk.detach().to(DEV), null_v=nv.detach().to(DEV))
    assert rel_err(o_dev, out.detach()) < 2e-2
    dqn, dkn, dv, dnk, dnv = ops.attention_bwd(d(q), d(k), d(v), o_dev, dout.to(DEV, bf16), qs.detach().to(DEV), ks.detach().to(DEV),
                                               nk.detach().to(DEV), nv.detach().to(DEV), key_mask=mask.to(DEV) if masked else None)
    tol = 3e-2
    assert rel_err(dqn.reshape(b, n, h, 64).permute(0, 2, 1, 3), qn.grad) < tol
    if j:
        assert rel_err(dkn.reshape(b, j, h, 64).permute(0, 2, 1, 3), kn.grad[:, :, 1:]) < tol
        assert rel_err(dv.reshape(b, j, h, 64).permute(0, 2, 1, 3), v.grad) < tol
    assert rel_err(dnk.reshape(b, h, 64), kn.grad[:, :, 0]) < tol
    assert rel_err(dnv.reshape(b, h, 64).sum(0), nv.grad) < tol
    # l2norm * scale chain rule: q side, k side, null key
    q_rows = q.detach().permute(0, 2, 1, 3).reshape(b * n, h * 64)
    dq, dqs = ops.qk_norm_bwd(q_rows.to(DEV, bf16).contiguous(), dqn.reshape(b * n, h * 64), qs.detach().to(DEV), h)
    assert rel_err(dq.reshape(b, n, h, 64).permute(0, 2, 1, 3), q.grad) < tol
    assert rel_err(dqs, qs.grad) < tol
    dnull, dks_null = ops.qk_norm_bwd(None, None, ks.detach().to(DEV), h, x_f32=nk.detach().to(DEV), dy_f32=dnk)
    assert rel_err(dnull.reshape(b, h, 64).sum(0), nk.grad) < tol
    if j:
        k_rows = k.detach().permute(0, 2, 1, 3).reshape(b * j, h * 64)
        dk, dks = ops.qk_norm_bwd(k_rows.to(DEV, bf16).contiguous(), dkn.reshape(b * j, h * 64).contiguous(), ks.detach().to(DEV), h)
        assert rel_err(dk.reshape(b, j, h, 64).permute(0, 2, 1, 3), k.grad) < tol
        assert rel_err(dks + dks_null, ks.grad) < tol
    else:
        assert rel_err(dks_null, ks.grad) < tol


@pytest.mark.parametrize('nparts', [1, 2, 5, 8, 13, 16, 32])
def test_wide_colsum_is_the_narrow_colsum_bit_for_bit(ops, nparts):
    """split-K slabs of a weight gradient (D = N x K >= 65536) take the 16-byte-per-lane kernel: the same 4-stream x 2-accumulator association as the
    64-column kernel, so the two agree bit for bit (the narrow kernel runs on each half, which is below the switch)."""
    g = torch.Generator().manual_seed(nparts)
    part = (torch.randn(nparts, 131072, generator=g) * torch.logspace(-3, 3, 131072)[None]).to(DEV)
    wide = ops.colsum(part)
    narrow = torch.cat([ops.colsum(part[:, :32768].contiguous()), ops.colsum(part[:, 32768:65536].contiguous()),
                        ops.colsum(part[:, 65536:98304].contiguous()), ops.colsum(part[:, 98304:].contiguous())])
    assert torch.equal(wide, narrow)
    assert (wide.double().cpu() - part.double().sum(0).cpu()).abs().max() <= 1e-5 * part.abs().sum(0).max().item()


@pytest.mark.parametrize('M,N,K', [(600, 256, 32768), (1100, 512, 65536), (512, 132, 32768)])
def test_split_k_on_the_256x128_tile_for_long_contractions(ops, M, N, K):
    """round 6: dX of the training head (rows x dim over the vocabulary) -- mm_gemm_wgrad_splits picks 256 x 128 tiles and enough splits for two workgroups per
    CU; the slabs are summed in a fixed order.  Against the fp64 product of the same bf16 operands; twice the same bits."""
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.5).to(torch.bfloat16)
    lib = ops.L.lib()
    s = lib.mm_gemm_wgrad_splits(M, N, K)
    assert s >= 4 and (K // 64) % s == 0 and K // s >= 4096
    a = ops.gemm_wgrad(x.to(DEV), w.to(DEV))
    b = ops.gemm_wgrad(x.to(DEV), w.to(DEV))
    assert torch.equal(a, b)
    ref = x.double() @ w.double().t()
    assert (a.cpu().double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize('rows,N,K', [(8192, 512, 512), (8192, 2816, 512), (8192, 512, 1408), (1024, 1024, 512), (5457, 1024, 512), (100, 128, 128), (8192, 1536, 512)])
def test_weight_gradient_without_transposed_copies(ops, rows, N, K):
    """round 6 (csrc/gemm_tn.hip): dW = dY^T X read straight from the row-major operands (LDS-DMA blocks + ds_read_b64_tr_b16), against the fp64 product and
    the transposed-copy form; row counts that are no multiple of the 64-row stage, strided operand rows; twice the same bits."""
    g = torch.Generator().manual_seed(rows + N + K)
    dy_full = (torch.randn(rows, N + 64, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    x_full = (torch.randn(rows, K + 128, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    dy, x = dy_full[:, :N], x_full[:, 64:64 + K]                      # strided rows, the second operand at a column offset
    a = ops.gemm_wgrad_tn(dy, x)
    b = ops.gemm_wgrad_tn(dy, x)
    assert torch.equal(a, b)
    ref = dy.double().cpu().t() @ x.double().cpu()
    assert (a.cpu().double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-4
    old = ops.gemm_wgrad(ops.transpose(dy.contiguous(), pad_to=64), ops.transpose(x.contiguous(), pad_to=64))
    assert (a - old).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-4


@pytest.mark.parametrize('R,V', [(37, 65536), (5, 40000), (3, 16388)])
def test_ce_bwd_on_long_rows_reads_every_row_once(ops, R, V):
    """round 6: rows of 16388 .. 65536 logits are held in registers between the statistics and the gradient sweep (train.hip ce_bwd_row_kernel: 1024 threads x 16
    float4) instead of being read twice; against torch's cross-entropy gradient, with labels at the row's first / last column, twice the same bits."""
    g = torch.Generator().manual_seed(R + V)
    logits = (torch.randn(R, V, generator=g) * 3).requires_grad_(True)
    labels = torch.randint(0, V, (R,), generator=g)
    labels[0], labels[-1] = 0, V - 1
    torch.nn.functional.cross_entropy(logits, labels).backward()
    a = ops.ce_bwd(logits.detach().to(DEV), labels.to(DEV), 1.0 / R)
    b = ops.ce_bwd(logits.detach().to(DEV), labels.to(DEV), 1.0 / R)
    assert torch.equal(a, b)
    assert (a.float().cpu() - logits.grad).abs().max() < 2 ** -8 * logits.grad.abs().max()      # (bf16 output: half an ulp of the label's (p - 1) / R entry)
    assert abs(a.float().sum(dim=1).cpu()).max() < 2 ** -7 / R          # rows of softmax - onehot sum to ~0 (bf16 rounding of 65536 entries)
