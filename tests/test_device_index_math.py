"""CPU models of three pieces of device-side index arithmetic added in round 4 (pure Python mirrors of the HIP code, no GPU, no library):

  * csrc/sampling_fused.hip, dense gather: a piece's 128-bit keep mask -> the granule number of every entry of its compacted slot, by rank (v_mbcnt) and two lane
    permutations (ds_permute).  Each permutation must be a bijection of the 64 lanes (a destination written twice would lose an entry) and entry e must receive
    the position of the e-th set bit -- including pieces with more than 64 kept granules, whose tail wraps around the second permutation.
  * csrc/cross_fold.hip, placement: workgroup id -> (sequence, query block) with all query blocks of a sequence on one XCD (workgroup id mod 8).
  * csrc/cross_fold.hip, packed operands: the fragment-major index maps of the pack kernels are bijections onto their buffers, and the folded algebra
    (P_h V_h) W_o,h^T == P_h (V_h W_o,h^T) reproduces the three-step cross-attention (muse_maskgit_pytorch.py:139-162) in fp64.

The GPU tests hold the kernels to the oracle; these pin the combinatorics they rely on."""
import random

import torch


def _gather_lane_model(mask128):
    """returns (granule number per list entry) as the finisher's lanes compute it"""
    m01, m23 = mask128 & ((1 << 64) - 1), mask128 >> 64
    c01, cnt = bin(m01).count('1'), bin(mask128).count('1')
    ja, jb = [None] * 64, [None] * 64
    for lane in range(64):
        ra = bin(m01 & ((1 << lane) - 1)).count('1')                       # v_mbcnt over the low mask half
        rb = bin(m23 & ((1 << lane) - 1)).count('1')
        da = ra if (m01 >> lane) & 1 else c01 + (lane - ra)                 # owners keep their rank, the others go behind the list
        db = ((c01 + rb) if (m23 >> lane) & 1 else (cnt + (lane - rb))) & 63
        assert ja[da] is None and jb[db] is None, 'a permutation destination is written twice'
        ja[da], jb[db] = lane, lane
    assert None not in ja and None not in jb
    out = [ja[e] if e < c01 else 64 + jb[e] for e in range(min(cnt, 64))]
    out += [64 + jb[e - 64] for e in range(64, cnt)]                         # the tail round reads the wrapped half of the second permutation
    return out


def test_dense_gather_recovers_every_granule_number_from_the_keep_mask():
    rng = random.Random(4)
    masks = [0, (1 << 128) - 1, (1 << 64) - 1, ((1 << 64) - 1) << 64, 1, 1 << 127, (1 << 65) - 1]
    for _ in range(4000):
        dens = rng.random()
        masks.append(sum(1 << j for j in range(128) if rng.random() < dens))
    for mk in masks:
        assert _gather_lane_model(mk) == [j for j in range(128) if (mk >> j) & 1]


def test_cross_fold_places_every_query_block_of_a_sequence_on_one_xcd():
    for seqs, nq in ((32, 256), (3, 80), (1, 7), (64, 256), (9, 135), (8, 32)):
        nqb = (nq + 31) // 32
        grid = 8 * ((seqs + 7) // 8) * nqb
        seen = {}
        for wg in range(grid):
            xcd, slot = wg & 7, wg >> 3
            b, qb = (slot // nqb) * 8 + xcd, slot % nqb
            if b >= seqs:
                continue
            assert (b, qb) not in seen
            seen[(b, qb)] = xcd
        assert len(seen) == seqs * nqb                                      # every (sequence, query block) exactly once
        for b in range(seqs):
            assert len({seen[(b, qb)] for qb in range(nqb)}) == 1           # ... and a sequence's blocks share one XCD


def _vwt_index(s, o, kfl):          # cross_fold_pack_kernel: element (feature o, flat key kfl = 36 head + key) of sequence s
    ob, fro, kb, fgk, j = o >> 4, o & 15, kfl >> 5, (kfl & 31) >> 3, kfl & 7
    return ((((s * 32 + ob) * 9 + kb) * 64) + fgk * 16 + fro) * 8 + j


def _khat_index(s, h, key, d):      # K^ fragments of the 16 x 16 x 16 MFMA
    kb, frk, ob, fgk, j = key >> 4, key & 15, d >> 4, (d & 15) >> 2, d & 3
    return (((s * 8 + h) * 12 + kb * 4 + ob) * 64 + fgk * 16 + frk) * 4 + j


def _wqf_index(o, k):               # q weight fragments
    h, ob, fro, kb, fgk, j = o >> 6, (o & 63) >> 4, o & 15, k >> 5, (k & 31) >> 3, k & 7
    return ((((h * 4 + ob) * 16 + kb) * 64) + fgk * 16 + fro) * 8 + j


def test_cross_fold_fragment_layouts_are_bijections():
    S = 2
    assert sorted(_vwt_index(s, o, k) for s in range(S) for o in range(512) for k in range(288)) == list(range(S * 512 * 288))
    assert sorted(_khat_index(s, h, key, d) for s in range(S) for h in range(8) for key in range(48) for d in range(64)) == list(range(S * 8 * 48 * 64))
    assert sorted(_wqf_index(o, k) for o in range(512) for k in range(512)) == list(range(512 * 512))
    # what a wave of the kernel reads with one 16-byte load per lane is one contiguous KiB: lane = 16 fg + fr takes 8 consecutive elements
    for ob, kb in ((0, 0), (5, 3), (31, 8)):
        idx = [_vwt_index(1, ob * 16 + (lane & 15), kb * 32 + (lane >> 4) * 8 + j) for lane in range(64) for j in range(8)]
        assert idx == list(range(idx[0], idx[0] + 512))


def test_output_projection_folds_into_the_values():
    """x += (softmax(8 q^ k^T + mask) V) W_o^T, head by head, equals one contraction of the flat probabilities [query][(head, key)] with VW = V_h W_o,h^T
    (null key / value first, 36 key slots per head, masked and padding keys with probability zero)"""
    g = torch.Generator().manual_seed(0)
    H, dh, D, m, nq = 8, 64, 512, 33, 5
    q = torch.randn(nq, H, dh, generator=g, dtype=torch.float64)
    k = torch.randn(m + 1, H, dh, generator=g, dtype=torch.float64)         # key 0 = the null key
    v = torch.randn(m + 1, H, dh, generator=g, dtype=torch.float64)
    wo = torch.randn(D, H * dh, generator=g, dtype=torch.float64)
    keep = torch.ones(m + 1, dtype=torch.bool)
    keep[20:] = False                                                        # zero-padded text rows
    keep[0] = True
    qn, kn = torch.nn.functional.normalize(q, dim=-1), torch.nn.functional.normalize(k, dim=-1)
    sim = 8. * torch.einsum('qhd,khd->hqk', qn, kn).masked_fill(~keep[None, None, :], float('-inf'))
    p = sim.softmax(dim=-1)                                                  # [H][nq][m + 1]
    ref = torch.einsum('hqk,khd->qhd', p, v).reshape(nq, H * dh) @ wo.t()   # attention, then the output projection
    vw = torch.zeros(H * 36, D, dtype=torch.float64)                        # flat (head, key) axis, 36 slots per head
    pf = torch.zeros(nq, H * 36, dtype=torch.float64)
    for h in range(H):
        vw[h * 36:h * 36 + m + 1] = v[:, h] @ wo[:, h * dh:(h + 1) * dh].t()
        pf[:, h * 36:h * 36 + m + 1] = p[h]
    assert torch.allclose(pf @ vw, ref, rtol=1e-12, atol=1e-12)
    # the null pass (every text key masked): one non-zero probability per head -> the sum over the heads of VW's null rows (k_cross_fold_null_row)
    null_row = sum(vw[h * 36] for h in range(H))
    assert torch.allclose(null_row, (v[0].reshape(-1) @ wo.t()), rtol=1e-12, atol=1e-12)
