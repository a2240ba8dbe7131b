"""CPU checks of the fp8 engine's self-defined oracle (oracle/muse_oracle.py e4m3_rows / Fp8Rounding): the fake quantiser is the rule the kernels implement
(scale = max |row| / 448, 1 for a zero row, round to nearest even on the e4m3 grid, idempotent), and it is applied at exactly the seven Linear inputs of a
layer the engine quantises -- not at the context's k | v projection of the cross-attention, not at to_logits."""
import torch

import muse_oracle as O


def test_e4m3_rows_rule():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 300, generator=g) * torch.tensor([1e-3, 1., 40., 0., 7.])[:, None]
    q = O.e4m3_rows(x)
    assert torch.equal(q[3], torch.zeros(300))                                   # zero row: scale 1, stays zero
    assert torch.equal(O.e4m3_rows(q), q)                                        # idempotent: the grid values are fixed points
    amax = x.abs().amax(-1)
    assert torch.equal(q.abs().amax(-1)[[0, 1, 2, 4]], amax[[0, 1, 2, 4]])       # the row maximum maps to 448 * scale exactly
    rel = ((q - x).abs() / amax.clamp_min(1e-30)[:, None])
    assert rel.max().item() <= 2 ** -4                                           # e4m3: 3 mantissa bits -> half a step of 2^-3 at the top binade
    # against torch's own cast with the same scale
    scale = torch.where(amax > 0, amax / 448., torch.ones_like(amax))
    assert torch.equal(q, (x / scale[:, None]).to(torch.float8_e4m3fn).float() * scale[:, None])


def test_fp8_rounding_hooks_the_seven_linear_inputs_of_a_layer():
    torch.manual_seed(0)
    D, H, dh, F, V, n, L = 128, 2, 64, 341, 64, 6, 3
    I = H * dh
    sd = {'token_emb.weight': torch.randn(V + 1, D), 'pos_emb.weight': torch.randn(n, D), 'to_logits.weight': torch.randn(V, D) * 0.1,
          'transformer_blocks.norm.gamma': torch.ones(D), 'transformer_blocks.norm.beta': torch.zeros(D)}
    for j in (0, 1):
        p = f'transformer_blocks.layers.0.{j}.'
        sd.update({p + 'norm.gamma': torch.ones(D), p + 'norm.beta': torch.zeros(D), p + 'to_q.weight': torch.randn(I, D) * 0.05,
                   p + 'to_kv.weight': torch.randn(2 * I, D) * 0.05, p + 'to_out.weight': torch.randn(D, I) * 0.05, p + 'null_kv': torch.randn(2, H, 1, dh),
                   p + 'q_scale': torch.ones(dh), p + 'k_scale': torch.ones(dh)})
    p = 'transformer_blocks.layers.0.2.'
    sd.update({p + '0.gamma': torch.ones(D), p + '0.beta': torch.zeros(D), p + '1.weight': torch.randn(2 * F, D) * 0.05, p + '3.gamma': torch.ones(F),
               p + '3.beta': torch.zeros(F), p + '4.weight': torch.randn(D, F) * 0.05})

    class Counting(O.Fp8Rounding):
        shapes = []

        @staticmethod
        def gq(t):
            Counting.shapes.append(tuple(t.shape))
            return O.e4m3_rows(t)
    ids = torch.randint(0, V, (2, n))
    te = torch.randn(2, L, D)
    out = O.transformer_forward(sd, dict(depth=1, heads=H), ids, te, 0., rp=Counting())
    assert out.shape == (2, n, V) and torch.isfinite(out).all()
    # self-attention: LN(x) (feeds q AND k|v: one quantised activation), attention output; cross-attention: LN(x) for q, attention output (the context is
    # NOT quantised); feed-forward: LN(x), LN(inner)  ->  6 quantiser calls for the 7 Linear weights of the layer
    assert Counting.shapes == [(2, n, D), (2, n, I), (2, n, D), (2, n, I), (2, n, D), (2, n, F)]
    plain = O.transformer_forward(sd, dict(depth=1, heads=H), ids, te, 0., rp=O.bf16_round)
    assert (out - plain).abs().max() > 0                                         # the mode is really on
