"""CPU study (build container, no GPU): does a split-bf16 GEMM operand format hold the reference's ids at BASELINE configs[1] size?

Runs the oracle's 18-step decode of tests/golden/base_c2.pt (B = 2, V = 65536, the reference's own noise) with every GEMM activation
operand rounded to the sum of `terms` bf16 values (terms = 2: x ~ h + l, 16-17 significant bits; terms = 3: h + m + l, 24 bits = fp32) and
reports the agreement with the reference run.  The fixture's weights are bf16-representable, so no weight-side split enters; the flag
--wsplit rounds an fp32-perturbed copy of the weights to 2 bf16 terms as well and drops the lo*lo product (what the general 3-product form does).
Attention (QK^T, softmax, PV) stays fp32, as in the tier under study.     python tools/split_precision_study.py --terms 2
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import golden_recipe as R  # noqa: E402
import muse_oracle as O  # noqa: E402


def split(t, terms):
    if terms <= 0:
        return t
    acc = torch.zeros_like(t)
    r = t
    for _ in range(terms):
        h = r.to(torch.bfloat16).float()
        acc = acc + h
        r = r - h
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--terms', type=int, default=2)
    ap.add_argument('--steps', type=int, default=R.T)
    args = ap.parse_args()
    import muse_maskgit_pytorch_amd as mm
    g = torch.load(os.path.join(ROOT, 'tests', 'golden', 'base_c2.pt'))
    tr = R.build_transformer(mm.MaskGitTransformer, peaky=True)
    sd = {k: v.detach() for k, v in tr.state_dict().items()}
    inp = R.inputs()
    te = inp['text_embeds']
    rp = lambda t: split(t, args.terms)
    depth, heads = 8, 8

    def lin(x, w):
        return rp(x) @ w.t()

    def attn(x, p, context=None, cmask=None):
        b, n, _ = x.shape
        xn = O.layer_norm(x, sd[p + 'norm.gamma'], sd[p + 'norm.beta'])
        kv_in = context if context is not None else xn
        q = lin(xn, sd[p + 'to_q.weight'])
        kv = lin(kv_in, sd[p + 'to_kv.weight'])
        k, v = kv.chunk(2, dim=-1)
        sp = lambda t: t.reshape(t.shape[0], t.shape[1], heads, -1).permute(0, 2, 1, 3)
        q, k, v = sp(q), sp(k), sp(v)
        nk, nv = sd[p + 'null_kv']
        k = torch.cat((nk[None].expand(b, -1, -1, -1), k), dim=-2)
        v = torch.cat((nv[None].expand(b, -1, -1, -1), v), dim=-2)
        q = F.normalize(q, dim=-1) * sd[p + 'q_scale']
        k = F.normalize(k, dim=-1) * sd[p + 'k_scale']
        mask = None
        if cmask is not None:
            mask = F.pad(cmask[:, None, None, :].expand(b, heads, n, -1), (1, 0), value=True)
        out = O.attend(q, k, v, mask=mask).permute(0, 2, 1, 3).reshape(b, n, -1)
        return lin(out, sd[p + 'to_out.weight'])

    def ff(x, p):
        h = O.layer_norm(x, sd[p + '0.gamma'], sd[p + '0.beta'])
        h = O.geglu(lin(h, sd[p + '1.weight']))
        h = O.layer_norm(h, sd[p + '3.gamma'], sd[p + '3.beta'])
        return lin(h, sd[p + '4.weight'])

    def forward(ids, drop):
        b, n = ids.shape
        cmask = (te != 0).any(dim=-1)
        if drop:
            cmask = cmask & torch.zeros((b, 1), dtype=torch.bool)
        x = sd['token_emb.weight'][ids] + sd['pos_emb.weight'][torch.arange(n)]
        for i in range(depth):
            p = f'transformer_blocks.layers.{i}.'
            x = attn(x, p + '0.') + x
            x = attn(x, p + '1.', context=te, cmask=cmask) + x
            x = ff(x, p + '2.') + x
        e = O.layer_norm(x, sd['transformer_blocks.norm.gamma'], sd['transformer_blocks.norm.beta'])
        return lin(e, sd['to_logits.weight'])

    noise = R.noise_stream()
    gen = g['generate']
    ref_in = gen['step_in_ids'].long()
    worst = [0.]
    t0 = time.time()

    def demask(ids, step):
        agree = (ids == ref_in[step]).float().mean().item()
        with torch.no_grad():
            lc = forward(ids, False)
            ln = forward(ids, True)
        print(f'step {step}: input state agreement {100 * agree:.3f} %  ({time.time() - t0:.0f}s)', flush=True)
        return ln + (lc - ln) * 3.

    def gumbel(step, shape):
        return O.gumbel_from_uniform(next(noise))

    ids = O.generate_ids(demask, R.B, R.N, 65536, gumbel, timesteps=R.T)
    final = (ids.reshape(gen['final_ids'].shape) == gen['final_ids']).float().mean().item()
    print(f'terms={args.terms}: final ids equal to the reference run: {100 * final:.3f} %')


if __name__ == '__main__':
    main()
