"""CPU study (build container, no GPU): does a split-bf16 GEMM operand format hold the reference's ids at BASELINE configs[1] size?

Runs the oracle's 18-step decode of tests/golden/base_c2.pt (B = 2, V = 65536, the reference's own noise) with every GEMM activation
operand rounded to the sum of `terms` bf16 values (terms = 2: x ~ h + l, 16-17 significant bits; terms = 3: h + m + l, 24 bits = fp32) and
reports the agreement with the reference run.  The fixture's weights are bf16-representable, so no weight-side split enters; the flag
--wsplit rounds an fp32-perturbed copy of the weights to 2 bf16 terms as well and drops the lo*lo product (what the general 3-product form does).
Attention (QK^T, softmax, PV) stays fp32, as in the tier under study.     python tools/split_precision_study.py --terms 2

Round 4: `--fixture base_c2_fp32.pt` runs the study on the GENERAL fp32 checkpoint (weights split as well), `--fmt f16` splits into fp16 terms
(11 significand bits each: two terms carry 22 bits) with `--wterms` weight terms and the term pairs i + j <= `--order` kept, `--ftz` flushes fp16
subnormal terms (what a matrix pipe without fp16 denormal support would do), `--wscale` multiplies each weight matrix by a power of two before
the split (and the product by its inverse).       python tools/split_precision_study.py --fixture base_c2_fp32.pt --fmt f16 --terms 2 --wterms 2 --order 1
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


def term_list(t, terms, fmt, ftz=False):
    """t as a list of `terms` fp32 tensors holding its successive bf16 / fp16 terms"""
    dt = torch.bfloat16 if fmt == 'bf16' else torch.float16
    out, r = [], t
    for _ in range(terms):
        h = r.to(dt).float()
        if ftz and fmt == 'f16':
            h = torch.where(h.abs() < 2.0 ** -14, torch.zeros_like(h), h)
        out.append(h)
        r = r - h
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--terms', type=int, default=2)
    ap.add_argument('--steps', type=int, default=R.T)
    ap.add_argument('--fixture', default='base_c2.pt')
    ap.add_argument('--fmt', default='bf16', choices=['bf16', 'f16'])
    ap.add_argument('--wterms', type=int, default=0, help='weight terms (0: weights as they are)')
    ap.add_argument('--order', type=int, default=2, help='keep the term pairs with i + j <= order')
    ap.add_argument('--ftz', action='store_true')
    ap.add_argument('--wscale', type=float, default=1.0)
    args = ap.parse_args()
    import muse_maskgit_pytorch_amd as mm
    g = torch.load(os.path.join(ROOT, 'tests', 'golden', args.fixture))
    tr = R.build_transformer(mm.MaskGitTransformer, peaky=True, bf16_weights=g['recipe'].get('bf16_weights', True))
    sd = {k: v.detach() for k, v in tr.state_dict().items()}
    inp = R.inputs(g['recipe'].get('input_seed'))
    te = inp['text_embeds']
    rp = lambda t: split(t, args.terms)
    depth, heads = 8, 8
    wcache = {}

    def lin(x, w):
        if args.wterms <= 0 and args.fmt == 'bf16':
            return rp(x) @ w.t()
        key = id(w)
        if key not in wcache:
            wcache[key] = [t.t().contiguous() for t in term_list(w * args.wscale, max(args.wterms, 1), args.fmt, args.ftz)] if args.wterms > 0 else [(w * args.wscale).t().contiguous()]
        xs = term_list(x, args.terms, args.fmt, args.ftz)
        acc = None
        for i, xi in enumerate(xs):
            for j, wj in enumerate(wcache[key]):
                if i + j <= args.order:
                    p = xi @ wj
                    acc = p if acc is None else acc + p
        return acc / args.wscale

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

    # forward logits against the reference's recorded ones (plain, not peaky, to_logits: the fixture's unit-scale logits)
    with torch.no_grad():
        wl = sd['to_logits.weight']
        sd['to_logits.weight'] = wl / R.PEAK
        lc = forward(inp['ids'], False).reshape(R.B * R.N, -1)
        sd['to_logits.weight'] = wl
        wcache.clear()
    fw = g['forward']['logits_cond']
    e1 = (lc[g['full_rows']] - fw['rows']).abs().max().item()
    e2 = (lc[:, ::g['col_stride']] - fw['cols']).abs().max().item()
    print(f'forward logits(cond) vs the reference: max abs err {max(e1, e2):.3g} (logits absmax {fw["absmax"]:.3g})', flush=True)
    if args.steps <= 0:
        return
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
    print(f'{args.fixture} fmt={args.fmt} terms={args.terms} wterms={args.wterms} order={args.order} ftz={args.ftz} wscale={args.wscale}: final ids equal to the reference run: {100 * final:.3f} %')


if __name__ == '__main__':
    main()
