"""Training step of the transformer on MI355X: forward WITH saved activations and a hand-written backward, as one
torch.autograd.Function, so that `loss = maskgit(images_or_ids, ...); loss.backward()` works as with the reference
(muse_maskgit_pytorch.py:623-741, which differentiates Transformer.forward mmp.py:279-348 with autograd).

Every arithmetic step is a C-ABI operator of libmuse_hip.so (include/muse_hip.h): bf16 MFMA GEMMs (activation gradients
bf16, parameter gradients fp32), the MFMA attention backward, LayerNorm / GEGLU / cross-entropy / embedding gradient kernels.
torch supplies memory, the autograd tape and the optimizer; parameters stay fp32 nn.Parameters (bf16 copies are packed per
step), gradients land in .grad like with the reference.  Covers the generator (cross-entropy), the TokenCritic (BCE),
self-conditioning and super-res conditioning ids.  Linear layers:  dX = dY W  and  dW = dY^T X  are NT GEMMs on transposed copies.
"""
import torch

from . import ops

bf16 = torch.bfloat16


def _pad64(n):
    return (n + 63) // 64 * 64


def _t(x):
    """bf16 [R, C] -> contiguous [C, R64] transposed copy whose padding columns are zero (a GEMM contraction dim)."""
    return ops.transpose(x, pad_to=64)


def _wgrad(dy, x):
    """dW [N, K] = dY^T X for dY [M, N], X [M, K] (bf16) -> fp32.  Where mm_train_step reads the operands as they are (csrc/gemm_tn.hip: the small projections, the
    head) this driver does too -- same kernel, same split count, same bits."""
    if dy.stride(1) == 1 and x.stride(1) == 1 and ops.L.lib().mm_gemm_wgrad_tn_prefer(dy.shape[0], dy.shape[1], x.shape[1], dy.stride(0), x.stride(0)):
        return ops.gemm_wgrad_tn(dy, x)
    return ops.gemm_wgrad(_t(dy), _t(x))


def _dgrad(dy, w):
    """dX [M, K] = dY W for dY [M, N], W [N, K] (bf16) -> bf16."""
    return ops.gemm(dy, _t(w))


def _dgrad_long_k(dy, w):
    """dX = dY W where the contraction is LONG and the output small (the head: N = the vocabulary, dX = [labelled rows, dim] is a few dozen tiles that
    cannot fill 256 CUs): split-K like a weight gradient -- fp32 slabs summed in a fixed order -- then bf16.  Falls back to _dgrad when one split is chosen."""
    wt = _t(w)
    if dy.shape[1] % 4 == 0 and ops.L.lib().mm_gemm_wgrad_splits(dy.shape[0], wt.shape[0], dy.shape[1]) > 1:
        return ops.to_bf16(ops.gemm_wgrad(dy, wt))
    return ops.gemm(dy, wt)


def _heads(t, b, n, h, col0=0):
    """[b*n, >= col0 + h*64] row tensor -> (b, h, n, 64) strided view."""
    return t[:, col0:col0 + h * 64].unflatten(0, (b, n)).unflatten(2, (h, 64)).permute(0, 2, 1, 3)


class _Params:
    """fixed order of the parameters that receive gradients"""

    def __init__(self, tr, head=None):
        self.names, self.tensors = [], []
        tb = tr.transformer_blocks

        def add(name, p):
            self.names.append(name)
            self.tensors.append(p)
        add('token_emb', tr.token_emb.weight)
        add('pos_emb', tr.pos_emb.weight)
        self.has_proj = isinstance(tr.text_embed_proj, torch.nn.Linear)
        if self.has_proj:
            add('text_proj', tr.text_embed_proj.weight)
        for i, (sa, ca, ff) in enumerate(tb.layers):
            for tag, a in (('sa', sa), ('ca', ca)):
                for nm in ('norm.gamma', 'to_q.weight', 'to_kv.weight', 'q_scale', 'k_scale', 'null_kv', 'to_out.weight'):
                    obj = a
                    for part in nm.split('.'):
                        obj = getattr(obj, part)
                    add(f'{i}.{tag}.{nm}', obj)
            add(f'{i}.ff.g1', ff[0].gamma)
            add(f'{i}.ff.w1', ff[1].weight)
            add(f'{i}.ff.g2', ff[3].gamma)
            add(f'{i}.ff.w2', ff[4].weight)
        add('final.gamma', tb.norm.gamma)
        if head is None:
            add('to_logits', tr.to_logits.weight)
        else:                                                  # SelfCritic: Linear(dim, 1) with bias on the generator's embed (mmp.py:352-374)
            add('head.weight', head.weight)
            add('head.bias', head.bias)
        if tr.self_cond:                                       # mmp.py:237-238, 325-328
            ff = tr.self_cond_to_init_embed
            add('sc.ff.g1', ff[0].gamma)
            add('sc.ff.w1', ff[1].weight)
            add('sc.ff.g2', ff[3].gamma)
            add('sc.ff.w2', ff[4].weight)


def _ff_forward(P, pre, betas, x_in, resid, dev):
    """FeedForward (mmp.py:79-89) on fp32 rows x_in, output added to `resid`; returns (y, weights, saved)."""
    f32 = lambda t: t.detach().float().contiguous()
    w1, w2 = P[pre + 'w1'].detach(), P[pre + 'w2'].detach()
    F, D = w2.shape[1], w1.shape[1]
    Fp = _pad64(F)
    w1p = torch.zeros(2 * Fp, D, dtype=bf16, device=dev)
    w1p[:F] = w1[:F]
    w1p[Fp:Fp + F] = w1[F:]
    w2p = ops.pad_cols(w2.to(bf16), 64)
    u = ops.layernorm(x_in, f32(P[pre + 'g1']), betas[pre + 'b1'])
    h = ops.gemm(u, w1p)
    z = ops.geglu_ln(h, F, f32(P[pre + 'g2']), betas[pre + 'b2'])
    y = ops.gemm(z, w2p, out_f32=True, resid=resid)
    return y, dict(w1p=w1p, w2p=w2p), dict(x_in=x_in, u=u, h=h, z=z, F=F)


def _ff_backward(P, pre, lw, ls, dres, G, need_dx=True):
    """dres: fp32 gradient of the FF's output rows; fills G[pre + ...]; adds the input gradient into dres when need_dx."""
    f32 = lambda t: t.float().contiguous()
    F = ls['F']
    Fp = _pad64(F)
    dy = ops.to_bf16(dres)
    dz = _dgrad(dy, lw['w2p'])
    G[pre + 'w2'] = _wgrad(dy, ls['z'])[:, :F]
    dh, G[pre + 'g2'] = ops.geglu_ln_bwd(ls['h'], dz, F, f32(P[pre + 'g2']))
    dw1p = _wgrad(dh, ls['u'])
    G[pre + 'w1'] = torch.cat([dw1p[:F], dw1p[Fp:Fp + F]], 0)
    du = _dgrad(dh, lw['w1p'])
    if need_dx:
        G[pre + 'g1'] = ops.layernorm_bwd(ls['x_in'], du, f32(P[pre + 'g1']), dres)
    else:
        scratch = torch.empty_like(dres)
        G[pre + 'g1'] = ops.layernorm_bwd(ls['x_in'], du, f32(P[pre + 'g1']), scratch, accumulate=False)


class TransformerTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, ids, te, ctx_mask, labels_rows, row_index, sce, cond_ids, *params):
        """cfg: dict(depth, heads, dim, F, names, betas ...); ids int64 [b, n]; te fp32 [b, L, td]; ctx_mask uint8 [b, L];
        row_index int32 [R] = flat positions with a label, labels_rows int64 [R]."""
        P = dict(zip(cfg['names'], params))
        dev = ids.device
        b, n = ids.shape
        D, H = cfg['dim'], cfg['heads']
        I = H * 64
        M = b * n
        Lt = te.shape[1]
        W = {}           # bf16 operand copies of this step
        sv = dict(b=b, n=n, Lt=Lt, layers=[])
        f32 = lambda t: t.detach().float().contiguous()

        x = ops.embed(ids, P['token_emb'].detach().to(bf16).contiguous(), P['pos_emb'].detach().to(bf16).contiguous())       # mmp.py:322-323
        te_b = ops.to_bf16(te.reshape(b * Lt, -1).contiguous())
        if cfg['has_proj']:
            W['tp'] = P['text_proj'].detach().to(bf16).contiguous()
            cx = ops.gemm(te_b, W['tp'])                                                                             # mmp.py:302
        else:
            cx = te_b
        nc = 0
        if cond_ids is not None:                                # mmp.py:314-318: conditioning image ids join the context, always attended
            nc = cond_ids.shape[1]
            cemb = P['token_emb'].detach().to(bf16)[cond_ids.reshape(-1)].reshape(b, nc, D)
            cx = torch.cat([cx.reshape(b, Lt, D), cemb], 1).reshape(b * (Lt + nc), D).contiguous()
            ctx_mask = torch.cat([ctx_mask, torch.ones(b, nc, dtype=ctx_mask.dtype, device=dev)], 1).contiguous()
        Lc = Lt + nc
        sv['te_b'], sv['cx'], sv['Lc'] = te_b, cx, Lc
        if cfg['self_cond']:                                    # mmp.py:325-328
            if sce is None:
                sce = torch.zeros(M, D, dtype=torch.float32, device=dev)
            x, lwsc, lssc = _ff_forward(P, 'sc.ff.', cfg['betas'], sce, x, dev)
            sv['sc'] = (lwsc, lssc)
        for i in range(cfg['depth']):
            lw, ls = {}, {}
            # ---- self attention (mmp.py:137-162, 186)
            a = f'{i}.sa.'
            lw['wqkv'] = torch.cat([P[a + 'to_q.weight'].detach(), P[a + 'to_kv.weight'].detach()], 0).to(bf16).contiguous()
            lw['wo'] = P[a + 'to_out.weight'].detach().to(bf16).contiguous()
            ls['x0'] = x
            u = ops.layernorm(x, f32(P[a + 'norm.gamma']), cfg['betas'][a])
            qkv = ops.gemm(u, lw['wqkv'])
            nk, nv = f32(P[a + 'null_kv'][0, :, 0]), f32(P[a + 'null_kv'][1, :, 0])
            o = ops.attend(_heads(qkv, b, n, H), _heads(qkv, b, n, H, I), _heads(qkv, b, n, H, 2 * I), normalize=True,
                           q_scale=f32(P[a + 'q_scale']), k_scale=f32(P[a + 'k_scale']), null_k=nk, null_v=nv, out_rows=True)
            x = ops.gemm(o, lw['wo'], out_f32=True, resid=x)
            ls.update(u=u, qkv=qkv, o=o)
            # ---- cross attention (mmp.py:139-141, 155-157, 187)
            c = f'{i}.ca.'
            lw['wq2'] = P[c + 'to_q.weight'].detach().to(bf16).contiguous()
            lw['wkv2'] = P[c + 'to_kv.weight'].detach().to(bf16).contiguous()
            lw['wo2'] = P[c + 'to_out.weight'].detach().to(bf16).contiguous()
            ls['x1'] = x
            u2 = ops.layernorm(x, f32(P[c + 'norm.gamma']), cfg['betas'][c])
            q2 = ops.gemm(u2, lw['wq2'])
            kv2 = ops.gemm(cx, lw['wkv2'])
            o2 = ops.attend(_heads(q2, b, n, H), _heads(kv2, b, Lc, H), _heads(kv2, b, Lc, H, I), key_mask=ctx_mask, normalize=True,
                            q_scale=f32(P[c + 'q_scale']), k_scale=f32(P[c + 'k_scale']), null_k=f32(P[c + 'null_kv'][0, :, 0]),
                            null_v=f32(P[c + 'null_kv'][1, :, 0]), out_rows=True)
            x = ops.gemm(o2, lw['wo2'], out_f32=True, resid=x)
            ls.update(u2=u2, q2=q2, kv2=kv2, o2=o2)
            # ---- feed forward (mmp.py:79-89, 188)
            x, lwf, lsf = _ff_forward(P, f'{i}.ff.', cfg['betas'], x, x, dev)
            lw['ff'], ls['ff'] = lwf, lsf
            sv['layers'].append((lw, ls))
        hw = 'head.weight' if cfg['ext_head'] else 'to_logits'
        W['wl'] = P[hw].detach().to(bf16).contiguous()
        V = W['wl'].shape[0]
        if not cfg['bce'] and V % 64:      # a vocabulary that is not a multiple of 64: zero weight rows pad it (V is a contraction dim of the backward)
            if V % 4:
                raise NotImplementedError('MI355X training path: num_tokens must be a multiple of 4')
            wl_p = torch.zeros(_pad64(V), D, dtype=bf16, device=dev)
            wl_p[:V] = W['wl']
            W['wl'] = wl_p
        if cfg['bce']:
            # ---- TokenCritic / SelfCritic head (mmp.py:345-346, 352-374, 383-386): one logit per position, BCE against float labels
            #      over ALL positions
            e = ops.layernorm(x, f32(P['final.gamma']), cfg['betas']['final'])
            if cfg['ext_head']:
                logits = ops.conv2d_nhwc(e.reshape(M, 1, 1, D), ops.pad_cols(W['wl'], 64), 1, 1, 1, 1, (0, 0), bias=f32(P['head.bias']),
                                         out_nchw_f32=True).reshape(M, 1)
            else:
                logits = ops.gemm(e, W['wl'], out_f32=True)                 # [M, 1]
            loss = ops.bce_loss(logits.reshape(-1), labels_rows)
        else:
            # ---- head on the rows that carry a label (mmp.py:330-343): rows with ignore_index contribute nothing to the loss
            e = ops.layernorm(x, f32(P['final.gamma']), cfg['betas']['final'], row_index=row_index)
            logits = ops.gemm(e, W['wl'], out_f32=True)[:, :V]            # (a view: the padding columns are never read)
            loss = ops.ce_loss(logits, labels_rows, -100)
        sv.update(xL=x, e=e, logits=logits, W=W)
        ctx.sv, ctx.cfg = sv, cfg
        ctx.P = {k: v.detach() for k, v in P.items()}
        ctx.ids, ctx.labels_rows, ctx.row_index, ctx.ctx_mask, ctx.cond_ids = ids, labels_rows, row_index, ctx_mask, cond_ids
        logits_out = logits.detach().contiguous()
        ctx.mark_non_differentiable(logits_out)
        return loss.clone(), logits_out

    @staticmethod
    def backward(ctx, gloss, _glogits=None):
        sv, cfg, P = ctx.sv, ctx.cfg, ctx.P
        b, n, Lt, Lc = sv['b'], sv['n'], sv['Lt'], sv['Lc']
        D, H = cfg['dim'], cfg['heads']
        I = H * 64
        M = b * n
        dev = ctx.ids.device
        G = {}
        need_dcx = cfg['has_proj'] or ctx.cond_ids is not None
        f32 = lambda t: t.float().contiguous()
        # ---- head
        dres = torch.zeros(M, D, dtype=torch.float32, device=dev)
        if cfg['bce']:
            de, dwl = ops.bce_head_bwd(sv['e'], sv['logits'].reshape(-1), ctx.labels_rows, sv['W']['wl'].float().reshape(-1))
            if cfg['ext_head']:
                G['head.weight'] = dwl.reshape(1, D)
                G['head.bias'] = ((torch.sigmoid(sv['logits'].reshape(-1)) - ctx.labels_rows).sum() / M).reshape(1)      # 1 scalar: host-side reduction
            else:
                G['to_logits'] = dwl.reshape(1, D)
            G['final.gamma'] = ops.layernorm_bwd(sv['xL'], de, f32(P['final.gamma']), dres, accumulate=False)
        else:
            R = ctx.row_index.numel()
            dl = ops.ce_bwd(sv['logits'], ctx.labels_rows, 1.0 / R, pad_to=64)
            G['to_logits'] = _wgrad(dl, sv['e'])[:sv['logits'].shape[1]]
            de = _dgrad_long_k(dl, sv['W']['wl'])
            G['final.gamma'] = ops.layernorm_bwd(sv['xL'], de, f32(P['final.gamma']), dres, accumulate=False, row_index=ctx.row_index)
        dcx = None
        sync = cfg.get('sync')
        if sync is not None:
            head_keys = [k for k in ('to_logits', 'head.weight', 'head.bias', 'final.gamma') if k in G]
            for k in head_keys:
                G[k] = G[k].contiguous()
            sync.push([G[k] for k in head_keys])
        for i in reversed(range(cfg['depth'])):
            lw, ls = sv['layers'][i]
            # ---- feed forward
            _ff_backward(P, f'{i}.ff.', lw['ff'], ls['ff'], dres, G)
            # ---- cross attention
            c = f'{i}.ca.'
            dy = ops.to_bf16(dres)
            do2 = _dgrad(dy, lw['wo2'])
            G[c + 'to_out.weight'] = _wgrad(dy, ls['o2'])
            qs, ks = f32(P[c + 'q_scale']), f32(P[c + 'k_scale'])
            nk, nv = f32(P[c + 'null_kv'][0, :, 0]), f32(P[c + 'null_kv'][1, :, 0])
            dqn, dkn, dv, dnk, dnv = ops.attention_bwd(_heads(ls['q2'], b, n, H), _heads(ls['kv2'], b, Lc, H), _heads(ls['kv2'], b, Lc, H, I),
                                                       _heads(ls['o2'], b, n, H), _heads(do2, b, n, H), qs, ks, nk, nv, key_mask=ctx.ctx_mask)
            dq2, dqs = ops.qk_norm_bwd(ls['q2'], dqn.reshape(M, I), qs, H)
            dk2, dks = ops.qk_norm_bwd(ls['kv2'], dkn.reshape(b * Lc, I), ks, H)
            dnull, dks_n = ops.qk_norm_bwd(None, None, ks, H, x_f32=nk, dy_f32=dnk)
            G[c + 'q_scale'], G[c + 'k_scale'] = dqs, dks + dks_n
            G[c + 'null_kv'] = torch.stack([ops.colsum(dnull.reshape(b, H * 64)), ops.colsum(dnv.reshape(b, H * 64))], 0).reshape(2, H, 1, 64)
            dkv2 = torch.cat([dk2, dv.reshape(b * Lc, I)], 1)
            du2 = _dgrad(dq2, lw['wq2'])
            G[c + 'to_q.weight'] = _wgrad(dq2, ls['u2'])
            G[c + 'to_kv.weight'] = _wgrad(dkv2, sv['cx'])
            if need_dcx:
                dcx = ops.gemm(dkv2, _t(lw['wkv2']), out_f32=True, resid=dcx)
            G[c + 'norm.gamma'] = ops.layernorm_bwd(ls['x1'], du2, f32(P[c + 'norm.gamma']), dres)
            # ---- self attention
            a = f'{i}.sa.'
            dy = ops.to_bf16(dres)
            do = _dgrad(dy, lw['wo'])
            G[a + 'to_out.weight'] = _wgrad(dy, ls['o'])
            qs, ks = f32(P[a + 'q_scale']), f32(P[a + 'k_scale'])
            nk, nv = f32(P[a + 'null_kv'][0, :, 0]), f32(P[a + 'null_kv'][1, :, 0])
            qkv = ls['qkv']
            dqn, dkn, dv, dnk, dnv = ops.attention_bwd(_heads(qkv, b, n, H), _heads(qkv, b, n, H, I), _heads(qkv, b, n, H, 2 * I),
                                                       _heads(ls['o'], b, n, H), _heads(do, b, n, H), qs, ks, nk, nv)
            dq, dqs = ops.qk_norm_bwd(qkv, dqn.reshape(M, I), qs, H)
            dk, dks = ops.qk_norm_bwd(qkv[:, I:], dkn.reshape(M, I), ks, H)
            dnull, dks_n = ops.qk_norm_bwd(None, None, ks, H, x_f32=nk, dy_f32=dnk)
            G[a + 'q_scale'], G[a + 'k_scale'] = dqs, dks + dks_n
            G[a + 'null_kv'] = torch.stack([ops.colsum(dnull.reshape(b, H * 64)), ops.colsum(dnv.reshape(b, H * 64))], 0).reshape(2, H, 1, 64)
            dqkv = torch.cat([dq, dk, dv.reshape(M, I)], 1)
            du = _dgrad(dqkv, lw['wqkv'])
            dwqkv = _wgrad(dqkv, ls['u'])
            G[a + 'to_q.weight'], G[a + 'to_kv.weight'] = dwqkv[:I], dwqkv[I:]
            G[a + 'norm.gamma'] = ops.layernorm_bwd(ls['x0'], du, f32(P[a + 'norm.gamma']), dres)
            if sync is not None:      # this layer's gradients are final: average them across ranks while the layers below run
                sync.push([G[k] for k in cfg['names'] if k.startswith(f'{i}.')])
        if cfg['self_cond']:
            _ff_backward(P, 'sc.ff.', sv['sc'][0], sv['sc'][1], dres, G, need_dx=False)       # the embed fed back is detached (mmp.py:707)
        # ---- embeddings / text projection
        G['token_emb'], G['pos_emb'] = ops.embed_bwd(ctx.ids, dres, P['token_emb'].shape[0])
        if ctx.cond_ids is not None:                            # the conditioning ids were embedded with the same table (mmp.py:316)
            nc = ctx.cond_ids.shape[1]
            dcond = dcx.reshape(b, Lc, D)[:, Lt:].reshape(b * nc, D).contiguous()
            ops.embed_bwd(ctx.cond_ids, dcond, P['token_emb'].shape[0], dtoken=G['token_emb'])
        if n < P['pos_emb'].shape[0]:
            full = torch.zeros_like(P['pos_emb'], dtype=torch.float32)
            full[:n] = G['pos_emb']
            G['pos_emb'] = full
        if cfg['has_proj']:
            G['text_proj'] = _wgrad(ops.to_bf16(dcx.reshape(b, Lc, D)[:, :Lt].reshape(b * Lt, D).contiguous()), sv['te_b'])
        if sync is not None:
            sync.push([G[k] for k in ('token_emb', 'pos_emb', 'text_proj') if k in G])
            sync.finish()
        grads = []
        for name in cfg['names']:
            g = G[name].to(P[name].dtype).reshape(P[name].shape)
            grads.append(g * gloss if gloss.numel() == 1 and float(gloss) != 1.0 else g)
        ctx.sv = None
        return (None, None, None, None, None, None, None, None, *grads)


class TrainStepFn(torch.autograd.Function):
    """The same step as TransformerTrainFn through ONE C call (mm_train_step, csrc/train_step.hip): forward, loss and the whole backward run
    on the stream without returning to Python; the gradients are kept and handed to autograd in backward() (scaled by the incoming gradient of
    the loss).  Bit-identical to the operator-by-operator driver above (tests/test_gpu_train_step.py)."""

    @staticmethod
    def forward(ctx, cfg, ids, te, ctx_mask, labels_rows, row_index, *params):
        from . import _lib as L
        import ctypes as C
        tr = cfg['tr']
        P = dict(zip(cfg['names'], params))
        dev = ids.device
        b, n = ids.shape
        tb = tr.transformer_blocks
        depth, D, H = cfg['depth'], cfg['dim'], cfg['heads']
        R = row_index.numel()
        V = P['to_logits'].shape[0]
        G = {k: torch.empty(v.shape, dtype=torch.float32, device=dev) for k, v in P.items()}
        keep = [v.detach().float().contiguous() for v in params]      # (fp32 contiguous parameters: no copies in the usual case)
        Pc = dict(zip(cfg['names'], keep))
        layers = (L.TrainLayer * depth)()
        betas = cfg['betas']
        for i in range(depth):
            for tag, fld in (('sa', layers[i].sa), ('ca', layers[i].ca)):
                a = f'{i}.{tag}.'
                fld.gamma, fld.beta = L.ptr(Pc[a + 'norm.gamma']), L.ptr(betas[a])
                fld.to_q, fld.to_kv, fld.to_out = L.ptr(Pc[a + 'to_q.weight']), L.ptr(Pc[a + 'to_kv.weight']), L.ptr(Pc[a + 'to_out.weight'])
                fld.q_scale, fld.k_scale, fld.null_kv = L.ptr(Pc[a + 'q_scale']), L.ptr(Pc[a + 'k_scale']), L.ptr(Pc[a + 'null_kv'])
                fld.d_gamma, fld.d_to_q, fld.d_to_kv, fld.d_to_out = L.ptr(G[a + 'norm.gamma']), L.ptr(G[a + 'to_q.weight']), L.ptr(G[a + 'to_kv.weight']), L.ptr(G[a + 'to_out.weight'])
                fld.d_q_scale, fld.d_k_scale, fld.d_null_kv = L.ptr(G[a + 'q_scale']), L.ptr(G[a + 'k_scale']), L.ptr(G[a + 'null_kv'])
            f = f'{i}.ff.'
            ff = layers[i].ff
            ff.g1, ff.b1, ff.w1 = L.ptr(Pc[f + 'g1']), L.ptr(betas[f + 'b1']), L.ptr(Pc[f + 'w1'])
            ff.g2, ff.b2, ff.w2 = L.ptr(Pc[f + 'g2']), L.ptr(betas[f + 'b2']), L.ptr(Pc[f + 'w2'])
            ff.d_g1, ff.d_w1, ff.d_g2, ff.d_w2 = L.ptr(G[f + 'g1']), L.ptr(G[f + 'w1']), L.ptr(G[f + 'g2']), L.ptr(G[f + 'w2'])
        d = L.TrainDesc()
        d.dim, d.depth, d.heads, d.ff_inner = D, depth, H, P['0.ff.w2'].shape[1]
        d.seq_len, d.vocab_rows, d.dim_out, d.text_dim = P['pos_emb'].shape[0], P['token_emb'].shape[0], V, te.shape[-1]
        d.token_emb, d.pos_emb, d.to_logits = L.ptr(Pc['token_emb']), L.ptr(Pc['pos_emb']), L.ptr(Pc['to_logits'])
        d.final_gamma, d.final_beta = L.ptr(Pc['final.gamma']), L.ptr(betas['final'])
        d.d_token_emb, d.d_pos_emb, d.d_final_gamma, d.d_to_logits = L.ptr(G['token_emb']), L.ptr(G['pos_emb']), L.ptr(G['final.gamma']), L.ptr(G['to_logits'])
        if cfg['has_proj']:
            d.text_proj, d.d_text_proj = L.ptr(Pc['text_proj']), L.ptr(G['text_proj'])
        d.layers = C.cast(layers, C.POINTER(L.TrainLayer))
        Lt = te.shape[1]
        lib = L.lib()
        wsb = lib.mm_train_step_workspace_bytes(C.byref(d), b, n, Lt, R)
        if wsb == 0:
            raise L.MuseHipError('mm_train_step: ' + (lib.mm_last_error() or b'').decode())
        ws = getattr(tr, '_train_ws', None)
        if ws is None or ws.numel() < wsb or ws.device != dev:
            ws = tr._train_ws = torch.empty(int(wsb), dtype=torch.uint8, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        logits = torch.empty(R, V, dtype=torch.float32, device=dev) if cfg['want_logits'] else None
        L.check(lib.mm_train_step(C.byref(d), L.stream(), L.ptr(ids), b, n, L.ptr(te), Lt, L.ptr(ctx_mask), L.ptr(row_index), L.ptr(labels_rows), R,
                                  L.ptr(loss), L.ptr(logits), L.ptr(ws), ws.numel()), 'mm_train_step')
        ctx.grads = [G[k] for k in cfg['names']]
        ctx.dtypes = [p.dtype for p in params]
        if logits is None:
            logits = torch.empty(0, device=dev)
        ctx.mark_non_differentiable(logits)
        return loss[0].clone(), logits

    @staticmethod
    def backward(ctx, gloss, _glogits=None):
        # the gradients were computed by the C step with d loss = 1; the incoming scale is applied ON THE DEVICE (one fused multi-tensor multiply) -- reading it on the
        # host, as round 4 did, is a synchronisation per step.  ctx.grads stays alive with the ctx: a second backward (retain_graph) sees the same gradients.
        grads = torch._foreach_mul(ctx.grads, gloss.reshape(()).to(ctx.grads[0].dtype))
        return (None, None, None, None, None, None, *[g.to(dt) for g, dt in zip(grads, ctx.dtypes)])


def _c_step_eligible(tr, ids, te, head, bce, sce, cond_ids, grad_sync):
    """shapes / variants mm_train_step covers (include/muse_hip.h); everything else runs on the operator-by-operator driver above"""
    import os
    if os.environ.get('MM_TRAIN_PY'):      # A/B and the bit-identity test
        return False
    if not torch.is_grad_enabled() or not any(p.requires_grad for p in tr.parameters()):
        return False                        # (evaluation / validation loss: the C entry computes the whole backward -- the operator driver runs the forward only)
    b, n = ids.shape
    return (head is None and not bce and sce is None and cond_ids is None and grad_sync is None and not tr.self_cond and n in (64, 128, 256)
            and (b * n) % 64 == 0 and tr.dim % 64 == 0 and tr.dim_out % 64 == 0 and te.shape[-1] % 64 == 0 and te.shape[1] > 0)


def transformer_loss(tr, ids, text_embeds, labels, ignore_index, cond_drop_prob, grad_sync=None, return_logits=False,
                     self_cond_embed=None, conditioning_token_ids=None, head=None):
    """Differentiable CE loss of Transformer.forward(labels=...) (mmp.py:337-346) on the MI355X training path.
    grad_sync: an optional parallel.GradBucketer -- data-parallel gradient averaging overlapped with the backward."""
    dev = tr.token_emb.weight.device
    ids = ids.to(device=dev, dtype=torch.long).contiguous()
    b, n = ids.shape
    if tr.transformer_blocks.cfg['dim_head'] != 64:
        raise NotImplementedError('the training path (attention backward) is written for dim_head 64; inference covers 32 / 64 / 128')
    assert (b * n) % 64 == 0, 'the training path needs batch * seq_len to be a multiple of 64'
    te = text_embeds.to(device=dev, dtype=torch.float32).contiguous()
    ctx_mask = (te != 0).any(dim=-1)                                                   # mmp.py:304
    if cond_drop_prob >= 1.:
        ctx_mask = torch.zeros_like(ctx_mask)
    elif cond_drop_prob > 0.:                                                          # mmp.py:308-310, 393-399
        ctx_mask = ctx_mask & (torch.rand((b, 1), device=dev) < (1. - cond_drop_prob))
    bce = tr.dim_out == 1 or head is not None      # `head`: an external Linear(dim, 1) on the embed (SelfCritic)
    if bce:                                                                            # TokenCritic: float targets at every position
        labels_rows = labels.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        row_index = torch.zeros(1, dtype=torch.int32, device=dev)
    else:
        labels = labels.to(device=dev, dtype=torch.long).reshape(-1)
        row_index = torch.nonzero(labels != ignore_index).reshape(-1).to(torch.int32).contiguous()
        assert row_index.numel() > 0, 'no position carries a label'
        labels_rows = labels[row_index.long()].contiguous()
    tr._last_train_rows = int(row_index.numel()) if not bce else b * n      # (bench.py: executed flops of the step)
    pr = _Params(tr, head)
    tb = tr.transformer_blocks
    betas = {'final': tb.norm.beta.float().contiguous()}
    for i, (sa, ca, ff) in enumerate(tb.layers):
        betas[f'{i}.sa.'] = sa.norm.beta.float().contiguous()
        betas[f'{i}.ca.'] = ca.norm.beta.float().contiguous()
        betas[f'{i}.ff.b1'] = ff[0].beta.float().contiguous()
        betas[f'{i}.ff.b2'] = ff[3].beta.float().contiguous()
    betas['sc.ff.b1'] = tr.self_cond_to_init_embed[0].beta.float().contiguous()
    betas['sc.ff.b2'] = tr.self_cond_to_init_embed[3].beta.float().contiguous()
    sce = None
    if tr.self_cond and self_cond_embed is not None:
        sce = self_cond_embed.detach().to(device=dev, dtype=torch.float32).reshape(b * n, tr.dim).contiguous()
    cond_ids = None
    if conditioning_token_ids is not None:
        cond_ids = conditioning_token_ids.reshape(b, -1).to(device=dev, dtype=torch.long).contiguous()
    cfg = dict(depth=tb.cfg['depth'], heads=tb.cfg['heads'], dim=tr.dim, names=pr.names, has_proj=pr.has_proj, betas=betas,
               sync=grad_sync, bce=bce, self_cond=bool(tr.self_cond), ext_head=head is not None)
    if _c_step_eligible(tr, ids, te, head, bce, sce, cond_ids, grad_sync):      # the whole step as one C call (mm_train_step)
        cfg.update(tr=tr, want_logits=return_logits)
        loss, logits = TrainStepFn.apply(cfg, ids, te, ctx_mask.to(torch.uint8).contiguous(), labels_rows, row_index, *pr.tensors)
        return (loss, logits, row_index) if return_logits else loss
    loss, logits = TransformerTrainFn.apply(cfg, ids, te, ctx_mask.to(torch.uint8).contiguous(), labels_rows, row_index, sce, cond_ids, *pr.tensors)
    if return_logits:
        return loss, logits, row_index          # logits of the labelled rows only (CE) / of every position (BCE)
    return loss
