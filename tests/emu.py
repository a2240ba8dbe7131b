"""CPU emulation (plain torch) of the SEMANTICS of the VAE-side C-ABI operators, used by the not-gpu tests to
validate the host orchestration in muse_maskgit_pytorch_amd/vqgan_vae.py (layer order, conv geometry, weight
packing, transposed-conv parity split) against the reference's golden outputs.  Test infrastructure only."""
import torch
import torch.nn.functional as F


def conv2d_nhwc(x, w_packed, cout, th, tw, stride=1, off=(0, 0), out_hw=None, os_=1, parity=(0, 0), full_hw=None, bias=None,
                act=False, resid=None, out=None, out_nchw_f32=False):
    x = x.float()
    B, H, W, Cin = x.shape
    Hv, Wv = out_hw if out_hw is not None else (H, W)
    Hout, Wout = full_hw if full_hw is not None else (Hv * os_, Wv * os_)
    K = th * tw * Cin
    cols = torch.zeros(B, Hv, Wv, w_packed.shape[1])
    ys = torch.arange(Hv)
    xs = torch.arange(Wv)
    for ty in range(th):
        for tx in range(tw):
            iy = ys * stride + ty + off[0]
            ix = xs * stride + tx + off[1]
            vy = (iy >= 0) & (iy < H)
            vx = (ix >= 0) & (ix < W)
            patch = x[:, iy.clamp(0, H - 1)][:, :, ix.clamp(0, W - 1)]
            patch = patch * (vy[:, None] & vx[None, :])[None, :, :, None]
            t = ty * tw + tx
            cols[..., t * Cin:(t + 1) * Cin] = patch
    assert w_packed.shape[1] >= K
    y = cols @ w_packed.float().t()
    if bias is not None:
        y = y + bias
    if act:
        y = F.leaky_relu(y, 0.1)
    if out is None:
        out = torch.zeros(B, cout, Hout, Wout) if out_nchw_f32 else torch.zeros(B, Hout, Wout, cout)
    oy = ys * os_ + parity[0]
    ox = xs * os_ + parity[1]
    if out_nchw_f32:
        assert resid is None
        out[:, :, oy[:, None], ox[None, :]] = y.permute(0, 3, 1, 2)
    else:
        if resid is not None:
            y = y + resid.float()[:, oy[:, None], ox[None, :]]
        out[:, oy[:, None], ox[None, :]] = y.to(out.dtype)
    return out


def glu_nhwc(x):
    return F.glu(x.float(), dim=-1)


def groupnorm_nhwc(x, groups, gamma, beta, act=False):
    y = F.group_norm(x.float().permute(0, 3, 1, 2), groups, gamma, beta).permute(0, 2, 3, 1)
    return F.leaky_relu(y, 0.1) if act else y


def lfq_decode(ids, bits, C, w=None, b=None):
    mask = 2 ** torch.arange(bits - 1, -1, -1)
    codes = ((ids[..., None] & mask) != 0).float() * 2 - 1
    return codes @ w.t() + b if w is not None else codes


def lfq_encode(x, bits, w_in=None, b_in=None, w_out=None, b_out=None):
    t = x.float() @ w_in.t() + b_in if w_in is not None else x.float()
    pos = t > 0
    mask = 2 ** torch.arange(bits - 1, -1, -1)
    ids = (pos.long() * mask).sum(-1)
    q = pos.float() * 2 - 1
    return ids, (q @ w_out.t() + b_out if w_out is not None else q)


def nchw_to_nhwc8(img):
    B, C, H, W = img.shape
    out = torch.zeros(B, H, W, 8)
    out[..., :C] = img.permute(0, 2, 3, 1)
    return out


def nhwc_to_nchw_f32(x):
    return x.float().permute(0, 3, 1, 2).contiguous()


def install(monkeypatch, ops_module):
    """route muse_maskgit_pytorch_amd.ops.<vae op> to the emulation for a host-logic test"""
    for name in ('conv2d_nhwc', 'glu_nhwc', 'groupnorm_nhwc', 'lfq_decode', 'lfq_encode', 'nchw_to_nhwc8', 'nhwc_to_nchw_f32'):
        monkeypatch.setattr(ops_module, name, globals()[name])
