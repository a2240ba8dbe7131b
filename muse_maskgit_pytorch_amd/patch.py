"""Drop-in mode 2: `patch_reference()` swaps the hot methods ON THE REFERENCE'S OWN CLASSES, in place, so that objects
built by unmodified user code -- `muse_maskgit_pytorch.MaskGit(vae=VQGanVAE(...), transformer=MaskGitTransformer(...))`, `Muse(...)`, anything
`@beartype`-checked with `isinstance` (muse_maskgit_pytorch.py:427, 745) -- run their hot path on MI355X.

Patched (reference file:line -> replacement):
    attend.py:109                   Attend.forward                     -> mm_attend (csrc/attention.hip)
    muse_maskgit_pytorch.py:279     Transformer.forward                -> Transformer.forward of this package (mm_transformer_forward / training.py)
    muse_maskgit_pytorch.py:240     Transformer.forward_with_cond_scale-> both passes + the fused guidance GEMM
    muse_maskgit_pytorch.py:491     MaskGit.generate                   -> mm_generate (whole decode loop in one call)
    vqgan_vae.py:422,427,440        VQGanVAE.encode / decode_from_ids / decode

Mechanism: every reference instance gets, on first use, a SHADOW instance of this package's class whose nn.Parameters / buffers ARE the
reference instance's tensors (same objects, matched by state-dict key: the key names are identical by construction), so there is no second
copy of the weights, optimizers keep updating the tensors the reference module owns, and the packed bf16 copies are rebuilt whenever
those tensors change (Transformer._pack_key).  Nothing here computes: without libmuse_hip.so or without a gfx950 device the patched
methods raise, exactly like mode 1.
"""
import torch
from torch import nn

_PATCHED = {}          # (class, attribute) -> original function
_SHADOW = '_mm_amd_shadow'


def _share_tensors(shadow, ref):
    """make shadow's parameters / buffers the very tensors of ref (matched by qualified name; strict)."""
    ref_params, ref_bufs = dict(ref.named_parameters()), dict(ref.named_buffers())
    own_params, own_bufs = dict(shadow.named_parameters()), dict(shadow.named_buffers())
    missing = (set(own_params) - set(ref_params)) | (set(own_bufs) - set(ref_bufs))
    if missing:
        raise RuntimeError(f'patch_reference: the reference module has no tensor for {sorted(missing)[:5]}')
    for name, t in list(ref_params.items()) + list(ref_bufs.items()):
        if name not in own_params and name not in own_bufs:
            continue                      # e.g. the reference VAE's discriminator / VGG: not on the hot path
        mod = shadow
        *path, leaf = name.split('.')
        for p in path:
            mod = getattr(mod, p)
        if tuple(getattr(mod, leaf).shape) != tuple(t.shape):
            raise RuntimeError(f'patch_reference: shape mismatch for {name}: {tuple(getattr(mod, leaf).shape)} vs {tuple(t.shape)}')
        if name in own_params:
            mod._parameters[leaf] = t
        else:
            mod._buffers[leaf] = t
    return shadow


def _transformer_shadow(ref):
    from . import muse_maskgit as M
    from .t5 import _KNOWN_DIMS
    sh = ref.__dict__.get(_SHADOW)
    if sh is not None:
        return sh
    layers = ref.transformer_blocks.layers
    heads = layers[0][0].heads
    inner = layers[0][0].to_q.weight.shape[0]
    F = layers[0][2][4].weight.shape[1]
    dim = ref.dim
    ff_mult = next((m for m in (4, 2, 8, F * 1.5 / dim, (F + 0.5) * 1.5 / dim) if int(dim * m * 2 / 3) == F), None)
    if ff_mult is None:
        raise RuntimeError(f'patch_reference: cannot express the feed-forward width {F} as int(dim * mult * 2 / 3)')
    text_dim = ref.text_embed_proj.weight.shape[1] if isinstance(ref.text_embed_proj, nn.Linear) else dim
    name = getattr(getattr(ref, 'encode_text', None), 'keywords', {}).get('name')
    if name is None or _KNOWN_DIMS.get(name, text_dim) != text_dim:
        name = next(k for k, v in _KNOWN_DIMS.items() if v == text_dim)
    cls = M.TokenCritic if ref.dim_out == 1 and ref.mask_id is None else (M.MaskGitTransformer if ref.mask_id is not None else M.Transformer)
    kw = dict(num_tokens=ref.num_tokens, dim=dim, seq_len=ref.seq_len, t5_name=name, self_cond=ref.self_cond, depth=len(layers), dim_head=inner // heads,
              heads=heads, ff_mult=ff_mult)
    if cls is M.Transformer:
        kw.update(dim_out=ref.dim_out, add_mask_id=False)
    with torch.device('meta'):
        sh = cls(**kw)                    # structure only; its tensors are replaced below
    _share_tensors(sh, ref)
    sh.train(ref.training)
    sh.__dict__['encode_text'] = lambda texts, _ref=ref: _ref.encode_text(texts)      # per-instance attribute in the reference (mmp.py:229)
    ref.__dict__[_SHADOW] = sh
    return sh


def _vae_shadow(ref):
    from .vqgan_vae import VQGanVAE, ResBlock
    sh = ref.__dict__.get(_SHADOW)
    if sh is not None:
        return sh
    ed = ref.enc_dec
    stem = ed.encoders[0]
    dim, layers = stem.out_channels, ed.layers
    downs, blocks, groups = [], [], 16
    for m in list(ed.encoders)[1:]:
        if isinstance(m, nn.Sequential):
            downs.append(m[0].out_channels)
            blocks.append(0)
        else:                              # the reference's ResBlock
            blocks[-1] += 1
            groups = m.net[1].num_groups
    mults = [c // dim for c in downs]
    with torch.device('meta'):
        sh = VQGanVAE(dim=dim, channels=stem.in_channels, layers=layers, codebook_size=ref.codebook_size, use_vgg_and_gan=False,
                      lookup_free_quantization=getattr(ref, 'lookup_free_quantization', True), encdec_layer_mults=mults,
                      encdec_num_resnet_blocks=tuple(blocks), encdec_resnet_groups=groups, encdec_first_conv_kernel_size=stem.kernel_size[0])
    _share_tensors(sh, ref)
    sh.eval()
    ref.__dict__[_SHADOW] = sh
    return sh


def _maskgit_shadow(ref):
    from . import muse_maskgit as M
    sh = ref.__dict__.get(_SHADOW)
    if sh is not None:
        return sh
    sh = M.MaskGit.__new__(M.MaskGit)
    nn.Module.__init__(sh)
    sh.vae = _vae_shadow(ref.vae) if ref.vae is not None else None
    sh.cond_vae = sh.vae if ref.cond_vae is ref.vae else (_vae_shadow(ref.cond_vae) if ref.cond_vae is not None else None)
    sh.transformer = _transformer_shadow(ref.transformer)
    tc = ref.token_critic
    if tc is None:
        sh.token_critic = None
    elif hasattr(tc, 'to_pred'):           # the reference's SelfCritic (mmp.py:352-374)
        crit = M.SelfCritic.__new__(M.SelfCritic)
        nn.Module.__init__(crit)
        crit.net, crit.to_pred = sh.transformer, tc.to_pred
        sh.token_critic = crit
    else:
        sh.token_critic = _transformer_shadow(tc)
    for k in ('image_size', 'cond_image_size', 'resize_image_for_cond_image', 'cond_drop_prob', 'self_cond', 'mask_id', 'noise_schedule',
              'critic_loss_weight', 'self_cond_prob', 'no_mask_token_prob'):
        setattr(sh, k, getattr(ref, k))
    sh._gen_ws = None
    sh._graphs = {}                   # (generate(graph=True) through a patched reference object: the attributes MaskGit.__init__ would have set)
    sh.fused_sampling_fallbacks = 0
    sh.fused_bound_switches = 0
    sh.fused_row_fallbacks = 0
    ref.__dict__[_SHADOW] = sh
    return sh


def patch_reference(package=None):
    """Swap the hot methods of the reference's classes (module `muse_maskgit_pytorch`, or the module object passed in).  Idempotent;
    `unpatch_reference()` restores the originals.  Returns the list of patched 'Class.method' names."""
    import importlib
    if package is None:
        package = importlib.import_module('muse_maskgit_pytorch')
    mmp = importlib.import_module(package.__name__ + '.muse_maskgit_pytorch')
    vaemod = importlib.import_module(package.__name__ + '.vqgan_vae')
    attmod = importlib.import_module(package.__name__ + '.attend')
    from .attend import Attend as AmdAttend

    def attend_forward(self, q, k, v, mask=None, force_non_flash=False):                       # attend.py:109
        return AmdAttend(scale=self.scale).forward(q, k, v, mask=mask)

    def tr_forward(self, x, *args, **kwargs):                                                     # mmp.py:279
        return _transformer_shadow(self).train(self.training).forward(x, *args, **kwargs)

    def tr_forward_cs(self, *args, **kwargs):                                                     # mmp.py:240
        return _transformer_shadow(self).train(self.training).forward_with_cond_scale(*args, **kwargs)

    def mg_generate(self, *args, **kwargs):                                                       # mmp.py:491
        return _maskgit_shadow(self).generate(*args, **kwargs)

    def vae_encode(self, fmap):                                                                   # vqgan_vae.py:422
        return _vae_shadow(self).encode(fmap)

    def vae_decode_from_ids(self, ids):                                                           # vqgan_vae.py:427
        return _vae_shadow(self).decode_from_ids(ids)

    def vae_decode(self, fmap):                                                                   # vqgan_vae.py:440
        return _vae_shadow(self).decode(fmap)

    table = [(attmod.Attend, 'forward', attend_forward), (mmp.Transformer, 'forward', tr_forward),
             (mmp.Transformer, 'forward_with_cond_scale', tr_forward_cs), (mmp.MaskGit, 'generate', mg_generate),
             (vaemod.VQGanVAE, 'encode', vae_encode), (vaemod.VQGanVAE, 'decode_from_ids', vae_decode_from_ids), (vaemod.VQGanVAE, 'decode', vae_decode)]
    done = []
    for cls, attr, fn in table:
        if (cls, attr) not in _PATCHED:
            _PATCHED[(cls, attr)] = cls.__dict__[attr]
            fn.__name__, fn.__qualname__, fn.__doc__ = attr, f'{cls.__name__}.{attr}', f'MI355X replacement installed by muse_maskgit_pytorch_amd.patch_reference(); original: {cls.__module__}.{cls.__name__}.{attr}'
            setattr(cls, attr, fn)
        done.append(f'{cls.__name__}.{attr}')
    return done


def unpatch_reference():
    for (cls, attr), orig in list(_PATCHED.items()):
        setattr(cls, attr, orig)
        del _PATCHED[(cls, attr)]
