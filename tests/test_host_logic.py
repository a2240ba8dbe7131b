"""CPU tests of the host-side logic: C-ABI library surface, checkpoint compatibility, weight packing, conv
geometry, schedule.  No kernel is launched here (there is no GPU in the build container)."""
import ctypes as C
import math
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

import emu
import muse_oracle as O
from conftest import ROOT, sd_f32

import muse_maskgit_pytorch_amd as mm
from muse_maskgit_pytorch_amd import _lib, ops


# ------------------------------------------------------------------------------------------------ C ABI surface
def test_library_loads_and_exports_every_header_symbol():
    lib = _lib.lib()
    declared = _lib.header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/muse_hip.h but not exported'
    assert set(declared) == set(_lib.SIGNATURES), 'ctypes signature table out of sync with the header'
    assert lib.mm_abi_version() == 9


def test_library_reads_no_environment_variable_and_carries_no_experimental_kernel():
    """VERDICT r5 weak 8 / ADVICE r5: MM_PP / MM_PP_ABL (tools switches of the rejected out-of-lock-step logits GEMMs) lived in the shipped library and could
    turn a product generate() into garbage; MM_TRAIN_SIDE was read per training step.  The library now does not IMPORT getenv at all, and the experimental
    kernels (tools/experiments/gemm_pp.hip) are linked only by tools/build_timing.sh."""
    so = _lib.LIB_PATH
    und = subprocess.run(['nm', '-D', '--undefined-only', so], capture_output=True, text=True, check=True).stdout
    assert 'getenv' not in und, 'libmuse_hip.so imports getenv'
    syms = subprocess.run(['nm', '-C', so], capture_output=True, text=True, check=True).stdout
    assert 'gemm_pp' not in syms and 'pp_variant' not in syms
    for f in os.listdir(os.path.join(ROOT, 'muse_maskgit_pytorch_amd', 'csrc')):
        assert 'getenv' not in open(os.path.join(ROOT, 'muse_maskgit_pytorch_amd', 'csrc', f)).read().replace('import getenv', ''), f


def test_abi_reports_errors_without_touching_the_gpu():
    lib = _lib.lib()
    # empty problems are no-ops
    assert lib.mm_gemm_bf16(None, None, 0, None, 0, 0, 0, 64, None, 0, 0, None) == 0
    # NULL operands -> MM_ERR_SHAPE with a message, no crash
    rc = lib.mm_gemm_bf16(None, None, 64, None, 64, 4, 4, 64, None, 4, 0, None)
    assert rc == -1 and b'NULL' in lib.mm_last_error()
    rc = lib.mm_transformer_create(None, None)
    assert rc == -1
    assert lib.mm_transformer_workspace_bytes(None, 1, 1, 1) == 0


def test_no_fallback_when_there_is_no_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.MuseHipError):
        ops.gemm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(4, 64, dtype=torch.bfloat16))
    t = mm.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=1, t5_name='t5-small')
    with pytest.raises(_lib.MuseHipError):
        t(torch.zeros(1, 4, dtype=torch.long), text_embeds=torch.randn(1, 3, 512))


def test_product_package_does_not_import_the_oracle():
    src = os.path.join(ROOT, 'muse_maskgit_pytorch_amd')
    for f in os.listdir(src):
        if f.endswith('.py'):
            text = open(os.path.join(src, f)).read()
            assert 'muse_oracle' not in text and 'oracle/' not in text and 'reference_harness' not in text, f


# ------------------------------------------------------------------------------------------------ checkpoints
def test_state_dict_keys_match_reference(golden):
    g, gv = golden('transformer_tiny.pt'), golden('vae_tiny.pt')
    t = mm.MaskGitTransformer(t5_name='t5-small', **g['cfg'])
    assert list(t.state_dict().keys()) == list(g['sd'].keys())
    for k, v in t.state_dict().items():
        assert v.shape == g['sd'][k].shape, k
    v = mm.VQGanVAE(**gv['cfg'])
    ve = v.copy_for_eval()
    assert list(ve.state_dict().keys()) == list(gv['sd'].keys())
    assert any(k.startswith('discr.') for k in v.state_dict())        # training checkpoints carry the discriminator
    mg = mm.MaskGit(vae=v, transformer=t, image_size=128)
    assert len(mg.state_dict()) == 45 + 45 + 58                       # vae.* + cond_vae.* (same object) + transformer.*
    assert mg.mask_id == 512


def test_maskgit_type_checks_like_beartype():
    t = mm.Transformer(num_tokens=512, seq_len=64, dim=128, depth=1, t5_name='t5-small')
    with pytest.raises(TypeError):
        mm.MaskGit(image_size=128, transformer=t, vae=mm.VQGanVAE(dim=16, codebook_size=512))


# ------------------------------------------------------------------------------------------------ schedule
def test_schedule_matches_oracle_and_reference(golden):
    for (T, n), cnt in golden('schedule.pt').items():
        assert ops.mask_counts(T, n) == cnt == O.mask_counts(T, n)
    assert ops.step_temperatures(18, 1.)[-1] == 1e-10
    assert ops.step_temperatures(4, 2.) == O.step_temperatures(4, 2.)
    t = mm.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=1, t5_name='t5-small')
    mg = mm.MaskGit(image_size=128, transformer=t, vae=mm.VQGanVAE(dim=16, codebook_size=512))
    assert mg._mask_counts(18, 256) == O.mask_counts(18, 256)


# ------------------------------------------------------------------------------------------------ conv packing / geometry
def test_conv_packing_matches_torch_convs():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 16, 6, 6, generator=g)
    xn = x.permute(0, 2, 3, 1).contiguous()
    # 3x3 pad 1
    w, b = torch.randn(24, 16, 3, 3, generator=g), torch.randn(24, generator=g)
    y = emu.conv2d_nhwc(xn, ops.pack_conv_weight(w).float(), 24, 3, 3, 1, (-1, -1), bias=b)
    assert torch.allclose(y.permute(0, 3, 1, 2), F.conv2d(x, w.bfloat16().float(), b, padding=1), atol=1e-4)
    # 4x4 stride 2 pad 1
    w = torch.randn(8, 16, 4, 4, generator=g)
    y = emu.conv2d_nhwc(xn, ops.pack_conv_weight(w).float(), 8, 4, 4, 2, (-1, -1), out_hw=(3, 3))
    assert torch.allclose(y.permute(0, 3, 1, 2), F.conv2d(x, w.bfloat16().float(), None, stride=2, padding=1), atol=1e-4)
    # ConvTranspose2d(4, 2, 1) as four parity 2x2 convolutions
    wt, bt = torch.randn(16, 8, 4, 4, generator=g), torch.randn(8, generator=g)
    out = torch.zeros(2, 12, 12, 8)
    for (py, px), wp in ops.pack_convT_weight(wt).items():
        emu.conv2d_nhwc(xn, wp.float(), 8, 2, 2, 1, (py - 1, px - 1), out_hw=(6, 6), os_=2, parity=(py, px), full_hw=(12, 12),
                        bias=bt, out=out)
    ref = F.conv_transpose2d(x, wt.bfloat16().float(), bt, stride=2, padding=1)
    assert torch.allclose(out.permute(0, 3, 1, 2), ref, atol=1e-4)
    # 5x5 stem on the NHWC8 image layout
    img = torch.randn(2, 3, 6, 6, generator=g)
    ws = torch.randn(8, 3, 5, 5, generator=g)
    y = emu.conv2d_nhwc(emu.nchw_to_nhwc8(img), ops.pack_conv_weight_cin8(ws).float(), 8, 5, 5, 1, (-2, -2))
    assert torch.allclose(y.permute(0, 3, 1, 2), F.conv2d(img, ws.bfloat16().float(), None, padding=2), atol=1e-4)


def test_vae_host_orchestration_against_reference_golden(golden, monkeypatch):
    """VQGanVAE.encode / decode_from_ids with every C-ABI op replaced by its CPU emulation must reproduce the
    reference's outputs: validates layer order, conv geometry and weight packing of the host code."""
    gv = golden('vae_tiny.pt')
    emu.install(monkeypatch, ops)
    v = mm.VQGanVAE(**gv['cfg']).copy_for_eval()
    v.load_state_dict(sd_f32(gv['sd']))
    v.composite = False                      # operator by operator: each operator is what the emulation replaces
    dec = v.decode_from_ids(gv['ids'])
    assert dec.shape == gv['decoded'].shape
    assert (dec - gv['decoded']).abs().max() < 2e-4 * gv['decoded'].abs().max().clamp(min=1)
    fmap, ids, aux = v.encode(gv['image'])
    assert torch.equal(ids, gv['enc_ids'])
    assert (fmap - gv['enc_fmap']).abs().max() < 1e-4
    assert v.get_encoded_fmap_size(128) == gv['fmap_size'] == 8


def test_ff_packing_layout():
    t = mm.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=1, t5_name='t5-small')
    ff = t.transformer_blocks.layers[0][2]
    keep = []
    fw, F_, Fp = t._pack_ff(ff, keep)
    assert (F_, Fp) == (341, 384)
    w1p, w2p = keep[0]['w1'].float(), keep[0]['w2'].float()
    w1 = ff[1].weight.detach().bfloat16().float()
    assert w1p.shape == (768, 128)
    # GEGLU-interleaved order: tile t, half w: 32 gelu-half rows then the 32 gate-half rows of the same output columns
    for t_, w_, j in [(0, 0, 0), (0, 1, 5), (2, 1, 31), (5, 0, 20)]:
        col = 64 * t_ + 32 * w_ + j
        gelu_row, gate_row = w1p[128 * t_ + 64 * w_ + j], w1p[128 * t_ + 64 * w_ + 32 + j]
        if col < 341:
            assert torch.equal(gelu_row, w1[col]) and torch.equal(gate_row, w1[341 + col])
        else:
            assert gelu_row.abs().sum() == 0 and gate_row.abs().sum() == 0
    # emulating the fused epilogue on the packed matrix reproduces GEGLU(x @ w1^T) in the first F columns, zeros after
    x = torch.randn(3, 128)
    y = (x @ w1p.t()).reshape(3, 6, 2, 2, 32)                    # [row][tile][half][gelu|gate][32]
    fused = (y[..., 1, :] * torch.nn.functional.gelu(y[..., 0, :])).reshape(3, 384)
    h = x @ w1.t()
    ref = h[:, 341:] * torch.nn.functional.gelu(h[:, :341])
    assert torch.allclose(fused[:, :341], ref, atol=1e-5) and fused[:, 341:].abs().sum() == 0
    assert w2p.shape == (128, 384) and w2p[:, 341:].abs().sum() == 0
    assert int(128 * 4 * 2 / 3) == 341 and int(512 * 4 * 2 / 3) == 1365
    # LayerNorm(inner) folded into w2 (mm_ff_weights.w2_folded / ln2_c1 / ln2_c2): emulating the two GEMM epilogues -- per-row sum and
    # sum of squares in partial groups of 64 columns, rstd * (a . W2g^T) - rstd * mean * c1 + c2 -- reproduces LayerNorm(a) @ w2^T
    with torch.no_grad():
        ff[3].gamma.copy_(1 + 0.3 * torch.randn(341))
    keep = []
    t._pack_ff(ff, keep)
    k = keep[0]
    a = torch.randn(5, 384).bfloat16().float()
    a[:, 341:] = 0
    parts = a.reshape(5, 6, 64)
    s1, s2 = parts.sum(-1).sum(-1, keepdim=True), (parts * parts).sum(-1).sum(-1, keepdim=True)
    mean = s1 / 341
    rstd = 1 / torch.sqrt((s2 / 341 - mean * mean).clamp(min=0) + 1e-5)
    folded = rstd * (a @ k['w2f'].float().t()) - rstd * mean * k['c1'] + k['c2']
    z = torch.nn.functional.layer_norm(a[:, :341], (341,), ff[3].gamma.detach(), None)
    ref2 = z @ ff[4].weight.detach().bfloat16().float().t()
    assert k['w2f'].shape == (128, 384) and k['w2f'][:, 341:].abs().sum() == 0
    assert (folded - ref2).abs().max() < 2e-2 * ref2.abs().max()      # = the bf16 rounding of w2 * gamma


# ------------------------------------------------------------------------------------------------ round-2 host logic
def test_text_condition_pair_contract():
    """t5.py:59-99 / mmp.py:304: the explicit (embeds, mask) pair <-> the zero-padded single tensor the hot path consumes."""
    from muse_maskgit_pytorch_amd import t5
    g = torch.Generator().manual_seed(0)
    e = torch.randn(3, 6, 16, generator=g)
    m = torch.tensor([[1, 1, 1, 1, 0, 0], [1, 1, 1, 1, 1, 1], [1, 0, 0, 0, 0, 0]], dtype=torch.bool)
    packed = t5.pack_text_condition(e, m)
    assert torch.equal(packed[m], e[m]) and (packed[~m] == 0).all()
    e2, m2 = t5.unpack_text_condition(packed)
    assert torch.equal(m2, m) and torch.equal(t5.derive_text_mask(packed), m)
    # a kept position whose embedding is all-zero cannot be expressed by the reference's zeros-are-padding contract: rejected, not dropped
    e[1, 2] = 0
    with pytest.raises(ValueError):
        t5.pack_text_condition(e, m)
    with pytest.raises(ValueError):
        t5.pack_text_condition(torch.randn(1, t5.MAX_LENGTH + 1, 8), torch.ones(1, t5.MAX_LENGTH + 1, dtype=torch.bool))
    # the oracle derives the same mask (it is the reference's expression)
    assert torch.equal((packed != 0).any(dim=-1), m)


def test_token_critic_all_zero_labels_still_take_the_differentiable_path(monkeypatch):
    """A TokenCritic's float labels may all be 0 == the default ignore_index; that must not demote the call to the no-grad branch
    (ADVICE r1).  And return_embed wins over labels like in the reference (mmp.py:334-335)."""
    from muse_maskgit_pytorch_amd import training
    calls = []
    monkeypatch.setattr(training, 'transformer_loss', lambda *a, **k: calls.append('train') or torch.zeros(()))
    critic = mm.TokenCritic(num_tokens=512, seq_len=64, dim=128, depth=1, t5_name='t5-small')
    critic(torch.zeros(1, 4, dtype=torch.long), text_embeds=torch.randn(1, 3, 512), labels=torch.zeros(1, 4))
    assert calls == ['train']
    gen = mm.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=1, t5_name='t5-small')
    calls.clear()
    with pytest.raises(_lib.MuseHipError):       # all rows ignored for the CE head: forward-only branch (which needs the GPU)
        gen(torch.zeros(1, 4, dtype=torch.long), text_embeds=torch.randn(1, 3, 512), labels=torch.zeros(1, 4, dtype=torch.long))
    with pytest.raises(_lib.MuseHipError):       # return_embed requested: never the loss branch
        gen(torch.zeros(1, 4, dtype=torch.long), text_embeds=torch.randn(1, 3, 512), labels=torch.ones(1, 4, dtype=torch.long), return_embed=True)
    assert calls == []


def test_packed_weight_cache_key_sees_storage_and_version_changes():
    t = mm.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=1, t5_name='t5-small')
    k0 = t._pack_key()
    with torch.no_grad():
        t.to_logits.weight.mul_(2.)                       # in-place: version bump
    k1 = t._pack_key()
    assert k1 != k0
    t.to_logits.weight.data = t.to_logits.weight.data.clone()      # new storage, same version
    assert t._pack_key() != k1
    t._handle, t._handle_key = object(), t._pack_key()
    t.load_state_dict(t.state_dict())
    assert t._handle is None                              # load_state_dict drops the packed copies
    t._handle = object()
    t.invalidate_packed_weights()
    assert t._handle is None
    v = mm.VQGanVAE(dim=16, codebook_size=512)
    v._packed = dict(key=v._pack_key())
    v.float()                                             # _apply
    assert v._packed is None


def test_vae_pack_walks_interleaved_resnet_blocks():
    """num_resnet_blocks as a tuple interleaves residual blocks with the down / up-sampling convolutions (vqgan_vae.py:218-231); the packer
    dispatches on the layer type instead of assuming the default layout."""
    v = mm.VQGanVAE(dim=16, codebook_size=512, encdec_num_resnet_blocks=(1, 0, 2, 1))
    kinds_e = [e['kind'] for e in v._pack()['enc']]
    kinds_d = [e['kind'] for e in v._pack()['dec']]
    assert kinds_e == ['stem', 'down', 'res', 'down', 'down', 'res', 'res', 'down', 'res']
    assert kinds_d == ['glu', 'up', 'glu', 'glu', 'up', 'up', 'glu', 'up', 'head']
    d = mm.VQGanVAE(dim=16, codebook_size=512)
    assert [e['kind'] for e in d._pack()['enc']] == ['stem', 'down', 'down', 'down', 'down', 'res']
    assert [e['kind'] for e in d._pack()['dec']] == ['glu', 'up', 'up', 'up', 'up', 'head']


def test_vae_composite_handle_and_workspace_queries():
    """mm_vae_create over the packed layer list (host structs only, no launch): the workspace queries grow with the batch and the map size,
    and the entry points reject NULL / undersized arguments with the documented codes."""
    import ctypes as C
    from muse_maskgit_pytorch_amd import _lib as L
    v = mm.VQGanVAE(dim=16, codebook_size=512, encdec_num_resnet_blocks=(1, 0, 2, 1))
    h = v._pack()['handle'].h
    lib = L.lib()
    d1, d2, d3 = (lib.mm_vae_decode_workspace_bytes(h, b, s, s) for b, s in ((1, 8), (2, 8), (2, 16)))
    assert 0 < d1 < d2 < d3
    e1, e2 = lib.mm_vae_encode_workspace_bytes(h, 1, 64, 64), lib.mm_vae_encode_workspace_bytes(h, 2, 128, 128)
    assert 0 < e1 < e2
    assert lib.mm_vae_decode_workspace_bytes(None, 1, 8, 8) == 0
    assert lib.mm_vae_decode_from_ids(None, None, None, 1, 8, 8, None, None, 0) == -1
    buf = C.create_string_buffer(64)
    assert lib.mm_vae_decode_from_ids(h, None, buf, 1, 8, 8, buf, buf, 64) == -6
    assert lib.mm_vae_encode(h, None, buf, 1, 64, 64, None, buf, buf, 64) == -6
    out = C.c_void_p()
    assert lib.mm_vae_create(None, C.byref(out)) == -1


def test_vae_half_decode_handle_packing_and_policy():
    """Round 6: the decoder on fp16 storage (VQGanVAE.decode_storage; csrc/vae_model.hip `half`).  Host side only: the policy (which precision / storage pairs take it),
    the fp16 packs (one fp16 term of scale * w, scale a power of two that puts the decoder's largest |w| into [2^13, 2^14); the value the kernels multiply, pack / scale,
    within 2^-11 of w), the decode-only C handle (encode on it is refused, its workspace query answers), and that invalidating the weights drops it."""
    import ctypes as C
    from muse_maskgit_pytorch_amd import _lib as L
    torch.manual_seed(2)
    v = mm.VQGanVAE(dim=16, codebook_size=512)
    assert v.decode_storage == 'f16' and v._half_decode()
    for prec, storage, want in (('bf16', 'f16', True), ('bf16', 'bf16', False), ('f16x2', 'f16', True), ('f16x2', 'terms', False), ('f16x2', 'bf16', True),
                                ('parity', 'f16', False), ('bf16x3', 'f16', False)):
        v.set_precision(prec).set_decode_storage(storage)
        assert v._half_decode() == want, (prec, storage)
    v.set_precision('bf16').set_decode_storage('f16')
    with pytest.raises(ValueError):
        v.set_decode_storage('fp8')
    H = v._pack_half()
    scale = H['scale']
    wmax = max(float(w.detach().abs().max()) for w in v._decoder_conv_weights())
    assert math.log2(scale) == round(math.log2(scale)) and 2 ** 13 <= wmax * scale < 2 ** 14
    assert [e['kind'] for e in H['dec']] == [e['kind'] for e in v._pack()['dec']] and H['enc'] == []
    head = v.enc_dec.decoders[-1]
    wp = H['dec'][-1]['w']
    assert wp.dtype == torch.float16 and wp.shape == (3, 64)                     # [Cout][Kp]: K = 16 padded to 64
    ref = head.weight.detach().reshape(3, 16)
    got = wp[:, :16].float() / scale
    assert (got - ref).abs().max() <= ref.abs().max() * 2.0 ** -11 and bool((wp[:, 16:] == 0).all())
    up = H['dec'][1]
    assert up['kind'] == 'up' and set(up['w']) == {(0, 0), (0, 1), (1, 0), (1, 1)} and all(t.dtype == torch.float16 for t in up['w'].values())
    lib, h = L.lib(), H['handle'].h
    assert lib.mm_vae_decode_workspace_bytes(h, 2, 8, 8) > 0
    buf = C.create_string_buffer(64)
    assert lib.mm_vae_encode(h, None, buf, 1, 64, 64, None, buf, buf, 1 << 30) == -7      # MM_ERR_UNSUPPORTED: the fp16-storage handle is decode-only
    assert v._pack_half() is H                                                    # cached beside the bf16 packs ...
    gen0 = v._pack_gen
    v.invalidate_packed_weights()
    assert v._packed is None and v._pack_gen == gen0 + 1 and v._pack_half() is not H      # ... and dropped with them


def test_seeded_construction_matches_the_base_golden_recipe(golden):
    """The base-size golden stores no checkpoint: both sides rebuild it from seeds (oracle/golden_recipe.py).  This package's classes must
    reproduce the reference's parameters exactly -- checked against the checksums the reference run stored."""
    import golden_recipe as R
    g = golden('base_c2.pt')
    tr = R.build_transformer(mm.MaskGitTransformer, peaky=False)
    got = R.state_checksum(tr)
    assert got == g['weight_checksum']
    with torch.no_grad():
        tr.to_logits.weight.mul_(R.PEAK)
    assert R.state_checksum(tr) == g['weight_checksum_peaky']
    inp = R.inputs()
    assert {k: R.checksum(v.float()) for k, v in inp.items()} == g['input_checksum']
    vae = R.build_vae(mm.VQGanVAE).copy_for_eval()
    assert R.state_checksum(vae) == g['vae_weight_checksum']
    u0 = next(iter(R.noise_stream()))
    assert R.checksum(u0) == g['generate']['noise_checksum'][0] and torch.equal(u0.flatten()[:8], g['generate']['noise_head'][0])


def test_patch_reference_mode_2_keeps_the_reference_classes():
    """SURVEY 8b mode 2: hot methods swapped on the reference's own classes; isinstance (what @beartype checks, mmp.py:427,745) still
    holds, the shadow shares the reference module's tensors, and without a GPU the patched methods raise (no fallback)."""
    import reference_harness as H
    if not H.reference_available():
        pytest.skip('the reference package only exists in the build container')
    pkg, mmp, vaemod, att = H.reference_modules()
    from muse_maskgit_pytorch_amd import patch
    orig = mmp.MaskGit.generate
    names = mm.patch_reference(pkg)
    try:
        assert 'MaskGit.generate' in names and 'Attend.forward' in names and mmp.MaskGit.generate is not orig
        assert mm.patch_reference(pkg) == names                              # idempotent
        tr = pkg.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=2, dim_head=64, heads=8, t5_name='t5-small')
        vae = pkg.VQGanVAE(dim=16, codebook_size=512)
        mg = pkg.MaskGit(vae=vae, transformer=tr, image_size=128)
        assert isinstance(mg.transformer, pkg.MaskGitTransformer) and isinstance(mg.vae, pkg.VQGanVAE) and isinstance(mg, pkg.MaskGit)
        sh = patch._transformer_shadow(tr)
        assert type(sh) is mm.MaskGitTransformer and patch._transformer_shadow(tr) is sh
        own = dict(sh.named_parameters())
        for k, p in tr.named_parameters():
            assert own[k] is p
        assert not any(t.is_meta for t in list(sh.parameters()) + list(sh.buffers()))
        vsh = patch._vae_shadow(mg.vae)
        assert all(dict(vsh.named_parameters())[k] is p for k, p in mg.vae.named_parameters() if k in dict(vsh.named_parameters()))
        assert not any(t.is_meta for t in list(vsh.parameters()) + list(vsh.buffers()))
        msh = patch._maskgit_shadow(mg)
        assert msh.transformer is sh and msh.vae is vsh and msh.mask_id == 512
        crit = pkg.TokenCritic(num_tokens=512, seq_len=64, dim=128, depth=1, dim_head=64, heads=8, t5_name='t5-small')
        assert type(patch._transformer_shadow(crit)) is mm.TokenCritic
        if not torch.cuda.is_available():
            with pytest.raises(_lib.MuseHipError):
                tr(torch.zeros(1, 4, dtype=torch.long), text_embeds=torch.randn(1, 3, 512))
            tr.encode_text = lambda texts: torch.randn(len(texts), 3, 512)      # per-instance attribute, as reference users override it (mmp.py:229)
            with pytest.raises(_lib.MuseHipError):
                mg.generate(['a'], timesteps=2)
    finally:
        mm.unpatch_reference()
    assert mmp.MaskGit.generate is orig


def test_mm_generate_rejects_bad_parameters_before_any_launch():
    """Argument validation of the decode entry point (host structs only: every error below is returned before the first kernel launch, so it
    runs without a GPU): schedule, outputs, critic combinations, workspaces."""
    import ctypes as C
    from muse_maskgit_pytorch_amd import _lib as L
    lib = L.lib()
    dummy = torch.zeros(64, dtype=torch.uint8)

    class H:      # a transformer handle over dummy pointers: mm_transformer_create copies the description, nothing dereferences it here
        def __init__(self, dim_out, vocab_rows):
            self.layers = (L.LayerWeights * 1)()
            d = L.TransformerDesc(dim=128, depth=1, heads=2, dim_head=64, ff_inner=341, ff_inner_padded=384, seq_len=64, num_tokens=512,
                                  vocab_rows=vocab_rows, dim_out=dim_out, text_dim=512, self_cond=0)
            d.token_emb = d.pos_emb = d.text_proj = d.final_gamma = d.final_beta = d.to_logits = L.ptr(dummy)
            d.layers = self.layers
            self.ptr = C.c_void_p()
            assert lib.mm_transformer_create(C.byref(d), C.byref(self.ptr)) == 0
            self.desc = d

        def __del__(self):
            lib.mm_transformer_destroy(self.ptr)

    h, hc = H(512, 513), H(1, 512)
    lib = L.lib()
    buf = torch.zeros(1 << 16, dtype=torch.uint8)
    B, n, T = 2, 64, 3

    def params(**over):
        p = L.GenerateParams()
        p.batch, p.n, p.timesteps, p.k_keep, p.nc, p.L = B, n, T, 52, 0, 5
        p.cond_scale, p.noise_kind = 3.0, L.MM_NOISE_PHILOX
        cnt, tmp = (C.c_int32 * T)(64, 40, 1), (C.c_float * T)(1.0, 0.6, 0.3)
        p.mask_counts, p.temperatures = cnt, tmp
        p.text_embeds = p.ids = p.scores = L.ptr(buf)
        p._keep = (cnt, tmp)
        for k, v in over.items():
            setattr(p, k, v)
        return p

    def rc(p, ws_bytes=0):
        return lib.mm_generate(h.ptr, None, C.byref(p), L.ptr(buf), ws_bytes)

    need = lib.mm_generate_workspace_bytes(h.ptr, B, n, 5, 0)
    assert need > 0 and lib.mm_generate_critic_workspace_bytes(hc.ptr, B, n, 5, 0) > 0 and lib.mm_generate_critic_workspace_bytes(None, B, n, 5, 0) == 0
    assert rc(params()) == -6                                            # workspace too small (nothing launched)
    assert rc(params(timesteps=0), need) == -1
    bad = params()
    bad.mask_counts = (C.c_int32 * T)(64, 10, 20)                        # must be non-increasing
    assert rc(bad, need) == -1
    bad = params()
    bad.mask_counts = (C.c_int32 * T)(32, 10, 1)                         # the first step masks everything
    assert rc(bad, need) == -1
    assert rc(params(noise_kind=L.MM_NOISE_GUMBEL), need) == -1          # tensor noise mode without a noise tensor
    assert rc(params(critic=hc.ptr, critic_head_w=L.ptr(buf), critic_head_b=L.ptr(buf), critic_noise=L.ptr(buf)), need) == -1      # exclusive
    assert rc(params(critic=hc.ptr), need) == -1                         # a critic needs its noise
    assert rc(params(critic_head_w=L.ptr(buf), critic_noise=L.ptr(buf)), need) == -1      # a self-critic head needs its bias
    assert rc(params(critic=hc.ptr, critic_noise=L.ptr(buf), critic_workspace=L.ptr(buf), critic_workspace_bytes=64), need) == -6
    assert rc(params(critic=h.ptr, critic_noise=L.ptr(buf), critic_workspace=L.ptr(buf), critic_workspace_bytes=1 << 16), need) == -1     # dim_out != 1
    assert b'dim_out' in lib.mm_last_error()
