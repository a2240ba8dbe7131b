"""TEST INFRASTRUCTURE ONLY.  The deterministic recipe of the BASE-SIZE (BASELINE configs[1]) golden run, shared by
oracle/make_golden_base.py (which applies it to the UNMODIFIED reference's classes in the build container) and by the tests (which apply
it to this package's classes on the GPU box -- the reference does not travel).  A base transformer has 103 M parameters, so the
checkpoint cannot be a committed fixture: both sides REBUILD it from `torch.manual_seed` + the module constructors (the reference's and
this package's constructors create the same parameters in the same order from the same generator stream -- asserted, parameter by
parameter, in make_golden_base.py and tests/test_host_logic.py) followed by the edits below, and the per-step Gumbel noise is rebuilt
from the CPU generator stream the reference consumed (mmp.py:406-408: one `zeros_like(logits).uniform_(0, 1)` per decode step, nothing
else draws).  Checksums of both are stored in the fixture and asserted before anything is compared.
"""
import torch

BASE_CFG = dict(num_tokens=65536, seq_len=256, dim=512, depth=8, dim_head=64, heads=8, ff_mult=4, t5_name='t5-small')      # README.md:61-70
VAE_CFG = dict(dim=256, codebook_size=65536)                                                                          # README.md:23-26
B, N, L, T = 2, 256, 32, 18
WEIGHT_SEED, VAE_SEED, EDIT_SEED, INPUT_SEED, NOISE_SEED = 0, 1, 4321, 77, 20260924
INPUT_SEED_FP32 = 83      # inputs of the general-fp32 fixture (base_c2_fp32.pt): chosen by tools/find_golden_input_seed.py -- with seed 77 that checkpoint's reference
                          # run has two confidences 4e-6 apart at a re-masking boundary, a coin flip for every implementation that is not bit-identical to it
PEAK = 8.0      # to_logits scale of the decode run: well-separated confidences (SURVEY 8c determinism control 3)


C4_CFG = dict(num_tokens=65536, seq_len=1024, dim=512, depth=8, dim_head=64, heads=8, ff_mult=4, t5_name='t5-small')      # BASELINE configs[3]
C4_T, C4_NOISE_SEED, C4_WEIGHT_SEED = 6, 20260925, 2


# BASELINE configs[4] shape (paper-scale base): dim 1024, depth 24, 16 heads, codebook 8192; t5-small embeddings (512) -> the text projection runs
C5_CFG = dict(num_tokens=8192, seq_len=256, dim=1024, depth=24, dim_head=64, heads=16, ff_mult=4, t5_name='t5-small')
C5_T, C5_NOISE_SEED, C5_WEIGHT_SEED = 5, 20260926, 3


def build_transformer(cls, peaky, cfg=None, seed=None, bf16_weights=True):
    """cls = MaskGitTransformer of the reference or of this package.  Module-default init under WEIGHT_SEED, learned scales / norm gains
    made non-trivial, optionally peaky logits, everything rounded to bf16-representable fp32 (exactly loadable by the bf16 engine) --
    or, `bf16_weights=False`, left as the GENERAL fp32 values the constructors and the edits produce (what every checkpoint the reference
    trains looks like, mmp.py:85,88,118-124,233: the `*_fp32.pt` fixtures)."""
    torch.manual_seed(WEIGHT_SEED if seed is None else seed)
    tr = cls(**(BASE_CFG if cfg is None else cfg))
    gen = torch.Generator().manual_seed(EDIT_SEED)
    with torch.no_grad():
        for name, p in tr.named_parameters():
            if name.endswith('q_scale') or name.endswith('k_scale') or name.endswith('gamma'):
                p.mul_(1 + 0.2 * torch.randn(p.shape, generator=gen))
        if peaky:
            tr.to_logits.weight.mul_(PEAK)
        if bf16_weights:
            for p in tr.parameters():
                p.copy_(p.to(torch.bfloat16).float())
    return tr.eval()


def build_vae(cls, bf16_weights=True):
    torch.manual_seed(VAE_SEED)
    vae = cls(**VAE_CFG)
    if bf16_weights:
        with torch.no_grad():
            for p in vae.parameters():
                p.copy_(p.to(torch.bfloat16).float())
    return vae.eval()


def inputs(seed=None):
    """ids with ~half the positions masked, zero-padded text embeddings (t5.py:93), VAE inputs.  `seed`: INPUT_SEED by default; the general-fp32
    fixture (base_c2_fp32.pt) uses INPUT_SEED_FP32 (fixture['recipe']['input_seed']), see tools/find_golden_input_seed.py for how it was chosen."""
    g = torch.Generator().manual_seed(INPUT_SEED if seed is None else seed)
    ids = torch.randint(0, BASE_CFG['num_tokens'], (B, N), generator=g)
    ids[torch.rand(B, N, generator=g) < 0.5] = BASE_CFG['num_tokens']          # mask id
    te = torch.randn(B, L, 512, generator=g)
    te[1, L - 5:] = 0
    vae_ids = torch.randint(0, VAE_CFG['codebook_size'], (B, 16, 16), generator=g)
    image = torch.randn(B, 3, 256, 256, generator=g)
    return dict(ids=ids, text_embeds=te, vae_ids=vae_ids, image=image)


def noise_stream(steps=T, seed=NOISE_SEED, shape=None):
    """the U(0,1) tensors the reference's gumbel_noise drew during generate(), step by step, from torch's CPU generator"""
    torch.manual_seed(seed)
    for _ in range(steps):
        yield torch.zeros(*(shape or (B, N, BASE_CFG['num_tokens']))).uniform_(0, 1)


def c4_inputs():
    """super-resolution case, batch 1: 1024 token ids (half masked), text, a 256 x 256 low-resolution condition image"""
    g = torch.Generator().manual_seed(INPUT_SEED + 1)
    ids = torch.randint(0, 65536, (1, 1024), generator=g)
    ids[torch.rand(1, 1024, generator=g) < 0.5] = 65536
    te = torch.randn(1, L, 512, generator=g)
    te[0, L - 3:] = 0
    cond_image = torch.randn(1, 3, 256, 256, generator=g)
    return dict(ids=ids, text_embeds=te, cond_image=cond_image)


def c5_inputs():
    """paper-scale case, batch 2: 256 token ids (half masked), zero-padded text embeddings"""
    g = torch.Generator().manual_seed(INPUT_SEED + 2)
    ids = torch.randint(0, 8192, (2, 256), generator=g)
    ids[torch.rand(2, 256, generator=g) < 0.5] = 8192
    te = torch.randn(2, L, 512, generator=g)
    te[1, L - 7:] = 0
    return dict(ids=ids, text_embeds=te)


def checksum(t):
    """order-sensitive EXACT checksum of a tensor's fp32 bit patterns (weights / inputs / noise reproduction check): integer arithmetic, so it
    does not depend on the summation order of the host's reduction kernels (an fp64 sum differed in the last digit between two CPUs)"""
    bits = t.detach().to(torch.float32).cpu().contiguous().view(torch.int32).flatten().to(torch.int64)
    w = torch.arange(1, bits.numel() + 1, dtype=torch.int64).remainder(1009) + 1
    return int((bits * w).sum().item())


def state_checksum(module):
    return {k: checksum(v) for k, v in module.state_dict().items() if v.is_floating_point()}
