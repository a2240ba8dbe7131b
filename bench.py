"""bench.py -- images/sec of the Muse base path on MI355X (BASELINE.json metric).

One "step" = one MaskGit.generate() of `--batch` images per GPU on the BASELINE configs[1] workload
(C2 base 256x256: seq_len=256 dim=512 depth=8 heads=8, codebook 65536, 18 decode steps, cond_scale 3, bf16) INCLUDING
the VQGanVAE decode, on synthetic random-init weights and random zero-padded "T5" embeddings already resident in HBM.
N>1: one process per GPU (torchrun), the batch is sharded (weak scaling: 32 images per GPU), zero communication inside
the decode loop, one RCCL all-gather of the generated token grids per step.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     : the dominant kernel = the fused to_logits+CFG MFMA GEMM, timed with HIP events on its launch stream
                 inside the timed region (mm_profile_*), algorithmic flops / measured time vs 2.5 PFLOP/s dense bf16;
  roofline_hbm : the HBM-bound sampling kernel the same way (algorithmic bytes = one fp32 read of each sampled row);
  cpu_baseline : the CPU oracle (port of the reference algorithm, torch fp32 on the host cores) on a bounded sample;
  parity_tier  : the same workload through precision 'f16x2' (round 3: 'bf16x3', still reported inside) -- the engine that meets the north star's tolerance (logits within 1e-3 of the fp32
                 reference, ids bit-exact: tests/test_gpu_base_size.py) inside the same mm_generate call -- timed after the main region.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

_RECORD_OUT = sys.stdout

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0          # HBM3E spec


CONFIGS = {
    # name: (transformer kwargs, vae kwargs, image_size, cond_image_size, default images per GPU, description)
    'c2': (dict(num_tokens=65536, seq_len=256, dim=512, depth=8, dim_head=64, heads=8, ff_mult=4), dict(dim=256, codebook_size=65536), 256, None, 32,
           'BASELINE configs[1]: base 256x256, MaskGit.generate (18 steps, cond_scale 3, top-k 0.9, Philox Gumbel noise) + VQGanVAE(dim=256, codebook 65536) decode'),
    'c4': (dict(num_tokens=65536, seq_len=1024, dim=512, depth=8, dim_head=64, heads=8, ff_mult=4), dict(dim=256, codebook_size=65536), 512, 256, 8,
           'BASELINE configs[3]: super-res 512x512 (1024 tokens, 256 low-res condition ids + text in the cross-attention context), VQGanVAE encode of the '
           '256x256 condition image + 18-step generate + 512x512 decode'),
    'c5': (dict(num_tokens=8192, seq_len=256, dim=1024, depth=24, dim_head=64, heads=16, ff_mult=4), dict(dim=256, codebook_size=8192), 256, None, 32,
           'BASELINE configs[4] shape: paper-scale base (dim 1024, depth 24, 16 heads, codebook 8192), 18-step generate + decode'),
}


def build_config(name, device):
    """MaskGit of a named BASELINE configuration (random init, torch.manual_seed(0))."""
    import muse_maskgit_pytorch_amd as mm
    tkw, vkw, image_size, cond_size, _, _ = CONFIGS[name]
    torch.manual_seed(0)
    vae = mm.VQGanVAE(**vkw)
    tr = mm.MaskGitTransformer(t5_name='t5-small', **tkw)
    kw = dict(cond_image_size=cond_size) if cond_size else {}
    mg = mm.MaskGit(vae=vae, transformer=tr, image_size=image_size, **kw)
    return mg.to(device).eval(), image_size


def build_models(device, tiny=False):
    import muse_maskgit_pytorch_amd as mm
    torch.manual_seed(0)
    if tiny:
        vae = mm.VQGanVAE(dim=64, codebook_size=512)
        tr = mm.MaskGitTransformer(num_tokens=512, seq_len=64, dim=128, depth=2, dim_head=64, heads=8, t5_name='t5-small')
        image_size = 128
    else:
        vae = mm.VQGanVAE(dim=256, codebook_size=65536)                                   # README.md:23-26
        tr = mm.MaskGitTransformer(num_tokens=65536, seq_len=256, dim=512, depth=8, dim_head=64, heads=8, ff_mult=4,
                                   t5_name='t5-small')                                     # README.md:61-70
        image_size = 256
    mg = mm.MaskGit(vae=vae, transformer=tr, image_size=image_size)
    return mg.to(device).eval(), image_size


def synth_text(total, L, dim, seed=0):
    g = torch.Generator().manual_seed(seed)
    te = torch.randn(total, L, dim, generator=g)
    tails = torch.randint(0, L // 2 + 1, (total,), generator=g)
    for i, t in enumerate(tails.tolist()):
        if t:
            te[i, L - t:] = 0          # t5.py:93: padded positions are exactly zero
    return te


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def graph_replay_leg(mg, B, T, te, steps, eager_s):
    """The same step through the product's graph mode, MaskGit.generate(graph=True): decode loop + VAE decode captured ONCE in a hipGraph (first call eager,
    second call captures) and replayed with the Philox keys read from a device buffer at execution time -- every timed replay draws a FRESH seed and re-executes
    all of the work; ids and pixels of a replay are bit-identical to the eager call with the same seed (tests/test_gpu_fused_sampling.py).  What it removes is
    the host's ~1190 launch calls per generate and the gaps between dependent launches.  Reported beside the headline (which times the eager call sequence)."""
    try:
        for i in range(2):                                   # warm-up (eager) + capture
            mg.generate([''] * B, timesteps=T, cond_scale=3, text_embeds=te, seed=7000 + i, return_ids='both', graph=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            ids, images = mg.generate([''] * B, timesteps=T, cond_scale=3, text_embeds=te, seed=8000 + i, return_ids='both', graph=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ids_e, _ = mg.generate([''] * B, timesteps=T, cond_scale=3, text_embeds=te, seed=8000 + steps - 1, return_ids='both')
        return {'value': B / dt, 'unit': 'images/sec', 'ms_per_step': dt * 1e3, 'steps': steps, 'x_eager_time': dt / eager_s, 'fresh_seed_per_replay': True,
                'last_replay_ids_equal_eager_call_with_the_same_seed': bool(torch.equal(ids, ids_e)), 'finite_images': bool(torch.isfinite(images).all().item()),
                'note': 'MaskGit.generate(graph=True): one captured generate + VAE decode per call signature, replayed with fresh Philox keys from a device buffer'}
    except Exception as e:      # never lets the extra leg cost the line
        return {'error': f'{type(e).__name__}: {e}'[:300]}


def _tier_traffic(substr):
    """HBM bytes per launch of a kernel of the f16x2 tier from the committed PMC passes of `bench.py --precision f16x2` (general fp32 checkpoint), or None"""
    try:
        with open(os.path.join(ROOT, 'profiles', 'r06_f16x2_pmc_summary.json')) as f:
            pmc = json.load(f)
    except OSError:
        return None
    for name, v in pmc.items():
        if substr in name:
            return v['hbm_bytes_per_launch']
    return None


def off_ideal_legs(mg, tr, args, B, T, rank, world, dev, eager_s):
    """What the headline is worth off its ideal case, each timed like the main region (eager, fresh seeds), reported BESIDE the value:
    text_len 77 / 256 -- the reference pads to the longest prompt up to MAX_LENGTH = 256 (t5.py:16,78-79), the headline runs L = 32 --, and a to_logits whose
    rows are far from the Gaussian the fused sampler's bound assumes: x 8 (peaky, the goldens' own recipe) with a heavy-tailed block of 4096 vocabulary rows x 4
    on top; `rows_finished_by_on_device_fallback` counts the rows the bound could not be verified for."""
    out = {}

    def timed(te, nsteps, seed0):
        mg.generate([''] * B, timesteps=T, cond_scale=3, text_embeds=te, seed=seed0 - 1, return_ids='both')
        torch.cuda.synchronize()
        r0, f0 = mg.fused_row_fallbacks, mg.fused_sampling_fallbacks
        t0 = time.perf_counter()
        for i in range(nsteps):
            mg.generate([''] * B, timesteps=T, cond_scale=3, text_embeds=te, seed=seed0 + i, return_ids='both')
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / nsteps
        return sec, (mg.fused_row_fallbacks - r0) / nsteps, mg.fused_sampling_fallbacks - f0

    nsteps = max(3, min(args.steps, 5))
    try:
        for L in (77, 256):
            te = synth_text(world * B, L, tr.text_embed_dim, seed=L)[rank * B:(rank + 1) * B].to(dev)
            sec, _, _ = timed(te, nsteps, 3000 + L)
            out[f'text_len_{L}'] = {'value': B / sec, 'unit': 'images/sec', 'ms_per_step': sec * 1e3, 'steps': nsteps, 'x_headline_time': sec / eager_s}
        te = synth_text(world * B, args.text_len, tr.text_embed_dim)[rank * B:(rank + 1) * B].to(dev)
        w = tr.to_logits.weight
        saved = w.detach().clone()
        try:
            with torch.no_grad():
                w.mul_(8.)
                w[1024:1024 + 4096].mul_(4.)
            sec, rows, falls = timed(te, nsteps, 5000)
            sampled_rows = B * sum(mg._mask_counts(T, tr.seq_len))
            out['non_gaussian_logits'] = {'value': B / sec, 'unit': 'images/sec', 'ms_per_step': sec * 1e3, 'steps': nsteps, 'x_headline_time': sec / eager_s,
                                          'to_logits': 'x 8 (peaky), vocabulary rows 1024 .. 5119 x 4 on top (heavy-tailed block)',
                                          'rows_finished_by_on_device_fallback_per_generate': rows, 'sampled_rows_per_generate': sampled_rows,
                                          'fallback_row_fraction': rows / sampled_rows, 'whole_call_fallbacks_to_logits_path': falls,
                                          'bound_in_use': getattr(tr._model(), 'auto_bound', None) if tr.fused_bound == 'auto' else tr.fused_bound, 'bound_switches': mg.fused_bound_switches}
        finally:
            with torch.no_grad():
                w.copy_(saved)
    except Exception as e:
        out['error'] = f'{type(e).__name__}: {e}'[:300]
    return out


def cpu_baseline(mg, te_two, timesteps, cond_scale, max_threads=32):
    """The reference algorithm on the host cores, beside the GPU number (never the thing measured as `value`).  kind = "port": the oracle
    (oracle/muse_oracle.py, a functional fp32 torch restatement pinned bit-exactly to goldens of the unmodified reference) -- the reference
    package itself is absent from the GPU box.  Protocol (bounded to ~20 s): batch 2, one untimed warm-up decode step, then
    ONE full generate -- all `timesteps` decode steps (each the full two-pass transformer and the full-vocabulary sampling tail) and the VAE
    decode of the batch -- timed end to end: no extrapolation."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import muse_oracle as O
    cores = min(os.cpu_count(), max_threads)   # torch's intra-op pool stops scaling (and regresses) far below 256 threads
    torch.set_num_threads(cores)
    tr = mg.transformer
    sd = {k: (v.detach().float().cpu() if v.is_floating_point() else v.detach().cpu()) for k, v in tr.state_dict().items()}
    vsd = {k: (v.detach().float().cpu() if v.is_floating_point() else v.detach().cpu()) for k, v in mg.vae.state_dict().items()}
    cfg = dict(depth=tr.transformer_blocks.cfg['depth'], heads=tr.transformer_blocks.cfg['heads'])
    n, V = tr.seq_len, tr.num_tokens
    Bc = te_two.shape[0]
    counts = O.mask_counts(timesteps, n)
    temps = O.step_temperatures(timesteps, 1.)
    g = torch.Generator().manual_seed(0)
    f = int(math.isqrt(n))

    def run(steps, decode):
        ids = torch.full((Bc, n), tr.mask_id, dtype=torch.long)
        scores = torch.zeros(Bc, n)
        t0 = time.perf_counter()
        for s in range(steps):
            sel = O.select_topk_stable(scores, counts[s])
            ids = torch.where(sel, torch.full_like(ids, tr.mask_id), ids)
            logits = O.forward_with_cond_scale(sd, cfg, ids, te_two, cond_scale)
            gum = O.gumbel_from_uniform(torch.rand(Bc, n, V, generator=g))
            ids, scores, _ = O.sample_step(logits, gum, ids, tr.mask_id, temps[s])
        t1 = time.perf_counter()
        if decode:
            O.vae_decode_from_ids(vsd, ids.clamp(max=V - 1).reshape(Bc, f, f))
        return t1 - t0, time.perf_counter() - t1

    with torch.no_grad():
        run(1, False)                                  # warm-up
        loop, dec = run(timesteps, True)
    per_batch = loop + dec
    return dict(value=Bc / per_batch, unit='images/sec', cores=cores, kind='port', cpu_model=_cpu_model(), host_cpus=os.cpu_count(),
                sample=f'batch {Bc}, 1 warm-up decode step, then one full generate timed end to end: all {timesteps} decode steps ({loop:.1f} s) + the VAE decode '
                       f'({dec:.1f} s), no extrapolation; oracle/muse_oracle.py (fp32 torch port of the reference: the reference package is not present on the GPU box), '
                       f'{cores} torch threads')


def roofline_hbm(fused_on, s_cnt, s_ms, s_bytes, traffic, pmc_file):
    """The sampling tail against the HBM roofline.  Logits path: one fp32 read of each sampled row (algorithmic bytes).  Fused path (default): the
    logits never reach HBM -- sample_fused_kernel reads the candidates the GEMM emitted, so `achieved` / `frac` are the bytes it REALLY moves (PMC
    FETCH + WRITE per launch from the committed rocprofv3 counter pass of this command) over its measured time; the logits-equivalent rate (4 V per
    row: what a logits-reading sampler would have to read in the same time) is reported beside it, labelled as such."""
    avg_s = s_ms * 1e-3 / s_cnt if s_cnt else None
    d = {'kernel': ('sample_fused_kernel (exact k-th largest + Gumbel argmax + confidence over the candidates emitted by the GEMM)' if fused_on
                    else 'sample_kernel (top-k + Gumbel argmax + confidence on materialised logits)'), 'bound': 'hbm', 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
         'traffic': traffic, 'launches': s_cnt, 'avg_launch_ms': s_ms / s_cnt if s_cnt else None}
    if not fused_on:
        d.update(achieved=s_bytes / (s_ms * 1e-3) / 1e9 if s_ms else None, frac=(s_bytes / (s_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if s_ms else None,
                 bytes_kind='algorithmic (one fp32 read of each sampled row)', algorithmic_bytes_per_launch=s_bytes / s_cnt if s_cnt else None)
        return d
    real = traffic / avg_s / 1e9 if (traffic and avg_s) else None
    d.update(achieved=real, frac=real / PEAK_HBM_GBS if real else None,
             bytes_kind=(f'HBM bytes actually moved per launch (profiles/{pmc_file}: rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE of this command) / the launch time measured here'
                         if real else 'no counter pass available for this configuration: achieved / frac not stated'),
             logits_equivalent={'gb_per_s': s_bytes / (s_ms * 1e-3) / 1e9 if s_ms else None, 'bytes_per_launch': s_bytes / s_cnt if s_cnt else None,
                                'note': '4 V bytes per sampled row -- what a sampler that reads materialised logits moves; NOT bytes this kernel moves, no roofline fraction'})
    return d


def executed_flops_per_generate(tr, B, n, m_text, nc, counts):
    """MFMA flops the fused engine actually executes for one generate of B images (GEMMs + attention products; SURVEY 8d asks for the
    EXECUTED count for the utilisation figure): 2B sequences through every layer, cross-attention on the cond half only when the null
    pass's is a constant (no condition ids), the last layer's row-wise tail and the logits only at the k_t sampled rows of each step."""
    cfgb = tr.transformer_blocks.cfg
    D, depth, H = tr.dim, cfgb['depth'], cfgb['heads']
    I = H * cfgb['dim_head']
    F = int(D * cfgb['ff_mult'] * 2 / 3)
    Fp = (F + 63) // 64 * 64
    V = tr.dim_out
    m = m_text + nc
    total = 2.0 * B * m * D * 2 * I * depth                                 # cross-attention K/V of every layer, once
    for k in counts:
        rows2, R2 = 2 * B * n, 2 * B * k
        cross_rows = (2 * B if nc else B) * n
        full = depth - 1 if k < n else depth
        per_layer = (2.0 * rows2 * D * 3 * I + 4.0 * 2 * B * H * n * (n + 1) * 64 + 2.0 * rows2 * I * D      # self: q|k|v, QK^T + PV, out
                     + 2.0 * cross_rows * D * I * 2 + 4.0 * (cross_rows // n) * H * n * (m + 1) * 64        # cross: q, out, attention
                     + 2.0 * rows2 * D * 2 * Fp + 2.0 * rows2 * Fp * D)                                     # FF
        total += full * per_layer
        if k < n:      # last layer: token mixing on all rows, everything behind it on the compacted rows
            cr = (2 * B if nc else B) * k
            total += (2.0 * rows2 * D * 3 * I + 4.0 * 2 * B * H * n * (n + 1) * 64 + 2.0 * R2 * I * D
                      + 2.0 * cr * D * I * 2 + 4.0 * (cr // max(k, 1)) * H * k * (m + 1) * 64 + 2.0 * R2 * D * 2 * Fp + 2.0 * R2 * Fp * D)
        total += 2.0 * B * k * V * D                                            # guidance logits: ONE pass over the mixed embeddings (round 3)
    return total


def reference_flops_per_generate(tr, B, n, m_text, nc, timesteps):
    """what the reference computes for the same call (SURVEY 8d formula): 2 full passes per step over all n rows, logits at every row"""
    cfgb = tr.transformer_blocks.cfg
    D, depth, H = tr.dim, cfgb['depth'], cfgb['heads']
    I = H * cfgb['dim_head']
    F = int(D * cfgb['ff_mult'] * 2 / 3)
    m = m_text + nc
    layer = 2.0 * n * D * I * 4 + 4.0 * H * n * (n + 1) * 64 + 2.0 * n * D * I * 2 + 2.0 * m * D * 2 * I + 4.0 * H * n * (m + 1) * 64 + 6.0 * n * D * F
    return B * 2 * timesteps * (depth * layer + 2.0 * n * D * tr.dim_out)


def parity_tier_leg(mg, tr, step, args, B, T, n, nc, image_size, counts, lib, bf16_s_per_step):
    """The same step through precision 'f16x2' (round 4; csrc/split.hip, common.h split2_f16): every GEMM / convolution operand as TWO fp16 terms of
    its fp32 value (22 significand bits), multiplied as term products on the fp16 MFMA (same rate as the bf16 one) with fp32 accumulation -- THREE
    products for general fp32 weights, TWO for a bf16-representable checkpoint --, fp32 everywhere else, self-attention as fp16 term products too, the VAE
    decode's convolutions the same way.  Round 5: the products of a k-block SHARE operands (xh.wh, xl.wh, xh.wl), so every GEMM of the tier stages each term plane
    once and runs all products from that staging (csrc/gemm_terms.hip and the NP forms of the other kernels) instead of one GEMM of depth 3 K over duplicated
    segments; FF w1 carries GEGLU + the term split + the LayerNorm(inner) statistics in its epilogue.  The tier that holds logits within 1e-3 of the reference's fp32 run and reproduces its ids bit for bit on BOTH kinds of
    checkpoint (tests/test_gpu_base_size.py, fixtures base_c2.pt and base_c2_fp32.pt).  Timed first on the raw fp32 initialisation of the main line
    (`fp32_checkpoint`: what the reference's own constructors produce), then on the same parameters rounded to bf16 -- the checkpoint the bf16
    engine effectively multiplies by.  `bf16x3` is round 3's tier (three bf16 terms per value, 6 / 3 products) on the same two checkpoints."""
    import ctypes as C
    from muse_maskgit_pytorch_amd import _lib

    def timed(steps):
        step(-100)
        torch.cuda.synchronize()
        lib.mm_profile_enable(1)
        t0 = time.perf_counter()
        marks = []
        for i in range(steps):
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            _, images, e_mid = step(i)
            marks.append((e0, e_mid))
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / steps
        lib.mm_profile_enable(0)
        prof = []
        for slot in range(2):
            cnt, ms, work = C.c_int64(), C.c_double(), C.c_double()
            _lib.check(lib.mm_profile_read(slot, C.byref(cnt), C.byref(ms), C.byref(work)), 'mm_profile_read')
            prof.append((cnt.value, ms.value, work.value))
        assert torch.isfinite(images).all() and images.shape == (B, 3, image_size, image_size)
        return sec, sum(a.elapsed_time(b) for a, b in marks) / steps, prof

    nsteps = max(5, min(args.steps, 10))
    try:
        # ---- general fp32 weights (the raw initialisation)
        mg.set_precision('f16x2')
        sec3, loop3, _ = timed(nsteps)
        p3 = tr._model().packed['P']
        lib.mm_debug_set2(2)          # A/B in the same process: rounds 4-5's concatenated-depth kernels (no term sharing, three-kernel feed-forward)
        try:
            sec3_old, loop3_old, _ = timed(3)
        finally:
            lib.mm_debug_set2(0)
        mg.set_precision('bf16x3')
        sec6, loop6, _ = timed(2)
        p6 = tr._model().packed['P']
        # ---- the same parameters rounded to bf16
        with torch.no_grad():
            for p_ in list(mg.transformer.parameters()) + list(mg.vae.parameters()):
                p_.copy_(p_.to(torch.bfloat16).float())
        sec_b3, loop_b3, _ = timed(2)
        pb3 = tr._model().packed['P']
        mg.set_precision('f16x2')
        sec, loop_ms, prof = timed(nsteps)
        P = tr._model().packed['P']
        lib.mm_debug_set2(2)
        try:
            sec_old, loop_old, _ = timed(3)
        finally:
            lib.mm_debug_set2(0)
        g_cnt, g_ms, g_flops = prof[0]
        ex = executed_flops_per_generate(tr, B, n, args.text_len, nc, counts)
        return {
            'precision': 'f16x2', 'value': B / sec, 'unit': 'images/sec', 'ms_per_step': sec * 1e3, 'decode_loop_ms_per_step': loop_ms, 'steps': nsteps,
            'x_bf16_engine_time': sec / bf16_s_per_step, 'term_products': P,
            'checkpoint': 'the main line\'s parameters rounded to bf16 (what the bf16 engine multiplies by): one fp16 term per weight, 2 term products',
            'tolerance': 'logits <= 1e-3 absolute (measured 4e-6) and ids bit-exact at every decode step against the reference fp32 run at this size, on a '
                         'bf16-representable AND on a general fp32 checkpoint (tests/test_gpu_base_size.py, precision f16x2, fixtures base_c2.pt / base_c2_fp32.pt); '
                         'VAE decode: single fp16 terms on fp16 storage (round 6; decoded pixels 1.7e-4 of the image scale against the reference, bound 1e-3 -- the three-product term split of rounds 4-5 stays selectable: VQGanVAE.set_decode_storage(\'terms\'))',
            'roofline': {'kernel': 'gemm_wide_fused_kernel<F16, NP> (term products of to_logits on the guidance-mixed embeddings, every term plane staged once per 32-deep k-block)', 'bound': 'mfma',
                         'achieved': g_flops / (g_ms * 1e-3) / 1e12 if g_ms else None, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': (g_flops / (g_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS) if g_ms else None,
                         'flops_kind': 'executed fp16 MFMA flops (term products x 2 R V D of the one mixed pass; same peak as bf16)',
                         'algorithmic_frac': (g_flops / P / (g_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS) if g_ms else None,
                         'launches': g_cnt, 'avg_launch_ms': g_ms / g_cnt if g_cnt else None, 'traffic': _tier_traffic('gemm_wide_fused_kernel'),
                         'traffic_source': 'profiles/r06_f16x2_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of `bench.py --precision f16x2`, bytes per launch, FETCH x2 gfx950 correction)'},
            'executed_f16_tflops_decode_loop_gemms_x_products': P * ex / (loop_ms * 1e-3) / 1e12,
            'without_term_sharing': {'value': B / sec_old, 'ms_per_step': sec_old * 1e3, 'decode_loop_ms_per_step': loop_old, 'steps': 3, 'x_time': sec_old / sec,
                                     'note': 'same process, mm_debug_set2(2): the depth-P*K GEMMs over duplicated term segments of rounds 4-5 (DESIGN 3.10)'},
            'fp32_checkpoint': {'term_products': p3, 'value': B / sec3, 'ms_per_step': sec3 * 1e3, 'decode_loop_ms_per_step': loop3, 'steps': nsteps,
                                'x_bf16_engine_time': sec3 / bf16_s_per_step,
                                'without_term_sharing': {'value': B / sec3_old, 'ms_per_step': sec3_old * 1e3, 'decode_loop_ms_per_step': loop3_old, 'steps': 3,
                                                         'x_time': sec3_old / sec3},
                                'note': 'same tier on the raw fp32 initialisation (general fp32 weights, what the reference\'s constructors / training produce): three term pairs'},
            'bf16x3': {'note': 'round 3\'s tier (bf16 terms) on the same two checkpoints, 2 timed steps each',
                       'fp32_checkpoint': {'term_products': p6, 'value': B / sec6, 'ms_per_step': sec6 * 1e3, 'decode_loop_ms_per_step': loop6},
                       'bf16_checkpoint': {'term_products': pb3, 'value': B / sec_b3, 'ms_per_step': sec_b3 * 1e3, 'decode_loop_ms_per_step': loop_b3}},
        }
    finally:
        mg.set_precision('bf16')


def train_bench(args, dev, rank, world, dist):
    """MaskGit.forward (muse_maskgit_pytorch.py:623-741: random masking, cond-dropped forward, cross-entropy on the masked positions) + the
    hand-written backward (training.py) + torch.optim.AdamW on the C2 base transformer, token ids as input (no VAE in the step).  One process per
    GPU; with N > 1 the gradients are averaged by parallel.GradBucketer (bucketed asynchronous RCCL all-reduces overlapped with the backward)."""
    import muse_maskgit_pytorch_amd as mm
    from muse_maskgit_pytorch_amd.parallel import GradBucketer
    torch.manual_seed(0)
    tr = mm.MaskGitTransformer(num_tokens=65536, seq_len=256, dim=512, depth=8, dim_head=64, heads=8, ff_mult=4, t5_name='t5-small').to(dev)
    mg = mm.MaskGit(vae=None, transformer=tr, image_size=256)
    if dist is not None:
        tr.grad_sync = GradBucketer(dist)
    B, n = args.batch or 32, 256
    g = torch.Generator().manual_seed(100 + rank)
    ids = torch.randint(0, 65536, (B, n), generator=g).to(dev)
    te = synth_text(world * B, args.text_len, tr.text_embed_dim)[rank * B:(rank + 1) * B].to(dev)
    try:
        opt = torch.optim.AdamW(tr.parameters(), lr=1e-4, fused=True)      # torch's single-kernel-per-group AdamW (the optimizer is torch's on every path)
    except (RuntimeError, TypeError):
        opt = torch.optim.AdamW(tr.parameters(), lr=1e-4)
    losses = []
    from muse_maskgit_pytorch_amd import training as _training
    c_step = _training._c_step_eligible(tr, ids, te, None, False, None, None, tr.grad_sync)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = mg(ids, text_embeds=te)
        loss.backward()
        opt.step()
        losses.append(loss.detach())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if args.train_phases:      # tools: device-side duration of the phases of a step (events on the stream) and the host's time in each, printed to stderr -- not the timed run
        evs, host = [], []
        for _ in range(8):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            h = [time.perf_counter()]
            opt.zero_grad(set_to_none=True)
            e[0].record()
            loss = mg(ids, text_embeds=te); e[1].record(); h.append(time.perf_counter())
            loss.backward(); e[2].record(); h.append(time.perf_counter())
            opt.step(); e[3].record(); h.append(time.perf_counter())
            evs.append(e); host.append(h)
        torch.cuda.synchronize()
        for i in range(2, 8):
            e, h = evs[i], host[i]
            sys.stderr.write('[train phases] device: MaskGit.forward (masking + C step) %.3f ms, backward() %.3f ms, AdamW %.3f ms, to the next step\'s first event %s ms | '
                             'host: forward %.3f, backward %.3f, optimizer %.3f ms\n' % (
                                 e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3]),
                                 ('%.3f' % e[3].elapsed_time(evs[i + 1][0])) if i + 1 < 8 else '-',
                                 (h[1] - h[0]) * 1e3, (h[2] - h[1]) * 1e3, (h[3] - h[2]) * 1e3))
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=dev)
    if dist is not None:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        sec = dt.item() / args.steps
        # executed MFMA flops of one step: the Linear layers three times (forward, dX, dW), the attention products once forward and 3.5x backward
        # (7 block products against 2), the head on the labelled rows only (their count of the last step)
        cfgb = tr.transformer_blocks.cfg
        D, H, I, F, V, M, Mc = tr.dim, cfgb['heads'], cfgb['heads'] * 64, int(tr.dim * 4 * 2 / 3), tr.dim_out, B * n, B * args.text_len
        R = int(getattr(tr, '_last_train_rows', M // 2))
        lin = cfgb['depth'] * (2.0 * M * D * 3 * I + 2.0 * M * I * D + 2.0 * M * D * I + 2.0 * M * I * D + 2.0 * Mc * D * 2 * I + 2.0 * M * D * 2 * F + 2.0 * M * F * D) + 2.0 * R * D * V
        att = cfgb['depth'] * (4.0 * B * H * n * (n + 1) * 64 + 4.0 * B * H * n * (args.text_len + 1) * 64)
        ex = 3.0 * lin + 4.5 * att
        _RECORD_OUT.write(json.dumps({
            'executed_tflops': ex / sec / 1e12, 'executed_mfma_frac': ex / sec / 1e12 / PEAK_BF16_TFLOPS, 'labelled_rows_last_step': R,
            'driver': 'mm_train_step (one C call per step: forward + loss + backward; dependent chain on the caller\'s stream, parameter preparation and the leaves of the backward -- dW GEMMs, reductions -- on a second stream; csrc/train_step.hip)' if c_step else 'training.py (operator by operator over the C ABI)',
            'metric': 'training tokens/sec (C2 base transformer: MaskGit.forward + backward + AdamW)', 'value': world * B * n / sec, 'unit': 'tokens/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': sec * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16 (fp32 master weights, fp32 gradients)', 'data': 'synthetic',
            'config': {'workload': 'training step of the BASELINE configs[1] transformer (dim 512, depth 8, 256 tokens, codebook 65536), token ids in, '
                                   'cosine-schedule random masking, cond_drop_prob 0.5, AdamW', 'sequences_per_gpu_per_step': B, 'global_batch': world * B,
                       'seq_len': n, 'parallelism': f'dp{world} (bucketed gradient all-reduce overlapped with the backward)'},
            'note': 'secondary line: BASELINE.json names no training metric',
            'loss_first': float(losses[0]), 'loss_last': float(losses[-1])}) + '\n')
        _RECORD_OUT.flush()
    if dist is not None:
        dist.destroy_process_group()
    return 0


def _respawn_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: become the launcher (one process per GPU over RCCL), like the driver's
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='c2', help='c2 = the metric configuration (BASELINE configs[1]); c4 super-res, c5 paper-scale shape')
    ap.add_argument('--batch', type=int, default=None, help='images per GPU per step (default: 32; 8 for c4)')
    ap.add_argument('--timesteps', type=int, default=18)
    ap.add_argument('--text-len', type=int, default=32)
    ap.add_argument('--tiny', action='store_true', help='configs[0] plumbing case instead of the metric config')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fused-sampling', action='store_true', help='materialise the logits (round-1 path) for A/B timing')
    ap.add_argument('--fp8', action='store_true', help="secondary line: run the transformer on the fp8 engine (precision 'fp8', BASELINE configs[4] \"fp8 MFMA weights\"; "
                    "use with --config c5).  Never the headline: the metric configuration is quoted in bf16")
    ap.add_argument('--no-graph-leg', action='store_true', help='skip the extra leg that replays one captured generate (hipGraph) -- reported beside, never as, the value')
    ap.add_argument('--no-off-ideal', action='store_true', help='skip the extra legs off the ideal case (text_len 77 / 256, non-Gaussian to_logits)')
    ap.add_argument('--no-parity-tier', action='store_true', help="skip the second timed leg (precision 'f16x2', the tolerance-meeting tier)")
    ap.add_argument('--precision', choices=['bf16', 'f16x2', 'bf16x3'], default='bf16', help="secondary line: run the timed region on a precision tier instead of the bf16 engine "
                    "(profiling the tier: tools/r5_kstats.sh); never the headline")
    ap.add_argument('--fused-bound', choices=['auto', 'quantile', 'gaussian'], default='auto', help="A/B: the fused sampler's bound of the k-th largest logit (Transformer.fused_bound); 'gaussian' = rounds 2-4")
    ap.add_argument('--bf16-round-weights', action='store_true', help='secondary line: round the random-init parameters to bf16 first (the bf16-representable checkpoint of the tier figures)')
    ap.add_argument('--vae-storage', choices=['f16', 'bf16'], default='f16', help="A/B: 16-bit storage of the VAE decoder (VQGanVAE.decode_storage; 'bf16' = rounds 1-5)")
    ap.add_argument('--train-phases', action='store_true', help='with --train: print device- and host-side durations of the phases of a step to stderr (tools)')
    ap.add_argument('--train', action='store_true', help='time the TRAINING step of the C2 base transformer (MaskGit.forward + backward + AdamW) instead of '
                    'generation: a second, separately labelled line -- not the BASELINE metric')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(_respawn_under_torchrun(args))
    # stdout carries exactly ONE line, the JSON record: libraries that chat on file descriptor 1 (RCCL prints a version banner when a communicator
    # is created) are sent to stderr, the record goes to the original descriptor
    global _RECORD_OUT
    sys.stdout.flush()
    _RECORD_OUT = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs (no CPU fallback exists)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1 or os.environ.get('MM_BENCH_FORCE_DIST'):      # (the env switch exercises the RCCL path on a single GPU)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from muse_maskgit_pytorch_amd import _lib
    from muse_maskgit_pytorch_amd.parallel import allgather_ids
    _lib.require_device()
    if args.train:
        return train_bench(args, dev, rank, world, dist)
    if args.tiny:
        mg, image_size = build_models(dev, tiny=True)
        desc, cond_size = 'C1 tiny plumbing config', None
    else:
        mg, image_size = build_config(args.config, dev)
        desc, cond_size = CONFIGS[args.config][5], CONFIGS[args.config][3]
    tr = mg.transformer
    if args.fp8:
        mg.set_precision('fp8')
        args.no_parity_tier = True
    if args.bf16_round_weights:
        with torch.no_grad():
            for p_ in list(mg.transformer.parameters()) + list(mg.vae.parameters()):
                p_.copy_(p_.to(torch.bfloat16).float())
    tr.fused_bound = args.fused_bound
    if mg.vae is not None:
        mg.vae.set_decode_storage(args.vae_storage)
    if args.precision != 'bf16':
        mg.set_precision(args.precision)
        args.no_parity_tier = args.no_graph_leg = True
    B = args.batch or (32 if args.tiny else CONFIGS[args.config][4])
    T = args.timesteps
    n = (image_size // 16) ** 2
    te_all = synth_text(world * B, args.text_len, tr.text_embed_dim)
    te = te_all[rank * B:(rank + 1) * B].to(dev)
    cond = None
    if cond_size:
        cond = torch.randn(world * B, 3, cond_size, cond_size, generator=torch.Generator().manual_seed(5))[rank * B:(rank + 1) * B].to(dev)
    nc = (cond_size // 16) ** 2 if cond_size else 0

    def step(i):
        # ids + images from the one call: the VAE decode is enqueued before generate() reads its 8-byte status (the step's one host synchronisation);
        # e_mid marks the end of the decode loop on the stream
        e_mid = torch.cuda.Event(enable_timing=True)
        ids, images = mg.generate([''] * B, timesteps=T, cond_scale=3, text_embeds=te, cond_images=cond, seed=1000 + i, row_offset=rank * B, return_ids='both',
                                  fused_sampling=not args.no_fused_sampling, loop_end_event=e_mid)
        all_ids = allgather_ids(ids, dist) if dist is not None else ids      # one RCCL all-gather of token grids
        return all_ids, images, e_mid

    for i in range(args.warmup):
        step(-1 - i)
    torch.cuda.synchronize()
    lib = _lib.lib()
    lib.mm_profile_enable(1)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for i in range(args.steps):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        _, images, e_mid = step(i)
        marks.append((e0, e_mid))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lib.mm_profile_enable(0)
    rank_times = None
    if dist is not None:
        # value uses the MAX over ranks; the per-rank spread and the number of ranks RCCL actually connected go into the line so that a
        # driver-run scaling point is self-checking
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_times = [t.item() for t in every]
        elapsed = max(rank_times)
    assert torch.isfinite(images).all() and images.shape == (B, 3, image_size, image_size)

    import ctypes as C
    prof = []
    for slot in range(2):
        cnt, ms, work = C.c_int64(), C.c_double(), C.c_double()
        _lib.check(lib.mm_profile_read(slot, C.byref(cnt), C.byref(ms), C.byref(work)), 'mm_profile_read')
        prof.append((cnt.value, ms.value, work.value))
    loop_ms = sum(a.elapsed_time(b) for a, b in marks)

    if rank == 0:
        # HBM traffic per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
        # (separate runs: counters cannot be collected inside the timed region); null when no summary is committed
        pmc, pmc_file = {}, None
        for cand in ('r06_bench_b32_pmc_summary.json', 'r05_bench_b32_pmc_summary.json', 'r04_bench_b32_pmc_summary.json', 'r03_bench_b32_pmc_summary.json', 'r02_bench_b32_pmc_summary.json', 'r01_bench_b32_pmc_summary.json'):
            try:
                with open(os.path.join(ROOT, 'profiles', cand)) as f:
                    pmc, pmc_file = json.load(f), cand
                break
            except OSError:
                continue

        def traffic_of(substr):
            for name, v in pmc.items():
                if substr in name:
                    return v['hbm_bytes_per_launch']
            return None
        metric_cfg = (not args.tiny) and args.config == 'c2' and B == 32 and not args.fp8 and args.precision == 'bf16' and args.text_len == 32
        fused_on = tr._model().fused_ready and not args.no_fused_sampling
        total_images = world * B * args.steps
        value = total_images / elapsed
        passes = 2 * T
        counts = mg._mask_counts(T, n)
        ex_flops = executed_flops_per_generate(tr, B, n, args.text_len, nc, counts)
        ref_flops = reference_flops_per_generate(tr, B, n, args.text_len, nc, T)
        loop_s = loop_ms / 1e3 / args.steps
        g_cnt, g_ms, g_flops = prof[0]
        s_cnt, s_ms, s_bytes = prof[1]
        out = {
            'metric': 'images/sec (256x256 base, 18 decode steps)' if args.config == 'c2' and not args.tiny else f'images/sec ({image_size}x{image_size}, {T} decode steps, config {args.config})',
            'value': value, 'unit': 'images/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('fp8 (e4m3 weights and activations on the fp8 MFMA for the blocks\' Linear layers; attention / to_logits / sampling / VAE bf16)' if args.fp8
                      else ('bf16' if args.precision == 'bf16' else f'{args.precision} (precision tier: fp16 / bf16 term products with fp32 accumulation; secondary line)')),
            'data': 'synthetic',
            'config': {'workload': desc + (' [fp8 engine]' if args.fp8 else ''), 'images_per_gpu_per_step': B, 'global_batch': world * B, 'seq_len': n, 'timesteps': T,
                       'text_len': args.text_len, 'cond_ids': nc, 'parallelism': f'dp{world} (batch-sharded, 1 all-gather of ids per step)',
                       'weights': 'random init (module defaults, torch.manual_seed(0))',
                       'vae_decode_storage': getattr(mg.vae, 'decode_storage', None) and {'f16': 'fp16 (single fp16 terms on the fp16 MFMA, fp32 accumulation: same rate as bf16, pixels 1.7e-4 of the image scale vs the reference)', 'bf16': 'bf16', 'terms': 'bf16'}[mg.vae.decode_storage]},
            # reference-equivalent: the reference's 2 * timesteps full passes over all n positions / the decode-loop time (comparable across
            # implementations); executed: token rows that actually pass through the transformer blocks here (2B sequences per step)
            'transformer_tok_per_s_per_gpu': B * n * passes / loop_s,
            'transformer_tok_per_s_per_gpu_kind': 'reference-equivalent (36 full passes per generate; 37 % of the logits rows and the null pass\'s cross-attention are provably dead work and skipped)',
            'executed_tok_per_s_per_gpu': 2 * B * n * T / loop_s,
            'generated_tok_per_s_per_gpu': B * n / loop_s,      # image tokens produced / decode-loop time
            'decode_loop_ms_per_step': loop_ms / args.steps,
            'executed_tflops_decode_loop': ex_flops / loop_s / 1e12,
            'executed_mfma_frac_decode_loop': ex_flops / loop_s / 1e12 / PEAK_BF16_TFLOPS,
            'reference_equivalent_tflops_decode_loop': ref_flops / loop_s / 1e12,
            # the guidance logits: since round 3 ONE pass over the mixed embeddings e_null + (e_cond - e_null) * s (to_logits is linear): `achieved` counts the
            # EXECUTED flops 2 R V D per launch; the reference's two passes + combine would be twice that for the same logits
            'roofline': {'kernel': 'gemm_wide_fused_kernel (to_logits of the guidance-mixed embeddings + fused-sampling emission from the accumulators; persistent 256-token x 256-column x 64-deep MFMA GEMM, one barrier per k-step)', 'bound': 'mfma',
                         'achieved': g_flops / (g_ms * 1e-3) / 1e12 if g_ms else None, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': (g_flops / (g_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS) if g_ms else None,
                         'traffic': traffic_of('gemm_wide_fused_kernel') if metric_cfg else None,
                         'traffic_source': f'profiles/{pmc_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch, FETCH x2 gfx950 correction)' if pmc_file else None,
                         'launches': g_cnt, 'avg_launch_ms': g_ms / g_cnt if g_cnt else None,
                         'algorithmic_flops_per_launch': g_flops / g_cnt if g_cnt else None,
                         'flops_kind': 'executed (one pass); reference-equivalent (two passes + combine) = 2x',
                         'reference_equivalent_tflops': 2 * g_flops / (g_ms * 1e-3) / 1e12 if g_ms else None},
            # the sampling tail.  With fused sampling (default) the logits never reach HBM: sample_fused_kernel works on the ~15 % candidates the
            # GEMM emitted; its rate is quoted in LOGITS-EQUIVALENT bytes (4 V per row: what a logits-reading sampler must read) for comparison
            # with round 1's sample_kernel, which is what runs when fused sampling is off
            'roofline_hbm': roofline_hbm(fused_on, s_cnt, s_ms, s_bytes, traffic_of('sample_fused_kernel' if fused_on else 'sample_kernel') if metric_cfg else None, pmc_file),
            'multi_gpu': ({'ranks_in_process_group': dist.get_world_size(), 'backend': dist.get_backend(), 'ids_gather': allgather_ids.last_transport, 'ids_gather_error': allgather_ids.last_error,
                           'devices_visible_to_rank0': torch.cuda.device_count(), 'per_rank_elapsed_s': {'min': min(rank_times), 'max': max(rank_times)},
                           'per_rank_images_per_s': [B * args.steps / t_ for t_ in rank_times]} if dist is not None else None),
            'fused_sampling': {'enabled': fused_on, 'bound': tr.fused_bound, 'bound_in_use': getattr(tr._model(), 'auto_bound', None) if tr.fused_bound == 'auto' else tr.fused_bound,
                               'fallbacks_to_logits_path': mg.fused_sampling_fallbacks, 'rows_finished_by_on_device_fallback': mg.fused_row_fallbacks},
        }
        if world == 1 and not args.tiny and args.config == 'c2' and fused_on and not args.no_graph_leg and not args.fp8:
            out['hip_graph_replay'] = graph_replay_leg(mg, B, T, te, args.steps, elapsed / args.steps)
        if world == 1 and not args.tiny and args.config == 'c2' and fused_on and not args.no_off_ideal and not args.fp8 and args.precision == 'bf16':
            out['off_ideal'] = off_ideal_legs(mg, tr, args, B, T, rank, world, dev, elapsed / args.steps)
        if world == 1 and not args.no_parity_tier and not args.tiny:
            out['parity_tier'] = parity_tier_leg(mg, tr, step, args, B, T, n, nc, image_size, counts, lib, elapsed / args.steps)
        if world == 1 and not args.no_cpu_baseline and not args.tiny and args.config == 'c2':
            out['cpu_baseline'] = cpu_baseline(mg, te_all[:2], T, 3.)
        _RECORD_OUT.write(json.dumps(out) + '\n')
        _RECORD_OUT.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
