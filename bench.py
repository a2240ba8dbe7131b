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
  cpu_baseline : the CPU oracle (port of the reference algorithm, torch fp32 on the host cores) on a bounded sample.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0          # HBM3E spec


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


def cpu_baseline(mg, te_one, timesteps, cond_scale, sample_steps=18, max_threads=32):
    """The reference algorithm (oracle port, fp32 torch on the host cores) on a bounded sample of the same workload:
    batch 1, `sample_steps` of the 18 decode steps (every reference step costs the same: it always runs the full
    2-pass transformer and the full-vocabulary tail) + one VAE decode, extrapolated to a full generate."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import muse_oracle as O
    cores = min(os.cpu_count(), max_threads)   # torch's intra-op pool stops scaling (and regresses) far below 256 threads
    torch.set_num_threads(cores)
    tr = mg.transformer
    sd = {k: (v.detach().float().cpu() if v.is_floating_point() else v.detach().cpu()) for k, v in tr.state_dict().items()}
    vsd = {k: (v.detach().float().cpu() if v.is_floating_point() else v.detach().cpu()) for k, v in mg.vae.state_dict().items()}
    cfg = dict(depth=tr.transformer_blocks.cfg['depth'], heads=tr.transformer_blocks.cfg['heads'])
    n, V = tr.seq_len, tr.num_tokens
    sample_steps = min(sample_steps, timesteps)
    counts = O.mask_counts(timesteps, n)
    temps = O.step_temperatures(timesteps, 1.)
    ids = torch.full((1, n), tr.mask_id, dtype=torch.long)
    scores = torch.zeros(1, n)
    g = torch.Generator().manual_seed(0)
    t0 = time.perf_counter()
    with torch.no_grad():
        for s in range(sample_steps):
            sel = O.select_topk_stable(scores, counts[s])
            ids = torch.where(sel, torch.full_like(ids, tr.mask_id), ids)
            logits = O.forward_with_cond_scale(sd, cfg, ids, te_one, cond_scale)
            gum = O.gumbel_from_uniform(torch.rand(1, n, V, generator=g))
            ids, scores, _ = O.sample_step(logits, gum, ids, tr.mask_id, temps[s])
        t1 = time.perf_counter()
        f = int(math.isqrt(n))
        O.vae_decode_from_ids(vsd, ids.clamp(max=V - 1).reshape(1, f, f))
        t2 = time.perf_counter()
    per_image = (t1 - t0) / sample_steps * timesteps + (t2 - t1)
    return dict(value=1.0 / per_image, unit='images/sec', cores=cores, kind='port',
                sample=f'batch 1, {sample_steps} of {timesteps} decode steps ({t1 - t0:.1f} s) + 1 VAE decode ({t2 - t1:.1f} s), '
                       f'extrapolated x{timesteps / sample_steps:g}; oracle/muse_oracle.py fp32 torch')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU per step')
    ap.add_argument('--timesteps', type=int, default=18)
    ap.add_argument('--text-len', type=int, default=32)
    ap.add_argument('--tiny', action='store_true', help='configs[0] plumbing case instead of the metric config')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}'
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs (no CPU fallback exists)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1 or os.environ.get('MM_BENCH_FORCE_DIST'):      # (the env switch exercises the RCCL path on a single GPU)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from muse_maskgit_pytorch_amd import _lib
    from muse_maskgit_pytorch_amd.parallel import allgather_ids
    _lib.require_device()
    mg, image_size = build_models(dev, tiny=args.tiny)
    tr = mg.transformer
    B, T = args.batch, args.timesteps
    n = (image_size // 16) ** 2
    te_all = synth_text(world * B, args.text_len, tr.text_embed_dim)
    te = te_all[rank * B:(rank + 1) * B].to(dev)

    def step(i):
        ids = mg.generate([''] * B, timesteps=T, cond_scale=3, text_embeds=te, seed=1000 + i, row_offset=rank * B, return_ids=True)
        e_mid = torch.cuda.Event(enable_timing=True)
        e_mid.record()
        all_ids = allgather_ids(ids, dist) if dist is not None else ids      # one RCCL all-gather of token grids
        images = mg.vae.decode_from_ids(ids)
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
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
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
        pmc = {}
        try:
            with open(os.path.join(ROOT, 'profiles', 'r01_bench_b32_pmc_summary.json')) as f:
                for name, v in json.load(f).items():
                    pmc[name] = v
        except OSError:
            pass

        def traffic_of(substr):
            for name, v in pmc.items():
                if substr in name:
                    return v['hbm_bytes_per_launch']
            return None
        total_images = world * B * args.steps
        value = total_images / elapsed
        passes = 2 * T
        tok_s_gpu = B * n * passes * args.steps / (loop_ms / 1e3)
        g_cnt, g_ms, g_flops = prof[0]
        s_cnt, s_ms, s_bytes = prof[1]
        out = {
            'metric': 'images/sec (256x256 base, 18 decode steps)', 'value': value, 'unit': 'images/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': ('C1 tiny plumbing config' if args.tiny else
                                    'BASELINE configs[1]: base 256x256, MaskGit.generate (18 steps, cond_scale 3, top-k 0.9, '
                                    'Philox Gumbel noise) + VQGanVAE(dim=256, codebook 65536) decode'),
                       'images_per_gpu_per_step': B, 'global_batch': world * B, 'seq_len': n, 'timesteps': T,
                       'text_len': args.text_len, 'parallelism': f'dp{world} (batch-sharded, 1 all-gather of ids per step)',
                       'weights': 'random init (module defaults, torch.manual_seed(0))'},
            'transformer_tok_per_s_per_gpu': tok_s_gpu,
            'decode_loop_ms_per_step': loop_ms / args.steps,
            'roofline': {'kernel': 'gemm_cfg2_kernel (to_logits + classifier-free guidance, persistent 128-token x 256-column MFMA GEMM)', 'bound': 'mfma',
                         'achieved': g_flops / (g_ms * 1e-3) / 1e12 if g_ms else None, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': (g_flops / (g_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS) if g_ms else None,
                         'traffic': traffic_of('gemm_cfg2_kernel') if not args.tiny and B == 32 else None,
                         'traffic_source': 'profiles/r01_bench_b32_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch, FETCH x2 gfx950 correction)',
                         'launches': g_cnt, 'avg_launch_ms': g_ms / g_cnt if g_cnt else None,
                         'algorithmic_flops_per_launch': g_flops / g_cnt if g_cnt else None},
            'roofline_hbm': {'kernel': 'sample_kernel (top-k + Gumbel argmax + confidence)', 'bound': 'hbm',
                             'achieved': s_bytes / (s_ms * 1e-3) / 1e9 if s_ms else None, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                             'frac': (s_bytes / (s_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if s_ms else None,
                             'traffic': traffic_of('sample_kernel') if not args.tiny and B == 32 else None,
                             'algorithmic_bytes_per_launch': s_bytes / s_cnt if s_cnt else None,
                             'launches': s_cnt, 'avg_launch_ms': s_ms / s_cnt if s_cnt else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(mg, te_all[:1], T, 3.)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
