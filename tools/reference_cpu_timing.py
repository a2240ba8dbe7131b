"""BUILD CONTAINER ONLY (needs /root/reference): time the UNMODIFIED reference's MaskGit.generate and the oracle port (oracle/muse_oracle.py,
what bench.py's cpu_baseline leg runs on the GPU box, where the reference package does not exist) on the SAME host cores, same workload, so that
the port's speed is itself pinned against the reference's.  BASELINE configs[1]: dim 512, depth 8, codebook 65536, 256 tokens,
batch 2, 18 steps, cond_scale 3, fp32, VQGanVAE(dim=256) decode.     python tools/reference_cpu_timing.py [--steps 18] [--runs 3]
Prints one JSON line; the numbers of this container go into DESIGN.md section 5."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import golden_recipe as R  # noqa: E402
from reference_harness import reference_modules  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=18)
    ap.add_argument('--runs', type=int, default=3)
    args = ap.parse_args()
    threads = os.cpu_count()
    torch.set_num_threads(threads)
    pkg, mmp, vaemod, att = reference_modules()
    tr = R.build_transformer(pkg.MaskGitTransformer, peaky=False)
    vae = R.build_vae(pkg.VQGanVAE)
    mg = pkg.MaskGit(vae=vae, transformer=tr, image_size=256).eval()
    inp = R.inputs()
    te = inp['text_embeds']
    tr.encode_text = lambda texts, te=te: te

    def ref_run():
        t0 = time.perf_counter()
        with torch.no_grad():
            img = mg.generate(['a', 'b'], timesteps=args.steps, cond_scale=3.)
        assert img.shape == (2, 3, 256, 256)
        return time.perf_counter() - t0

    ref_run()                                     # warm-up
    ref_t = sorted(ref_run() for _ in range(args.runs))

    # the oracle port on the same parameters (what cpu_baseline times on the GPU box)
    sys.path.insert(0, ROOT)
    import muse_oracle as O
    sd = {k: v.detach() for k, v in tr.state_dict().items()}
    vsd = {k: v.detach().float() for k, v in mg.vae.state_dict().items() if v.is_floating_point()}
    vsd.update({k: v for k, v in mg.vae.state_dict().items() if not v.is_floating_point()})
    cfg = dict(depth=8, heads=8)
    g = torch.Generator().manual_seed(0)

    def port_run():
        t0 = time.perf_counter()
        with torch.no_grad():
            ids = O.generate_ids(lambda i, s: O.forward_with_cond_scale(sd, cfg, i, te, 3.), 2, 256, 65536,
                                 lambda s, shape: O.gumbel_from_uniform(torch.rand(shape, generator=g)), timesteps=args.steps)
            O.vae_decode_from_ids(vsd, ids.reshape(2, 16, 16))
        return time.perf_counter() - t0

    port_run()
    port_t = sorted(port_run() for _ in range(args.runs))
    med = lambda v: v[len(v) // 2]
    cpu = 'unknown'
    try:
        with open('/proc/cpuinfo') as f:
            cpu = next(line.split(':', 1)[1].strip() for line in f if line.startswith('model name'))
    except (OSError, StopIteration):
        pass
    print(json.dumps(dict(workload='BASELINE configs[1], batch 2, fp32, generate + VAE decode', steps=args.steps, torch_threads=threads, cpu_model=cpu,
                          reference_s=med(ref_t), reference_images_per_s=2 / med(ref_t), reference_runs_s=ref_t,
                          oracle_port_s=med(port_t), oracle_port_images_per_s=2 / med(port_t), oracle_port_runs_s=port_t,
                          port_over_reference_time=med(port_t) / med(ref_t))))


if __name__ == '__main__':
    main()
