"""Runs MaskGit.generate on the other BASELINE.json configurations at full size (tools only: sanity + timing, bf16).
C4 super-res 512x512 (seq_len 1024, cond_image_size 256, B = 8) and the C5 paper-scale shape (dim 1024, depth 24, heads 16, codebook 8192,
B = 32) -- C5 both in bf16 (fused engine) and with fp8 weights (W8A16, stepwise operator path)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import muse_maskgit_pytorch_amd as mm  # noqa: E402


def timed(fn, n=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


def main():
    dev = 'cuda'
    torch.manual_seed(0)
    which = sys.argv[1:] or ['c4', 'c5']
    if 'c4' in which:
        vae = mm.VQGanVAE(dim=256, codebook_size=65536)
        tr = mm.MaskGitTransformer(num_tokens=65536, seq_len=1024, dim=512, depth=8, dim_head=64, heads=8, t5_name='t5-small')
        mg = mm.MaskGit(vae=vae, transformer=tr, image_size=512, cond_image_size=256).to(dev).eval()
        B = 8
        te = bench.synth_text(B, 32, 512).to(dev)
        cond = torch.randn(B, 3, 256, 256, device=dev)
        dt, img = timed(lambda: mg.generate([''] * B, cond_images=cond, text_embeds=te, timesteps=18, seed=1))
        assert img.shape == (B, 3, 512, 512) and torch.isfinite(img).all()
        print(f'C4 super-res 512x512 (n=1024, 256 cond ids + text, B={B}): {dt * 1e3:.1f} ms per generate = {B / dt:.1f} images/s', flush=True)
        del mg, vae, tr
    if 'c5' in which:
        tr = mm.MaskGitTransformer(num_tokens=8192, seq_len=256, dim=1024, depth=24, dim_head=64, heads=16, t5_name='t5-small')
        vae = mm.VQGanVAE(dim=256, codebook_size=8192)
        mg = mm.MaskGit(vae=vae, transformer=tr, image_size=256).to(dev).eval()
        B = 32
        te = bench.synth_text(B, 32, 512).to(dev)
        dt, img = timed(lambda: mg.generate([''] * B, text_embeds=te, timesteps=18, seed=1))
        assert img.shape == (B, 3, 256, 256) and torch.isfinite(img).all()
        n, D, I, F, V, Lt, depth = 256, 1024, 1024, 2730, 8192, 32, 24
        layer = 2 * n * D * I * 4 + 4 * 16 * n * (n + 1) * 64 + 2 * n * D * I * 2 + 4 * 16 * n * (Lt + 1) * 64 + 6 * n * D * F
        flops = B * 36 * depth * layer
        print(f'C5 shape in bf16 (dim 1024, depth 24, heads 16, V=8192, B={B}): {dt * 1e3:.1f} ms per generate = {B / dt:.1f} images/s, '
              f'~{flops / dt / 1e12:.0f} TFLOP/s on the transformer blocks', flush=True)
        tr.quantize_weights_fp8()
        dt8, img8 = timed(lambda: mg.generate([''] * B, text_embeds=te, timesteps=18, seed=1), n=1)
        assert img8.shape == (B, 3, 256, 256) and torch.isfinite(img8).all()
        print(f'C5 with fp8 (e4m3) weights, W8A16, operator-by-operator path: {dt8 * 1e3:.1f} ms per generate = {B / dt8:.1f} images/s', flush=True)


if __name__ == '__main__':
    main()
