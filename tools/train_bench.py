"""Training-step microbenchmark on the C2 base transformer (tools only): forward + hand-written backward (training.py), B = 32.
Prints ms per step and the achieved model TFLOP/s (3 x forward flops of the executed rows; the logits head only runs on labelled rows)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import muse_maskgit_pytorch_amd as mm  # noqa: E402


def main():
    dev = 'cuda'
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    torch.manual_seed(0)
    tr = mm.MaskGitTransformer(num_tokens=65536, seq_len=256, dim=512, depth=8, dim_head=64, heads=8, ff_mult=4, t5_name='t5-small').to(dev)
    mg = mm.MaskGit(vae=None, transformer=tr, image_size=256)
    ids = torch.randint(0, 65536, (B, 256), device=dev)
    te = bench.synth_text(B, 32, 512).to(dev)
    opt = torch.optim.AdamW(tr.parameters(), lr=1e-4)

    def step(do_opt):
        opt.zero_grad(set_to_none=True)
        loss = mg(ids, text_embeds=te)
        loss.backward()
        if do_opt:
            opt.step()
        return loss

    for _ in range(2):
        l0 = step(True)
    torch.cuda.synchronize()
    for name, do_opt in (('fwd+bwd', False), ('fwd+bwd+AdamW', True)):
        t0 = time.perf_counter()
        N = 5
        for _ in range(N):
            l = step(do_opt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N
        n, D, I, F, V, Lt, depth = 256, 512, 512, 1365, 65536, 32, 8
        layer = 2 * n * D * I * 4 + 4 * 8 * n * (n + 1) * 64 + 2 * n * D * I * 2 + 2 * Lt * D * 2 * I + 4 * 8 * n * (Lt + 1) * 64 + 6 * n * D * F
        fwd = B * (depth * layer + 2 * n * 0.5 * D * V)          # about half the rows carry a label on average
        print(f'{name:16s} B={B}: {dt * 1e3:8.2f} ms/step  {B * n / dt / 1e3:8.1f} k tok/s  ~{3 * fwd / dt / 1e12:6.1f} TFLOP/s  loss {l.item():.4f} (first {l0.item():.4f})', flush=True)


if __name__ == '__main__':
    main()
