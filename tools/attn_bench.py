"""GPU microbenchmark / ablation of attention_kernel at the C2 self-attention shape (tools only)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_maskgit_pytorch_amd import _lib, ops

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters

shapes = [(64, 256, 256, 'self'), (32, 256, 32, 'cross')]
if os.environ.get('ATTN_SWEEP'):      # workgroup-count sweep of the self-attention shape
    shapes = [(b_, 256, 256, 'self') for b_ in (8, 16, 32, 64, 128)]
for (b, n, j, tag) in shapes:
    h = 8
    qkv = torch.randn(b, n, 3 * h * 64, device='cuda').bfloat16()
    kv = torch.randn(b, j, 2 * h * 64, device='cuda').bfloat16()
    qs = torch.ones(64, device='cuda'); ks = torch.ones(64, device='cuda')
    nk = torch.randn(h, 64, device='cuda'); nv = torch.randn(h, 64, device='cuda')
    if tag == 'self':
        q4 = qkv.view(b, n, 3 * h, 64)[:, :, :h].permute(0, 2, 1, 3)
        k4 = qkv.view(b, n, 3 * h, 64)[:, :, h:2 * h].permute(0, 2, 1, 3)
        v4 = qkv.view(b, n, 3 * h, 64)[:, :, 2 * h:].permute(0, 2, 1, 3)
    else:
        q4 = qkv.view(b, n, 3 * h, 64)[:, :, :h].permute(0, 2, 1, 3)
        k4 = kv.view(b, j, 2 * h, 64)[:, :, :h].permute(0, 2, 1, 3)
        v4 = kv.view(b, j, 2 * h, 64)[:, :, h:].permute(0, 2, 1, 3)
    line = f'{tag:6s} b={b:3d}'
    for fl in [int(a) for a in (sys.argv[1:] or ['0'])]:
        _lib.lib().mm_debug_set(fl | (49 << 16))          # 50 launches per call from C
        t = timeit(lambda: ops.attend(q4, k4, v4, None, 8.0, True, qs, ks, nk, nv), 4) / 50
        line += f' | dbg{fl}: {t*1e6:7.1f} us'
    _lib.lib().mm_debug_set(0)
    print(line, flush=True)
