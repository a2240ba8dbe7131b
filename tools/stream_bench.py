"""Does running two half-size GEMMs on two streams beat one full-size GEMM? (tools only)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_maskgit_pytorch_amd import _lib, ops

def run(shapes, iters=30):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for name, M, N, K, f32 in shapes:
        x = torch.randn(M, K, device='cuda').bfloat16()
        w = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
        out = torch.empty(M, N, device='cuda', dtype=torch.float32 if f32 else torch.bfloat16)
        resid = torch.randn(M, N, device='cuda') if f32 else None
        h = M // 2
        def one():
            ops.gemm(x, w, out_f32=f32, resid=resid, out=out)
        def two():
            with torch.cuda.stream(s1):
                ops.gemm(x[:h], w, out_f32=f32, resid=resid[:h] if f32 else None, out=out[:h])
            with torch.cuda.stream(s2):
                ops.gemm(x[h:], w, out_f32=f32, resid=resid[h:] if f32 else None, out=out[h:])
        res = []
        for fn in (one, two):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(iters): fn()
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / iters * 1e6)
        print(f'{name:6s} one stream {res[0]:7.1f} us | two streams x half {res[1]:7.1f} us', flush=True)

run([('qkv', 16384, 1536, 512, False), ('w1', 16384, 2816, 512, False), ('out', 16384, 512, 512, True), ('big4k', 32768, 4096, 512, False)])
