"""GPU microbenchmark of the MFMA GEMM family on the shapes of the C2 base config (tools only, not a test)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_maskgit_pytorch_amd import _lib, ops

SHAPES = [('xq', 8192, 512, 512), ('xkv', 1024, 1024, 512), ('qkv', 16384, 1536, 512), ('out', 16384, 512, 512), ('w1', 16384, 2816, 512), ('w1geglu', 16384, 2816, 512), ('w1geglu', 16384, 5632, 1024), ('w2', 16384, 512, 1408),
          ('big', 8192, 8192, 8192), ('logits', 8192, 65536, 512)]


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    dev = 'cuda'
    args = sys.argv[1:]
    pad = 0
    if args and args[0].startswith('pad'):      # operand row stride = K + pad elements (de-aligns the power-of-two row pitch)
        pad = int(args[0][3:]); args = args[1:]
    flags = [int(f) for f in (args or ['0', '4096', '8'])]
    for name, M, N, K in SHAPES:
        x = torch.randn(M, K + pad, device=dev).bfloat16()[:, :K]
        w = (torch.randn(N, K + pad, device=dev) * 0.05).bfloat16()[:, :K]
        out_bf = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        line = f'{name:8s} M={M:6d} N={N:6d} K={K:5d}'
        for fl in flags:
            _lib.lib().mm_debug_set(fl)
            if name == 'logits':
                o32 = torch.empty(M // 2, N, device=dev, dtype=torch.float32)
                t = timeit(lambda: ops.gemm_cfg_logits(x[:M // 2], x[M // 2:], w, 3.0, out=o32), 5)
            elif name == 'w1geglu':
                t = timeit(lambda: ops.gemm_geglu(x, w))
            else:
                t = timeit(lambda: ops.gemm(x, w, out=out_bf))
            line += f' | dbg{fl}: {t * 1e6:8.1f} us {2 * M * N * K / t / 1e12:7.1f} TF'
        _lib.lib().mm_debug_set(0)
        print(line, flush=True)


if __name__ == '__main__':
    main()
