"""Vendor-library yardstick for the GEMM shapes of the C2 decode loop (TOOLS ONLY: nothing under muse_maskgit_pytorch_amd/ calls a library GEMM).

torch.mm / F.linear on ROCm dispatch to hipBLASLt (or rocBLAS/Tensile) for bf16: timing them on the same [tokens][K] x [N][K]^T shapes, back to back
with this repo's kernels through `ops.gemm`, says how far the hand-written kernels are from what the vendor's tuned assembly reaches on this box at this
clock.  The library produces a plain bf16 (or fp32) matrix; the repo's kernels at these call sites also carry GEGLU / LayerNorm statistics / fp32 residual /
the sampling emission, so the comparison is a floor for the bare contraction, not a like-for-like replacement.

usage (GPU box):  python tools/hipblaslt_yardstick.py            # prints one line per shape, JSON at the end
                  rocprofv3 --kernel-trace --stats -- python tools/hipblaslt_yardstick.py   # to see which library kernels (macro-tiles) ran
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_maskgit_pytorch_amd import ops

# name, tokens M, out features N, K, what the repo's kernel at this call site additionally does
SHAPES = [
    ('ff_w1', 16384, 2816, 512, 'GEGLU + LayerNorm(inner) statistics, LN(dim) fold'),
    ('self_qkv', 16384, 1536, 512, 'LN(dim) fold'),
    ('self_out', 16384, 512, 512, 'fp32 residual in/out, bf16 rows + LN statistics'),
    ('ff_w2', 16384, 512, 1408, 'LN(inner) fold, fp32 residual in/out, bf16 rows + LN statistics'),
    ('cross_q', 8192, 512, 512, 'LN(dim) fold'),
    ('cross_out', 8192, 512, 512, 'fp32 residual in/out, bf16 rows + LN statistics'),
    ('to_logits', 5140, 65536, 512, 'no logits written: softmax statistics + top-k candidates emitted from the accumulators'),
]


def time_us(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    dev = 'cuda'
    torch.manual_seed(0)
    rows = []
    for name, M, N, K, extra in SHAPES:
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        iters = 20 if N > 10000 else 100
        out_dtype = torch.float32 if name == 'to_logits' else torch.bfloat16
        lib_out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_lib = time_us(lambda: torch.mm(x, w.t(), out=lib_out), iters)
        if name == 'to_logits':
            o = torch.empty(M, N, device=dev, dtype=torch.float32)
            t_own = time_us(lambda: ops.gemm(x, w, out_f32=True, out=o), iters)
        else:
            o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            t_own = time_us(lambda: ops.gemm(x, w, out=o), iters)
        err = (o.float() - lib_out.float()).abs().max().item()
        fl = 2.0 * M * N * K
        rows.append(dict(shape=name, M=M, N=N, K=K, library_us=t_lib, library_tflops=fl / t_lib / 1e6, plain_kernel_us=t_own, plain_kernel_tflops=fl / t_own / 1e6,
                         plain_over_library=t_own / t_lib, max_abs_diff=err, call_site_extra=extra))
        print(f'{name:10s} {M:6d} x {N:6d} x {K:5d}  library {t_lib:8.1f} us {fl / t_lib / 1e6:7.1f} TF/s | this repo (plain epilogue) {t_own:8.1f} us {fl / t_own / 1e6:7.1f} TF/s'
              f' | ratio {t_own / t_lib:5.2f} | max diff {err:.3g}', flush=True)
    print(json.dumps(dict(device=torch.cuda.get_device_name(0), torch=torch.__version__, rows=rows)))


if __name__ == '__main__':
    main()
