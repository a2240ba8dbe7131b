"""Per-shape timing of the 'f16x2' tier's GEMMs: term-sharing kernels (gemm_terms.hip, the NP forms of gemm_big.hip / gemm.hip) against the concatenated-depth
form of rounds 4-5 (mm_debug_set2(2)), same process, 50 back-to-back launches each.  usage: python tools/terms_gemm_timing.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_maskgit_pytorch_amd import _lib as L      # noqa: E402
from muse_maskgit_pytorch_amd import ops            # noqa: E402

DEV = 'cuda'


def timed(fn, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3      # us


def main():
    shapes = [('q|k|v', 16384, 1536, 512, False, False), ('FF w1 (fp32 out)', 16384, 2816, 512, False, False), ('FF w1 + GEGLU + split', 16384, 2816, 512, False, True),
              ('attention out + residual', 16384, 512, 512, True, False), ('FF w2 + residual', 16384, 512, 1408, True, False), ('cross q', 8192, 512, 512, False, False),
              ('cross out + residual', 8192, 512, 512, True, False)]
    print('| GEMM (tokens x N x K) | products | concatenated depth (rounds 4-5) | term sharing (round 5) | x |')
    print('|---|---|---|---|---|')
    for P in (3, 2):
        code = ops.MM_SPLIT_F16 | P
        for name, M, N, K, resid, geglu in shapes:
            g = torch.Generator().manual_seed(M + N + K)
            x = torch.randn(M, K, generator=g).to(DEV)
            w = (torch.randn(N, K, generator=g) / K ** 0.5)
            if P == 2:
                w = w.bfloat16().float()
            w = w.to(DEV)
            sc = ops.f16_weight_scale([w])
            xs, ws = ops.split_rows(x, code), ops.split_pack_weight(w, code, 64, sc)
            r = torch.randn(M, N, device=DEV) if resid else None
            flops = 2. * M * N * K * P
            if geglu:
                fn_new = lambda: ops.gemm_split_geglu(xs, ws, code, 1.0 / sc)
                h = torch.empty(M, N, dtype=torch.float32, device=DEV)
                fn_old = None
            else:
                fn_new = lambda: ops.gemm_split(xs, ws, code, 1.0 / sc, resid=r, shared=True)
                fn_old = lambda: ops.gemm_split(xs, ws, code, 1.0 / sc, resid=r, shared=False)
            t_new = timed(fn_new)
            t_old = timed(fn_old) if fn_old else float('nan')
            print(f'| {name} {M} x {N} x {K} | {P} | {t_old:.1f} us ({flops / t_old / 1e6:.0f} TFLOP/s) | {t_new:.1f} us ({flops / t_new / 1e6:.0f} TFLOP/s) | {t_old / t_new:.2f} |')


if __name__ == '__main__':
    main()
