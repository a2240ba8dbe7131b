"""tools: the weight-gradient GEMM without transposed copies (csrc/gemm_tn.hip) against transpose + transpose + NT GEMM, per shape of the training step
(C2 transformer, B = 32: 8192 rows; the head: ~5500 labelled rows x vocabulary).  usage: python tools/tn_gemm_timing.py"""
import sys
import torch
sys.path.insert(0, '.')
from muse_maskgit_pytorch_amd import ops


def t_us(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    dev = 'cuda'
    for name, rows, N, K in (('out-proj dW', 8192, 512, 512), ('w1 dW', 8192, 2816, 512), ('w2 dW', 8192, 512, 1408), ('qkv dW', 8192, 1536, 512),
                             ('cross kv dW', 1024, 1024, 512), ('head dW', 5457, 65536, 512)):
        dy = (torch.randn(rows, N, device=dev) * 0.1).to(torch.bfloat16)
        x = (torch.randn(rows, K, device=dev) * 0.1).to(torch.bfloat16)
        tn = t_us(lambda: ops.gemm_wgrad_tn(dy, x))
        tr = t_us(lambda: (ops.transpose(dy, pad_to=64), ops.transpose(x, pad_to=64)))
        dyt, xt = ops.transpose(dy, pad_to=64), ops.transpose(x, pad_to=64)
        nt = t_us(lambda: ops.gemm_wgrad(dyt, xt))
        tr1 = t_us(lambda: ops.transpose(dy, pad_to=64))
        flops = 2.0 * rows * N * K
        print(f'{name:12s} rows {rows:5d} N {N:5d} K {K:4d}: TN {tn:7.1f} us ({flops / tn / 1e6:6.1f} TFLOP/s) | NT {nt:7.1f} + transposes {tr:6.1f} (dY alone {tr1:5.1f}) = {nt + tr:7.1f} us (the step pays NT + dY transpose = {nt + tr1:7.1f})')


if __name__ == '__main__':
    main()
