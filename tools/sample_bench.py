"""GPU microbenchmark / ablation of sample_kernel (tools only)."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_maskgit_pytorch_amd import _lib, ops

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters

R, V = 8192, 65536
logits = torch.randn(R, V, device='cuda') * 0.6
k = math.ceil(0.1 * V)
for fl in [int(a) for a in (sys.argv[1:] or ['0'])]:
    _lib.lib().mm_debug_set(fl)
    t = timeit(lambda: ops.sample_rows(logits, k, 0.7, noise_kind=_lib.MM_NOISE_PHILOX, seed=1))
    print(f'dbg{fl:4d}: {t*1e6:8.1f} us  {R*V*4/t/1e9:7.1f} GB/s', flush=True)
_lib.lib().mm_debug_set(0)
