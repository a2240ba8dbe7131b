"""tools only: time the pieces of the fused sampling path at the bench shape (R = 5140 rows, V = 65536): bound estimate, finishing kernel
with / without noise, against sample_rows on the same logits."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_maskgit_pytorch_amd import _lib, ops

DEV = 'cuda'
R, V, D = 5140, 65536, 512
torch.manual_seed(0)
W = (torch.randn(V, D) * (D ** -0.5)).to(torch.bfloat16).to(DEV)
ec, en = torch.randn(R, D).to(torch.bfloat16).to(DEV), torch.randn(R, D).to(torch.bfloat16).to(DEV)
Wf = W.float()
wmean = Wf.mean(0).contiguous()
wcov = ((Wf.t() @ Wf) / V - torch.outer(wmean, wmean)).to(torch.bfloat16).contiguous()
k = math.ceil(0.1 * V)
z = ops.fused_z(k, V)
logits = ops.gemm_cfg_logits(ec, en, W, 3.0)
fb = ops.fused_buffers(R, V, DEV)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


thr = ops.fused_threshold(ec, en, 3.0, wmean, wcov, z)
print(f'threshold estimate      {timeit(lambda: ops.fused_threshold(ec, en, 3.0, wmean, wcov, z)):.3f} ms')
print(f'plain guidance GEMM     {timeit(lambda: ops.gemm_cfg_logits(ec, en, W, 3.0, out=logits)):.3f} ms')
print(f'fused guidance GEMM     {timeit(lambda: ops.gemm_cfg_logits_fused(ec, en, W, 3.0, thr, fb)):.3f} ms')
for name, kw in (('philox', dict(noise_kind=_lib.MM_NOISE_PHILOX, seed=1)), ('no noise', dict())):
    print(f'sample_rows   ({name:8s}) {timeit(lambda: ops.sample_rows(logits, k, 1.0, **kw)):.3f} ms')
    print(f'fused_sample  ({name:8s}) {timeit(lambda: ops.fused_sample(fb, thr, R, V, k, 1.0, **kw)):.3f} ms')
print('fail flag', int(fb['fail'].item()), ' candidates per row', float((logits >= thr[:, None]).sum(1).float().mean()))
