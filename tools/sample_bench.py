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
args = sys.argv[1:]
if args and args[0] == 'insitu':
    # the logits the C2 bench model actually produces at step 0 (all positions masked), not a synthetic Gaussian
    args = args[1:]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    mg, _ = bench.build_models('cuda')
    te = bench.synth_text(32, 16, 512).cuda()
    ids = torch.full((32, 256), mg.transformer.mask_id, device='cuda', dtype=torch.long)
    logits = mg.transformer.forward_with_cond_scale(ids, text_embeds=te, cond_scale=3.).reshape(R, V).contiguous()
    print('in-situ logits: mean %.3f std %.3f min %.3f max %.3f' % (logits.mean().item(), logits.std().item(), logits.min().item(), logits.max().item()))
    r0 = logits[0]
    print('row0 std %.3f skew %.3f kurt %.3f' % (r0.std().item(), (((r0 - r0.mean()) / r0.std()) ** 3).mean().item(), (((r0 - r0.mean()) / r0.std()) ** 4).mean().item()))
else:
    logits = torch.randn(R, V, device='cuda') * 0.6
k = math.ceil(0.1 * V)
Rs = [R]
if args and args[0] == 'rows':
    Rs = [256, 512, 1024, 2048, 8192]; args = args[1:]
for fl in [int(a) for a in (args or ['0'])]:
    _lib.lib().mm_debug_set(fl)
    for Rn in Rs:
        t = timeit(lambda: ops.sample_rows(logits[:Rn], k, 0.7, noise_kind=_lib.MM_NOISE_PHILOX, seed=1))
        print(f'dbg{fl:4d} rows {Rn:5d}: {t*1e6:8.1f} us  {Rn*V*4/t/1e9:7.1f} GB/s  {t*1e6/((Rn+255)//256):6.1f} us per row-round', flush=True)
_lib.lib().mm_debug_set(0)
