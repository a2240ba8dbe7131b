"""Cycle stamps inside gemm_cfg2_kernel<WIDE_MIXF> (the single-pass guidance-logits GEMM with the accumulator emission); tools only.
Build the timing variant first:  tools/build_exp.sh 7 gemm_cfg -DMM_GEMM_TIMING   then   MM_LIB=.../libmuse_exp7.so python tools/mixf_timing.py"""
import ctypes, math, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_maskgit_pytorch_amd import _lib, ops
R, V, D = int(sys.argv[1]) if len(sys.argv) > 1 else 5140, 65536, int(sys.argv[2]) if len(sys.argv) > 2 else 512
KT = D // 32
torch.manual_seed(0)
W = (torch.randn(V, D, device='cuda') * D ** -0.5).bfloat16()
em = torch.randn(R, D, device='cuda').bfloat16()
Wf = W.float()
wmean = Wf.mean(0).contiguous()
wcov = ((Wf.t() @ Wf) / V - torch.outer(wmean, wmean)).bfloat16().contiguous()
k_keep = math.ceil(0.1 * V)
thr = ops.fused_threshold(em, em, 1.0, wmean, wcov, ops.fused_z(k_keep, V))
fb = ops.fused_buffers(R, V, 'cuda')
for _ in range(3):
    ops.gemm_cfg_logits_fused(em, None, W, 1.0, thr, fb)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.gemm_cfg_logits_fused(em, None, W, 1.0, thr, fb)
e1.record(); torch.cuda.synchronize()
print(f'launch: {e0.elapsed_time(e1) / 10:.4f} ms for R = {R}')
buf = (ctypes.c_ulonglong * 1024)()
_lib.lib().mm_debug_cfg2_stamps(buf, 1024)
ts = np.array(buf[:], dtype=np.int64)
per = KT + 5          # KT step stamps, loop end, barrier passed, statistics written, exchange barrier passed, stores issued
TROWS = 256 if (R >= 1024 and not (int(os.environ.get('MM_DEBUG', '0'), 0) & ((1 << 26) | (1 << 28)))) else 128
ntiles = min(((R + TROWS - 1) // TROWS) * (V // 256) // 256, 1023 // per)
print('rows per tile:', TROWS)
t = ts[1:1 + per * ntiles].reshape(ntiles, per)
d = np.diff(t, axis=1)
mid = slice(2, ntiles - 1)
print('prologue:', t[0, 0] - ts[0], 'cycles')
print('mean cycles per k-step index:', np.round(d[mid, :KT - 0].mean(0)[:KT], 0))
names = ['loop end -> tile-end barrier passed', 'statistics + candidate stores + LDS publish', 'exchange barrier', 'statistics records']
for i, nme in enumerate(names):
    print(f'{nme:45s}', round(d[mid, KT + i].mean()))
nxt = t[1:, 0] - t[:-1, -1]
print('emission end -> first step of the next tile:', round(nxt[1:].mean()))
print('tile total:', round((t[3:, 0] - t[2:-1, 0]).mean()), ' k-steps:', round(d[mid, :KT].sum(1).mean()))
