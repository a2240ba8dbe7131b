"""Cycle stamps inside gemm_cfg2_kernel (build with MM_GEMM_TIMING=1); tools only."""
import ctypes, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_maskgit_pytorch_amd import _lib, ops
M, N, K = 4096, 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 512
KT = K // 32
x = torch.randn(2 * M, K, device='cuda').bfloat16()
w = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
o = torch.empty(M, N, device='cuda')
for _ in range(3):
    ops.gemm_cfg_logits(x[:M], x[M:], w, 3.0, out=o)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 1024)()
rc = _lib.lib().mm_debug_cfg2_stamps(buf, 1024)
ts = np.array(buf[:], dtype=np.int64)
per = KT + 3
ntiles = min((M // 128) * (N // 256) // 256, 1023 // (KT + 3))
t = ts[1:1 + per * ntiles].reshape(ntiles, per)
print('prologue (kernel start -> first step):', t[0, 0] - ts[0], 'ticks (100 MHz s_memtime => x10 ns)')
d = np.diff(t, axis=1)                       # [tile][KT steps ..., loop end->barrier, barrier->combine done]
nxt = t[1:, 0] - t[:-1, -1]                  # combine done -> first step of the next tile
mid = slice(2, ntiles - 1)
print('mean ticks per k-step index:', np.round(d[mid, :KT].mean(0), 1))
print('last step end -> barrier passed:', d[mid, KT].mean(), ' barrier -> combine/ct written:', d[mid, KT + 1].mean(), ' -> next tile first step:', nxt[1:].mean())
print('tile total:', (t[3:, 0] - t[2:-1, 0]).mean(), ' sum of steps:', d[mid, :KT].sum(1).mean())
