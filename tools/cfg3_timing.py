"""Cycle stamps inside gemm_cfg3_kernel (tools/build_exp.sh <N> gemm_cfg3 -DMM_GEMM_TIMING, run with MM_LIB=.../libmuse_exp<N>.so MM_DEBUG=0x4000000);
tools only.  The stamps perturb the phases they measure: use them for the shape of a step, ablation builds (MM_EXP) for magnitudes.
Per k-step and wave group: C start | MFMAs issued | (wait, barrier) L start | DMA issued | load-phase work done | waits done | (barrier)."""
import ctypes, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_maskgit_pytorch_amd import _lib, ops
M, N, K = 5140, 65536, 512
KT = K // 32
x = torch.randn(2 * M, K, device='cuda').bfloat16()
w = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
o = torch.empty(M, N, device='cuda')
for _ in range(3):
    ops.gemm_cfg_logits(x[:M], x[M:], w, 3.0, out=o)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
ctypes.CDLL(_lib.LIB_PATH).mm_debug_cfg3_stamps(buf, 4096)      # same handle as the loaded library
ts = np.array(buf[:], dtype=np.int64).reshape(2, 2048)
NS = 6
for grp, name in ((0, 'A'), (1, 'B')):
    t = ts[grp][: (2048 // NS) * NS].reshape(-1, NS)
    steps = t.shape[0]
    tiles = steps // KT
    t = t[: tiles * KT].reshape(tiles, KT, NS)
    seg = np.diff(t, axis=2)                                     # [tile][kt][5]: C, wait+barrier, DMA issue, L work, final wait
    mid = slice(2, tiles - 1)
    names = ['C (MFMA issue)', 'wait+barrier', 'DMA issue', 'L work', 'final wait']
    print(f'group {name}: {tiles} tiles; shader-clock cycles (each stamp itself costs ~100: s_memtime + lgkmcnt(0) + one store)')
    for i, nm in enumerate(names):
        print(f'  {nm:16s}', np.round(seg[mid, :, i].mean(0), 1))
    bar = t[mid, 1:, 0] - t[mid, :-1, -1]
    print(f'  {"end barrier":16s}', np.round(bar.mean(0), 1))
    print('  step total      ', np.round((t[mid, 1:, 0] - t[mid, :-1, 0]).mean(0), 1), ' tile:', (t[3:, 0, 0] - t[2:-1, 0, 0]).mean())
