"""Cycle stamps inside the 128x128 gemm_kernel (build with MM_GEMM_TIMING=1); tools only."""
import ctypes, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muse_maskgit_pytorch_amd import _lib, ops
M, N, K = 8192, 512, 512
x = torch.randn(M, K, device='cuda').bfloat16()
w = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
res = torch.randn(M, N, device='cuda')
for mode in ('bf16 out', 'fp32 out + residual'):
    for _ in range(3):
        if mode == 'bf16 out':
            ops.gemm(x, w)
        else:
            ops.gemm(x, w, out_f32=True, resid=res, out=torch.empty(M, N, device='cuda'))
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    _lib.lib().mm_debug_gemm_stamps(buf, 64)
    ts = np.array(buf[:], dtype=np.int64)
    KT = K // 64
    print(mode)
    print('  setup (addresses)          ', ts[1] - ts[0])
    print('  first tile DMA -> landed   ', ts[2] - ts[1])
    print('  k-steps                    ', np.diff(np.concatenate(([ts[2]], ts[3:3 + KT - 1]))), ' last compute', ts[40] - ts[3 + KT - 2])
    print('  barrier before epilogue    ', ts[41] - ts[40])
    print('  acc -> LDS tile            ', ts[42] - ts[41])
    print('  write-out (+ residual)     ', ts[43] - ts[42])
    print('  total                      ', ts[43] - ts[0])
