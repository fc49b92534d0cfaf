"""Probe: cost of the value-plane (Vt) epilogue of the q|k|v projection as a function of
how many of the column tiles take it.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import ops  # noqa: E402


def timeit(fn, iters=60, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


M, C, T = 4096, 512, 512
g = torch.Generator().manual_seed(0)
a = ops.split_rows((torch.randn(M, C, generator=g) * 1.3).cuda())
wq = ops.split_rows((torch.randn(3 * C, C, generator=g) * 0.05).cuda())
bq = torch.randn(3 * C).cuda()
s3 = ops.split_rows_empty(M, 3 * C, 'cuda')
for rep in range(2):
    line = []
    for col0 in (None, 1024, 512, 0):
        if col0 is None:
            t = timeit(lambda: ops.gemm_split(a, wq, M, 3 * C, C, out_split=s3, bias=bq))
        else:
            vt = ops.vt_empty(M // T, (3 * C - col0) // 64, T, 'cuda')
            t = timeit(lambda: ops.gemm_split(a, wq, M, 3 * C, C, out_split=s3, bias=bq, vt=vt, vt_col0=col0, vt_T=T))
        line.append(f'vt_col0={col0}: {t:5.1f} us')
    print(' | '.join(line))
