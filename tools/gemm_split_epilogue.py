"""Cost of the split GEMM's epilogue variants on the sampler shapes (B=8).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import ops  # noqa: E402


def timeit(fn, iters=40, warm=5):
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


M, C, T, H = 4096, 512, 512, 8
g = torch.Generator().manual_seed(0)
a = ops.split_rows((torch.randn(M, C, generator=g) * 1.3).cuda())
a4 = ops.split_rows((torch.randn(M, 4 * C, generator=g) * 0.7).cuda())
wq = ops.split_rows((torch.randn(3 * C, C, generator=g) * 0.05).cuda())
w1 = ops.split_rows((torch.randn(4 * C, C, generator=g) * 0.05).cuda())
w2 = ops.split_rows((torch.randn(C, 4 * C, generator=g) * 0.05).cuda())
bq, b1, b2 = torch.randn(3 * C).cuda(), torch.randn(4 * C).cuda(), torch.randn(C).cuda()
o3, o4, o1 = torch.empty(M, 3 * C).cuda(), torch.empty(M, 4 * C).cuda(), torch.empty(M, C).cuda()
res = torch.randn(M, C).cuda()
s3, s4 = ops.split_rows_empty(M, 3 * C, 'cuda'), ops.split_rows_empty(M, 4 * C, 'cuda')
vt = ops.vt_empty(M // T, H, T, 'cuda')
print(f'qkv  fp32 out {timeit(lambda: ops.gemm_split(a, wq, M, 3 * C, C, out=o3, bias=bq)):6.1f} us | '
      f'split out {timeit(lambda: ops.gemm_split(a, wq, M, 3 * C, C, out_split=s3, bias=bq)):6.1f} | '
      f'split q,k + Vt {timeit(lambda: ops.gemm_split(a, wq, M, 3 * C, C, out_split=s3, bias=bq, vt=vt, vt_col0=2 * C, vt_T=T)):6.1f}')
print(f'fc1  fp32 out {timeit(lambda: ops.gemm_split(a, w1, M, 4 * C, C, out=o4, bias=b1)):6.1f} us | '
      f'+GELU {timeit(lambda: ops.gemm_split(a, w1, M, 4 * C, C, out=o4, bias=b1, act=ops.ACT_GELU)):6.1f} | '
      f'GELU + split out {timeit(lambda: ops.gemm_split(a, w1, M, 4 * C, C, out_split=s4, bias=b1, act=ops.ACT_GELU)):6.1f} | '
      f'split out, no GELU {timeit(lambda: ops.gemm_split(a, w1, M, 4 * C, C, out_split=s4, bias=b1)):6.1f}')
print(f'fc2  fp32 out {timeit(lambda: ops.gemm_split(a4, w2, M, C, 4 * C, out=o1, bias=b2)):6.1f} us | '
      f'+residual {timeit(lambda: ops.gemm_split(a4, w2, M, C, 4 * C, out=o1, bias=b2, residual=res)):6.1f}')
x = torch.randn(M, C).cuda()
gg, bb = torch.ones(C).cuda(), torch.zeros(C).cuda()
hs = ops.split_rows_empty(M, C, 'cuda')
print(f'layernorm -> split rows {timeit(lambda: ops.layernorm_split(x, gg, bb, hs)):6.1f} us')
