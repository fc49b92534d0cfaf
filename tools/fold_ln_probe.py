"""Probe: what do the folded-LayerNorm producer / consumer epilogues cost per launch?  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import ops  # noqa: E402


def timeit(fn, iters=60, warm=8):
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
a4 = ops.split_rows((torch.randn(M, 4 * C, generator=g) * 0.7).cuda())
wq = ops.split_rows((torch.randn(3 * C, C, generator=g) * 0.05).cuda())
w1 = ops.split_rows((torch.randn(4 * C, C, generator=g) * 0.05).cuda())
wp = ops.split_rows((torch.randn(C, C, generator=g) * 0.05).cuda())
w2 = ops.split_rows((torch.randn(C, 4 * C, generator=g) * 0.05).cuda())
bq, b1, bp = torch.randn(3 * C).cuda(), torch.randn(4 * C).cuda(), torch.randn(C).cuda()
x = torch.randn(M, C).cuda()
xs = ops.split_rows_empty(M, C, 'cuda')
part = torch.rand(M, C // 32, 2).cuda() + 1.0
cs3, cs4 = torch.randn(3 * C).cuda(), torch.randn(4 * C).cuda()
s3, s4 = ops.split_rows_empty(M, 3 * C, 'cuda'), ops.split_rows_empty(M, 4 * C, 'cuda')
vt = ops.vt_empty(M // T, 8, T, 'cuda')
for rep in range(2):
    print(f'proj  plain {timeit(lambda: ops.gemm_split(a, wp, M, C, C, out=x, bias=bp, residual=x)):5.1f} us | '
          f'+split out {timeit(lambda: ops.gemm_split(a, wp, M, C, C, out=x, out_split=xs, bias=bp, residual=x)):5.1f} | '
          f'+split out +partials {timeit(lambda: ops.gemm_split(a, wp, M, C, C, out=x, out_split=xs, bias=bp, residual=x, ln_part_out=part)):5.1f}')
    print(f'fc2   plain {timeit(lambda: ops.gemm_split(a4, w2, M, C, 4 * C, out=x, bias=bp, residual=x)):5.1f} us | '
          f'+split out +partials {timeit(lambda: ops.gemm_split(a4, w2, M, C, 4 * C, out=x, out_split=xs, bias=bp, residual=x, ln_part_out=part)):5.1f}')
    print(f'qkv   plain {timeit(lambda: ops.gemm_split(a, wq, M, 3 * C, C, out_split=s3, bias=bq, vt=vt, vt_col0=2 * C, vt_T=T)):5.1f} us | '
          f'folded LN in {timeit(lambda: ops.gemm_split(a, wq, M, 3 * C, C, out_split=s3, bias=bq, vt=vt, vt_col0=2 * C, vt_T=T, ln_part=part, ln_colsum=cs3)):5.1f}')
    print(f'fc1   plain {timeit(lambda: ops.gemm_split(a, w1, M, 4 * C, C, out_split=s4, bias=b1, act=ops.ACT_GELU)):5.1f} us | '
          f'folded LN in {timeit(lambda: ops.gemm_split(a, w1, M, 4 * C, C, out_split=s4, bias=b1, act=ops.ACT_GELU, ln_part=part, ln_colsum=cs4)):5.1f}')
hs = ops.split_rows_empty(M, C, 'cuda')
gg, bb = torch.ones(C).cuda(), torch.zeros(C).cuda()
print(f'layernorm -> split rows {timeit(lambda: ops.layernorm_split(x, gg, bb, hs)):5.1f} us')
