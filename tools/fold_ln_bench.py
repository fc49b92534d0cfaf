"""What the folded LayerNorm costs each GEMM (B=8 shapes): proj / fc2 with and without the split(x) +
row-moment outputs, q|k|v / fc1 on LayerNorm'ed rows vs on split(x) with the epilogue correction, and
the LayerNorm kernel they replace.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import ops, weights  # noqa: E402

DEV = 'cuda'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M, C = 512 * B, 512
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


x = r(M, C).to(DEV)
gam, bet = (r(C) * 0.1 + 1).to(DEV), (r(C) * 0.1).to(DEV)
hs, xs, part = ops.split_rows_empty(M, C, DEV), ops.split_rows_empty(M, C, DEV), ops.ln_partials_empty(M, C, DEV)
print(f'layernorm_split            {timeit(lambda: ops.layernorm_split(x, gam, bet, hs)):6.1f} us')
for name, K in (('proj', 512), ('fc2', 2048)):
    a_s = ops.split_rows(r(M, K).to(DEV))
    w_s = ops.pack_split_rows_host(r(C, K, sc=0.05)).to(DEV)
    b = r(C).to(DEV)
    x0 = x.clone()
    t0 = timeit(lambda: ops.gemm_split(a_s, w_s, M, C, K, out=x, bias=b, residual=x0))
    t1 = timeit(lambda: ops.gemm_split(a_s, w_s, M, C, K, out=x, bias=b, residual=x0, out_split=xs))
    t2 = timeit(lambda: ops.gemm_split(a_s, w_s, M, C, K, out=x, bias=b, residual=x0, out_split=xs, ln_part_out=part))
    t3 = timeit(lambda: ops.gemm_split(a_s, w_s, M, C, K, out=x, bias=b, residual=x0, ln_part_out=part))
    t4 = timeit(lambda: ops.gemm_split(a_s, w_s, M, C, K, out=x, bias=b, residual=x0))
    print(f'{name:5s} plain {t0:6.1f} | + split(x) {t1:6.1f} | + split(x) + moments {t2:6.1f} | moments only {t3:6.1f} | plain again {t4:6.1f} us')
x.copy_(r(M, C))
ops.gemm_split(ops.split_rows(r(M, C).to(DEV)), ops.pack_split_rows_host(r(C, C, sc=0.05)).to(DEV), M, C, C, out=x,
               residual=x, out_split=xs, ln_part_out=part)
ops.layernorm_split(x, gam, bet, hs)
for name, N, act in (('qkv', 1536, ops.ACT_NONE), ('fc1', 2048, ops.ACT_GELU)):
    w, b = r(N, C, sc=0.05), r(N)
    wf, cs, bf = weights.fold_layernorm(w, b, gam.cpu(), bet.cpu())
    w_s, wf, cs, bf, b = ops.pack_split_rows_host(w).to(DEV), wf.to(DEV), cs.to(DEV), bf.to(DEV), b.to(DEV)
    o = ops.split_rows_empty(M, N, DEV)
    t0 = timeit(lambda: ops.gemm_split(hs, w_s, M, N, C, out_split=o, bias=b, act=act))
    t1 = timeit(lambda: ops.gemm_split(xs, wf, M, N, C, out_split=o, bias=bf, act=act, ln_in=(part, cs)))
    t2 = timeit(lambda: ops.gemm_split(hs, w_s, M, N, C, out_split=o, bias=b, act=act))
    t3 = timeit(lambda: ops.gemm_split(xs, wf, M, N, C, out_split=o, bias=bf, act=act, ln_in=(part, cs)))
    print(f'{name:5s} on LayerNorm rows {t0:6.1f} {t2:6.1f} | on split(x), folded {t1:6.1f} {t3:6.1f} us')
