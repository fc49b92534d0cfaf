"""The sampler's four Linears at B = 8 / 32 on fp16-plane operands (three fp16 products) against x8 operands (fp16
hi*hi + one 8-bit instruction for both cross terms), real data and scales, every epilogue the sampler uses; for fc1
also the x8 kernel with an fp16-plane output (what the 8-bit output planes cost).  Interleaved repetitions, median.
GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import ops  # noqa: E402

DEV = 'cuda'


def timeit(fn, iters=40, warm=4):
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


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    M, C = 512 * B, 512
    g = torch.Generator().manual_seed(0)
    for name, (n, k, gelu, res, split_out) in dict(qkv=(1536, 512, False, False, True), proj=(512, 512, False, True, False),
                                                   fc1=(2048, 512, True, False, True), fc2=(512, 2048, False, True, False)).items():
        a = (torch.randn(M, k, generator=g) * 1.2).to(DEV)
        w = (torch.randn(n, k, generator=g) * 0.05).to(DEV)
        bias = torch.randn(n, generator=g).to(DEV)
        sa, sw = ops.x8_scale_for(float(a.abs().max())), ops.x8_scale_for(float(w.abs().max()), 256.0)
        a1, w1 = ops.split_rows(a), ops.split_rows(w)
        a8, w8 = ops.split_rows_x8(a, sa), ops.split_rows_x8(w, sw)
        x = torch.randn(M, n, generator=g).to(DEV)
        o_s = ops.split_rows_empty(M, n, DEV)
        act = ops.ACT_GELU if gelu else ops.ACT_NONE
        kw = dict(bias=bias, act=act)
        if res:
            kw.update(out=x, residual=x)
        else:
            kw.update(out_split=o_s)
        vt = ops.vt_empty(B, 8, 512, DEV) if name == 'qkv' else None
        if vt is not None:
            kw.update(vt=vt, vt_col0=1024, vt_T=512)
        variants = {'fp16 planes': lambda: ops.gemm_split(a1, w1, M, n, k, **kw),
                    'x8': lambda: ops.gemm_split(a8, w8, M, n, k, x8=(sa, sw), **(dict(kw, out_x8_scale=16.0) if name == 'fc1' else kw))}
        if name == 'fc1':
            variants['x8, fp16-plane output'] = lambda: ops.gemm_split(a8, w8, M, n, k, x8=(sa, sw), **kw)
        times = {v: [] for v in variants}
        for rep in range(5):
            for v, fn in variants.items():
                times[v].append(timeit(fn))
        line = f'{name:4s} M{M} N{n} K{k}:'
        for v in variants:
            t = sorted(times[v])[2]
            line += f'  {v}: {t:6.1f} us [{min(times[v]):.1f}-{max(times[v]):.1f}]'
        print(line, flush=True)
    assert ops.split_overflow_bits(reset=True) == 0


if __name__ == '__main__':
    main()
