"""A/B of the split-precision (2xfp16, 3 products) GEMM against the exact-fp32
MFMA GEMM on the sampler shapes: accuracy vs an fp64 reference and time.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import _lib, ops  # noqa: E402

DEV = 'cuda'


def timeit(fn, iters=20, warm=3):
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
    CFGS = [int(c) for c in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 6, 8]
    M = B * 512
    g = torch.Generator().manual_seed(0)
    shapes = {'qkv': (M, 1536, 512), 'proj': (M, 512, 512), 'fc1': (M, 2048, 512), 'fc2': (M, 512, 2048)}
    lib = _lib.load()
    for name, (m, n, k) in shapes.items():
        a = (torch.randn(m, k, generator=g) * 1.3).to(DEV)
        w = (torch.randn(n, k, generator=g) * 0.05).to(DEV)
        bias = torch.randn(n, generator=g).to(DEV)
        ref = (a.double() @ w.double().t() + bias.double())
        scale = ref.abs().mean().item()
        out = torch.empty(m, n, device=DEV)
        ops.gemm(a, w, out=out, bias=bias)
        e32 = (out.double() - ref).abs().max().item()
        t32 = timeit(lambda: ops.gemm(a, w, out=out, bias=bias))
        a_s = ops.split_rows(a)
        w_s = ops.pack_split_rows_host(w.cpu()).to(DEV)
        w_s2 = ops.split_rows(w)
        assert torch.equal(w_s, w_s2.view_as(w_s)), 'device split != host split'
        t_split3 = timeit(lambda: ops.split_rows(a, out=a_s))
        line = f'{name:5s} M{m} N{n} K{k} | fp32: {t32:6.1f} us err {e32:.2e} | split(A) {t_split3:5.1f} us |'
        for cfg in CFGS:
            lib.t2h_gemm_split_force_config(cfg)
            ops.gemm_split(a_s, w_s, m, n, k, out=out, bias=bias)
            es = (out.double() - ref).abs().max().item()
            ts = timeit(lambda: ops.gemm_split(a_s, w_s, m, n, k, out=out, bias=bias))
            line += f' cfg{cfg}: {ts:6.1f} us ({2.0 * m * n * k / ts / 1e6:6.1f} TF/s eq) err {es:.2e} |'
        lib.t2h_gemm_split_force_config(-1)
        # split-row output round trip
        o_s = ops.split_rows_empty(m, n, DEV)
        ops.gemm_split(a_s, w_s, m, n, k, out_split=o_s, bias=bias, act=ops.ACT_GELU)
        planes = ops.unsplit_rows_host(o_s, m, n)
        eg = (planes.double() - torch.nn.functional.gelu(ref.cpu())).abs().max().item()
        print(line + f' gelu+split-out err {eg:.2e} | ref scale {scale:.2f}')


if __name__ == '__main__':
    main()
