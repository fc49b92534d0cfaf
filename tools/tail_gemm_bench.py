"""The three Linears of the last layer's tail on M changed rows (default 16): tiled 64x64 kernel
(config 2) vs the few-rows kernel (config 9).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import _lib, ops  # noqa: E402

DEV = 'cuda'
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator().manual_seed(0)
lib = _lib.load()


def timeit(fn, iters=50):
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


for name, (n, k) in dict(proj=(512, 512), fc1=(2048, 512), fc2=(512, 2048)).items():
    a = (torch.randn(M, k, generator=g)).to(DEV)
    w = (torch.randn(n, k, generator=g) * 0.05).to(DEV)
    b = torch.randn(n, generator=g).to(DEV)
    ref = a.double() @ w.double().t() + b.double()
    a_s, w_s = ops.split_rows(a), ops.split_rows(w)
    out = torch.empty(M, n, device=DEV)
    line = f'{name:5s} M{M} N{n} K{k} |'
    for cfg in (2, 9):
        lib.t2h_gemm_split_force_config(cfg)
        ops.gemm_split(a_s, w_s, M, n, k, out=out, bias=b)
        err = (out.double() - ref).abs().max().item()
        t = timeit(lambda: ops.gemm_split(a_s, w_s, M, n, k, out=out, bias=b))
        line += f' cfg{cfg}: {t:6.1f} us err {err:.1e} |'
    lib.t2h_gemm_split_force_config(-1)
    print(line)
