"""Within-process A/B of the GEMM tile configurations on the hot shapes
(transformer linears at B=8 and the dominant decode conv).  GPU only."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import ops, weights  # noqa: E402

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
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    M = B * 512
    g = torch.Generator().manual_seed(0)
    res = {}
    shapes = {'qkv': (M, 1536, 512), 'proj': (M, 512, 512), 'fc1': (M, 2048, 512), 'fc2': (M, 512, 2048)}
    for name, (m, n, k) in shapes.items():
        a = torch.randn(m, k, generator=g).to(DEV)
        w = (torch.randn(n, k, generator=g) * 0.05).to(DEV)
        bias = torch.randn(n, generator=g).to(DEV)
        out = torch.empty(m, n, device=DEV)
        ops.gemm_force_config(2)
        ref = ops.gemm(a, w, bias=bias).clone()
        for cfg in (2, 3, 4, 5, 6, 7, 8):
            ops.gemm_force_config(cfg)
            ops.gemm(a, w, out=out, bias=bias)
            err = (out - ref).abs().max().item()
            us = timeit(lambda: ops.gemm(a, w, out=out, bias=bias))
            tf = 2.0 * m * n * k / us / 1e6
            res[f'{name}/{ops.GEMM_CFG_NAMES[cfg]}'] = dict(us=round(us, 1), tflops=round(tf, 1), err=err)
    # decode conv 128->128 @512x256 with GroupNorm+swish prologue (one image)
    n_img, h, w_, c = 2, 512, 256, 128
    x = torch.randn(n_img * h * w_, c, generator=g).to(DEV)
    wt = weights.pack_conv3x3(torch.randn(c, c, 3, 3, generator=g) * 0.03).to(DEV)
    sc = (torch.rand(n_img, c, generator=g) + 0.5).to(DEV)
    sh = torch.randn(n_img, c, generator=g).to(DEV)
    bias = torch.randn(c, generator=g).to(DEV)
    out = torch.empty(n_img * h * w_, c, device=DEV)
    ops.gemm_force_config(3)
    ref = ops.conv3x3(x, wt, n_img, h, w_, c, bias=bias, pro=(sc, sh, 1)).clone()
    for cfg in (2, 3, 4, 5, 7, 8):
        ops.gemm_force_config(cfg)
        ops.conv3x3(x, wt, n_img, h, w_, c, out=out, bias=bias, pro=(sc, sh, 1))
        err = (out - ref).abs().max().item()
        us = timeit(lambda: ops.conv3x3(x, wt, n_img, h, w_, c, out=out, bias=bias, pro=(sc, sh, 1)), iters=5, warm=1)
        tf = 2.0 * n_img * h * w_ * c * 9 * c / us / 1e6
        res[f'conv128/{ops.GEMM_CFG_NAMES[cfg]}'] = dict(us=round(us, 1), tflops=round(tf, 1), err=err)
    ops.gemm_force_config(-1)
    # attention + layernorm for reference
    qkv = torch.randn(M, 1536, generator=g).to(DEV)
    y = torch.empty(M, 512, device=DEV)
    us = timeit(lambda: ops.mha_noncausal(qkv, B, 512, 8, out=y))
    res['mha'] = dict(us=round(us, 1), tflops=round(4.0 * B * 8 * 512 * 512 * 64 / us / 1e6, 1))
    for k, v in res.items():
        print(k.ljust(28), v)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
