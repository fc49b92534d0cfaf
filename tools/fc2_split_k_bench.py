"""Would fc2 (M 4096, N 512, K 2048: 64 tiles of 256x128) be faster as a 4-way split-K over 256 workgroups on fc1's
ping-pong LDS-DMA loop, with the fp32 partial tiles reduced inside the following LayerNorm launch?  (VERDICT r04
item 5.)  Measured with the product kernels, no new code: a split-K workgroup IS a 256x128 tile over K = 512 that
stores an fp32 tile, i.e. the K-slices stacked as rows = ONE launch of the ping-pong kernel at M = 4 x 4096, N = 512,
K = 512 with fp32 output (256 tiles, one per CU, 33.5 MB of partial sums written); the reduce is bounded from below
by a LayerNorm that reads 4 partial tensors + the residual stream instead of one tensor (timed here as the existing
LayerNorm kernel over 5 x the rows: the same bytes).  Against: today's fc2 (128x64 tiles, two K groups) + LayerNorm.
Same for proj (K 512: 2-way split leaves K = 256 per workgroup).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import _lib, ops  # noqa: E402

DEV = 'cuda'


def timeit(fn, iters=50, warm=5):
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
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    M, N = 4096, 512
    x = torch.randn(M, N, generator=g).to(DEV)
    gam, bet = torch.ones(N, device=DEV), torch.zeros(N, device=DEV)
    hs = ops.split_rows_empty(M, N, DEV)
    t_ln = timeit(lambda: ops.layernorm_split(x, gam, bet, hs))
    x5 = torch.randn(5 * M, N, generator=g).to(DEV)
    hs5 = ops.split_rows_empty(5 * M, N, DEV)
    t_ln5 = timeit(lambda: ops.layernorm_split(x5, gam, bet, hs5))
    x3 = x5[:3 * M].contiguous()
    hs3 = ops.split_rows_empty(3 * M, N, DEV)
    t_ln3 = timeit(lambda: ops.layernorm_split(x3, gam, bet, hs3))
    print(f'LayerNorm (split rows out): {t_ln:.1f} us at 4096 rows; {t_ln3:.1f} us over 3x the rows, {t_ln5:.1f} us over 5x '
          f'(= reading 2 / 4 partial tensors + x; a lower bound of the fused reduce: it writes 1x, not 3x / 5x)')
    for name, K, ways in (('fc2', 2048, 4), ('proj', 512, 2)):
        a = (torch.randn(M, K, generator=g) * 1.2).to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.05).to(DEV)
        bias = torch.randn(N, generator=g).to(DEV)
        a_s, w_s = ops.split_rows(a), ops.split_rows(w)
        out = torch.empty(M, N, device=DEV)
        t_now = timeit(lambda: ops.gemm_split(a_s, w_s, M, N, K, out=x, bias=bias, residual=x))
        cfg_now = lib.t2h_gemm_split_tile_config
        # the split-K launch: `ways` K slices of width K / ways stacked as rows
        Ks = K // ways
        a2 = (torch.randn(ways * M, Ks, generator=g) * 1.2).to(DEV)
        w2 = (torch.randn(N, Ks, generator=g) * 0.05).to(DEV)
        a2_s, w2_s = ops.split_rows(a2), ops.split_rows(w2)
        part = torch.empty(ways * M, N, device=DEV)
        lib.t2h_gemm_split_force_config(8)
        try:
            t_part = timeit(lambda: ops.gemm_split(a2_s, w2_s, ways * M, N, Ks, out=part))
        finally:
            lib.t2h_gemm_split_force_config(-1)
        t_red = t_ln5 if ways == 4 else t_ln3
        print(f'{name}: today {t_now:.1f} us (+ LayerNorm {t_ln:.1f} = {t_now + t_ln:.1f}) | {ways}-way split-K on the '
              f'ping-pong loop: partial tiles {t_part:.1f} us + reduce-in-LayerNorm >= {t_red:.1f} us = {t_part + t_red:.1f} us')


if __name__ == '__main__':
    main()
