"""A/B of the sampler attention at B=8: exact-fp32 MFMA kernel vs the three-product
fp16 kernel (q, k split rows + transposed v planes).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import ops  # noqa: E402


def timeit(fn, iters=30, warm=3):
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
    T, H, C = 512, 8, 512
    g = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B * T, 3 * C, generator=g) * 1.2).cuda()
    ys = ops.split_rows_empty(B * T, C, 'cuda')
    t32 = timeit(lambda: ops.mha_noncausal_split(qkv, B, T, H, ys))
    qk_s = ops.split_rows(qkv)
    vt = ops.vt_empty(B, H, T, 'cuda')
    vt.zero_()
    ts = timeit(lambda: ops.mha_split(qk_s, 3 * C, vt, B, T, H, out_split=ys))
    fl = 4.0 * T * T * 64 * H * B
    print(f'B={B}: fp32 mha {t32:6.1f} us ({fl / t32 / 1e6:5.1f} TF/s) | three-product fp16 mha {ts:6.1f} us '
          f'({fl / ts / 1e6:5.1f} TF/s fp32-equivalent, {3 * fl / ts / 1e6:6.1f} TF/s fp16)')


if __name__ == '__main__':
    main()
