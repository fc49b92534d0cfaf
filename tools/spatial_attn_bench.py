"""Decoder AttnBlock attention: flash-style t2h_spatial_attention_f32 vs the materialised
bmm -> softmax -> bmm form, at the shapes of the decode (B=8: N=512 top, N=2048 bottom-res;
1024x512: N=2048 / 8192 in chunks of 2 images).  GPU only.

    python tools/spatial_attn_bench.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import ops  # noqa: E402

DEV = 'cuda'


def timeit(fn, iters=10):
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
    C = 512
    for n_img, N in ((8, 512), (8, 2048), (2, 2048), (2, 8192)):
        qkv = torch.randn(n_img * N, 3 * C, device=DEV) * 0.7
        out = torch.empty(n_img * N, C, device=DEV)
        q3 = qkv.view(n_img, N, 3 * C)
        s = torch.empty((n_img, N, N), device=DEV)
        o3 = out.view(n_img, N, C)

        def old():
            ops.bgemm(q3[:, :, :C], q3[:, :, C:2 * C], s, alpha=float(C**-0.5))
            ops.softmax_rows_(s)
            ops.bgemm(s, q3[:, :, 2 * C:], o3, b_trans=True)

        t_new, t_old = timeit(lambda: ops.spatial_attention(qkv, n_img, N, C, out=out)), timeit(old)
        fl = 4.0 * n_img * N * N * C
        print(f'n_img={n_img} N={N:5d}: flash {t_new:8.1f} us ({fl / t_new / 1e6:6.1f} TF fp32) | materialised {t_old:8.1f} us '
              f'({fl / t_old / 1e6:6.1f} TF), N x N tensor {n_img * N * N * 4 / 1e6:7.1f} MB')


if __name__ == '__main__':
    main()
