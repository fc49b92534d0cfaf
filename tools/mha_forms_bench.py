"""The two forms of the split-precision attention kernel (csrc/attention.hip, t2h_mha_split_force_form) at the
batch sizes the sampler runs: 2 = 128-query workgroups whose wave groups take the two key halves and merge,
1 = 256-query workgroups whose eight waves each walk all keys.  Time per launch (HIP events over back-to-back
launches), max abs difference between the forms, error of each against fp64 at B = 2.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import _lib, ops  # noqa: E402


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
    T, H, C = 512, 8, 512
    lib = _lib.load()
    for B in (8, 12, 16, 20, 24, 28, 32):
        g = torch.Generator().manual_seed(B)
        qkv = (torch.randn(B * T, 3 * C, generator=g) * 1.2).cuda()
        qk_s = ops.split_rows(qkv)
        # Vt planes through the product path: the q|k|v GEMM's value routing with an identity weight
        a_s = ops.split_rows(qkv[:, 2 * C:].contiguous())
        w_s = ops.pack_split_rows_host(torch.cat([torch.zeros(2 * C, C), torch.eye(C)])).cuda()
        vt = ops.vt_empty(B, H, T, 'cuda')
        scratch = ops.split_rows_empty(B * T, 3 * C, 'cuda')
        ops.gemm_split(a_s, w_s, B * T, 3 * C, C, out_split=scratch, vt=vt, vt_col0=2 * C, vt_T=T)
        ys = ops.split_rows_empty(B * T, C, 'cuda')
        out = {}
        line = f'B={B:2d}:'
        times = {2: [], 1: [], 0: []}
        for rep in range(4):  # interleaved: the clock / power state drifts over a run of back-to-back launches
            for form in (2, 1, 0):
                lib.t2h_mha_split_force_form(form)
                if rep == 0:
                    y = torch.empty(B * T, C, device='cuda')
                    ops.mha_split(qk_s, 3 * C, vt, B, T, H, out=y)
                    out[form] = y
                times[form].append(timeit(lambda: ops.mha_split(qk_s, 3 * C, vt, B, T, H, out_split=ys), iters=40))
        fl = 3 * 4.0 * T * T * 64 * H * B
        for form in (2, 1, 0):
            t = sorted(times[form])[1]
            line += f'  form {form}: {t:6.1f} us [{min(times[form]):.1f}-{max(times[form]):.1f}] ({fl / t / 1e6:5.0f} TF/s executed)'
        lib.t2h_mha_split_force_form(0)
        line += f'  | forms differ by {(out[1] - out[2]).abs().max().item():.1e}; auto == form {1 if torch.equal(out[0], out[1]) else 2}'
        if B == 8:
            q, k, v = [t.view(B, T, H, 64).transpose(1, 2).double() for t in qkv.split(C, dim=1)]
            ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1) @ v).transpose(1, 2).reshape(B * T, C)
            line += f'  | err vs fp64: form 2 {(out[2].double() - ref).abs().max().item():.1e}, form 1 {(out[1].double() - ref).abs().max().item():.1e}'
        print(line, flush=True)


if __name__ == '__main__':
    main()
