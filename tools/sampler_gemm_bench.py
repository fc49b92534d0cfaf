"""The four Linears of one sampler layer exactly as engine.SamplerNet issues them (q|k|v with
split-row + Vt outputs, proj with the residual, fc1 with the GELU split-row epilogue, fc2 with the
residual) timed per tile configuration of t2h_gemm_split_f32: interleaved rounds in one process,
median per call, plus the back-to-back chain of all four.  GPU only.

    python tools/sampler_gemm_bench.py [batch=8] [cfgs=-1,0,6,8] [rounds=7]
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import _lib, ops  # noqa: E402

DEV = 'cuda'
C, H, T = 512, 8, 512


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cfgs = [int(c) for c in sys.argv[2].split(',')] if len(sys.argv) > 2 else [-1, 0, 6, 8]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    M = B * T
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    lib = _lib.load()
    h_s, y_s = ops.split_rows(rnd(M, C)), ops.split_rows(rnd(M, C))
    u_s = ops.split_rows(rnd(M, 4 * C))
    w = {k: ops.pack_split_rows_host(torch.randn(n, kk, generator=g) * 0.05).to(DEV)
         for k, (n, kk) in dict(qkv=(3 * C, C), proj=(C, C), fc1=(4 * C, C), fc2=(C, 4 * C)).items()}
    bias = {k: rnd(n) for k, n in dict(qkv=3 * C, proj=C, fc1=4 * C, fc2=C).items()}
    x = rnd(M, C)
    qk_s, uo_s = ops.split_rows_empty(M, 3 * C, DEV), ops.split_rows_empty(M, 4 * C, DEV)
    vt = ops.vt_empty(B, H, T, DEV)
    calls = {
        'qkv': lambda: ops.gemm_split(h_s, w['qkv'], M, 3 * C, C, out_split=qk_s, bias=bias['qkv'], vt=vt,
                                      vt_col0=2 * C, vt_T=T, vt_hd=C // H),
        'proj': lambda: ops.gemm_split(y_s, w['proj'], M, C, C, out=x, bias=bias['proj'], residual=x),
        'fc1': lambda: ops.gemm_split(h_s, w['fc1'], M, 4 * C, C, out_split=uo_s, bias=bias['fc1'],
                                      act=ops.ACT_GELU),
        'fc2': lambda: ops.gemm_split(u_s, w['fc2'], M, C, 4 * C, out=x, bias=bias['fc2'], residual=x),
    }
    flops = dict(qkv=2.0 * M * 3 * C * C, proj=2.0 * M * C * C, fc1=2.0 * M * 4 * C * C, fc2=2.0 * M * 4 * C * C)

    def usable(cfg, name):
        k = 4 * C if name == 'fc2' else C
        if cfg == 6:
            return k % 64 == 0
        if cfg in (5, 9):   # 256-column tiles do not divide the value-head boundary; 9 = few-rows kernel (no Vt)
            return name != 'qkv'
        return True

    def time_fn(fn, iters=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    res = {(c, n): [] for c in cfgs for n in list(calls) + ['chain']}
    for r in range(rounds + 1):
        for cfg in cfgs:
            lib.t2h_gemm_split_force_config(cfg)
            for name, fn in calls.items():
                if not usable(cfg, name):
                    continue
                t = time_fn(fn)
                if r:
                    res[(cfg, name)].append(t)
            if all(usable(cfg, n) for n in calls):
                t = time_fn(lambda: [f() for f in calls.values()], iters=10)
                if r:
                    res[(cfg, 'chain')].append(t)
    lib.t2h_gemm_split_force_config(-1)
    print(f'B={B} M={M}: median us over {rounds} interleaved rounds (executed TFLOP/s = 3 x fp32-equivalent)')
    for cfg in cfgs:
        line = f'cfg {cfg:3d} |'
        for name in list(calls) + ['chain']:
            v = res[(cfg, name)]
            if not v:
                line += f' {name:5s}    --          |'
                continue
            med = statistics.median(v)
            tf = '' if name == 'chain' else f'({3 * flops[name] / med / 1e6:5.0f} TF)'
            line += f' {name:5s} {med:6.1f} {tf:10s} |'
        print(line)


if __name__ == '__main__':
    main()
