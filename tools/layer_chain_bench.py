"""Does it matter where the weights come from?  The per-layer kernel chain of the sampler (LayerNorm,
q|k|v, attention, proj, LayerNorm, fc1, fc2) over 24 layers with ONE weight set reused by every layer
(resident in the Infinity Cache) vs 24 distinct sets (302 MB: streamed from HBM every step, as in the
model), and the chain of only the four Linears.  GPU only.

    python tools/layer_chain_bench.py [batch=8]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import ops  # noqa: E402

DEV = 'cuda'
C, H, T, L = 512, 8, 512, 24
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M = B * T
g = torch.Generator().manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)


def weights():
    return dict(qkv=ops.split_rows(rnd(3 * C, C, sc=0.05)), proj=ops.split_rows(rnd(C, C, sc=0.02)),
                fc1=ops.split_rows(rnd(4 * C, C, sc=0.05)), fc2=ops.split_rows(rnd(C, 4 * C, sc=0.01)),
                bq=rnd(3 * C), bp=rnd(C), b1=rnd(4 * C), b2=rnd(C), g=torch.ones(C, device=DEV), b=torch.zeros(C, device=DEV))


sets = [weights() for _ in range(L)]
x0 = rnd(M, C)
x = x0.clone()
hs, ys, us = ops.split_rows_empty(M, C, DEV), ops.split_rows_empty(M, C, DEV), ops.split_rows_empty(M, 4 * C, DEV)
qks, vt = ops.split_rows_empty(M, 3 * C, DEV), ops.vt_empty(B, H, T, DEV)


def layer(w, linears_only=False):
    if not linears_only:
        ops.layernorm_split(x, w['g'], w['b'], hs)
    ops.gemm_split(hs, w['qkv'], M, 3 * C, C, out_split=qks, bias=w['bq'], vt=vt, vt_col0=2 * C, vt_T=T, vt_hd=64)
    if not linears_only:
        ops.mha_split(qks, 3 * C, vt, B, T, H, out_split=ys)
    ops.gemm_split(ys, w['proj'], M, C, C, out=x, bias=w['bp'], residual=x)
    if not linears_only:
        ops.layernorm_split(x, w['g'], w['b'], hs)
    ops.gemm_split(hs, w['fc1'], M, 4 * C, C, out_split=us, bias=w['b1'], act=ops.ACT_GELU)
    ops.gemm_split(us, w['fc2'], M, C, 4 * C, out=x, bias=w['b2'], residual=x)


def run(distinct, linears_only, steps=6):
    ops.layernorm_split(x0, sets[0]['g'], sets[0]['b'], hs)
    ops.layernorm_split(x0, sets[0]['g'], sets[0]['b'], ys)
    for rep in range(steps + 1):
        if rep == 1:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        x.copy_(x0)
        for i in range(L):
            layer(sets[i if distinct else 0], linears_only)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps / L * 1e3


for lin in (False, True):
    for rnd_ in range(2):
        a, b = run(False, lin), run(True, lin)
        print(f'{"four Linears" if lin else "whole layer "}: one weight set {a:7.1f} us/layer | 24 distinct sets {b:7.1f} us/layer', flush=True)
