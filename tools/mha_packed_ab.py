"""A/B of the attention kernel's softmax arithmetic (round 6): the product source built twice in the
build container (tools/build_mha_ab.sh): tools/_tb/mha_base.so (-DT2H_MHA_PACKED=0) and tools/_tb/mha_packed.so (=1: the
exponent arguments and the row sum on register pairs, v_pk_fma_f32 / v_pk_add_f32).  Same inputs, interleaved timing
(alternating blocks of 40 launches, 12 blocks; a third build, mha_early.so = -DT2H_MHA_EARLY_S=1, joins when present), x8 output like the product path; the outputs are compared.  GPU only.

    python tools/mha_packed_ab.py [batch=8]
"""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T, H, C = 512, 8, 512
g = torch.Generator().manual_seed(0)
qk = (torch.randn(B * T * 3 * C * 2, generator=g) * 0.5).half().view(torch.int16).cuda()
vt = (torch.randn(B * H * 2 * 64 * T, generator=g) * 0.5).half().view(torch.int16).cuda()
ovf = torch.zeros(1, dtype=torch.int32, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
libs, outs = {}, {}
VARIANTS = tuple(v for v in ('base', 'packed', 'early') if os.path.exists(os.path.join(ROOT, 'tools', '_tb', f'mha_{v}.so')))
for name in VARIANTS:
    libs[name] = ctypes.CDLL(os.path.join(ROOT, 'tools', '_tb', f'mha_{name}.so'))
    libs[name].t2h_mha_split_x8_f32.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float,
                                                ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    outs[name] = torch.zeros(B * T * C * 2, dtype=torch.int16, device='cuda')
call = lambda n: libs[n].t2h_mha_split_x8_f32(P(qk), 3 * C, P(vt), P(outs[n]), 8.0, B, T, H, P(ovf), st)
for n in libs:
    for _ in range(5):
        assert call(n) == 0
torch.cuda.synchronize()
times = {n: [] for n in libs}
for blk in range(12):
    for n in (VARIANTS if blk % 2 == 0 else VARIANTS[::-1]):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            call(n)
        e1.record()
        torch.cuda.synchronize()
        times[n].append(e0.elapsed_time(e1) / 40 * 1e3)
for n in libs:
    print(f'B={B} {n:7s}: median {statistics.median(times[n]):6.2f} us  [min {min(times[n]):.2f}, max {max(times[n]):.2f}] over 12 blocks of 40 launches')
hi = lambda t: t.view(-1, 128)[:, :64].contiguous().view(torch.float16).float()
a = outs['base'].view(torch.uint8).cpu()
for v in VARIANTS[1:]:
    b = outs[v].view(torch.uint8).cpu()
    d = (hi(a) - hi(b)).abs()
    print(f'{v} vs base: {int((a != b).sum())} of {a.numel()} bytes differ; fp16 hi plane max |diff| {d.max().item():.3e} on values up to {hi(a).abs().max().item():.3f}')
