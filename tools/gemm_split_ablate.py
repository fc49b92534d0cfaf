"""Debug: which pipe bounds the split-precision GEMM main loop?  Builds ablated
variants of gemm_split.hip on the GPU box (loads / LDS stores / LDS fragment reads /
matrix instructions removed one at a time; results are garbage) and times the four
sampler shapes at B=8 with every tile config.  GPU only."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2human_amd._lib import GemmSplitArgs  # noqa: E402

csrc = os.path.join(ROOT, 'text2human_amd', 'csrc')
VARIANTS = (('full', []), ('nogload', ['-DT2H_SDBG_NOGLOAD']), ('noput', ['-DT2H_SDBG_NOPUT']),
            ('nofrag', ['-DT2H_SDBG_NOFRAG']), ('nomma', ['-DT2H_SDBG_NOMMA']),
            ('nogload+noput', ['-DT2H_SDBG_NOGLOAD', '-DT2H_SDBG_NOPUT']),
            ('mma only', ['-DT2H_SDBG_NOGLOAD', '-DT2H_SDBG_NOPUT', '-DT2H_SDBG_NOFRAG']))
CFGS = [int(c) for c in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0]
M = 4096
for tag, extra in VARIANTS:
    so = f'/tmp/libt2h_sab_{tag.replace("+", "_").replace(" ", "_")}.so'
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-comment',
                    f'-I{ROOT}/include', *extra, os.path.join(csrc, 'api.hip'), os.path.join(csrc, 'gemm_split.hip'),
                    '-o', so], check=True)
    lib = ctypes.CDLL(so)
    line = f'{tag:14s}'
    for (n, k) in ((1536, 512), (512, 512), (2048, 512), (512, 2048)):
        a = torch.zeros(M * k * 2, dtype=torch.int16, device='cuda')
        w = torch.zeros(n * k * 2, dtype=torch.int16, device='cuda')
        out = torch.empty(M, n, device='cuda')
        g = GemmSplitArgs()
        g.A, g.B, g.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
        g.M, g.N, g.K, g.ldc = M, n, k, n
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for cfg in CFGS:
            lib.t2h_gemm_split_force_config(cfg)
            for _ in range(3):
                assert lib.t2h_gemm_split_f32(ctypes.byref(g), st) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.t2h_gemm_split_f32(ctypes.byref(g), st)
            e1.record()
            torch.cuda.synchronize()
            line += f' | N{n} K{k} cfg{cfg} {e0.elapsed_time(e1) * 50:6.1f} us'
    print(line, flush=True)
