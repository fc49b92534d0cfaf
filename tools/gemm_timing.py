"""Debug: per-phase shader-cycle breakdown of the GEMM main loop (block 8,
thread 0) from a -DT2H_GEMM_TIMING build of gemm.hip.  GPU only.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DT2H_GEMM_TIMING \
        text2human_amd/csrc/{api,gemm}.hip -o /tmp/libt2h_timing.so
"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2human_amd._lib import GemmArgs  # noqa: E402

csrc = os.path.join(ROOT, 'text2human_amd', 'csrc')
names = ['64x64x32', '128x32x32', '128x64x32', '128x128x32', '128x64x64', '128x128x64', '128x64x32w8',
         '128x64x64w8', '128x128x64w8']
for tag, extra in (('full', []), ('noload', ['-DT2H_DBG_NOLOAD']), ('nostore', ['-DT2H_DBG_NOSTORE']),
                   ('noload+nostore', ['-DT2H_DBG_NOLOAD', '-DT2H_DBG_NOSTORE'])):
    so = f'/tmp/libt2h_timing_{tag.replace("+", "_")}.so'
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-comment',
                    '-DT2H_GEMM_TIMING', *extra, os.path.join(csrc, 'api.hip'), os.path.join(csrc, 'gemm.hip'),
                    '-o', so], check=True)
    lib = ctypes.CDLL(so)
    buf = torch.zeros(8, dtype=torch.int64, device='cuda')
    assert lib.t2h_debug_set_timing_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
    for (m, n, k) in [(4096, 512, 2048), (4096, 2048, 512)]:
        a = torch.randn(m, k, device='cuda')
        w = torch.randn(n, k, device='cuda') * 0.05
        out = torch.empty(m, n, device='cuda')
        for cfg in (2, 6):
            lib.t2h_gemm_force_config(cfg)
            g = GemmArgs()
            g.A, g.B, g.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
            g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.alpha, g.batch = m, n, k, k, k, n, 1.0, 1
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(3):
                assert lib.t2h_gemm_f32(ctypes.byref(g), st) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                assert lib.t2h_gemm_f32(ctypes.byref(g), st) == 0
            e1.record()
            torch.cuda.synchronize()
            t = buf.cpu().tolist()
            nk = max(1, t[5])
            us = e0.elapsed_time(e1) * 100
            print(f'{tag:15s} M{m} N{n} K{k} {names[cfg]:12s} wall {us:7.1f} us | barrier {t[3] / nk:5.0f} '
                  f'| loop {t[4] / nk:6.0f} cyc/tile | implied clock {t[4] / (us * 1e-6) / 1e9:5.2f} GHz (upper bound)')
