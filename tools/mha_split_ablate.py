"""Debug: which part of the three-product attention kernel costs what?  Builds ablated
variants of attention.hip on the GPU box (results are garbage) and times B=8.  GPU only."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, 'text2human_amd', 'csrc')
VARIANTS = (('full', []), ('full+timing', ['-DT2H_MHA_TIMING']), ('noexp', ['-DT2H_MDBG_NOEXP']), ('nosplit', ['-DT2H_MDBG_NOSPLIT']),
            ('noexp+nosplit', ['-DT2H_MDBG_NOEXP', '-DT2H_MDBG_NOSPLIT']), ('nomma', ['-DT2H_MDBG_NOMMA']),
            ('nostage', ['-DT2H_MDBG_NOSTAGE']),
            ('mma only', ['-DT2H_MDBG_NOEXP', '-DT2H_MDBG_NOSPLIT', '-DT2H_MDBG_NOSTAGE']))
if len(sys.argv) > 1 and sys.argv[1] == 'flags':
    VARIANTS = (('full', []), ('no SLP vectorizer', ['-fno-slp-vectorize']), ('no setprio', ['-DT2H_MHA_NOPRIO']),
                ('no setprio+timing', ['-DT2H_MHA_NOPRIO', '-DT2H_MHA_TIMING']), ('full', []), ('no setprio', ['-DT2H_MHA_NOPRIO']))
if len(sys.argv) > 1 and sys.argv[1] == 'pipe2':
    VARIANTS = (('pipelined', []), ('pipelined, no setprio', ['-DT2H_MHA_NOPRIO']),
                ('pipelined, nt loads', ['-DT2H_MHA_DMA_POLICY=" nt"']), ('pipelined', []),
                ('pipelined, no setprio', ['-DT2H_MHA_NOPRIO']), ('pipelined, nt loads', ['-DT2H_MHA_DMA_POLICY=" nt"']))
if len(sys.argv) > 1 and sys.argv[1] == 'pipe':
    VARIANTS = (('register-staged', ['-DT2H_MHA_PIPE_DEFAULT=0']), ('pipelined + LDS-DMA', ['-DT2H_MHA_PIPE_DEFAULT=1']),
                ('pipelined+timing', ['-DT2H_MHA_PIPE_DEFAULT=1', '-DT2H_MHA_TIMING']),
                ('register-staged', ['-DT2H_MHA_PIPE_DEFAULT=0']), ('pipelined + LDS-DMA', ['-DT2H_MHA_PIPE_DEFAULT=1']))
if len(sys.argv) > 1 and sys.argv[1] == 'timing':
    VARIANTS = (('full', []), ('full+timing', ['-DT2H_MHA_TIMING']), ('nomma', ['-DT2H_MDBG_NOMMA']), ('mma only', ['-DT2H_MDBG_NOEXP', '-DT2H_MDBG_NOSPLIT', '-DT2H_MDBG_NOSTAGE']))
B, T, H, C = 8, 512, 8, 512
for tag, extra in VARIANTS:
    so = '/tmp/libt2h_mab_' + ''.join(ch if ch.isalnum() else '_' for ch in tag) + '.so'
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-comment',
                    f'-I{ROOT}/include', *extra, os.path.join(csrc, 'api.hip'), os.path.join(csrc, 'attention.hip'),
                    '-o', so], check=True)
    lib = ctypes.CDLL(so)
    tbuf = torch.zeros(16, dtype=torch.int64, device='cuda')
    if 'timing' in tag:
        assert lib.t2h_debug_set_mha_timing_buffer(ctypes.c_void_p(tbuf.data_ptr())) == 0
    g = torch.Generator().manual_seed(0)
    qk = (torch.randn(B * T * 3 * C * 2, generator=g) * 0.5).half().cuda()  # random planes
    vt = (torch.randn(B * H * 2 * 64 * T, generator=g) * 0.5).half().cuda()
    ys = torch.empty(B * T * C * 2, dtype=torch.int16, device='cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (ctypes.c_void_p(qk.data_ptr()), 3 * C, ctypes.c_void_p(vt.data_ptr()), ctypes.c_void_p(0),
            ctypes.c_void_p(ys.data_ptr()), B, T, H, st)
    for _ in range(3):
        assert lib.t2h_mha_split_f32(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        lib.t2h_mha_split_f32(*args)
    e1.record()
    torch.cuda.synchronize()
    print(f'{tag:14s} {e0.elapsed_time(e1) / 30 * 1e3:6.1f} us', flush=True)
    if 'timing' in tag:
        t = tbuf.cpu().tolist()
        for w, o in ((0, 0), (4, 8)):
            print(f'    wave {w}: total {t[o]} | prologue {t[o + 1]} | staging+barriers {t[o + 2]} | compute '
                  f'{t[o + 3]} | loop {t[o + 5]} | merge+epilogue {t[o + 4]}  (s_memtime ticks)')
