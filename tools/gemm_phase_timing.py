"""Where does a split-GEMM launch spend its time, and at which clock?  Phase stamps of
t2h_gemm_split_probe_next_launch (s_memrealtime at 100 MHz + s_memtime = shader clock, per workgroup): entry,
prologue done, main loop done, epilogue stores issued.  Prints the median phase lengths, the dispatch skew, the
span first-entry -> last stamp next to the event-timed launch, and the shader clock of each phase.  Runs on the
product library; ablation builds (T2H_TIMING_DEFS / T2H_TIMING_SO) drop the DMA, the matrix instructions or the
fragment reads.  GPU only.

    python tools/gemm_phase_timing.py [cfgs=6,8] [batch=8]
"""
import ctypes
import os
import statistics
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2human_amd._lib import GemmSplitArgs  # noqa: E402

csrc = os.path.join(ROOT, 'text2human_amd', 'csrc')
DEFS = [d for d in os.environ.get('T2H_TIMING_DEFS', '').split(',') if d]  # ablation builds, e.g. T2H_SDBG_NOGLOAD,T2H_SDBG_NOMMA
PREBUILT = os.environ.get('T2H_TIMING_SO')  # an ablation build made in the build container (tools/build_timing_variants.sh)
if PREBUILT:
    so = os.path.join(ROOT, PREBUILT)
elif not DEFS and not os.environ.get('T2H_DMA_POLICY'):
    so = os.path.join(ROOT, 'text2human_amd', 'libt2h_hip.so')  # the product library carries the phase stamps
else:
    so = '/tmp/libt2h_gemm_timing.so'
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-comment', f'-I{ROOT}/include',
                    *[f'-D{d}' for d in DEFS],
                    *(["-DT2H_DMA_POLICY=\" " + os.environ['T2H_DMA_POLICY'].replace('-', '') + "\""] if os.environ.get('T2H_DMA_POLICY') else []),
                    os.path.join(csrc, 'api.hip'), os.path.join(csrc, 'gemm_split.hip'), '-o', so], check=True)
lib = ctypes.CDLL(so)
# a configuration "11/2" = tile configuration 11 with a 2-way split over K across workgroups (t2h_gemm_split_args.ksplit:
# partial fp32 tiles; proj / fc2 shapes with x8 operands only)
CFGS = [(int(c.split('/')[0]), int(c.split('/')[1]) if '/' in c else 0)
        for c in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['6', '8'])]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
M = 512 * B
g0 = torch.Generator().manual_seed(0)
SHAPES = os.environ.get('T2H_TIMING_SHAPES', 'fc1,qkv_nov,proj,fc2').split(',')
print('defs:', DEFS or 'none', flush=True)
for name, (n, k, gelu, res) in dict(fc1=(2048, 512, True, False), qkv_nov=(1536, 512, False, False),
                                    proj=(512, 512, False, True), fc2=(512, 2048, False, True)).items():
    if name not in SHAPES:
        continue
    z = 0.0 if os.environ.get('T2H_TIMING_ZERO') == '1' else 1.0  # zero operands: the DVFS-free reference point
    a = (torch.randn(M * k * 2, generator=g0) * 0.5 * z).half().view(torch.int16).cuda()
    w = (torch.randn(n * k * 2, generator=g0) * 0.05 * z).half().view(torch.int16).cuda()
    out = torch.zeros(4 * M, n, device='cuda')  # (room for up to 4 partial tiles per output tile)
    osp = torch.empty(M * n * 2, dtype=torch.int16, device='cuda')
    bias = torch.randn(n, generator=g0).cuda()
    g = GemmSplitArgs()
    g.A, g.B, g.bias = a.data_ptr(), w.data_ptr(), bias.data_ptr()
    g.M, g.N, g.K, g.ldc, g.ldr = M, n, k, n, n
    if res:
        g.C, g.residual = out.data_ptr(), out.data_ptr()
    else:
        g.C_split = osp.data_ptr()
    g.epi_act = 1 if gelu else 0
    if os.environ.get('T2H_TIMING_X8') == '1':  # x8 operands (random bytes: timing only) and, for fc1, x8 output
        g.fmt, g.lo_mul = 1, 1.0 / 2048.0 / 64.0
        if not res:
            g.out_fmt, g.out_scale = (1, 4.0) if os.environ.get('T2H_TIMING_X8_OUT', '1') == '1' else (0, 0.0)
    ovf = torch.zeros(1, dtype=torch.int32, device='cuda')
    g.overflow_flag = ovf.data_ptr()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for cfg, ks in CFGS:
        if ks > 1 and not (res and g.fmt == 1):
            continue
        g.ksplit = ks
        lib.t2h_gemm_split_force_config(cfg)
        tb = torch.zeros(16 * 4096, dtype=torch.int64, device='cuda')
        setter = lib.t2h_gemm_split_probe_next_launch
        for _ in range(3):
            assert lib.t2h_gemm_split_f32(ctypes.byref(g), st) == 0, name
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.t2h_gemm_split_f32(ctypes.byref(g), st)
        e1.record()
        torch.cuda.synchronize()
        t_evt = e0.elapsed_time(e1) / 20 * 1e3
        setter(ctypes.c_void_p(tb.data_ptr()))
        lib.t2h_gemm_split_f32(ctypes.byref(g), st)
        torch.cuda.synchronize()
        raw = tb.view(-1, 16).cpu()
        raw = raw[raw[:, 0] != 0].double()
        t = raw[:, :8] * 0.01  # us
        clk = raw[:, 8:]       # shader-clock ticks at the same marks
        ghz = lambda a, b: statistics.median(((clk[:, b] - clk[:, a]) / ((t[:, b] - t[:, a]) * 1e3)).tolist())
        nb = t.shape[0]
        first = t[:, 0].min().item()
        med = lambda v: statistics.median(v.tolist())
        print(f'{name:8s} cfg{cfg:2d}{"/" + str(ks) if ks > 1 else "  "}: launch {t_evt:6.1f} us | {nb:4d} blocks | entry skew med {med(t[:, 0] - first):5.2f} '
              f'max {(t[:, 0] - first).max().item():5.2f} | prologue {med(t[:, 1] - t[:, 0]):5.2f} | main loop '
              f'{med(t[:, 2] - t[:, 1]):5.2f} (max {(t[:, 2] - t[:, 1]).max().item():5.2f}) | epilogue issue '
              f'{med(t[:, 3] - t[:, 2]):5.2f} | first entry -> last epilogue stamp '
              f'{(t[:, 3].max().item() - first):6.2f} | last main-loop end {(t[:, 2].max().item() - first):6.2f} | shader clock GHz: '
              f'prologue {ghz(0, 1):.2f} main loop {ghz(1, 2):.2f} epilogue {ghz(2, 3):.2f}',
              flush=True)
    lib.t2h_gemm_split_force_config(-1)
