"""Where does a workgroup of the halo-staged convolution spend its time?  s_memtime (shader clock) stamps of a debug
build (tools/_tb/halo_probe.so = api.hip + conv_halo.hip with -DT2H_HALO_PROBE, made in the build container): five
points per tap for one wave of each SIMD-sharing pair (waves 0 and 4) of the first 64 workgroups, plus entry /
prologue done / main loop done / end.  Prints median cycles per phase and tap.  GPU only.

    python tools/conv_halo_phase_timing.py [variant=1]
"""
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2human_amd._lib import GemmArgs  # noqa: E402
from text2human_amd import ops  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, 'tools', '_tb', os.environ.get('HALO_PROBE_SO', 'halo_probe.so')))
print('library:', os.environ.get('HALO_PROBE_SO', 'halo_probe.so'))
lib.t2h_conv_halo_f32.argtypes = [ctypes.POINTER(GemmArgs), ctypes.c_void_p, ctypes.c_void_p]
lib.t2h_conv_halo_probe_next_launches.argtypes = [ctypes.c_void_p]
lib.t2h_conv_halo_force_variant(int(sys.argv[1]) if len(sys.argv) > 1 else 1)

n_img, h, w, cin, cout = 8, 512, 256, 128, 128
x = torch.randn(n_img * h * w, cin, device='cuda')
ws = ops.split_rows(torch.randn(cout, 9 * cin, device='cuda') * 0.05)
bias = torch.randn(cout, device='cuda')
sc, sh = torch.rand(n_img, cin, device='cuda') + 0.5, torch.randn(n_img, cin, device='cuda') * 0.3
out = torch.empty(n_img * h * w, cout, device='cuda')
part = torch.empty(n_img, h * w // 128, 2, cout, device='cuda', dtype=torch.float64)
ovf = torch.zeros(1, dtype=torch.int32, device='cuda')
probe = torch.zeros(64 * 2 * 40 * 8, dtype=torch.int64, device='cuda')
g = GemmArgs()
g.A, g.B, g.C, g.bias = x.data_ptr(), ws.data_ptr(), out.data_ptr(), bias.data_ptr()
g.M, g.N, g.K = n_img * h * w, cout, 9 * cin
g.lda, g.ldc = cin, cout
g.a_mode, g.alpha, g.batch = 1, 1.0, 1
g.Hin, g.Win, g.Cin, g.Hout, g.Wout = h, w, cin, h, w
g.stride, g.pad = 1, 1
g.pro_scale, g.pro_shift, g.pro_ld, g.pro_act = sc.data_ptr(), sh.data_ptr(), cin, 1
g.gn_part_out = part.data_ptr()
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    if it == 2:
        lib.t2h_conv_halo_probe_next_launches(probe.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.t2h_conv_halo_f32(ctypes.byref(g), ovf.data_ptr(), st) == 0
    e1.record()
    torch.cuda.synchronize()
    print(f'launch {it}: {e0.elapsed_time(e1) * 1e3:.1f} us')
P = probe.cpu().view(64, 2, 40, 8)
med = lambda v: statistics.median(v)  # noqa: E731
for wsel, name in ((0, 'wave 0 (converts in even taps)'), (1, 'wave 4 (converts in odd taps)')):
    print(f'--- {name}: median cycles over 64 workgroups x 4 channel groups, per tap of a group')
    print(f'{"tap":>3s} {"top+conv":>9s} {"R0 rd+mfma(R1)":>15s} {"R1 rd+mfma(R0)":>15s} {"B write":>8s} {"barrier":>8s} {"tap total":>9s}')
    tot = 0
    for t in range(9):
        cols = [[] for _ in range(6)]
        for b in range(64):
            for gi in range(4):
                k = gi * 9 + t
                s = P[b, wsel, k]
                nxt = P[b, wsel, k + 1][0] if k + 1 < 36 else P[b, wsel, 36][2]
                d = [int(s[1] - s[0]), int(s[2] - s[1]), int(s[3] - s[2]), int(s[4] - s[3]), int(nxt - s[4]), int(nxt - s[0])]
                for c, v in zip(cols, d):
                    c.append(v)
        m = [med(c) for c in cols]
        tot += m[5]
        print(f'{t:3d} {m[0]:9.0f} {m[1]:15.0f} {m[2]:15.0f} {m[3]:8.0f} {m[4]:8.0f} {m[5]:9.0f}')
    print(f'sum over the nine taps: {tot:.0f} cycles; ideal (48 matrix instructions x 32 cycles per SIMD and tap) 13824')
pe = P[:, 0, 36, :4]
print(f'prologue {med([int(v[1] - v[0]) for v in pe]):.0f}, main loop {med([int(v[2] - v[1]) for v in pe]):.0f}, '
      f'epilogue {med([int(v[3] - v[2]) for v in pe]):.0f} cycles (median over 64 workgroups)')
