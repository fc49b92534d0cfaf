"""Where does a launch of the three-product attention spend its time?  Debug build (-DT2H_MHA_TIMING) with s_memtime /
s_memrealtime stamps in workgroup 8, waves 0 and 4 (the two key halves): prologue, K/V staging + barriers, compute,
merge + epilogue, and the shader clock of prologue / loop.  GPU only; builds with hipcc unless T2H_TIMING_SO names a
prebuilt library (tools/build_timing_variants.sh).

    python tools/mha_phase_timing.py [batch=8]
"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, 'text2human_amd', 'csrc')
so = os.environ.get('T2H_TIMING_SO')
if so:
    so = os.path.join(ROOT, so)
else:
    so = '/tmp/libt2h_mha_timing.so'
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-comment', f'-I{ROOT}/include',
                    '-DT2H_MHA_TIMING', os.path.join(csrc, 'api.hip'), os.path.join(csrc, 'attention.hip'), '-o', so], check=True)
lib = ctypes.CDLL(so)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T, H, C = 512, 8, 512
g = torch.Generator().manual_seed(0)
qk = (torch.randn(B * T * 3 * C * 2, generator=g) * 0.5).half().view(torch.int16).cuda()
vt = (torch.randn(B * H * 2 * 64 * T, generator=g) * 0.5).half().view(torch.int16).cuda()
ys = torch.empty(B * T * C * 2, dtype=torch.int16, device='cuda')
ovf = torch.zeros(1, dtype=torch.int32, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
call = lambda: lib.t2h_mha_split_f32(P(qk), 3 * C, P(vt), None, P(ys), B, T, H, P(ovf), st)
lib.t2h_debug_set_mha_timing_buffer(ctypes.c_void_p(0))
for _ in range(3):
    assert call() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    call()
e1.record()
torch.cuda.synchronize()
print(f'B={B}: launch {e0.elapsed_time(e1) / 30 * 1e3:.1f} us (30 back to back)')
tb = torch.zeros(16, dtype=torch.int64, device='cuda')
lib.t2h_debug_set_mha_timing_buffer(P(tb))
call()
torch.cuda.synchronize()
lib.t2h_debug_set_mha_timing_buffer(ctypes.c_void_p(0))
t = tb.cpu().tolist()
for w, o in ((0, t[:8]), (4, t[8:])):
    rt_pro, rt_loop = o[7] & 0xFFFFFFFF, o[7] >> 32
    print(f'  wave {w}: total {o[0]} | prologue {o[1]} | staging+barriers {o[2]} | compute {o[3]} | loop {o[5]} | merge+epilogue '
          f'{o[4]} (shader cycles) | wall: total {o[6] * 0.01:.2f} us, prologue {rt_pro * 0.01:.2f}, loop {rt_loop * 0.01:.2f} | '
          f'clock GHz: total {o[0] / (o[6] * 10):.2f} prologue {o[1] / max(1, rt_pro * 10):.2f} loop {o[5] / max(1, rt_loop * 10):.2f}')
