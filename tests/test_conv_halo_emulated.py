"""csrc/conv_halo.hip on the CPU: the kernel's SOURCE compiled for the host against tests/emu/hip_emu.h (a workgroup =
512 OS threads, __syncthreads = a barrier, __shared__ = static storage, the matrix instruction restated as a
wave-collective) and called through its product entry point t2h_conv_halo_f32 with host pointers.  Checks what can be
checked without a GPU: halo / tap / border index algebra, buffer rotation over channel groups, nearest-x2 staging,
clamped column tiles, the epilogue's pixel mapping, the GroupNorm partials, the overflow word -- against an fp64
convolution.  (tests/test_gpu_conv_halo.py is the parity test proper, on the hardware.)"""
import ctypes
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
import build_emu  # noqa: E402

from text2human_amd import ops, weights  # noqa: E402
from text2human_amd._lib import GemmArgs  # noqa: E402

pytestmark = pytest.mark.skipif(not build_emu.available(), reason='no host clang++ for the emulation build')


# (kernel variant, landing of the explicit vector-memory requests: at issue / as late as the counted waits allow --
# tests/emu/hip_emu.h, emu_set_deferred: the second is what checks the s_waitcnt vmcnt(N) counts)
@pytest.fixture(scope='module', params=[(1, 0), (1, 1), (0, 1)],
                ids=['lds-dma-kernel', 'lds-dma-kernel-late-landing', 'first-version-late-landing'])
def lib(request):
    so = ctypes.CDLL(build_emu.build('conv_halo.hip'))
    so.t2h_conv_halo_force_variant(request.param[0])
    so.emu_set_deferred(request.param[1])
    so.t2h_conv_halo_f32.restype = ctypes.c_int
    so.t2h_conv_halo_f32.argtypes = [ctypes.POINTER(GemmArgs), ctypes.c_void_p, ctypes.c_void_p]
    so.emu_last_error.restype = ctypes.c_char_p
    return so


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def split_rows_cpu(w):
    """[rows, C] fp32 -> split rows [rows, C/32, 2, 32] fp16 (as int16), ops.split_rows' layout"""
    hi, lo = ops.split_planes_host(w)
    r, C = w.shape
    return torch.stack([hi.view(r, C // 32, 32), lo.view(r, C // 32, 32)], dim=2).contiguous().view(torch.int16)


def run(lib, n_img, cin, cout, h, w, mode, use_pro, x_scale=1.0, residual=True):
    x = rnd(n_img, cin, h, w, seed=13) * x_scale
    wt, b = rnd(cout, cin, 3, 3, seed=14, scale=0.1), rnd(cout, seed=15)
    sc, sh = (rnd(n_img, cin, seed=16) * 0.3 + 1).contiguous(), (rnd(n_img, cin, seed=17) * 0.3).contiguous()
    ws = split_rows_cpu(weights.pack_conv3x3(wt))
    rows = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
    xin = x.double()
    if use_pro:
        xin = xin * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]
        xin = xin * torch.sigmoid(xin)
    ups = 1 if mode == 'up' else 0
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode='nearest')
    ref = F.conv2d(xin, wt.double(), b.double(), 1, 1)
    ho, wo = ref.shape[2:]
    ref = ref.permute(0, 2, 3, 1).reshape(-1, cout)
    M = n_img * ho * wo
    res = rnd(M, cout, seed=18)
    if residual:
        ref = ref + res.double()
    out = torch.full((M, cout), float('nan'))
    part = torch.full((n_img, ho * wo // 128, 2, cout), float('nan'), dtype=torch.float64)
    ovf = torch.zeros(1, dtype=torch.int32)
    g = GemmArgs()
    g.A, g.B, g.C, g.bias = rows.data_ptr(), ws.data_ptr(), out.data_ptr(), b.data_ptr()
    g.residual = res.data_ptr() if residual else None
    g.M, g.N, g.K = M, cout, 9 * cin
    g.lda, g.ldb, g.ldc, g.ldr = cin, 0, cout, cout if residual else 0
    g.a_mode, g.epi_act, g.alpha, g.res_pre = 1, 0, 1.0, 0
    g.Hin, g.Win, g.Cin, g.Hout, g.Wout = h, w, cin, ho, wo
    g.stride, g.pad, g.ups, g.batch = 1, 1, ups, 1
    if use_pro:
        g.pro_scale, g.pro_shift, g.pro_ld, g.pro_act = sc.data_ptr(), sh.data_ptr(), cin, 1
    g.gn_part_out = part.data_ptr()
    rc = lib.t2h_conv_halo_f32(ctypes.byref(g), ovf.data_ptr(), None)
    assert rc == 0, lib.emu_last_error()
    return out, ref, part, int(ovf[0]), (n_img, ho, wo)


@pytest.mark.parametrize('n_img,cin,cout,h,w,mode,use_pro,residual', [
    (1, 32, 128, 16, 16, 'same', False, True),   # one tile, one channel group: borders on all four sides
    (1, 32, 128, 16, 16, 'same', False, False),  # ... without a residual: the GroupNorm sums come from the staged values
    (1, 64, 96, 32, 16, 'same', True, True),     # two groups (both halo buffers), tiles stacked in y, clamped column tile
    (1, 96, 128, 8, 16, 'up', True, True),       # three groups (odd), nearest-x2 staging, tiles side by side in x
])
def test_emulated_conv_halo_vs_fp64(lib, n_img, cin, cout, h, w, mode, use_pro, residual):
    out, ref, part, ovf, (n, ho, wo) = run(lib, n_img, cin, cout, h, w, mode, use_pro, residual=residual)
    err = (out.double() - ref).abs()
    bad = ~(err <= 2e-5 + 2e-5 * ref.abs())
    where = [(r // (ho * wo), (r % (ho * wo)) // wo, r % wo, c) for r, c in bad.nonzero()[:8].tolist()]
    assert not bad.any(), f'max err {err.max().item():.3e}; first (img, y, x, channel): {where}'
    assert ovf == 0
    # the GroupNorm partials cover every pixel exactly once
    su = out.double().view(n, ho * wo, cout).sum(1)
    sq = (out.double() ** 2).view(n, ho * wo, cout).sum(1)
    assert (part[:, :, 0].sum(1) - su).abs().max().item() < 1e-9 * max(1.0, float(su.abs().max()))
    assert (part[:, :, 1].sum(1) - sq).abs().max().item() < 1e-9 * max(1.0, float(sq.abs().max()))


def test_emulated_conv_halo_raises_the_overflow_word(lib):
    _, _, _, ovf, _ = run(lib, 1, 32, 128, 16, 16, 'same', False, x_scale=1e5)
    assert ovf == 1


def test_emulated_entry_point_rejects_what_the_kernel_does_not_serve(lib):
    g = GemmArgs()
    buf = torch.zeros(64)
    ovf = torch.zeros(1, dtype=torch.int32)
    g.A = g.B = g.C = buf.data_ptr()
    g.M, g.N, g.K, g.Cin = 24 * 16, 128, 9 * 32, 32
    g.lda, g.ldc, g.a_mode, g.alpha, g.batch = 32, 128, 1, 1.0, 1
    g.Hin, g.Win, g.Hout, g.Wout, g.stride, g.pad = 24, 16, 24, 16, 1, 1
    assert lib.t2h_conv_halo_f32(ctypes.byref(g), ovf.data_ptr(), None) != 0
    assert b'multiples of 16' in lib.emu_last_error()


def test_kernel_choice_is_a_function_of_the_image_geometry_only(monkeypatch):
    """ops.conv_halo_ok decides between two kernels that sum K in different orders: it must not look at the batch."""
    monkeypatch.delenv('T2H_HALO_CONV', raising=False)
    for shape in ((512, 256, 128, 128), (256, 128, 256, 128), (128, 64, 256, 256), (64, 32, 512, 512),
                  (64, 32, 256, 256), (32, 16, 512, 512), (1024, 512, 128, 128)):
        assert len({ops.conv_halo_ok(n, *shape) for n in (1, 2, 3, 4, 8, 32)}) == 1, shape
    assert ops.conv_halo_ok(1, 512, 256, 128, 128) and ops.conv_halo_ok(1, 64, 32, 512, 512)
    assert not ops.conv_halo_ok(8, 64, 32, 256, 256) and not ops.conv_halo_ok(8, 32, 16, 512, 512)
    assert not ops.conv_halo_ok(8, 24, 16, 32, 128) and not ops.conv_halo_ok(8, 512, 256, 128, 128, 'down')
    assert ops.conv_halo_ok(8, 2048, 1024, 128, 128) and not ops.conv_halo_ok(8, 4096, 2048, 128, 128)   # 32-bit offsets per image
    monkeypatch.setenv('T2H_HALO_CONV', '0')
    assert not ops.conv_halo_ok(8, 512, 256, 128, 128)
    monkeypatch.setenv('T2H_HALO_CONV', '2')
    assert ops.conv_halo_ok(1, 32, 16, 512, 512)


def test_late_landing_catches_a_wait_count_that_is_one_too_permissive():
    """negative control of the late-landing mode: the LDS-DMA kernel with `s_waitcnt vmcnt(N + 1)` in front of its barrier
    (the weight tile of the next tap may still be in flight when the other waves read it) must give a wrong result."""
    def relax(text):
        old = 'ch_wait_vm_all<(t >= 1 && t <= 6) ? 3 : 2>();'
        assert old in text
        return text.replace(old, 'ch_wait_vm_all<(t >= 1 && t <= 6) ? 4 : 3>();')
    so = ctypes.CDLL(build_emu.build('conv_halo.hip', transform=relax, tag='_relaxed_wait'))
    so.t2h_conv_halo_f32.restype = ctypes.c_int
    so.t2h_conv_halo_f32.argtypes = [ctypes.POINTER(GemmArgs), ctypes.c_void_p, ctypes.c_void_p]
    so.emu_last_error.restype = ctypes.c_char_p
    so.emu_set_deferred(0)
    out, ref, _, _, _ = run(so, 1, 64, 128, 16, 16, 'same', True)
    assert ((out.double() - ref).abs() <= 2e-5 + 2e-5 * ref.abs()).all()   # landing at issue: the count is never tested
    so.emu_set_deferred(1)
    out, ref, _, _, _ = run(so, 1, 64, 128, 16, 16, 'same', True)
    assert not ((out.double() - ref).abs() <= 2e-5 + 2e-5 * ref.abs()).all(), 'a too permissive wait went unnoticed'
