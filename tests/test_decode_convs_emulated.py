"""The two-kernel decode path on the CPU -- csrc/norm.hip (GroupNorm finalize, GroupNorm apply + swish + split) and
csrc/conv_split.hip (implicit-GEMM convolution over split rows) compiled for the host against tests/emu/hip_emu.h and
called through their product entry points -- against fp64, and against csrc/conv_halo.hip emulated the same way: the
two paths the decoders choose between (ops.conv_halo_ok) compute the same convolution.  Test infrastructure only; the
parity tests proper run on the hardware (tests/test_gpu_conv_split.py, tests/test_gpu_conv_halo.py)."""
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
c_vp, c_i32, c_i64, c_f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float


def _load(kernel_file, sigs):
    so = ctypes.CDLL(build_emu.build(kernel_file))
    for name, args in sigs.items():
        getattr(so, name).restype = ctypes.c_int
        getattr(so, name).argtypes = args
    so.emu_last_error.restype = ctypes.c_char_p
    return so


@pytest.fixture(scope='module')
def norm():
    return _load('norm.hip', {
        't2h_gn_apply_split_f32': [c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp],
        't2h_groupnorm_finalize_f32': [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp]})


@pytest.fixture(scope='module', params=[0, 1], ids=['requests-land-at-issue', 'requests-land-at-the-wait'])
def conv_split(request):
    so = _load('conv_split.hip', {'t2h_conv_split_f32': [ctypes.POINTER(GemmArgs), c_vp],
                                  't2h_conv_split_force_tile': [ctypes.c_int]})
    so.emu_set_deferred(request.param)   # (its staged pieces are explicit requests with exact vmcnt waits)
    yield so
    so.emu_set_deferred(0)


@pytest.fixture(scope='module')
def conv_halo():
    return _load('conv_halo.hip', {'t2h_conv_halo_f32': [ctypes.POINTER(GemmArgs), c_vp, c_vp]})


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def split_rows_cpu(w):
    hi, lo = ops.split_planes_host(w)
    r, C = w.shape
    return torch.stack([hi.view(r, C // 32, 32), lo.view(r, C // 32, 32)], dim=2).contiguous().view(torch.int16)


def gn_apply_split(norm, rows, sc, sh, hw, act):
    n, C = rows.shape
    out = torch.zeros((n, C // 32, 2, 32), dtype=torch.int16)
    ovf = torch.zeros(1, dtype=torch.int32)
    rc = norm.t2h_gn_apply_split_f32(rows.data_ptr(), C, sc.data_ptr() if sc is not None else None,
                                     sh.data_ptr() if sh is not None else None, sc.shape[1] if sc is not None else 0,
                                     out.data_ptr(), n, hw, C, act, ovf.data_ptr(), None)
    assert rc == 0, norm.emu_last_error()
    return out, int(ovf[0])


def conv_args(a_ptr, ws, out, bias, res, part, n_img, h, w, cin, cout, ups):
    ho, wo = h << ups, w << ups
    g = GemmArgs()
    g.A, g.B, g.C, g.bias = a_ptr, ws.data_ptr(), out.data_ptr(), bias.data_ptr()
    g.residual = res.data_ptr() if res is not None else None
    g.M, g.N, g.K = n_img * ho * wo, cout, 9 * cin
    g.lda, g.ldb, g.ldc, g.ldr = cin, 0, cout, cout if res is not None else 0
    g.a_mode, g.epi_act, g.alpha, g.res_pre = 1, 0, 1.0, 0
    g.Hin, g.Win, g.Cin, g.Hout, g.Wout = h, w, cin, ho, wo
    g.stride, g.pad, g.ups, g.batch = 1, 1, ups, 1
    g.gn_part_out = part.data_ptr()
    return g


def problem(n_img, cin, cout, h, w, mode):
    x = rnd(n_img, cin, h, w, seed=13)
    wt, b = rnd(cout, cin, 3, 3, seed=14, scale=0.1), rnd(cout, seed=15)
    sc, sh = (rnd(n_img, cin, seed=16) * 0.3 + 1).contiguous(), (rnd(n_img, cin, seed=17) * 0.3).contiguous()
    rows = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
    xin = x.double() * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]
    xin = xin * torch.sigmoid(xin)
    act_rows = xin.permute(0, 2, 3, 1).reshape(-1, cin)
    if mode == 'up':
        xin = F.interpolate(xin, scale_factor=2.0, mode='nearest')
    ref = F.conv2d(xin, wt.double(), b.double(), 1, 1)
    ho, wo = ref.shape[2:]
    res = rnd(n_img * ho * wo, cout, seed=18)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, cout) + res.double()
    return dict(rows=rows, sc=sc, sh=sh, ws=split_rows_cpu(weights.pack_conv3x3(wt)), b=b, res=res, ref=ref,
                act_rows=act_rows, ho=ho, wo=wo)


def test_emulated_gn_apply_split_is_the_split_of_the_activated_rows(norm):
    P = problem(2, 64, 128, 16, 16, 'same')
    xs, ovf = gn_apply_split(norm, P['rows'], P['sc'], P['sh'], 256, 1)
    got = ops.unsplit_rows_host(xs, 512, 64).double()
    assert ovf == 0 and ((got - P['act_rows']).abs() <= 2e-6 + 2e-6 * P['act_rows'].abs()).all()
    plain, _ = gn_apply_split(norm, P['rows'], None, None, 0, 0)
    assert torch.equal(plain.view(-1), split_rows_cpu(P['rows']).view(-1))   # no tables: bitwise the host split
    _, ovf = gn_apply_split(norm, P['rows'] * 1e5, None, None, 0, 0)
    assert ovf == 1


def test_emulated_groupnorm_finalize_vs_numpy(norm):
    n_img, hw, C, groups, eps = 2, 512, 128, 32, 1e-6
    x = rnd(n_img, hw, C, seed=5) * 1.7 + 0.4
    gamma, beta = rnd(C, seed=6), rnd(C, seed=7)
    chunks = hw // 128
    xc = x.double().view(n_img, chunks, 128, C)
    part = torch.stack([xc.sum(2), (xc * xc).sum(2)], dim=2).contiguous()   # [n_img][chunk][2][C]
    scale, shift = torch.zeros(n_img, C), torch.zeros(n_img, C)
    rc = norm.t2h_groupnorm_finalize_f32(part.data_ptr(), chunks, gamma.data_ptr(), beta.data_ptr(), scale.data_ptr(),
                                         shift.data_ptr(), n_img, hw, C, groups, eps, None)
    assert rc == 0, norm.emu_last_error()
    xg = x.double().view(n_img, hw, groups, C // groups)
    mean = xg.mean(dim=(1, 3), keepdim=True)
    var = xg.var(dim=(1, 3), unbiased=False, keepdim=True)
    rstd = (1.0 / torch.sqrt(var + eps)).expand(n_img, 1, groups, C // groups).reshape(n_img, C)
    mean = mean.expand(n_img, 1, groups, C // groups).reshape(n_img, C)
    want_scale = rstd * gamma.double()
    want_shift = beta.double() - mean * want_scale
    assert (scale.double() - want_scale).abs().max().item() < 1e-5
    assert (shift.double() - want_shift).abs().max().item() < 1e-5


@pytest.mark.parametrize('tile', [128, 256])
@pytest.mark.parametrize('mode', ['same', 'up'])
def test_emulated_conv_split_vs_fp64_and_vs_the_halo_kernel(norm, conv_split, conv_halo, tile, mode):
    n_img, cin, cout, h, w = 1, 64, 128, (16 if mode == 'same' else 8), (16 if mode == 'same' else 8)
    P = problem(n_img, cin, cout, h, w, mode)
    ups = 1 if mode == 'up' else 0
    M = n_img * P['ho'] * P['wo']
    # two-kernel path
    xs, _ = gn_apply_split(norm, P['rows'], P['sc'], P['sh'], h * w, 1)
    out_a = torch.full((M, cout), float('nan'))
    part_a = torch.full((n_img, M // n_img // 128, 2, cout), float('nan'), dtype=torch.float64)
    conv_split.t2h_conv_split_force_tile(tile)
    try:
        rc = conv_split.t2h_conv_split_f32(ctypes.byref(conv_args(xs.data_ptr(), P['ws'], out_a, P['b'], P['res'], part_a,
                                                                  n_img, h, w, cin, cout, ups)), None)
    finally:
        conv_split.t2h_conv_split_force_tile(0)
    assert rc == 0, conv_split.emu_last_error()
    err = (out_a.double() - P['ref']).abs()
    assert (err <= 2e-5 + 2e-5 * P['ref'].abs()).all(), err.max().item()
    # halo path on the same raw rows and tables
    out_b = torch.full((M, cout), float('nan'))
    part_b = torch.full_like(part_a, float('nan'))
    ovf = torch.zeros(1, dtype=torch.int32)
    g = conv_args(P['rows'].data_ptr(), P['ws'], out_b, P['b'], P['res'], part_b, n_img, h, w, cin, cout, ups)
    g.pro_scale, g.pro_shift, g.pro_ld, g.pro_act = P['sc'].data_ptr(), P['sh'].data_ptr(), cin, 1
    rc = conv_halo.t2h_conv_halo_f32(ctypes.byref(g), ovf.data_ptr(), None)
    assert rc == 0, conv_halo.emu_last_error()
    assert (out_a - out_b).abs().max().item() < 1e-5 * max(1.0, float(P['ref'].abs().max()))
    # both leave GroupNorm partials that add up to the same sums (their 128-pixel chunks are different pixel sets)
    assert (part_a.sum(1) - part_b.sum(1)).abs().max().item() < 1e-3


@pytest.mark.parametrize('C', [256, 512, 1024])
def test_emulated_layernorm_with_its_dpp_reduction(C):
    """csrc/norm.hip's LayerNorm (one row per wave, the two reductions by DPP + v_readlane: tests/emu/hip_emu.h restates
    the four DPP controls as lane exchanges) against torch, fp32 out and split rows out."""
    so = _load('norm.hip', {'t2h_layernorm_f32': [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_vp],
                            't2h_layernorm_split_f32': [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_vp, c_vp]})
    rows = 11   # (a partly filled second workgroup)
    x = rnd(rows, C, seed=21) * 2.0 + 0.3
    g, b = rnd(C, seed=22) * 0.2 + 1.0, rnd(C, seed=23) * 0.1
    want = F.layer_norm(x.double(), (C,), g.double(), b.double(), 1e-5)
    y = torch.full((rows, C), float('nan'))
    assert so.t2h_layernorm_f32(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), rows, C, 1e-5, None) == 0, so.emu_last_error()
    assert (y.double() - want).abs().max().item() < 2e-5
    ys = torch.zeros((rows, C // 32, 2, 32), dtype=torch.int16)
    ovf = torch.zeros(1, dtype=torch.int32)
    assert so.t2h_layernorm_split_f32(x.data_ptr(), g.data_ptr(), b.data_ptr(), ys.data_ptr(), rows, C, 1e-5, ovf.data_ptr(),
                                      None) == 0, so.emu_last_error()
    got = ops.unsplit_rows_host(ys, rows, C).double()
    assert int(ovf[0]) == 0 and (got - want).abs().max().item() < 2e-5
    assert (got - y.double()).abs().max().item() < 2e-6   # the split rows carry the fp32 result to 22 bits
