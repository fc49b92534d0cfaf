"""The exact-fp32 matrix kernels on the CPU (tests/emu): t2h_gemm_f32 in its plain, batched and implicit-convolution
modes (csrc/gemm.hip: every nn.Linear / conv of the tokenizer, the index-prediction UNet and the AttnBlock products --
the part of the path whose results feed argmin / argmax decisions), t2h_conv3x3_small_f32 (conv_out) and the
flash-style AttnBlock attention (csrc/spatial_attn.hip), each through its product entry point against fp64."""
import ctypes
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
import build_emu  # noqa: E402

from text2human_amd import weights  # noqa: E402
from text2human_amd._lib import GemmArgs  # noqa: E402

pytestmark = pytest.mark.skipif(not build_emu.available(), reason='no host clang++ for the emulation build')
c_vp, c_i32, c_f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float


def _load(kernel_file, sigs):
    so = ctypes.CDLL(build_emu.build(kernel_file))
    for name, args in sigs.items():
        getattr(so, name).restype = ctypes.c_int
        getattr(so, name).argtypes = args
    so.emu_last_error.restype = ctypes.c_char_p
    return so


@pytest.fixture(scope='module', params=[0, 1], ids=['requests-land-at-issue', 'requests-land-at-the-wait'])
def gemm(request):
    so = _load('gemm.hip', {'t2h_gemm_f32': [ctypes.POINTER(GemmArgs), c_vp], 't2h_gemm_force_config': [ctypes.c_int]})
    so.emu_set_deferred(request.param)   # (register pieces in flight across two K tiles, counted waits)
    yield so
    so.emu_set_deferred(0)


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def close(got, ref, tol=2e-5):
    err = (got.double() - ref).abs()
    assert (err <= tol + tol * ref.abs()).all(), err.max().item()


@pytest.mark.parametrize('M,N,K,act,b_trans', [(70, 96, 64, 0, False),      # ragged tiles on both sides
                                               (128, 64, 160, 1, False),    # GELU epilogue (the sampler's fc1)
                                               (33, 40, 32, 2, True)])      # ReLU, weights given as [K, N]
def test_emulated_plain_gemm(gemm, M, N, K, act, b_trans):
    a, w, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2), rnd(N, seed=3), rnd(M, N, seed=4)
    wt = w.t().contiguous() if b_trans else w
    out = torch.full((M, N), float('nan'))
    g = GemmArgs()
    g.A, g.B, g.C, g.bias, g.residual = a.data_ptr(), wt.data_ptr(), out.data_ptr(), bias.data_ptr(), res.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = M, N, K, K, (N if b_trans else K), N, N
    g.a_mode, g.b_trans, g.epi_act, g.alpha, g.batch = 0, int(b_trans), act, 0.5, 1
    assert gemm.t2h_gemm_f32(ctypes.byref(g), None) == 0, gemm.emu_last_error()
    y = 0.5 * (a.double() @ w.double().t()) + bias.double()
    y = F.gelu(y) if act == 1 else (F.relu(y) if act == 2 else y)
    close(out, y + res.double())


def test_emulated_batched_gemm_is_the_attnblock_bmm(gemm):
    nb, M, N, K = 3, 64, 64, 32
    q, k = rnd(nb, M, K, seed=5), rnd(nb, N, K, seed=6)
    s = torch.full((nb, M, N), float('nan'))
    g = GemmArgs()
    g.A, g.B, g.C = q.data_ptr(), k.data_ptr(), s.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc = M, N, K, K, K, N
    g.strideA, g.strideB, g.strideC = M * K, N * K, M * N
    g.alpha, g.batch = float(K) ** -0.5, nb
    assert gemm.t2h_gemm_f32(ctypes.byref(g), None) == 0, gemm.emu_last_error()
    close(s, torch.bmm(q.double(), k.double().transpose(1, 2)) * float(K) ** -0.5)


@pytest.mark.parametrize('mode', ['same', 'up', 'down'])
def test_emulated_implicit_conv_with_the_groupnorm_prologue(gemm, mode):
    """vqgan_arch.py: 3x3 'same' (ResnetBlock), nearest x2 + 3x3 (Upsample), zero-pad right / bottom + stride 2
    (Downsample, :573-580) -- GroupNorm apply + swish fused into the operand load, residual in the epilogue"""
    n_img, cin, cout, h, w = 2, 32, 48, 8, 12
    x = rnd(n_img, cin, h, w, seed=7)
    wt, b = rnd(cout, cin, 3, 3, seed=8, scale=0.1), rnd(cout, seed=9)
    sc, sh = (rnd(n_img, cin, seed=10) * 0.3 + 1).contiguous(), (rnd(n_img, cin, seed=11) * 0.3).contiguous()
    xin = x.double() * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]
    xin = xin * torch.sigmoid(xin)
    if mode == 'up':
        ref = F.conv2d(F.interpolate(xin, scale_factor=2.0, mode='nearest'), wt.double(), b.double(), 1, 1)
        stride, pad, ups = 1, 1, 1
    elif mode == 'down':
        ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), wt.double(), b.double(), 2, 0)
        stride, pad, ups = 2, 0, 0
    else:
        ref = F.conv2d(xin, wt.double(), b.double(), 1, 1)
        stride, pad, ups = 1, 1, 0
    ho, wo = ref.shape[2:]
    ref = ref.permute(0, 2, 3, 1).reshape(-1, cout)
    res = rnd(ref.shape[0], cout, seed=12)
    rows = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
    wp = weights.pack_conv3x3(wt)
    out = torch.full((ref.shape[0], cout), float('nan'))
    g = GemmArgs()
    g.A, g.B, g.C, g.bias, g.residual = rows.data_ptr(), wp.data_ptr(), out.data_ptr(), b.data_ptr(), res.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = ref.shape[0], cout, 9 * cin, cin, 9 * cin, cout, cout
    g.a_mode, g.epi_act, g.alpha, g.batch = 1, 0, 1.0, 1
    g.Hin, g.Win, g.Cin, g.Hout, g.Wout, g.stride, g.pad, g.ups = h, w, cin, ho, wo, stride, pad, ups
    g.pro_scale, g.pro_shift, g.pro_ld, g.pro_act = sc.data_ptr(), sh.data_ptr(), cin, 1
    assert gemm.t2h_gemm_f32(ctypes.byref(g), None) == 0, gemm.emu_last_error()
    close(out, ref + res.double())


def test_emulated_conv_split_over_k_for_the_deep_unet_levels(gemm):
    """t2h_gemm_args.splitk_ws / ksplit (round 6; unet_arch.py:470-481: 4 x 2 pixels per image at 512 channels): K
    slices on their own workgroups, partial tiles in the caller's workspace, summed in slice order with bias / ReLU /
    residual behind.  The automatic slice count depends on (K, N, pixels per image) only: one image computed alone
    and inside a batch gives the same bits."""
    gemm.t2h_gemm_ksplit.restype = ctypes.c_int
    gemm.t2h_gemm_ksplit.argtypes = [ctypes.POINTER(GemmArgs)]
    n_img, cin, cout, h, w = 2, 32, 40, 4, 2
    x = rnd(n_img, cin, h, w, seed=31)
    wt, b = rnd(cout, cin, 3, 3, seed=32, scale=0.1), rnd(cout, seed=33)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), 1, 1)).permute(0, 2, 3, 1).reshape(-1, cout)
    res = rnd(ref.shape[0], cout, seed=34)
    rows = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
    wp = weights.pack_conv3x3(wt)

    def run(n, ksplit, ws_floats=None, rows_=rows):
        M = n * h * w
        out = torch.full((M, cout), float('nan'))
        g = GemmArgs()
        g.A, g.B, g.C, g.bias, g.residual = rows_.data_ptr(), wp.data_ptr(), out.data_ptr(), b.data_ptr(), res.data_ptr()
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = M, cout, 9 * cin, cin, 9 * cin, cout, cout
        g.a_mode, g.epi_act, g.alpha, g.batch = 1, 2, 1.0, 1
        g.Hin, g.Win, g.Cin, g.Hout, g.Wout, g.stride, g.pad, g.ups = h, w, cin, h, w, 1, 1, 0
        ws = torch.full((ws_floats if ws_floats is not None else 16 * M * cout,), float('nan'))
        g.splitk_ws, g.splitk_ws_floats, g.ksplit = ws.data_ptr(), ws.numel(), ksplit
        ks = gemm.t2h_gemm_ksplit(ctypes.byref(g))
        rc = gemm.t2h_gemm_f32(ctypes.byref(g), None)
        return rc, ks, out

    rc, ks_auto, out_auto = run(n_img, 0)
    assert rc == 0 and ks_auto == 2, (rc, ks_auto, gemm.emu_last_error())  # K = 288: 9 K tiles, at least four per slice
    close(out_auto, ref + res.double())
    for ks in (1, 3, 4):   # no split; slices of equal (9 K tiles / 3) and unequal (9 / 4 -> 2, 2, 2, 3) length
        rc, got, out = run(n_img, ks)
        assert rc == 0 and got == ks, gemm.emu_last_error()
        close(out, ref + res.double())
    rc, ks1, out1 = run(1, 0)   # image 0 alone: same slices, same bits
    assert rc == 0 and ks1 == ks_auto and torch.equal(out1, out_auto[:h * w])
    rc, _, _ = run(n_img, 0, ws_floats=ks_auto * n_img * h * w * cout - 1)
    assert rc != 0 and b'workspace' in gemm.emu_last_error()


def test_emulated_conv_out_on_the_vector_alu():
    so = _load('conv_small.hip', {'t2h_conv3x3_small_f32': [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_i32,
                                                            c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]})
    n_img, cin, cout, h, w = 2, 32, 3, 12, 20
    x = rnd(n_img, cin, h, w, seed=13)
    wt, b = rnd(cout, cin, 3, 3, seed=14, scale=0.1), rnd(cout, seed=15)
    sc, sh = (rnd(n_img, cin, seed=16) * 0.3 + 1).contiguous(), (rnd(n_img, cin, seed=17) * 0.3).contiguous()
    xin = x.double() * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]
    ref = F.conv2d(xin * torch.sigmoid(xin), wt.double(), b.double(), 1, 1).permute(0, 2, 3, 1).reshape(-1, cout)
    rows = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous()
    wp = weights.pack_conv3x3(wt)
    out = torch.full((n_img * h * w, 4), float('nan'))
    rc = so.t2h_conv3x3_small_f32(rows.data_ptr(), cin, wp.data_ptr(), b.data_ptr(), sc.data_ptr(), sh.data_ptr(), cin, 1,
                                  out.data_ptr(), 4, n_img, h, w, cin, cout, None)
    assert rc == 0, so.emu_last_error()
    close(out[:, :cout], ref)


def test_emulated_flash_style_spatial_attention():
    so = _load('spatial_attn.hip', {'t2h_spatial_attention_f32': [c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp]})
    n_img, N, C = 2, 96, 256
    qkv = rnd(n_img * N, 3 * C, seed=18) * 0.5
    out = torch.full((n_img * N, C), float('nan'))
    scale = float(int(C) ** -0.5)
    assert so.t2h_spatial_attention_f32(qkv.data_ptr(), 3 * C, out.data_ptr(), C, n_img, N, C, scale, None) == 0, so.emu_last_error()
    q, k, v = (t.double().view(n_img, N, C) for t in qkv.split(C, dim=1))
    ref = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * scale, dim=2) @ v
    close(out, ref.view(-1, C))
