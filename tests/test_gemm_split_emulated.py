"""The dominant kernel's SOURCE on the CPU: csrc/gemm_split.hip through tests/emu -- every tile configuration of the
split-precision GEMM (the ping-pong LDS-DMA loops, the register-staged tiles with and without the in-block K split,
the few-rows kernel), with two-fp16-plane operands and with x8 operands (cross terms on the scaled 8-bit matrix
instruction, restated here with an e4m3 decoder), the GELU / residual / split-row epilogues -- through
t2h_gemm_split_f32 against fp64 arithmetic on the planes the kernel was given."""
import ctypes
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
import build_emu  # noqa: E402

from text2human_amd import ops  # noqa: E402
from text2human_amd._lib import GemmSplitArgs  # noqa: E402

pytestmark = pytest.mark.skipif(not build_emu.available(), reason='no host clang++ for the emulation build')


# (landing of the kernels' explicit vector-memory requests -- register loads, LDS-DMA: at issue, or only when a counted
# wait forces it; the second mode is what checks the s_waitcnt vmcnt(N) counts, tests/emu/hip_emu.h)
@pytest.fixture(scope='module', params=[0, 1], ids=['requests-land-at-issue', 'requests-land-at-the-wait'])
def lib(request):
    so = build_emu.load('gemm_split.hip')
    so.emu_set_deferred(request.param)
    yield so
    so.emu_set_deferred(0)


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def split_rows_cpu(w):
    hi, lo = ops.split_planes_host(w)
    r, C = w.shape
    return torch.stack([hi.view(r, C // 32, 32), lo.view(r, C // 32, 32)], dim=2).contiguous().view(torch.int16)


def x8_rows_cpu(w, scale):
    """[rows, C] fp32 -> x8 rows [rows][C/32][hi16 (64 B) | hi8 (32 B) | lo8 (32 B)] (csrc/common.h)"""
    r, C = w.shape
    hi = w.half()
    lo = (w - hi.float()) * ops.SPLIT_LO_SCALE
    h8 = (hi.float() * scale).to(torch.float8_e4m3fn).view(torch.uint8).view(r, C // 32, 32)
    l8 = (lo * scale).to(torch.float8_e4m3fn).view(torch.uint8).view(r, C // 32, 32)
    out = torch.empty((r, C // 32, 128), dtype=torch.uint8)
    out[:, :, :64] = hi.view(r, C // 32, 32).view(torch.uint8).view(r, C // 32, 64)
    out[:, :, 64:96] = h8
    out[:, :, 96:] = l8
    return out.view(torch.int16).view(r, C // 32, 2, 32)


def run(lib, cfg, M, N, K, act, x8, split_out=False):
    a, w = rnd(M, K, seed=1) * 1.3, rnd(N, K, seed=2, scale=0.25)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    g = GemmSplitArgs()
    if x8:
        sa, sb = ops.x8_scale_for(a.abs().max()), ops.x8_scale_for(w.abs().max(), 256.0)
        A, B = x8_rows_cpu(a, sa), x8_rows_cpu(w, sb)
        g.fmt, g.lo_mul = 1, 1.0 / (ops.SPLIT_LO_SCALE * sa * sb)
        ah, a8, al8 = ops.unpack_x8_rows_host(A, M, K, sa)
        bh, b8, bl8 = ops.unpack_x8_rows_host(B, N, K, sb)
        exact = ah.double() @ bh.double().t() + (a8.double() @ bl8.double().t() + al8.double() @ b8.double().t()) / ops.SPLIT_LO_SCALE
    else:
        A, B = split_rows_cpu(a), split_rows_cpu(w)
        ah, al = ops.split_planes_host(a)
        bh, bl = ops.split_planes_host(w)
        exact = ah.double() @ bh.double().t() + (ah.double() @ bl.double().t() + al.double() @ bh.double().t()) / ops.SPLIT_LO_SCALE
    out = torch.full((M, N), float('nan'))
    osp = torch.zeros((M, N // 32, 2, 32), dtype=torch.int16) if split_out else None
    ovf = torch.zeros(1, dtype=torch.int32)
    g.A, g.B, g.C, g.bias, g.residual = A.data_ptr(), B.data_ptr(), out.data_ptr(), bias.data_ptr(), res.data_ptr()
    g.M, g.N, g.K, g.ldc, g.ldr, g.epi_act = M, N, K, N, N, act
    if split_out:
        g.C_split, g.overflow_flag, g.residual, g.ldr = osp.data_ptr(), ovf.data_ptr(), None, 0
    old = lib.t2h_gemm_split_force_config(cfg)
    try:
        assert lib.t2h_gemm_split_tile_config(ctypes.byref(g)) == (cfg if cfg >= 0 else 9)
        rc = lib.t2h_gemm_split_f32(ctypes.byref(g), None)
    finally:
        lib.t2h_gemm_split_force_config(old)
    assert rc == 0, lib.emu_last_error()
    y = exact + bias.double()
    y = F.gelu(y) if act == 1 else (F.relu(y) if act == 2 else y)
    if not split_out:
        y = y + res.double()
    err = (out.double() - y).abs()
    assert (err <= 3e-5 + 3e-5 * y.abs()).all(), (cfg, x8, err.max().item())
    # and the planes carry the fp32 product to ~22 (x8: ~15) bits
    true = a.double() @ w.double().t()
    assert (exact - true).abs().max().item() < (2e-3 if x8 else 2e-5) * float(true.abs().max())
    if split_out:
        got = ops.unsplit_rows_host(osp, M, N).double()
        assert int(ovf[0]) == 0 and (got - out.double()).abs().max().item() < 2e-6 * max(1.0, float(out.abs().max()))


# (cfg, M, N, K, epilogue): one or a few workgroups per configuration, ragged edges where the tile allows them
CASES = [(8, 256, 128, 128, 1),    # 256 x 128 ping-pong LDS-DMA loop (fc1: GELU)
         (10, 128, 192, 96, 0),    # 128 x 192 on the same loop (q | k | v), odd number of K tiles
         (6, 128, 64, 256, 0),     # 128 x 64 with the in-block K split (proj / fc2)
         (1, 130, 136, 64, 2),     # 128 x 128 register-staged, ragged M and N
         (2, 70, 72, 64, 0),       # 64 x 64
         (3, 128, 64, 32, 0),      # 128 x 64, a single K tile
         (5, 128, 256, 64, 0),     # 128 x 256
         (-1, 18, 48, 160, 1)]     # M <= 64: the few-rows kernel (16x16x32 straight from global memory, K over 8 waves)


# (x8 operands are dispatched to configurations 0, 2, 6, 8, 10 and the few-rows kernel)
@pytest.mark.parametrize('cfg,M,N,K,act,x8', [c + (False,) for c in CASES] + [c + (True,) for c in CASES if c[0] not in (1, 3, 5)])
def test_emulated_split_gemm(lib, cfg, M, N, K, act, x8):
    run(lib, cfg, M, N, K, act, x8)


def test_emulated_split_row_output(lib):
    run(lib, 8, 256, 128, 64, 0, False, split_out=True)


def test_emulated_split_row_producers_write_the_host_packers_bytes(lib):
    """t2h_split_rows_f32 / t2h_split_rows_x8_f32 (the weight packers' device twins; the x8 one swaps its 8-bit halves
    between lane pairs by DPP to store 16 bytes per lane) produce the bytes of the host-side packers above, and
    LayerNorm's x8 output (csrc/norm.hip) the x8 rows of its own fp32 result."""
    rows, C = 9, 256
    x = rnd(rows, C, seed=5) * 3.0
    out = torch.zeros((rows, C // 32, 2, 32), dtype=torch.int16)
    ovf = torch.zeros(1, dtype=torch.int32)
    assert lib.t2h_split_rows_f32(x.data_ptr(), C, out.data_ptr(), rows, C, ovf.data_ptr(), None) == 0, lib.emu_last_error()
    assert torch.equal(out, split_rows_cpu(x)) and int(ovf[0]) == 0
    scale = ops.x8_scale_for(x.abs().max())
    out8 = torch.zeros_like(out)
    assert lib.t2h_split_rows_x8_f32(x.data_ptr(), C, out8.data_ptr(), rows, C, scale, ovf.data_ptr(), None) == 0, lib.emu_last_error()
    assert torch.equal(out8.view(torch.uint8), x8_rows_cpu(x, scale).view(torch.uint8)) and int(ovf[0]) == 0
    assert lib.t2h_split_rows_x8_f32(x.data_ptr(), C, out8.data_ptr(), rows, C, scale * 64, ovf.data_ptr(), None) == 0
    assert int(ovf[0]) == 2   # bit 1: beyond the 8-bit planes' range (bit 0 would be fp16's)
    norm = build_emu.load('norm.hip')
    C = 512
    x = rnd(rows, C, seed=6) * 2.0 + 0.5
    g, b = rnd(C, seed=7) * 0.2 + 1.0, rnd(C, seed=8) * 0.1
    y = torch.zeros(rows, C)
    assert norm.t2h_layernorm_f32(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), rows, C, 1e-5, None) == 0
    y8 = torch.zeros((rows, C // 32, 2, 32), dtype=torch.int16)
    ovf.zero_()
    scale = ops.x8_scale_for(y.abs().max())
    rc = norm.t2h_layernorm_x8_f32(x.data_ptr(), g.data_ptr(), b.data_ptr(), y8.data_ptr(), rows, C, 1e-5, scale, ovf.data_ptr(), None)
    assert rc == 0, norm.emu_last_error()
    assert torch.equal(y8.view(torch.uint8), x8_rows_cpu(y, scale).view(torch.uint8)) and int(ovf[0]) == 0


def test_emulated_x8_producers_flag_the_fp16_range_under_a_small_scale():
    """with a tensor scale <= 2^-8 a value in [65504, 448 / s) overflows the fp16 hi plane while the 8-bit
    planes still hold it -- the 8-column store must raise bit 0 (fp16 range) on its own, not only behind the 8-bit
    range test.  LayerNorm with a huge gamma produces such values; `x8_scale_for` itself never returns a scale that
    small any more (448 / s stays below 65504)."""
    norm = build_emu.load('norm.hip')
    rows, C = 5, 512
    x = rnd(rows, C, seed=11)
    g, b = torch.full((C, ), 4.0e4), torch.zeros(C)   # outputs up to ~1.3e5
    y8 = torch.zeros((rows, C // 32, 2, 32), dtype=torch.int16)
    ovf = torch.zeros(1, dtype=torch.int32)
    rc = norm.t2h_layernorm_x8_f32(x.data_ptr(), g.data_ptr(), b.data_ptr(), y8.data_ptr(), rows, C, 1e-5, 2.0 ** -9,
                                   ovf.data_ptr(), None)
    assert rc == 0, norm.emu_last_error()
    assert int(ovf[0]) & 1, int(ovf[0])
    # and values inside fp16's range but beyond the 8-bit planes' raise bit 1 only
    ovf.zero_()
    g = torch.full((C, ), 100.0)
    rc = norm.t2h_layernorm_x8_f32(x.data_ptr(), g.data_ptr(), b.data_ptr(), y8.data_ptr(), rows, C, 1e-5, 4.0, ovf.data_ptr(), None)
    assert rc == 0 and int(ovf[0]) == 2, int(ovf[0])
    assert ops.x8_scale_for(1.0e5) == 2.0 ** -7 and 448.0 / ops.x8_scale_for(1.0e9) < 65504.0


@pytest.mark.parametrize('cfg,M,N,K,ks', [(11, 128, 128, 128, 2), (8, 256, 128, 256, 4), (11, 128, 256, 64, 1)])
def test_emulated_split_k_partial_tiles(lib, cfg, M, N, K, ks):
    """ksplit (round-6 experiment, include/t2h_hip.h): slice s of K goes to its own workgroups and lands as a partial
    fp32 tile at C + s * M * ldc; bias + residual travel with slice 0; the slices sum to the one-pass result.  Also
    the 128x128 ping-pong instantiation (configuration 11) without a split."""
    a, w = rnd(M, K, seed=21) * 1.3, rnd(N, K, seed=22, scale=0.25)
    bias, res = rnd(N, seed=23), rnd(M, N, seed=24)
    sa, sb = ops.x8_scale_for(a.abs().max()), ops.x8_scale_for(w.abs().max(), 256.0)
    A, B = x8_rows_cpu(a, sa), x8_rows_cpu(w, sb)
    ah, a8, al8 = ops.unpack_x8_rows_host(A, M, K, sa)
    bh, b8, bl8 = ops.unpack_x8_rows_host(B, N, K, sb)
    out = torch.full((ks * M, N), float('nan'))
    g = GemmSplitArgs()
    g.fmt, g.lo_mul, g.ksplit = 1, 1.0 / (ops.SPLIT_LO_SCALE * sa * sb), ks
    g.A, g.B, g.C, g.bias, g.residual = A.data_ptr(), B.data_ptr(), out.data_ptr(), bias.data_ptr(), res.data_ptr()
    g.M, g.N, g.K, g.ldc, g.ldr = M, N, K, N, N
    old = lib.t2h_gemm_split_force_config(cfg)
    try:
        rc = lib.t2h_gemm_split_f32(ctypes.byref(g), None)
    finally:
        lib.t2h_gemm_split_force_config(old)
    assert rc == 0, lib.emu_last_error()
    kk = K // ks
    for s in range(ks):
        sl = slice(s * kk, (s + 1) * kk)
        part = ah[:, sl].double() @ bh[:, sl].double().t() + (a8[:, sl].double() @ bl8[:, sl].double().t()
                                                              + al8[:, sl].double() @ b8[:, sl].double().t()) / ops.SPLIT_LO_SCALE
        if s == 0:
            part = part + bias.double() + res.double()
        err = (out[s * M:(s + 1) * M].double() - part).abs()
        assert (err <= 3e-5 + 3e-5 * part.abs()).all(), (cfg, s, err.max().item())
    # a split needs the ping-pong loop, x8 operands and a plain fp32 output
    g.ksplit = 2
    old = lib.t2h_gemm_split_force_config(6)
    try:
        assert lib.t2h_gemm_split_f32(ctypes.byref(g), None) != 0 and b'ksplit' in lib.emu_last_error()
    finally:
        lib.t2h_gemm_split_force_config(old)
