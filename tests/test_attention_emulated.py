"""The transformer's attention on the CPU (csrc/attention.hip through tests/emu): the split-precision flash attention in
both of its forms (128-query workgroups whose wave groups take the two key halves and merge; 256-query workgroups whose
eight waves walk all keys) -- q | k read as split rows, v from the transposed value planes with the keys in the order
the P.V matrix instruction contracts them, K / Vt tiles staged by LDS-DMA, the key halves synchronised by polled LDS
counters -- and the exact-fp32 attention, through their product entry points against fp64 softmax(q k^T / 8) v."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
import build_emu  # noqa: E402

from text2human_amd import ops  # noqa: E402

pytestmark = pytest.mark.skipif(not build_emu.available(), reason='no host clang++ for the emulation build')


@pytest.fixture(scope='module', params=[0, 1], ids=['requests-land-at-issue', 'requests-land-at-the-wait'])
def lib(request):
    so = build_emu.load('attention.hip')
    so.emu_set_deferred(request.param)   # (K / Vt tiles by LDS-DMA: landing at issue, or only at the kernel's own waits)
    yield so
    so.emu_set_deferred(0)


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def split_rows_cpu(w):
    hi, lo = ops.split_planes_host(w)
    r, C = w.shape
    return torch.stack([hi.view(r, C // 32, 32), lo.view(r, C // 32, 32)], dim=2).contiguous().view(torch.int16)


def reference(q, k, v, B, T, H):
    qh, kh, vh = (t.double().view(B, T, H, 64).permute(0, 2, 1, 3) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(2, 3) / 8.0, dim=3)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B * T, H * 64)


@pytest.mark.parametrize('form,B,T,H', [(2, 1, 256, 2),    # key halves + merge (what fills the chip at B = 8)
                                        (1, 1, 256, 1),    # all keys, 256-query workgroups
                                        (2, 2, 128, 1)])   # one K tile per key half
def test_emulated_split_attention(lib, form, B, T, H):
    C = 64 * H
    q, k, v = rnd(B * T, C, seed=1) * 1.2, rnd(B * T, C, seed=2) * 1.2, rnd(B * T, C, seed=3)
    qk = split_rows_cpu(torch.cat([q, k], dim=1))          # [B*T][2C/32][2][32]: q at columns [0, C), k at [C, 2C)
    # the value planes as the q|k|v projection writes them: Vt[B][H][plane][d][key position]
    vh, vl = ops.split_planes_host(v)
    pos = ops.vt_key_positions(T)
    vt = torch.zeros(B, H, 2, 64, T, dtype=torch.float16)
    for pl, plane in enumerate((vh, vl)):
        pv = plane.view(B, T, H, 64).permute(0, 2, 3, 1)   # [B][H][d][key]
        vt[:, :, pl, :, pos] = pv
    vt = vt.contiguous().view(torch.int16)
    y = torch.full((B * T, C), float('nan'))
    ys = torch.zeros((B * T, C // 32, 2, 32), dtype=torch.int16)
    ovf = torch.zeros(1, dtype=torch.int32)
    old = lib.t2h_mha_split_force_form(form)
    try:
        rc = lib.t2h_mha_split_f32(qk.data_ptr(), 2 * C, vt.data_ptr(), y.data_ptr(), ys.data_ptr(), B, T, H, ovf.data_ptr(), None)
    finally:
        lib.t2h_mha_split_force_form(old)
    assert rc == 0, lib.emu_last_error()
    # the operands the kernel saw are the split planes' values (22 bits of the fp32 ones)
    unsplit = lambda t: ops.unsplit_rows_host(split_rows_cpu(t), t.shape[0], t.shape[1])  # noqa: E731
    ref = reference(unsplit(q), unsplit(k), unsplit(v), B, T, H)
    assert (y.double() - ref).abs().max().item() < 2e-5
    assert int(ovf[0]) == 0 and (ops.unsplit_rows_host(ys, B * T, C).double() - y.double()).abs().max().item() < 2e-6


def test_emulated_exact_fp32_attention(lib):
    B, T, H = 1, 128, 2
    C = 64 * H
    qkv = rnd(B * T, 3 * C, seed=4)
    y = torch.full((B * T, C), float('nan'))
    assert lib.t2h_mha_noncausal_f32(qkv.data_ptr(), y.data_ptr(), B, T, H, None) == 0, lib.emu_last_error()
    q, k, v = qkv.split(C, dim=1)
    assert (y.double() - reference(q, k, v, B, T, H)).abs().max().item() < 2e-5


def test_emulated_attention_writes_x8_rows_of_its_output(lib):
    """t2h_mha_split_x8_f32: the attention output as the x8 rows proj reads (fp16 plane + two e4m3 planes)"""
    B, T, H = 1, 128, 1
    C = 64 * H
    q, k, v = rnd(B * T, C, seed=5), rnd(B * T, C, seed=6), rnd(B * T, C, seed=7)
    qk = split_rows_cpu(torch.cat([q, k], dim=1))
    vh, vl = ops.split_planes_host(v)
    pos = ops.vt_key_positions(T)
    vt = torch.zeros(B, H, 2, 64, T, dtype=torch.float16)
    for pl, plane in enumerate((vh, vl)):
        vt[:, :, pl, :, pos] = plane.view(B, T, H, 64).permute(0, 2, 3, 1)
    vt = vt.contiguous().view(torch.int16)
    y = torch.full((B * T, C), float('nan'))
    ovf = torch.zeros(1, dtype=torch.int32)
    assert lib.t2h_mha_split_f32(qk.data_ptr(), 2 * C, vt.data_ptr(), y.data_ptr(), None, B, T, H, None, None) == 0
    scale = ops.x8_scale_for(y.abs().max())
    y8 = torch.zeros((B * T, C // 32, 2, 32), dtype=torch.int16)
    rc = lib.t2h_mha_split_x8_f32(qk.data_ptr(), 2 * C, vt.data_ptr(), y8.data_ptr(), scale, B, T, H, ovf.data_ptr(), None)
    assert rc == 0, lib.emu_last_error()
    hi, h8, l8 = ops.unpack_x8_rows_host(y8, B * T, C, scale)
    assert int(ovf[0]) == 0 and torch.equal(hi, y.half().float())
    lo = (y - hi) * ops.SPLIT_LO_SCALE
    assert (h8 - hi).abs().max().item() <= 2.0**-4 * float(hi.abs().max()) and (l8 - lo).abs().max().item() <= 2.0**-4 * float(lo.abs().max())
