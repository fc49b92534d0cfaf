"""Host-side pieces of the split-precision path (CPU): the two-plane fp16 representation,
its row layout, and the key order of the transposed value planes."""
import torch

from text2human_amd import ops


def test_split_planes_carry_22_bits_and_survive_small_magnitudes():
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(4096, generator=g) * s for s in (1e-6, 1e-3, 1.0, 50.0, 1e4)])  # |x| < 65504
    hi, lo = ops.split_planes_host(x)
    back = hi.double() + lo.double() / ops.SPLIT_LO_SCALE
    err = (back - x.double()).abs()
    assert (err <= x.double().abs() * 2.0**-21 + 2.0**-36).all()
    assert torch.isfinite(hi.float()).all() and torch.isfinite(lo.float()).all()
    # the residual plane is stored scaled by 2^11, so it is a NORMAL fp16 number whenever
    # the value itself is above fp16's normal range times 2^-11
    big = x.abs() > 2.0**-3
    nz = lo[big].float().abs()
    assert (nz[nz > 0] >= 2.0**-14).all()


def test_pack_split_rows_layout_round_trip():
    g = torch.Generator().manual_seed(1)
    w = torch.randn(96, 64, generator=g) * 0.05
    packed = ops.pack_split_rows_host(w)                     # int16 [rows, K/32, 2, 32]
    assert packed.shape == (96, 2, 2, 32) and packed.dtype == torch.int16
    hi, lo = ops.split_planes_host(w)
    pl = packed.view(torch.float16)
    assert torch.equal(pl[:, 1, 0], hi[:, 32:64]) and torch.equal(pl[:, 0, 1], lo[:, 0:32])
    back = ops.unsplit_rows_host(packed, 96, 64)
    assert (back - w).abs().max().item() <= 2.0**-21 * w.abs().max().item()


def test_value_plane_key_order_is_the_mfma_contraction_order():
    """include/t2h_hip.h: inside every 32-key group, k16-step j of lane half h contracts the keys
    {16j + 4h + (e & 3) + 8(e >> 2)}, e = 0..7, and they must sit at positions 16j + 8h + e."""
    T = 128
    pos = ops.vt_key_positions(T)
    assert sorted(pos.tolist()) == list(range(T))
    for grp in range(T // 32):
        for j in range(2):
            for h in range(2):
                for e in range(8):
                    key = 32 * grp + 16 * j + 4 * h + (e & 3) + 8 * (e >> 2)
                    assert pos[key].item() == 32 * grp + 16 * j + 8 * h + e


def test_out_of_range_weights_are_rejected_loudly():
    import pytest
    with pytest.raises(ValueError):
        ops.pack_split_rows_host(torch.full((32, 32), 7.0e4))
