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


def test_fold_layernorm_is_the_same_affine_map_and_its_column_sums_cancel_the_mean():
    """weights.fold_layernorm (t2h_gemm_split_args.ln_part_in): rstd (x W'^T - mean colsum) + b' with the
    split-rounded W' is LayerNorm(x) W^T + b; the (mean, M2) partials combine to the row moments the way
    the kernel's prologue does (Chan et al.)."""
    import torch.nn.functional as F
    from text2human_amd import weights
    g = torch.Generator().manual_seed(5)
    C, N, M = 512, 96, 40
    x = torch.randn(M, C, generator=g) * 3.0 + torch.randn(M, 1, generator=g) * 2.0
    w, b = torch.randn(N, C, generator=g) * 0.05, torch.randn(N, generator=g)
    gam, beta = torch.randn(C, generator=g) * 0.2 + 1.0, torch.randn(C, generator=g) * 0.3
    wf_split, cs, bf = weights.fold_layernorm(w, b, gam, beta)
    wf = ops.unsplit_rows_host(wf_split, N, C).double()
    assert (cs.double() - wf.sum(1)).abs().max().item() < 1e-6      # fp32 rounding of the fp64 sum
    # partials per 32 columns -> row moments
    xc = x.double().view(M, C // 32, 32)
    mu_i = xc.mean(-1)
    m2_i = ((xc - mu_i[..., None]) ** 2).sum(-1)
    mean = mu_i.mean(-1)
    var = (m2_i.sum(-1) + 32.0 * ((mu_i - mean[:, None]) ** 2).sum(-1)) / C
    assert (mean - x.double().mean(-1)).abs().max().item() < 1e-12
    assert (var - x.double().var(-1, unbiased=False)).abs().max().item() < 1e-10
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    folded = rstd[:, None] * (x.double() @ wf.t() - mean[:, None] * cs.double()[None, :]) + bf.double()
    ref = F.layer_norm(x.double(), (C, ), gam.double(), beta.double(), 1e-5) @ w.double().t() + b.double()
    assert (folded - ref).abs().max().item() < 2e-5                  # W' carries 22 bits
