"""Host-side pieces of the split-precision path (CPU): the two-plane fp16 representation,
its row layout, and the key order of the transposed value planes."""
import pytest
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


_B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
_B128_GROUPS += [[l + 32 for l in g] for g in _B128_GROUPS]   # lanes a ds_read_b128 serves in one LDS cycle


def _conflict_free(addr_of_lane):
    for g in _B128_GROUPS:
        banks = set()
        for lane in g:
            for d in range(4):
                bank = (addr_of_lane(lane) // 4 + d) % 64
                assert bank not in banks
                banks.add(bank)


def test_lds_dma_swizzle_algebra_of_the_attention_kernel():
    """mha_split_pipe_kernel (csrc/attention.hip): a DMA instruction fills 1 KiB of LDS lane-linearly, so
    the tile images are unpadded and an XOR swizzle is applied on the SOURCE piece the lane requests and
    on the fragment read address.  This replays the kernel's index formulas: every fragment read must
    find the piece it means, in a bank-conflict-free pattern."""
    # K tile: 64 keys x 16 pieces of 16 B; piece p of key r at p ^ (r & 15)
    k_slot = {}
    for w in range(4):                       # wave of the key half
        for i in range(4):                   # its chunks 4w + i: rows 16w + 4i + (lane >> 4)
            for lane in range(64):
                rr, pp = lane >> 4, lane & 15
                r = 16 * w + 4 * i + rr
                src_piece = (((pp ^ rr) * 16) ^ (64 * i)) // 16          # k_b16 ^ 64 i
                lds_slot = ((4 * w + i) * 1024 + lane * 16) // 16        # lane-linear destination
                assert lds_slot == r * 16 + pp and src_piece == pp ^ (r & 15)
                k_slot[lds_slot] = (r, src_piece)
    assert len(k_slot) == 64 * 16
    for ks in range(2):
        for kk in range(4):
            for pl in range(2):
                const_p = (kk >> 1) * 8 + pl * 4 + (kk & 1) * 2
                addr = lambda lane: (ks * 32 + (lane & 31)) * 256 + ((const_p * 16) ^ (((lane >> 5) ^ (lane & 15)) * 16))
                for lane in range(64):
                    assert k_slot[addr(lane) // 16] == (ks * 32 + (lane & 31), const_p + (lane >> 5))
                _conflict_free(addr)
    # Vt tile: 128 (plane, d) rows x 8 pieces; piece p of row r at p ^ ((r >> 1) & 7)
    v_slot = {}
    for w in range(4):
        for i in range(4):                   # rows 32w + 8i + (lane >> 3)
            for lane in range(64):
                rr, pp = lane >> 3, lane & 7
                vr = 32 * w + 8 * i + rr
                src_piece = (((pp ^ (lane >> 4)) * 16) ^ (64 * (i & 1))) // 16   # v_b16 ^ 64 (i & 1)
                lds_slot = ((4 * w + i) * 1024 + lane * 16) // 16
                assert lds_slot == vr * 8 + pp and src_piece == pp ^ ((vr >> 1) & 7)
                v_slot[lds_slot] = (vr, src_piece)
    assert len(v_slot) == 128 * 8
    for pl in range(2):
        for dt in range(2):
            for ks in range(2):
                for j in range(2):
                    const_q = ks * 4 + j * 2
                    addr = lambda lane: ((pl * 64 + dt * 32 + (lane & 31)) * 128 +
                                         ((const_q * 16) ^ (((lane >> 5) ^ (((lane & 31) >> 1) & 7)) * 16)))
                    for lane in range(64):
                        assert v_slot[addr(lane) // 16] == (pl * 64 + dt * 32 + (lane & 31), const_q + (lane >> 5))
                    _conflict_free(addr)


@pytest.mark.parametrize('BM,BN,WM,WN', [(256, 128, 64, 64), (128, 192, 32, 96)])
def test_lds_dma_swizzle_algebra_of_the_gemm_tile_image(BM, BN, WM, WN):
    """gemm_split_kernel<.., PP = 2> (csrc/gemm_split.hip): 8-row groups of 128-byte rows, logical piece
    c of row r at c ^ ((r >> 1) & 7); fragment piece = plane * 4 + u * 2 + (lane >> 5).  Both tile
    shapes of the loop: 256x128 (wave tile 64x64) and 128x192 (wave tile 32x96)."""
    rows = BM + BN                               # A rows + B rows of a K tile
    slot = {}
    for wave in range(8):
        for i in range(rows // 64):
            g = wave + 8 * i
            for lane in range(64):
                r = g * 8 + (lane >> 3)
                pc = (lane & 7) ^ ((r >> 1) & 7)                          # source piece
                slot[(g * 1024 + lane * 16) // 16] = (r, pc)
                assert (r < BM) == (i < BM // 64)                         # one operand per DMA round i
    assert len(slot) == rows * 8
    frag_rows = [wm0 + ti * 32 for wm0 in range(0, BM, WM) for ti in range(WM // 32)]
    frag_rows += [BM + wn0 + tj * 32 for wn0 in range(0, BN, WN) for tj in range(WN // 32)]
    assert sorted(frag_rows) == list(range(0, rows, 32))
    for row0 in frag_rows:                       # MFMA-tile row offsets (multiples of 32)
        for c2 in range(4):                      # c2 = plane * 2 + u
            addr = lambda lane: (row0 + (lane & 31)) * 128 + (((c2 * 2 + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)) * 16)
            for lane in range(64):
                assert slot[addr(lane) // 16] == (row0 + (lane & 31), c2 * 2 + (lane >> 5))
            _conflict_free(addr)
