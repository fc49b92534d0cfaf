"""More of the path's kernels on the CPU through tests/emu (the kernel SOURCE compiled for the host, called through its
product entry point): the segm tokenizer's codebook argmin (T-3, vqgan_arch.py:88-92: first minimum wins), the
texture-routed argmin of the encode side, the texture-routed codebook gather (R-1), the image epilogue (D-5,
sample_model.py:245-254: the uint8 rounding bit for bit) and the texture map (P-3).  Decisions are compared with fp64
where the margin to the runner-up is not a rounding matter, exactly otherwise."""
import ctypes
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
import build_emu  # noqa: E402

pytestmark = pytest.mark.skipif(not build_emu.available(), reason='no host clang++ for the emulation build')
c_vp, c_i32, c_i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64


def _load(kernel_file, sigs):
    so = ctypes.CDLL(build_emu.build(kernel_file))
    for name, args in sigs.items():
        getattr(so, name).restype = ctypes.c_int
        getattr(so, name).argtypes = args
    so.emu_last_error.restype = ctypes.c_char_p
    return so


@pytest.fixture(scope='module')
def vq():
    return _load('vq.hip', {
        't2h_vq_l2_argmin_f32': [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp],
        't2h_vq_argmin_tex_f32': [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
        't2h_codebook_gather_tex_f32': [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]})


@pytest.fixture(scope='module')
def misc():
    return _load('misc.hip', {'t2h_image_epilogue': [c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp],
                              't2h_texture_map': [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]})


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def check_argmin(got, z, book):
    """got[row] is the fp64 argmin unless the two best distances are within fp32 rounding of each other"""
    d = (z.double()[:, None, :] - book.double()[None]).pow(2).sum(-1)
    best = d.argmin(1)
    two = d.topk(2, dim=1, largest=False).values
    clear = (two[:, 1] - two[:, 0]) > 1e-4 * two[:, 1].abs().clamp_min(1.0)
    assert clear.float().mean().item() > 0.9
    assert torch.equal(got[clear], best[clear])
    near = ~clear  # a near-tie may go either way, but to one of the two best
    assert ((d[near, got[near]] - two[near, 0]).abs() <= 1e-4 * two[near, 1].abs().clamp_min(1.0)).all()


@pytest.mark.parametrize('d', [32, 64])
def test_emulated_codebook_argmin(vq, d):
    n, n_e = 70, 512   # (a partly filled third workgroup; two codebook chunks)
    z, book = rnd(n, d, seed=1), rnd(n_e, d, seed=2)
    z[5] = book[77]                       # an exact hit
    book[300] = book[20]                  # a duplicated code: the FIRST one wins (vqgan_arch.py:91, torch.argmin)
    z[6] = book[20] + 1e-3
    idx = torch.full((n,), -7, dtype=torch.int64)
    assert vq.t2h_vq_l2_argmin_f32(z.data_ptr(), book.data_ptr(), idx.data_ptr(), n, n_e, d, None) == 0, vq.emu_last_error()
    assert idx[5] == 77 and idx[6] == 20
    check_argmin(idx, z, book)


def test_emulated_texture_routed_argmin_and_gather(vq):
    n, n_books, n_e, d = 12, 3, 64, 256
    z = rnd(n, d, seed=3)
    books = rnd(n_books, n_e, d, seed=4)
    tex = torch.randint(0, n_books, (n,), generator=torch.Generator().manual_seed(5))
    lists = torch.full((n_books, n), -9, dtype=torch.int64)
    rc = vq.t2h_vq_argmin_tex_f32(z.data_ptr(), books.data_ptr(), tex.data_ptr(), lists.data_ptr(), n, n_books, n_e, d, 0, 0, None)
    assert rc == 0, vq.emu_last_error()
    for h in range(n_books):
        rows = (tex == h).nonzero().flatten()
        assert (lists[h][tex != h] == -1).all()
        if len(rows):
            check_argmin(lists[h][rows], z[rows], books[h])
    out = torch.full((n, d), float('nan'))
    rc = vq.t2h_codebook_gather_tex_f32(lists.data_ptr(), tex.data_ptr(), books.data_ptr(), out.data_ptr(), n, n_books, n_e, d, None)
    assert rc == 0, vq.emu_last_error()
    want = torch.stack([books[tex[r], lists[tex[r], r]] for r in range(n)])
    assert torch.equal(out, want)


def test_emulated_image_epilogue_rounds_like_the_reference(misc):
    """img = ((dec + 1) / 2).clamp(0, 1); u8 = img.mul(255).add(0.5).clamp(0, 255).byte(): bit for bit"""
    B, HW, ld = 2, 300, 4
    dec = rnd(B * HW, ld, seed=8) * 0.8
    dec[0, :3] = torch.tensor([-1.0, 1.0, 0.0])
    dec[1, :3] = torch.tensor([-3.0, 3.0, 1.0 / 255.0 - 1.0])
    img = torch.full((B, 3, HW), float('nan'))
    u8 = torch.zeros((B, HW, 3), dtype=torch.uint8)
    assert misc.t2h_image_epilogue(dec.data_ptr(), ld, img.data_ptr(), u8.data_ptr(), B, HW, None) == 0, misc.emu_last_error()
    want = ((dec[:, :3] + 1) / 2).clamp(0, 1).view(B, HW, 3)
    assert torch.equal(img, want.permute(0, 2, 1))
    assert torch.equal(u8, want.mul(255).add(0.5).clamp(0, 255).to(torch.uint8))


def test_emulated_texture_map(misc):
    B, HW = 3, 200
    g = torch.Generator().manual_seed(9)
    segm = torch.randint(0, 24, (B, HW), generator=g)
    upper, lower, outer = torch.tensor([3, 17, 0]), torch.tensor([17, 5, 9]), torch.tensor([1, 2, 17])
    mask = torch.full((B, HW), float('nan'))
    rc = misc.t2h_texture_map(segm.data_ptr(), upper.data_ptr(), lower.data_ptr(), outer.data_ptr(), mask.data_ptr(), B, HW, None)
    assert rc == 0, misc.emu_last_error()
    want = torch.zeros(B, HW)
    for b in range(B):   # generate_texture_map (sample_model.py:443-471): upper {1, 4}, lower {3, 5, 21}, outer {2}; 17 = none
        for classes, attr in (((1, 4), upper[b]), ((3, 5, 21), lower[b]), ((2,), outer[b])):
            if attr != 17:
                for c in classes:
                    want[b][segm[b] == c] = float(attr + 1)
    assert torch.equal(mask, want)
