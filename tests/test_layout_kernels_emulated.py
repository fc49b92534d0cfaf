"""The layout / glue kernels of the path on the CPU (csrc/misc.hip, csrc/vq.hip, csrc/sampler.hip through tests/emu,
entry points typed from the product's own signature table): one-hot, NCHW <-> NHWC, MaxPool2d(2), bilinear x2, channel
argmax, the attribute embedding, the texture-routed index-prediction head, the folded bottom gather, q_sample, the
masked cross entropy of the training-time forward, row gather and the round cursor -- against torch on the same data."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
import build_emu  # noqa: E402

pytestmark = pytest.mark.skipif(not build_emu.available(), reason='no host clang++ for the emulation build')


@pytest.fixture(scope='module')
def misc():
    return build_emu.load('misc.hip')


@pytest.fixture(scope='module')
def vq():
    return build_emu.load('vq.hip')


@pytest.fixture(scope='module')
def sampler():
    return build_emu.load('sampler.hip')


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def p(t):
    return t.data_ptr()


def ok(lib, rc):
    assert rc == 0, lib.emu_last_error()


def test_onehot_and_layout_round_trip(misc):
    B, H, W, n_cls, Cpad = 2, 6, 10, 24, 32
    segm = torch.randint(0, n_cls, (B, H, W), generator=torch.Generator().manual_seed(1)).float()
    out = torch.full((B * H * W, Cpad), float('nan'))
    ok(misc, misc.t2h_onehot_nhwc_f32(p(segm), p(out), B * H * W, n_cls, Cpad, None))
    want = F.one_hot(segm.long().view(-1), n_cls).float()
    assert torch.equal(out[:, :n_cls], want) and (out[:, n_cls:] == 0).all()
    x = rnd(B, 7, H * W, seed=2)
    y = torch.full((B * H * W, 8), float('nan'))
    ok(misc, misc.t2h_nchw_to_nhwc_f32(p(x), p(y), B, 7, H * W, 8, None))
    assert torch.equal(y[:, :7], x.permute(0, 2, 1).reshape(-1, 7))
    back = torch.full((B, 7, H * W), float('nan'))
    ok(misc, misc.t2h_nhwc_to_nchw_f32(p(y), 8, p(back), B, 7, H * W, None))
    assert torch.equal(back, x)


def test_maxpool_bilinear_and_argmax(misc):
    B, H, W, C = 2, 8, 12, 8
    x = rnd(B, C, H, W, seed=3)
    rows = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous()
    y = torch.full((B * (H // 2) * (W // 2), C), float('nan'))
    ok(misc, misc.t2h_maxpool2_nhwc_f32(p(rows), C, p(y), B, H, W, C, None))
    assert torch.equal(y, F.max_pool2d(x, 2).permute(0, 2, 3, 1).reshape(-1, C))
    up = torch.full((B * 4 * H * W, C), float('nan'))
    ok(misc, misc.t2h_bilinear_up2_nhwc_f32(p(rows), p(up), B, H, W, C, None))
    want = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False).permute(0, 2, 3, 1).reshape(-1, C)
    assert (up - want).abs().max().item() < 1e-6
    idx = torch.full((rows.shape[0],), -3, dtype=torch.int64)
    ok(misc, misc.t2h_argmax_rows_f32(p(rows), C, p(idx), rows.shape[0], C, None))
    assert torch.equal(idx, rows.argmax(1))


def test_shape_attr_embedding(misc):
    """shape_attr_embedding_arch.py:23-35: per attribute one-hot -> Linear(cls, 8) -> LeakyReLU -> Linear(8, 8); concat ->
    Linear(n_attr * 8, 128) -> LeakyReLU -> Linear(128, 128)"""
    B, dim, out_dim, cls = 3, 8, 128, [6, 5, 4, 7]
    n_attr = len(cls)
    g = torch.Generator().manual_seed(4)
    attr = torch.stack([torch.randint(0, c, (B,), generator=g) for c in cls], dim=1)
    w0 = [torch.randn(dim, c, generator=g) * 0.3 for c in cls]
    b0, w1, b1 = torch.randn(n_attr, dim, generator=g) * 0.1, torch.randn(n_attr, dim, dim, generator=g) * 0.3, torch.randn(n_attr, dim, generator=g) * 0.1
    f0, fb0 = torch.randn(out_dim, n_attr * dim, generator=g) * 0.2, torch.randn(out_dim, generator=g) * 0.1
    f1, fb1 = torch.randn(out_dim, out_dim, generator=g) * 0.1, torch.randn(out_dim, generator=g) * 0.1
    w0t = torch.cat([w.t() for w in w0], dim=0).contiguous()
    cls_off = torch.tensor([sum(cls[:a]) for a in range(n_attr)], dtype=torch.int32)
    out = torch.full((B, out_dim), float('nan'))
    ok(misc, misc.t2h_shape_attr_embed_f32(p(attr), p(cls_off), p(w0t), p(b0), p(w1), p(b1), p(f0), p(fb0), p(f1), p(fb1),
                                           p(out), B, n_attr, dim, out_dim, None))
    feats = []
    for a in range(n_attr):
        h = F.leaky_relu(F.one_hot(attr[:, a], cls[a]).double() @ w0[a].double().t() + b0[a].double(), 0.01)
        feats.append(h @ w1[a].double().t() + b1[a].double())
    h = F.leaky_relu(torch.cat(feats, 1) @ f0.double().t() + fb0.double(), 0.01)
    want = h @ f1.double().t() + fb1.double()
    assert (out.double() - want).abs().max().item() < 1e-5


def test_routed_head_argmax_and_folded_gather(vq):
    n, n_heads, Cf, n_class = 40, 3, 32, 64
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(n, n_heads * Cf, generator=g)
    w, b = torch.randn(n_heads, n_class, Cf, generator=g) * 0.3, torch.randn(n_heads, n_class, generator=g) * 0.1
    tex = torch.randint(0, n_heads, (n,), generator=g)
    lists = torch.full((n_heads, n), -9, dtype=torch.int64)
    ok(vq, vq.t2h_routed_head_argmax(p(feat), n_heads * Cf, p(w), p(b), p(tex), p(lists), n, n_heads, Cf, n_class, None))
    for r in range(n):
        h = int(tex[r])
        sc = w[h].double() @ feat[r, h * Cf:(h + 1) * Cf].double() + b[h].double()
        top = sc.topk(2)
        assert (lists[:, r] >= 0).sum() == 1 and lists[h, r] in top.indices.tolist()
        if top.values[0] - top.values[1] > 1e-4:
            assert lists[h, r] == top.indices[0]
    # bottom codebook gather + F.fold(k = 2, s = 2) (vqgan_arch.py:463-486): entries are [c, kh, kw] patches
    B, hh, ww, n_books, n_e, C = 2, 3, 4, 2, 16, 8
    books = torch.randn(n_books, n_e, C * 4, generator=g)
    tex2 = torch.randint(0, n_books, (B * hh * ww,), generator=g)
    idx = torch.full((n_books, B * hh * ww), -1, dtype=torch.int64)
    pick = torch.randint(0, n_e, (B * hh * ww,), generator=g)
    idx[tex2, torch.arange(B * hh * ww)] = pick
    out = torch.full((B, 2 * hh, 2 * ww, C), float('nan'))
    ok(vq, vq.t2h_codebook_gather_fold_f32(p(idx), p(tex2), p(books), p(out), B, hh, ww, n_books, n_e, C, None))
    ent = books[tex2, pick].view(B, hh * ww, C * 4).permute(0, 2, 1)
    want = F.fold(ent, (2 * hh, 2 * ww), kernel_size=2, stride=2).permute(0, 2, 3, 1)
    assert torch.equal(out, want.contiguous())


def test_q_sample_masked_ce_gather_and_round_cursor(sampler, misc):
    B, T, num_t, mask_id = 3, 16, 256, 1024
    g = torch.Generator().manual_seed(6)
    x0 = torch.randint(0, 1024, (B, T), generator=g)
    u = torch.rand(B, T, generator=g)
    t = torch.tensor([1, 128, 256])
    x_t = torch.full((B, T), -1, dtype=torch.int64)
    mask = torch.full((B, T), 7, dtype=torch.uint8)
    ok(sampler, sampler.t2h_q_sample(p(x0), p(u), p(t), num_t, mask_id, p(x_t), p(mask), B, T, None))
    want_mask = u < (t.float() / num_t)[:, None]
    assert torch.equal(mask.bool(), want_mask) and torch.equal(x_t, torch.where(want_mask, torch.full_like(x0, mask_id), x0))
    # masked cross entropy over the routed heads (transformer_model.py:258-274)
    C, n_class, n_heads = 512, 48, 3
    hidden = torch.randn(B * T, C, generator=g)
    gamma, beta = torch.randn(C, generator=g) * 0.2 + 1, torch.randn(C, generator=g) * 0.1
    w = torch.randn(n_heads, n_class, C, generator=g) * 0.05
    tex = torch.randint(0, n_heads, (B * T,), generator=g)
    gt = torch.full((n_heads, B * T), -1, dtype=torch.int64)
    gt[tex, torch.arange(B * T)] = torch.randint(0, n_class, (B * T,), generator=g)
    ce_rows, ce_samples = torch.full((B * T,), float('nan')), torch.full((B,), float('nan'))
    ok(sampler, sampler.t2h_masked_ce_heads(p(hidden), p(gamma), p(beta), p(w), p(tex), p(mask), p(gt), p(ce_rows), p(ce_samples),
                                            B, T, C, n_class, n_heads, None))
    y = F.layer_norm(hidden.double(), (C,), gamma.double(), beta.double(), 1e-5)
    want = torch.zeros(B * T, dtype=torch.float64)
    for h in range(n_heads):
        ce = F.cross_entropy(y @ w[h].double().t(), gt[h], ignore_index=-1, reduction='none')
        want += ce
    want = want * mask.view(-1).double()
    assert (ce_rows.double() - want).abs().max().item() < 1e-4
    assert (ce_samples.double() - want.view(B, T).sum(1)).abs().max().item() < 1e-3
    # row gather and the round cursor of a padded schedule table
    src = torch.randn(20, 8, generator=g)
    rows = torch.tensor([3, 3, 19, 0, 7], dtype=torch.int32)
    dst = torch.full((5, 8), float('nan'))
    ok(misc, misc.t2h_gather_rows(p(src), p(rows), p(dst), 5, 32, None))
    assert torch.equal(dst, src[rows.long()])
    maxr, rounds = 4, 3
    tbl = torch.arange(rounds * maxr, dtype=torch.int32).view(rounds, maxr)
    aux = (torch.arange(rounds * maxr, dtype=torch.int64) * 4).view(rounds, maxr)
    ctr = torch.zeros(1, dtype=torch.int32)
    cur, cur64 = torch.full((maxr,), -1, dtype=torch.int32), torch.full((maxr,), -1, dtype=torch.int64)
    for r in range(rounds):
        ok(sampler, sampler.t2h_schedule_advance(p(tbl), p(aux), None, p(ctr), p(cur), p(cur64), None, maxr, None))
        assert torch.equal(cur, tbl[r]) and torch.equal(cur64, aux[r]) and int(ctr[0]) == r + 1


def test_absmax_of_a_matrix_and_of_a_split_row_hi_plane(misc):
    """t2h_absmax_f32 / t2h_split_rows_absmax (load-time plumbing of the x8 scales: weights' and calibration
    activations' maxima, read without a torch reduction): the bits of the fp32 maximum, atomicMax'ed into the caller's
    word -- exact, order independent."""
    from text2human_amd import ops
    x = rnd(37, 96, seed=21, scale=3.0)
    x[11, 5] = -17.25
    bits = torch.zeros(2, dtype=torch.int32)
    assert misc.t2h_absmax_f32(p(x), 96, 37, 96, p(bits), None) == 0, misc.emu_last_error()
    assert bits.view(torch.float32)[0].item() == 17.25 and int(bits[1]) == 0
    # a strided view (ldx > C), accumulated into a word that already holds a smaller / larger maximum
    bits = torch.tensor([0, 0], dtype=torch.int32)
    bits.view(torch.float32)[1] = 100.0
    v = x[:, :64]
    assert misc.t2h_absmax_f32(p(v), 96, 37, 64, p(bits), None) == 0
    assert misc.t2h_absmax_f32(p(v), 96, 37, 64, p(bits[1:]), None) == 0
    assert bits.view(torch.float32)[0].item() == 17.25 and bits.view(torch.float32)[1].item() == 100.0
    hi, lo = ops.split_planes_host(x)
    sp = torch.stack([hi.view(37, 3, 32), (lo * 0 + 999).half().view(37, 3, 32)], dim=2).contiguous().view(torch.int16)
    bits = torch.zeros(1, dtype=torch.int32)
    assert misc.t2h_split_rows_absmax(p(sp), 37, 96, p(bits), None) == 0, misc.emu_last_error()
    assert bits.view(torch.float32)[0].item() == hi.float().abs().max().item()   # (the lo plane's 999 is not looked at)
