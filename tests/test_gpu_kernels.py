"""Per-kernel parity (-m gpu): every C-ABI entry point against a plain PyTorch
fp32/fp64 CPU reference of the same op on seeded, ASYMMETRIC random inputs.

Tolerances: the matrix kernels are exact-fp32 fma chains (v_mfma_f32_32x32x2),
so errors are summation-order only: |err| <= 2e-5 * (1 + |ref|) for K <= 4608;
integer outputs must be bit-exact except where a float near-tie is proven.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from text2human_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def assert_close(got, ref, rtol=2e-5, atol=2e-5, what=''):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = err > bound
    assert not bad.any(), (f'{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance, '
                           f'max abs err {err.max().item():.3e}, ref absmax {ref.abs().max().item():.3e}')


# ------------------------------------------------------------------ GEMM


@pytest.mark.parametrize('M,N,K', [(512, 512, 512), (4096, 1536, 512), (1024, 2048, 512),
                                   (640, 512, 2048), (200, 96, 64), (40, 24, 32), (130, 3, 128),
                                   (2, 1024, 1024), (8192, 128, 1152)])
def test_gemm_plain(M, N, K):
    a, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3), rnd(M, N, seed=4)
    ref = a.double() @ w.double().t() + b.double()
    out = ops.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV))
    assert_close(out, ref, what='bias')
    out = ops.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV), act=ops.ACT_GELU, residual=r.to(DEV))
    assert_close(out, F.gelu(ref) + r.double(), what='gelu+res')
    out = ops.gemm(a.to(DEV), w.to(DEV), act=ops.ACT_RELU, alpha=0.5)
    assert_close(out, F.relu(0.5 * (a.double() @ w.double().t())), what='relu alpha')


def test_gemm_inplace_residual_and_strided_views():
    M, K, N = 1024, 512, 512
    x, w = rnd(M, N, seed=5).to(DEV), rnd(N, K, seed=6, scale=0.1).to(DEV)
    big = rnd(M, 3 * K, seed=7).to(DEV)
    a = big[:, K:2 * K]                      # lda = 3K view
    ref = x.cpu().double() + a.cpu().double() @ w.cpu().double().t()
    ops.gemm(a, w, out=x, residual=x)
    assert_close(x, ref)
    outbig = torch.zeros(M, 2 * N, device=DEV)
    ops.gemm(a, w, out=outbig[:, N:])         # ldc = 2N view
    assert_close(outbig[:, N:], a.cpu().double() @ w.cpu().double().t())
    assert outbig[:, :N].abs().max().item() == 0.0


def test_gemm_prologue_plain():
    n_img, hw, C, N = 3, 512, 256, 768
    x, w = rnd(n_img * hw, C, seed=8), rnd(N, C, seed=9, scale=0.1)
    sc, sh = rnd(n_img, C, seed=10) * 0.5 + 1, rnd(n_img, C, seed=11)
    xa = x.view(n_img, hw, C) * sc[:, None] + sh[:, None]
    for pact, f in ((ops.PRO_NONE, lambda t: t), (ops.PRO_SWISH, lambda t: t * torch.sigmoid(t))):
        ref = f(xa.double()).view(-1, C) @ w.double().t()
        out = ops.gemm(x.to(DEV), w.to(DEV), pro=(sc.to(DEV), sh.to(DEV), hw, pact))
        assert_close(out, ref, what=f'pro{pact}')


@pytest.mark.parametrize('nb,M,N,K', [(3, 512, 512, 512), (2, 2048, 2048, 512), (2, 96, 64, 64)])
def test_bgemm_nt_and_trans(nb, M, N, K):
    qkv = rnd(nb, M, 3 * K, seed=12).to(DEV)
    q, k, v = qkv[:, :, :K], qkv[:, :, K:2 * K], qkv[:, :, 2 * K:]
    s = torch.empty(nb, M, M, device=DEV)
    ops.bgemm(q, k, s, alpha=K**-0.5)
    ref = (q.cpu().double() @ k.cpu().double().transpose(1, 2)) * K**-0.5
    assert_close(s, ref, what='QK^T')
    p = torch.softmax(ref, -1).float().to(DEV)
    o = torch.empty(nb, M, K, device=DEV)
    ops.bgemm(p, v, o, b_trans=True)
    assert_close(o, p.cpu().double() @ v.cpu().double(), what='PV (b_trans)')


@pytest.mark.parametrize('mode', ['same', 'up', 'down'])
@pytest.mark.parametrize('cin,cout,h,w', [(64, 128, 16, 8), (32, 3, 32, 16), (256, 64, 8, 4)])
def test_conv3x3(mode, cin, cout, h, w):
    from text2human_amd import weights
    n_img = 2
    x = rnd(n_img, cin, h, w, seed=13)
    wt, b = rnd(cout, cin, 3, 3, seed=14, scale=0.1), rnd(cout, seed=15)
    sc, sh = rnd(n_img, cin, seed=16) * 0.3 + 1, rnd(n_img, cin, seed=17) * 0.3
    xd = x.double()
    for use_pro in (False, True):
        xin = xd
        if use_pro:
            xin = xd * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]
            xin = xin * torch.sigmoid(xin)
        if mode == 'same':
            ref = F.conv2d(xin, wt.double(), b.double(), 1, 1)
        elif mode == 'up':
            ref = F.conv2d(F.interpolate(xin, scale_factor=2.0, mode='nearest'), wt.double(), b.double(), 1, 1)
        else:
            ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), wt.double(), b.double(), 2, 0)
        ho, wo = ref.shape[2:]
        res = rnd(n_img * ho * wo, cout, seed=18)
        rows = ops.nchw_to_nhwc(x.to(DEV))
        out = ops.conv3x3(rows, weights.pack_conv3x3(wt).to(DEV), n_img, h, w, cin, bias=b.to(DEV),
                          residual=res.to(DEV), mode=mode,
                          pro=(sc.to(DEV), sh.to(DEV), ops.PRO_SWISH) if use_pro else None)
        ref_rows = ref.permute(0, 2, 3, 1).reshape(-1, cout) + res.double()
        assert_close(out, ref_rows, what=f'{mode} pro={use_pro}')


@pytest.mark.parametrize('cin,cout,h,w', [(512, 1024, 2, 1), (1024, 512, 4, 2), (256, 256, 16, 8), (64, 64, 32, 16)])
def test_conv3x3_split_over_k_at_the_deep_unet_levels(cin, cout, h, w):
    """unet_arch.py:470-481,657-674: the index-prediction / parsing UNets' deep levels are a handful of pixels per image
    against K = 9 * 512 .. 9 * 1024 -- t2h_gemm_f32 splits K across workgroups there (t2h_gemm_args.ksplit, round 6).
    Against fp64; no further from it than the single pass; an image's values do not depend on its batch (the slice
    count is a function of the layer's geometry only) -- bit for bit."""
    from text2human_amd import _lib, weights
    n_img = 8
    x = rnd(n_img, cin, h, w, seed=51)
    wt, b = rnd(cout, cin, 3, 3, seed=52, scale=(9 * cin) ** -0.5), rnd(cout, seed=53)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), 1, 1)).permute(0, 2, 3, 1).reshape(-1, cout)
    rows = ops.nchw_to_nhwc(x.to(DEV))
    wp = weights.pack_conv3x3(wt).to(DEV)
    auto = ops.conv3x3(rows, wp, n_img, h, w, cin, bias=b.to(DEV), act=ops.ACT_RELU)
    one = ops.conv3x3(rows, wp, n_img, h, w, cin, bias=b.to(DEV), act=ops.ACT_RELU, ksplit=1)
    forced = ops.conv3x3(rows, wp, n_img, h, w, cin, bias=b.to(DEV), act=ops.ACT_RELU, ksplit=3)
    g = _lib.GemmArgs()
    g.M, g.N, g.K, g.a_mode, g.batch, g.Hout, g.Wout, g.Cin = n_img * h * w, cout, 9 * cin, 1, 1, h, w, cin
    import ctypes
    g.splitk_ws = ctypes.c_void_p(1)
    assert _lib.load().t2h_gemm_ksplit(ctypes.byref(g)) > 1      # these geometries ARE split
    e_auto, e_one = (auto.cpu().double() - ref).abs().max().item(), (one.cpu().double() - ref).abs().max().item()
    assert_close(auto, ref, what='split over K')
    assert_close(forced, ref, what='3 slices')
    assert e_auto <= e_one + 1e-6, (e_auto, e_one)
    alone = ops.conv3x3(rows[:h * w].contiguous(), wp, 1, h, w, cin, bias=b.to(DEV), act=ops.ACT_RELU)
    assert torch.equal(alone, auto[:h * w]), 'an image depends on its batch'


def test_conv3x3_channel_padding_and_relu_prebias():
    from text2human_amd import weights
    n_img, cin, cout, h, w = 2, 24, 64, 16, 16
    x, wt, b = rnd(n_img, cin, h, w, seed=19), rnd(cout, cin, 3, 3, seed=20, scale=0.2), rnd(cout, seed=21)
    bm = rnd(n_img * h * w, cout, seed=22)
    rows = ops.nchw_to_nhwc(x.to(DEV), cpad=32)
    out = ops.conv3x3(rows, weights.pack_conv3x3(wt).to(DEV), n_img, h, w, 32, bias=b.to(DEV),
                      act=ops.ACT_RELU, residual=bm.to(DEV), res_pre=True)
    ref = F.conv2d(x.double(), wt.double(), b.double(), 1, 1).permute(0, 2, 3, 1).reshape(-1, cout)
    assert_close(out, F.relu(ref + bm.double()))


# ------------------------------------------------------------------ norms / softmax


def test_layernorm():
    x, g, b = rnd(1000, 512, seed=23) * 3 + 1, rnd(512, seed=24), rnd(512, seed=25)
    ref = F.layer_norm(x.double(), (512, ), g.double(), b.double(), 1e-5)
    assert_close(ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV)), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('C,hw', [(128, 4096), (512, 512), (64, 1024), (256, 2050)])
def test_groupnorm_tables(C, hw):
    n_img = 3
    x = rnd(n_img, C, hw, seed=26) * 2 + 0.7
    g, b = rnd(C, seed=27), rnd(C, seed=28)
    ref = F.group_norm(x.double(), 32, g.double(), b.double(), 1e-6)
    rows = x.permute(0, 2, 1).reshape(-1, C).contiguous().to(DEV)
    sc, sh = ops.groupnorm_tables(rows, g.to(DEV), b.to(DEV), n_img, hw)
    got = rows.view(n_img, hw, C) * sc[:, None] + sh[:, None]
    assert_close(got.permute(0, 2, 1), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('n', [512, 2048, 100])
def test_softmax_rows(n):
    x = rnd(300, n, seed=29) * 4
    got = ops.softmax_rows_(x.clone().to(DEV))
    assert_close(got, torch.softmax(x.double(), -1), rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------ sampler pieces


def test_embed_sum4():
    g = torch.Generator().manual_seed(30)
    B, T, C = 3, 512, 512
    idx = torch.randint(0, 18433, (B, T), generator=g)
    seg = torch.randint(0, 1024, (B, T), generator=g)
    tex = torch.randint(0, 18, (B, T), generator=g)
    te, pe, se, xe = rnd(18433, C, seed=31), rnd(T, C, seed=32), rnd(1024, C, seed=33), rnd(18, C, seed=34)
    ref = te[idx] + pe[None] + se[seg] + xe[tex]
    got = ops.embed_sum4(idx.to(DEV), seg.to(DEV), tex.to(DEV), te.to(DEV), pe.to(DEV), se.to(DEV), xe.to(DEV))
    assert torch.equal(got.cpu().view(B, T, C), ref)


@pytest.mark.parametrize('B,T', [(1, 512), (3, 512), (2, 128)])
def test_mha_noncausal(B, T):
    H, hd = 8, 64
    qkv = rnd(B * T, 3 * H * hd, seed=35) * 1.5
    q, k, v = [t.view(B, T, H, hd).transpose(1, 2).double() for t in qkv.split(H * hd, dim=1)]
    att = torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(hd), -1)
    ref = (att @ v).transpose(1, 2).reshape(B * T, H * hd)
    got = ops.mha_noncausal(qkv.to(DEV), B, T, H)
    assert_close(got, ref, rtol=2e-5, atol=2e-5)


def test_mha_rescale_branch_with_spiked_keys():
    """Forces the online-softmax max to jump late (rule: a rare data-dependent
    branch needs its own test): key rows late in the sequence dominate."""
    B, T, H, hd = 1, 512, 8, 64
    qkv = rnd(B * T, 3 * H * hd, seed=36)
    qkv[450:460, 512:1024] *= 25.0
    qkv[3, 512:1024] *= 40.0
    q, k, v = [t.view(B, T, H, hd).transpose(1, 2).double() for t in qkv.split(H * hd, dim=1)]
    ref = (torch.softmax(q @ k.transpose(-2, -1) / 8.0, -1) @ v).transpose(1, 2).reshape(B * T, H * hd)
    assert_close(ops.mha_noncausal(qkv.to(DEV), B, T, H), ref, rtol=5e-5, atol=5e-5)


def test_unmask_step():
    n = 2048
    g = torch.Generator().manual_seed(37)
    r = torch.rand(n, generator=g)
    um = (torch.rand(n, generator=g) < 0.3)
    tex = torch.randint(0, 18, (n, ), generator=g)
    for t in (1, 3, 256):
        unm = um.to(torch.uint8).to(DEV)
        ch = torch.zeros(n, dtype=torch.uint8, device=DEV)
        cnt = torch.zeros(18, dtype=torch.int32, device=DEV)
        ops.unmask_step(r.to(DEV), t, unm, ch, tex.to(DEV), cnt)
        changes = r < 1 / torch.tensor(float(t))
        changes = torch.bitwise_xor(changes, torch.bitwise_and(changes, um))
        assert torch.equal(ch.cpu().bool(), changes)
        assert torch.equal(unm.cpu().bool(), um | changes)
        assert torch.equal(cnt.cpu().long(), torch.bincount(tex[changes], minlength=18))
        # compact list of the changed rows (any order) + their total in cnt[18]
        unm2 = um.to(torch.uint8).to(DEV)
        cnt2 = torch.zeros(19, dtype=torch.int32, device=DEV)
        rows = torch.full((n, ), -1, dtype=torch.int32, device=DEV)
        ops.unmask_step(r.to(DEV), t, unm2, ch, tex.to(DEV), cnt2, rows, 18)
        k = int(cnt2[18])
        assert k == int(changes.sum()) and torch.equal(cnt2[:18].cpu(), cnt.cpu())
        assert torch.equal(rows[:k].cpu().long().sort().values, torch.nonzero(changes).flatten())


def test_sample_head_matches_categorical_race():
    n, C, K = 256, 512, 1024
    hidden, g, b = rnd(n, C, seed=38) * 2, rnd(C, seed=39) * 0.1 + 1, rnd(C, seed=40) * 0.1
    w = rnd(K, C, seed=41, scale=0.15)
    gen = torch.Generator().manual_seed(42)
    expo = torch.empty(n, K).exponential_(1.0, generator=gen)
    tex = torch.randint(0, 18, (n, ), generator=gen)
    changes = (torch.rand(n, generator=gen) < 0.5)
    head = 7
    tex[:40] = head
    logits = F.layer_norm(hidden.double(), (C, ), g.double(), b.double(), 1e-5) @ w.double().t()
    probs = torch.softmax(logits, -1)
    score = probs / expo.double()
    ref = score.argmax(-1)
    top2 = score.topk(2, -1).values
    margin = (top2[:, 0] - top2[:, 1]) / top2[:, 0]
    x_t = torch.full((n, ), 18432, dtype=torch.int64, device=DEV)
    out = torch.full((n, ), -1, dtype=torch.int64, device=DEV)
    ops.sample_head(hidden.to(DEV), g.to(DEV), b.to(DEV), w.to(DEV), expo.to(DEV),
                    changes.to(torch.uint8).to(DEV), tex.to(DEV), head, 1.0, x_t, out)
    sel = changes & (tex == head)
    assert sel.sum() > 10
    got, xt = out.cpu(), x_t.cpu()
    assert (got[~sel] == -1).all() and (xt[~sel] == 18432).all()
    mism = sel & (got != ref)
    assert (margin[mism] < 1e-5).all(), f'{int(mism.sum())} mismatches beyond near-tie margin'
    assert torch.equal(xt[sel & ~mism], ref[sel & ~mism] + 1024 * head)


def test_sample_heads_one_launch_equals_per_head_launches():
    n, C, K, H = 512, 512, 1024, 18
    hidden, g, b = rnd(n, C, seed=45) * 2, rnd(C, seed=46) * 0.1 + 1, rnd(C, seed=47) * 0.1
    w = rnd(H, K, C, seed=48, scale=0.15)
    gen = torch.Generator().manual_seed(49)
    tex = torch.randint(0, H, (n, ), generator=gen)
    tex[tex == 11] = 3                                   # head 11 inactive
    changes = (torch.rand(n, generator=gen) < 0.2)
    active = sorted(set(tex[changes].tolist()))
    expo = {h: torch.empty(n, K).exponential_(1.0, generator=gen).to(DEV) for h in active}
    dv = lambda t: t.to(DEV)
    hidden, g, b, w, texd, chd = dv(hidden), dv(g), dv(b), dv(w), dv(tex), dv(changes.to(torch.uint8))
    x_a = torch.full((n, ), 18432, dtype=torch.int64, device=DEV)
    out_a = torch.full((H, n), -1, dtype=torch.int64, device=DEV)
    for h in active:
        ops.sample_head(hidden, g, b, w[h], expo[h], chd, texd, h, 1.0, x_a, out_a[h])
    x_b, out_b = torch.full_like(x_a, 18432), torch.full_like(out_a, -1)
    rows = torch.nonzero(changes).flatten().flip(0).to(torch.int32).to(DEV)   # any order
    ops.sample_heads(hidden, g, b, w, expo, rows, rows.numel(), texd, 1.0, x_b, out_b, split=False)
    assert (out_a >= 0).sum() == int(changes.sum())
    assert torch.equal(x_a, x_b) and torch.equal(out_a, out_b)
    # the default two-launch form (8 workgroups per row + race kernel) gives the same bits
    x_c, out_c = torch.full_like(x_a, 18432), torch.full_like(out_a, -1)
    ops.sample_heads(hidden, g, b, w, expo, rows, rows.numel(), texd, 0.7, x_c, out_c)
    x_d, out_d = torch.full_like(x_a, 18432), torch.full_like(out_a, -1)
    ops.sample_heads(hidden, g, b, w, expo, rows, rows.numel(), texd, 0.7, x_d, out_d, split=False)
    assert torch.equal(x_c, x_d) and torch.equal(out_c, out_d)
    ops.sample_heads(hidden, g, b, w, expo, rows, rows.numel(), texd, 1.0, x_c, out_c)
    assert torch.equal(x_a, x_c) and torch.equal(out_a, out_c)


@pytest.mark.parametrize('n', [4096, 512, 1536, 3 * 512 + 7])
def test_philox_exponential_is_torchs_draw_bit_for_bit(n):
    """t2h_philox_exponential_f32 == torch.empty(n, 1024).exponential_() on the device generator: every
    element, several generator offsets, and the generator's own offset bookkeeping."""
    K = 1024
    torch.cuda.manual_seed_all(20210924 + n)
    cur = torch.cuda.current_device()
    gen = torch.cuda.default_generators[cur]
    torch.rand(8, 512, device=DEV)                       # move the offset off zero like the sampler does
    for _ in range(3):
        seed, off = gen.initial_seed(), gen.get_offset()
        ref = torch.empty(n, K, device=DEV).exponential_(1.0)
        _, inc = ops.torch_draw_geometry(n * K)
        assert gen.get_offset() == off + inc
        got = ops.philox_exponential(seed, off, n * K, DEV).view(n, K)
        assert torch.equal(got, ref), f'{int((got != ref).sum())} of {n * K} elements differ'


def test_sample_heads_philox_mode_equals_explicit_draws():
    n, C, K, H = 4096, 512, 1024, 18
    hidden, g, b = rnd(n, C, seed=45) * 2, rnd(C, seed=46) * 0.1 + 1, rnd(C, seed=47) * 0.1
    w = rnd(H, K, C, seed=48, scale=0.15)
    gen_c = torch.Generator().manual_seed(50)
    tex = torch.randint(0, H, (n, ), generator=gen_c)
    rows = torch.randperm(n, generator=gen_c)[:40].to(torch.int32).to(DEV)
    active = sorted(set(tex[rows.cpu().long()].tolist()))
    dv = lambda t: t.to(DEV)
    hidden, g, b, w, texd = dv(hidden), dv(g), dv(b), dv(w), dv(tex)
    from text2human_amd import engine
    noise = engine.TorchDeviceNoise(DEV)
    torch.cuda.manual_seed_all(77)
    expo = {h: noise.exponential(1, h, (n, K)) for h in active}
    x_a, out_a = torch.full((n, ), 18432, dtype=torch.int64, device=DEV), torch.full((H, n), -1, dtype=torch.int64, device=DEV)
    ops.sample_heads(hidden, g, b, w, expo, rows, rows.numel(), texd, 1.0, x_a, out_a)
    end_off = noise.generator()[0].get_offset()
    torch.cuda.manual_seed_all(77)
    gen = noise.generator()[0]
    _, inc = ops.torch_draw_geometry(n * K)
    philox = (gen.initial_seed(), {h: gen.get_offset() + i * inc for i, h in enumerate(active)})
    assert gen.get_offset() + len(active) * inc == end_off         # the draws moved the generator exactly that far
    x_b, out_b = torch.full_like(x_a, 18432), torch.full_like(out_a, -1)
    ops.sample_heads(hidden, g, b, w, {}, rows, rows.numel(), texd, 1.0, x_b, out_b, philox=philox)
    assert torch.equal(x_a, x_b) and torch.equal(out_a, out_b)
    # per-listed-row noise (row lists that mix sampling steps): offsets per row, explicit compact rows
    offs = torch.tensor([philox[1][int(tex[r])] for r in rows.cpu().tolist()], dtype=torch.int64, device=DEV)
    x_c, out_c = torch.full_like(x_a, 18432), torch.full_like(out_a, -1)
    ops.sample_heads(hidden, g, b, w, {}, rows, rows.numel(), texd, 1.0, x_c, out_c, row_noise=('philox', philox[0], offs))
    assert torch.equal(x_a, x_c) and torch.equal(out_a, out_c)
    perm = torch.randperm(rows.numel(), generator=gen_c)
    erows = torch.empty(rows.numel(), K, device=DEV)
    for i, r in enumerate(rows.cpu().tolist()):
        erows[int(perm[i])] = expo[int(tex[r])][r]
    x_d, out_d = torch.full_like(x_a, 18432), torch.full_like(out_a, -1)
    ops.sample_heads(hidden, g, b, w, {}, rows, rows.numel(), texd, 1.0, x_d, out_d,
                     row_noise=('explicit', erows, perm.to(torch.int32).to(DEV)))
    assert torch.equal(x_a, x_d) and torch.equal(out_a, out_d)


@pytest.mark.parametrize('n', [4096, 16384, 512, 3 * 512 + 7])
def test_philox_uniform_is_torchs_rand_bit_for_bit(n):
    """t2h_philox_uniform_f32 == torch.rand(n) on the device generator, at several offsets."""
    torch.cuda.manual_seed_all(4242 + n)
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    torch.empty(1000, device=DEV).exponential_(1.0)        # move the offset off zero
    for _ in range(3):
        seed, off = gen.initial_seed(), gen.get_offset()
        ref = torch.rand(n // 512, 512, device=DEV) if n % 512 == 0 else torch.rand(n, device=DEV)
        _, inc = ops.torch_draw_geometry(n)
        assert gen.get_offset() == off + inc
        got = ops.philox_uniform(seed, off, n, DEV)
        assert torch.equal(got, ref.reshape(-1)), f'{int((got != ref.reshape(-1)).sum())} of {n} elements differ'
    assert float(got.min()) >= 0.0 and float(got.max()) < 1.0


@pytest.mark.parametrize('B,steps', [(8, 256), (3, 17), (32, 256), (1, 1)])
def test_unmask_schedule_replays_the_reference_loop(B, steps):
    """t2h_unmask_schedule (one launch, no transformer) == the reference's loop over real torch draws
    (models/sample_model.py:279-306): same step for every token, same active heads per step, and the
    generator ends where the reference's would."""
    from text2human_amd import engine, schedule
    T, H, K = 512, 18, 1024
    n = B * T
    gen_c = torch.Generator().manual_seed(B * 1000 + steps)
    tex = torch.randint(0, H, (B, T), generator=gen_c)
    tex[tex == 5] = 6                                      # a head that never samples
    texd = tex.to(DEV)
    torch.cuda.manual_seed_all(99)
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    # reference loop on real draws
    unmasked = torch.zeros(B, T, dtype=torch.bool, device=DEV)
    want_step = torch.zeros(n, dtype=torch.int32, device=DEV)
    want_mask = [0] * (steps + 1)
    for t in range(steps, 0, -1):
        ch = torch.rand((B, T), device=DEV) < 1 / torch.tensor(float(t), device=DEV)
        ch = torch.bitwise_xor(ch, torch.bitwise_and(ch, unmasked))
        unmasked |= ch
        want_step[ch.view(-1)] = t
        for h in range(H):
            if int((texd.view(-1)[ch.view(-1)] == h).sum()) > 0:
                want_mask[t] |= 1 << h
                torch.empty((n, K), device=DEV).exponential_(1.0)
    end_off = gen.get_offset()
    # the product path's schedule on a re-seeded generator
    torch.cuda.manual_seed_all(99)
    noise = engine.TorchDeviceNoise(DEV)
    assert noise.emulation_ok(n, K)
    sched = engine.build_schedule(texd, steps, H, K, noise, compact=True)
    assert gen.get_offset() == end_off
    step_dev, mask_dev, rand_inc, expo_inc = ops.unmask_schedule(gen.initial_seed(), 0, texd.view(-1).contiguous(), steps, H, K)
    assert torch.equal(step_dev, want_step)
    assert mask_dev.cpu().tolist()[1:] == want_mask[1:]
    # rounds: every row once, each sample walks its own steps in descending order, offsets of the row's head
    rows = sched.rows.cpu().long()
    assert torch.equal(rows.sort().values, torch.arange(n))
    _, expo_off, final = schedule.draw_offsets(want_mask, steps, 0, rand_inc, expo_inc, H)
    assert final == end_off
    ws = want_step.cpu()
    for r in range(sched.n_rounds):
        rr = rows[int(sched.start[r]):int(sched.start[r + 1])]
        for b in range(B):
            mine = rr[(rr // T) == b]
            t_b = int(sched.round_steps[r, b])
            assert (ws[mine] == t_b).all() if t_b else mine.numel() == 0
    offs = sched.offsets.cpu()
    assert torch.equal(offs, torch.from_numpy(expo_off)[ws[rows].long(), tex.view(-1)[rows]])
    assert sched.n_rounds <= steps and (sched.round_steps > 0).sum() == sum(len(set(ws.view(B, T)[b].tolist())) for b in range(B))


# ------------------------------------------------------------------ quantizer pieces


def test_vq_l2_argmin_first_min_and_margin():
    n, n_e, d = 1000, 1024, 32
    z, cb = rnd(n, d, seed=43), rnd(n_e, d, seed=44)
    cb[77] = cb[5]  # exact duplicate: the first minimum (index 5) must win
    z[0] = cb[5] + 1e-3
    dist = (z.double()**2).sum(1, keepdim=True) + (cb.double()**2).sum(1) - 2 * z.double() @ cb.double().t()
    ref = dist.argmin(1)
    got = ops.vq_l2_argmin(z.to(DEV), cb.to(DEV)).cpu()
    assert got[0].item() == 5
    s = dist.sort(1).values
    mism = got != ref
    assert ((s[:, 1] - s[:, 0])[mism] < 1e-4).all()
    assert mism.float().mean() < 0.01
    assert ops.vq_l2_argmin(z[:7].contiguous().to(DEV), cb.to(DEV)).cpu().tolist() == got[:7].tolist()


def test_codebook_gathers():
    g = torch.Generator().manual_seed(45)
    B, h, w = 2, 32, 16
    n = B * h * w
    tex = torch.randint(0, 18, (n, ), generator=g)
    top_books, bot_books = rnd(18, 1024, 256, seed=46), rnd(18, 512, 1024, seed=47)
    top_idx = torch.randint(0, 1024, (18, n), generator=g)
    bot_idx = torch.randint(0, 512, (18, n), generator=g)
    got = ops.codebook_gather_tex(top_idx.to(DEV), tex.to(DEV), top_books.to(DEV)).cpu()
    ref = torch.stack([top_books[tex[i], top_idx[tex[i], i]] for i in range(n)])
    assert torch.equal(got, ref)
    got = ops.codebook_gather_fold(bot_idx.to(DEV), tex.to(DEV), bot_books.to(DEV), B, h, w).cpu()
    zq = torch.stack([bot_books[tex[i], bot_idx[tex[i], i]] for i in range(n)])
    ref = F.fold(zq.view(B, h * w, 1024).permute(0, 2, 1), (2 * h, 2 * w), kernel_size=2, stride=2)
    assert torch.equal(got.view(B, 2 * h, 2 * w, 256).permute(0, 3, 1, 2), ref)


def test_routed_head_argmax():
    g = torch.Generator().manual_seed(48)
    n, nh, cf, nc = 700, 18, 64, 512
    feat, w, b = rnd(n, nh * cf, seed=49), rnd(nh, nc, cf, seed=50), rnd(nh, nc, seed=51)
    tex = torch.randint(0, nh, (n, ), generator=g)
    got = ops.routed_head_argmax(feat.to(DEV), w.to(DEV), b.to(DEV), tex.to(DEV), nh, cf, nc).cpu()
    for i in range(0, n, 37):
        t = tex[i].item()
        lg = w[t].double() @ feat[i, t * cf:(t + 1) * cf].double() + b[t].double()
        assert got[t, i].item() == lg.argmax().item()
        assert (got[:, i] == -1).sum().item() == nh - 1


# ------------------------------------------------------------------ layout / misc


def test_layout_pool_resample_epilogue():
    B, C, H, W = 2, 64, 16, 8
    x = rnd(B, C, H, W, seed=52)
    rows = ops.nchw_to_nhwc(x.to(DEV))
    assert torch.equal(rows.cpu().view(B, H, W, C).permute(0, 3, 1, 2), x)
    assert torch.equal(ops.nhwc_to_nchw(rows, B, H, W).cpu(), x)
    mp = ops.maxpool2(rows, B, H, W).cpu().view(B, H // 2, W // 2, C).permute(0, 3, 1, 2)
    assert torch.equal(mp, F.max_pool2d(x, 2))
    up = ops.bilinear_up2(rows, B, H, W).cpu().view(B, 2 * H, 2 * W, C).permute(0, 3, 1, 2)
    assert_close(up, F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False),
                 rtol=1e-6, atol=1e-6)
    am = ops.argmax_rows(rows).cpu()
    assert torch.equal(am, rows.cpu().argmax(1))
    seg = torch.randint(0, 24, (B * H * W, ), generator=torch.Generator().manual_seed(53)).float()
    oh = ops.onehot_nhwc(seg.to(DEV), 24, 32).cpu()
    assert torch.equal(oh[:, :24], F.one_hot(seg.long(), 24).float()) and oh[:, 24:].abs().sum() == 0
    dec = rnd(B * H * W, 3, seed=54)
    img, u8 = ops.image_epilogue(dec.to(DEV), B, H, W, want_u8=True)
    ref = ((dec.view(B, H, W, 3).permute(0, 3, 1, 2) + 1) / 2).clamp(0, 1)
    assert torch.equal(img.cpu(), ref)
    assert torch.equal(u8.cpu(), ref.mul(255).add(0.5).clamp(0, 255).permute(0, 2, 3, 1).to(torch.uint8))


def test_texture_map():
    from text2human_amd import synthetic
    g = torch.Generator().manual_seed(55)
    segm = torch.randint(0, 24, (3, 1, 32, 16), generator=g)
    up, lo, ou = torch.tensor([3, 17, 0]), torch.tensor([17, 5, 2]), torch.tensor([1, 1, 17])
    got = ops.texture_map(segm.to(DEV), up.to(DEV), lo.to(DEV), ou.to(DEV)).cpu()
    assert torch.equal(got, synthetic.texture_mask_from_segm(segm.float(), up, lo, ou))


def test_error_reporting():
    from text2human_amd._lib import T2HError
    a, w = torch.zeros(64, 48, device=DEV), torch.zeros(32, 48, device=DEV)
    with pytest.raises(T2HError, match='multiple of 32'):
        ops.gemm(a, w)


# ------------------------------------------------------------------ decoder AttnBlock, flash-style


@pytest.mark.parametrize('n_img,N,C', [(2, 512, 512), (3, 2048, 512), (1, 8192, 512), (2, 512, 256), (1, 96, 256)])
def test_spatial_attention_matches_fp64_and_the_materialised_form(n_img, N, C):
    """t2h_spatial_attention_f32 (no N x N tensor) vs an fp64 reference of vqgan_arch.py:645-656, next to
    the materialised bmm -> softmax -> bmm form the encoders keep; a spiked key forces the online-softmax
    rescale."""
    qkv = rnd(n_img * N, 3 * C, seed=60 + N % 7) * 0.7
    qkv[5, C:2 * C] *= 5.0                                  # one key far above the rest
    qkv[N // 2 + 3, C:2 * C] *= -4.0
    q, k, v = [t.reshape(n_img, N, C).double() for t in qkv.split(C, dim=1)]
    ref = (torch.softmax(q @ k.transpose(1, 2) * float(int(C)**(-0.5)), -1) @ v).reshape(n_img * N, C)
    d = qkv.to(DEV)
    got = ops.spatial_attention(d, n_img, N, C).cpu().double()
    q3 = d.view(n_img, N, 3 * C)
    s = torch.empty((n_img, N, N), device=DEV)
    ops.bgemm(q3[:, :, :C], q3[:, :, C:2 * C], s, alpha=float(int(C)**(-0.5)))
    ops.softmax_rows_(s)
    mat = torch.empty((n_img, N, C), device=DEV)
    ops.bgemm(s, q3[:, :, 2 * C:], mat, b_trans=True)
    e_new, e_old = (got - ref).abs().max().item(), (mat.cpu().double().view(-1, C) - ref).abs().max().item()
    assert e_new < 2e-5 + 3 * e_old, (e_new, e_old)
    # strided output (a column block of a wider buffer)
    wide = torch.zeros(n_img * N, C + 64, device=DEV)
    ops.spatial_attention(d, n_img, N, C, out=wide[:, :C])
    assert torch.equal(wide[:, :C].cpu().double(), got) and (wide[:, C:] == 0).all()


@pytest.mark.parametrize('n_img,H,W,cin,cout,pro', [(2, 32, 64, 128, 3, True), (1, 13, 37, 64, 3, True), (3, 8, 32, 32, 4, False),
                                                   (1, 24, 40, 128, 1, True)])
def test_conv3x3_small_matches_fp64(n_img, H, W, cin, cout, pro):
    """t2h_conv3x3_small_f32 (conv_out: a few output channels, GroupNorm-apply + swish prologue, zero padding
    of the ACTIVATED tensor) vs an fp64 reference and next to the matrix-core kernel it replaces."""
    x = rnd(n_img, cin, H, W, seed=70 + H) * 1.5
    w = rnd(cout, cin, 3, 3, seed=71, scale=0.08)
    b = rnd(cout, seed=72)
    sc, sh = rnd(n_img, cin, seed=73) * 0.3 + 1.0, rnd(n_img, cin, seed=74) * 0.5
    a = x.double()
    if pro:
        a = a * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]
        a = a * torch.sigmoid(a)
    ref = F.conv2d(a, w.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    rows = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().to(DEV)
    wp = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(DEV)      # [Cout][tap][cin]
    p = (sc.to(DEV), sh.to(DEV), ops.PRO_SWISH) if pro else None
    got = ops.conv3x3_small(rows, wp, n_img, H, W, cin, bias=b.to(DEV), pro=p)
    old = ops.conv3x3(rows, wp, n_img, H, W, cin, bias=b.to(DEV), pro=p)
    e_new, e_old = (got.cpu().double() - ref).abs().max().item(), (old.cpu().double() - ref).abs().max().item()
    assert e_new < 2e-5 + 3 * e_old, (e_new, e_old)


@pytest.mark.parametrize('C', [96, 192, 384, 128])
def test_groupnorm_finalize_from_epilogue_partials_any_channels_per_group(C):
    """t2h_groupnorm_finalize_f32 (statistics from a producing conv's per-128-row partial sums) for channels-per-group
    counts that do NOT divide 256 (cpg = 3, 6, 12: a checkpoint with ch = 96) as well as one that does, against
    torch's GroupNorm folded into (scale, shift) tables.  (t2h_groupnorm_tables_f32, the pass over the tensor itself,
    serves C / 4 | 256 only: ADVICE r04.)"""
    n_img, hw, groups, eps = 2, 512, 32, 1e-6
    g = torch.Generator().manual_seed(C)
    x = torch.randn(n_img * hw, C, generator=g) * 1.7 + 0.4
    gamma, beta = torch.randn(C, generator=g) * 0.3 + 1.0, torch.randn(C, generator=g) * 0.2
    xc = x.view(n_img, hw // 128, 128, C).double()
    part = torch.stack([xc.sum(2), (xc * xc).sum(2)], 2).contiguous()          # [n_img, chunks, 2, C] fp64
    xd = x.to(DEV)
    xd._t2h_gn_part = part.to(DEV)
    scale, shift = ops.groupnorm_tables(xd, gamma.to(DEV), beta.to(DEV), n_img, hw, groups=groups, eps=eps)
    xg = x.view(n_img, hw, groups, C // groups).double()
    mean = xg.mean((1, 3), keepdim=True)
    var = xg.var((1, 3), unbiased=False, keepdim=True)
    rstd = (1.0 / torch.sqrt(var + eps)).expand(n_img, 1, groups, C // groups).reshape(n_img, C)
    mean = mean.expand(n_img, 1, groups, C // groups).reshape(n_img, C)
    want_scale = gamma.double() * rstd
    want_shift = beta.double() - mean * want_scale
    assert (scale.cpu().double() - want_scale).abs().max().item() < 1e-5
    assert (shift.cpu().double() - want_shift).abs().max().item() < 1e-5
    # applied: the normalised tensor equals torch's GroupNorm
    ref = torch.nn.functional.group_norm(x.view(n_img, hw, C).permute(0, 2, 1), groups, gamma, beta, eps)
    got = x.view(n_img, hw, C) * scale.cpu().unsqueeze(1) + shift.cpu().unsqueeze(1)
    assert (got.permute(0, 2, 1) - ref).abs().max().item() < 2e-5
