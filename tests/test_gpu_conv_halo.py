"""t2h_conv_halo_f32 (-m gpu): the decoders' 3x3 convolutions with GroupNorm apply + swish + split folded into the
staging of an 18 x 18-pixel halo (csrc/conv_halo.hip), against an fp64 reference with the accuracy bar of the
exact-fp32 kernel, against the two-kernel path it replaces (t2h_gn_apply_split_f32 + t2h_conv_split_f32), and end to
end: decode vs the golden image of the unmodified reference with the kernel forced onto every level it serves."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from text2human_amd import _lib, ops, weights

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(params=[1, 0], ids=['two-fragment-sets', 'first-version'])
def variant(request):
    """both main loops of the kernel (the library's default is 1)"""
    lib = _lib.load()
    lib.t2h_conv_halo_force_variant(request.param)
    yield request.param
    lib.t2h_conv_halo_force_variant(1)


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def reference(x, wt, b, sc, sh, mode, use_pro):
    xin = x.double()
    if use_pro:
        xin = xin * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]
        xin = xin * torch.sigmoid(xin)
    if mode == 'up':
        xin = F.interpolate(xin, scale_factor=2.0, mode='nearest')
    ref = F.conv2d(xin, wt.double(), b.double(), 1, 1)
    return ref.permute(0, 2, 3, 1).reshape(-1, wt.shape[0]), ref.shape[2], ref.shape[3]


def describe_errors(got, ref, n_img, ho, wo):
    """where the largest errors sit (image, y, x, channel): a wrong halo / tap / border shows as a pattern"""
    err = (got.cpu().double() - ref).abs()
    bound = 2e-5 + 2e-5 * ref.abs()
    bad = (err > bound).nonzero()
    msg = [f'max err {err.max().item():.3e}, {bad.shape[0]} of {err.numel()} elements beyond the bound']
    for r, c in bad[:12].tolist():
        i, rem = divmod(r, ho * wo)
        msg.append(f'  img {i} y {rem // wo} x {rem % wo} ch {c}: got {got[r, c].item():.6f} want {ref[r, c].item():.6f}')
    return '\n'.join(msg)


@pytest.mark.parametrize('mode', ['same', 'up'])
@pytest.mark.parametrize('cin,cout,h,w,n_img', [(32, 128, 16, 16, 2),     # one tile per image, one channel group
                                               (64, 128, 32, 16, 2),     # two groups, tiles stacked in y
                                               (128, 96, 16, 48, 3),     # N not a multiple of the column tile
                                               (96, 256, 32, 32, 1),     # three groups (odd), two column tiles
                                               (256, 128, 16, 32, 2)])   # eight groups
def test_conv_halo_vs_fp64_and_vs_the_two_kernel_path(mode, cin, cout, h, w, n_img, variant):
    x = rnd(n_img, cin, h, w, seed=13)
    wt, b = rnd(cout, cin, 3, 3, seed=14, scale=0.1), rnd(cout, seed=15)
    sc, sh = rnd(n_img, cin, seed=16) * 0.3 + 1, rnd(n_img, cin, seed=17) * 0.3
    wp = weights.pack_conv3x3(wt).to(DEV)
    ws = ops.split_rows(wp)
    rows = ops.nchw_to_nhwc(x.to(DEV))
    for use_pro in (False, True):
        ref, ho, wo = reference(x, wt, b, sc, sh, mode, use_pro)
        res = rnd(n_img * ho * wo, cout, seed=18)
        ref = ref + res.double()
        pro = (sc.to(DEV), sh.to(DEV)) if use_pro else None
        ops.split_overflow(reset=True)
        out = ops.conv_halo(rows, ws, n_img, h, w, cin, cout, bias=b.to(DEV), residual=res.to(DEV), mode=mode, pro=pro,
                            gn_stats=True)
        assert not ops.split_overflow(reset=True)
        err = (out.cpu().double() - ref).abs()
        assert (err <= 2e-5 + 2e-5 * ref.abs()).all(), f'{mode} pro={use_pro}:\n' + describe_errors(out.cpu(), ref, n_img, ho, wo)
        # the path it replaces: same operands, K summed in another order
        xs = (ops.gn_apply_split(rows, sc.to(DEV), sh.to(DEV), rows_per_img=h * w, act=ops.PRO_SWISH) if use_pro
              else ops.gn_apply_split(rows))
        old = ops.conv_split(xs, ws, n_img, h, w, cin, cout, bias=b.to(DEV), residual=res.to(DEV), mode=mode,
                             gn_stats=True)
        assert (out - old).abs().max().item() < 1e-5 * max(1.0, float(ref.abs().max()))
        # no worse than twice the exact-fp32 kernel's own distance from fp64
        o32 = ops.conv3x3(rows, wp, n_img, h, w, cin, bias=b.to(DEV), residual=res.to(DEV), mode=mode,
                          pro=(sc.to(DEV), sh.to(DEV), ops.PRO_SWISH) if use_pro else None)
        e32 = (o32.cpu().double() - ref).abs().max().item()
        assert err.max().item() <= 2 * e32 + 1e-6, (err.max().item(), e32)
        # GroupNorm tables from the epilogue's partial sums == tables from a pass over the tensor
        if cout % 128 == 0:
            gam, bet = rnd(cout, seed=30).to(DEV), rnd(cout, seed=31).to(DEV)
            assert hasattr(out, '_t2h_gn_part')
            sc_e, sh_e = ops.groupnorm_tables(out, gam, bet, n_img, ho * wo)
            sc_p, sh_p = ops.groupnorm_tables(out.clone(), gam, bet, n_img, ho * wo)
            assert (sc_e - sc_p).abs().max().item() < 1e-6 and (sh_e - sh_p).abs().max().item() < 1e-6


def test_conv_halo_input_with_a_row_stride_and_no_bias_or_residual():
    """x rows wider than Cin (a channel slice of a concatenated tensor), bias / residual / partials absent"""
    n_img, cin, cout, h, w = 2, 64, 128, 16, 16
    wide = rnd(n_img * h * w, cin + 32, seed=3).to(DEV)
    wt = rnd(cout, cin, 3, 3, seed=4, scale=0.1)
    ws = ops.split_rows(weights.pack_conv3x3(wt).to(DEV))
    out = ops.conv_halo(wide, ws, n_img, h, w, cin, cout)
    x = wide[:, :cin].cpu().view(n_img, h, w, cin).permute(0, 3, 1, 2)
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, cout)
    assert ((out.cpu().double() - ref).abs() <= 2e-5 + 2e-5 * ref.abs()).all()
    assert not hasattr(out, '_t2h_gn_part')


def test_conv_halo_overflow_word_and_rejections():
    n_img, cin, cout, h, w = 1, 32, 128, 16, 16
    x = rnd(n_img * h * w, cin, seed=5).to(DEV)
    ws = ops.split_rows(weights.pack_conv3x3(rnd(cout, cin, 3, 3, seed=6)).to(DEV))
    ops.split_overflow(reset=True)
    ops.conv_halo(x * 1e5, ws, n_img, h, w, cin, cout)
    assert ops.split_overflow(reset=True)
    ops.conv_halo(x, ws, n_img, h, w, cin, cout)
    assert not ops.split_overflow(reset=True)
    with pytest.raises(_lib.T2HError, match='multiples of 16'):
        ops.conv_halo(rnd(24 * 16, cin, seed=7).to(DEV), ws, 1, 24, 16, cin, cout)
    assert not ops.conv_halo_ok(8, 24, 16, 32, 128) and not ops.conv_halo_ok(8, 512, 256, 128, 128, 'down')
    assert ops.conv_halo_ok(8, 512, 256, 128, 128) and not ops.conv_halo_ok(1, 32, 16, 512, 512)
    # the choice between the two kernels never depends on the batch (an image must not depend on its neighbours)
    for shape in ((512, 256, 128, 128), (128, 64, 256, 256), (64, 32, 512, 512), (64, 32, 256, 256), (32, 16, 512, 512)):
        assert len({ops.conv_halo_ok(n, *shape) for n in (1, 2, 3, 8, 32)}) == 1, shape


@pytest.mark.parametrize('knob', ['2', '0'])
def test_decode_vs_reference_golden_with_the_halo_kernel_on_every_level_it_serves(knob, monkeypatch):
    """tests/test_gpu_conv_split.py's decode check with T2H_HALO_CONV=2 (every 3x3 convolution whose height and
    width are multiples of 16, whatever the grid) and =0 (the two-kernel path everywhere)."""
    from oracle.make_golden import WEIGHT_SEED, golden_inputs
    from text2human_amd import defaults, options, synthetic
    from text2human_amd.models import SampleFromParsingModel
    monkeypatch.setenv('T2H_HALO_CONV', knob)
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'decode_b1.npz'))
    gi = golden_inputs('decoder')
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    sds = synthetic.make_state_dicts(opt, seed=WEIGHT_SEED)
    model = SampleFromParsingModel(opt, state_dicts=sds)
    calls = []
    real = ops.conv_halo
    monkeypatch.setattr(ops, 'conv_halo', lambda *a, **k: (calls.append(a[3:7]), real(*a, **k))[1])
    zb = ops.nchw_to_nhwc(gi['zb'].to(DEV))
    z = ops.nchw_to_nhwc(gi['z'].to(DEV))
    bot_h = model.bot_decoder_res.decode_res(zb, 1, 64, 32)
    dec, ho, wo = model.decoder.decode(z, 1, 32, 16, bot_h=bot_h)
    out = ops.nhwc_to_nchw(dec, 1, 512, 256)
    err = float((out[0, :, ::4, ::4].cpu().double() - torch.as_tensor(g['dec_sample']).double()).abs().max())
    assert not ops.split_overflow(reset=True)
    assert err < 2e-4, err
    assert (len(calls) > 20) if knob == '2' else (len(calls) == 0), len(calls)
