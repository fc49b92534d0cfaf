"""t2h_conv_split_f32 (-m gpu): the decoders' stride-1 convolutions on the fp16 matrix cores
(2 x fp16 planes, three products) against an fp64 reference, with the accuracy bar of the
exact-fp32 kernel (tests/test_gpu_kernels.py: test_conv3x3), and end to end: decode with split
convolutions vs the golden image of the unmodified reference."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from text2human_amd import ops, weights

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def assert_close(got, ref, rtol=2e-5, atol=2e-5, what=''):
    err = (got.cpu().double() - ref.double()).abs()
    bound = atol + rtol * ref.double().abs()
    assert (err <= bound).all(), f'{what}: max err {err.max().item():.3e}'


@pytest.fixture(params=[128, 256])
def tile(request):
    """both tile shapes of the kernel (the host picks by problem size otherwise)"""
    from text2human_amd import _lib
    lib = _lib.load()
    lib.t2h_conv_split_force_tile(request.param)
    yield request.param
    lib.t2h_conv_split_force_tile(0)


@pytest.mark.parametrize('mode', ['same', 'up'])
@pytest.mark.parametrize('cin,cout,h,w', [(64, 128, 16, 16), (128, 96, 32, 16), (512, 512, 16, 16), (256, 128, 8, 32),
                                          (96, 128, 16, 8)])
def test_conv3x3_split(mode, cin, cout, h, w, tile):
    n_img = 2
    if tile == 256 and mode == 'same' and (h * w) % 256:
        pytest.skip('256-pixel tiles need 256 | pixels per image')
    x = rnd(n_img, cin, h, w, seed=13)
    wt, b = rnd(cout, cin, 3, 3, seed=14, scale=0.1), rnd(cout, seed=15)
    sc, sh = rnd(n_img, cin, seed=16) * 0.3 + 1, rnd(n_img, cin, seed=17) * 0.3
    wp = weights.pack_conv3x3(wt).to(DEV)
    ws = ops.split_rows(wp)
    rows = ops.nchw_to_nhwc(x.to(DEV))
    for use_pro in (False, True):
        xin = x.double()
        if use_pro:
            xin = xin * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]
            xin = xin * torch.sigmoid(xin)
        # the elementwise pass is bitwise the split of the fp32 kernel's operand prologue
        xs = (ops.gn_apply_split(rows, sc.to(DEV), sh.to(DEV), rows_per_img=h * w, act=ops.PRO_SWISH) if use_pro
              else ops.gn_apply_split(rows))
        got = ops.unsplit_rows_host(xs, n_img * h * w, cin).double()
        want = xin.permute(0, 2, 3, 1).reshape(-1, cin)
        assert ((got - want).abs() <= 2e-6 + 2e-6 * want.abs()).all()
        if mode == 'up':
            xin = F.interpolate(xin, scale_factor=2.0, mode='nearest')
        ref = F.conv2d(xin, wt.double(), b.double(), 1, 1)
        ho, wo = ref.shape[2:]
        assert ops.conv_split_ok(ho * wo, mode)
        res = rnd(n_img * ho * wo, cout, seed=18)
        out = ops.conv_split(xs, ws, n_img, h, w, cin, cout, bias=b.to(DEV), residual=res.to(DEV), mode=mode,
                             gn_stats=True)
        ref_rows = ref.permute(0, 2, 3, 1).reshape(-1, cout) + res.double()
        assert_close(out, ref_rows, what=f'{mode} pro={use_pro}')
        # GroupNorm tables from the epilogue's partial sums == tables from a pass over the tensor
        gam, bet = rnd(cout, seed=30).to(DEV), rnd(cout, seed=31).to(DEV)
        if cout % 128 == 0:                                      # (channel counts the plain pass serves)
            sc_e, sh_e = ops.groupnorm_tables(out, gam, bet, n_img, ho * wo)
            plain = out.clone()                                  # (a clone carries no partials)
            assert not hasattr(plain, '_t2h_gn_part') and hasattr(out, '_t2h_gn_part')
            sc_p, sh_p = ops.groupnorm_tables(plain, gam, bet, n_img, ho * wo)
            assert (sc_e - sc_p).abs().max().item() < 1e-6 and (sh_e - sh_p).abs().max().item() < 1e-6
        # and no worse than twice the exact-fp32 kernel's own distance from fp64 (one pass over K: with the K slices the
        # small geometries get since round 6 its blocked sum is closer to fp64 than any single fp32 accumulator)
        o32 = ops.conv3x3(rows, wp, n_img, h, w, cin, bias=b.to(DEV), residual=res.to(DEV), mode=mode,
                          pro=(sc.to(DEV), sh.to(DEV), ops.PRO_SWISH) if use_pro else None, ksplit=1)
        e32 = (o32.cpu().double() - ref_rows).abs().max().item()
        es = (out.cpu().double() - ref_rows).abs().max().item()
        assert es <= 2 * e32 + 1e-6, (es, e32)


def test_conv1x1_split_with_and_without_groupnorm(tile):
    n_img, hw, C, N = 3, 512, 256, 768
    x, w, b = rnd(n_img * hw, C, seed=8), rnd(N, C, seed=9, scale=0.1), rnd(N, seed=12)
    sc, sh = rnd(n_img, C, seed=10) * 0.5 + 1, rnd(n_img, C, seed=11)
    ws = ops.split_rows(w.to(DEV))
    xa = (x.view(n_img, hw, C) * sc[:, None] + sh[:, None]).double().view(-1, C)
    xs = ops.gn_apply_split(x.to(DEV), sc.to(DEV), sh.to(DEV), rows_per_img=hw)
    out = ops.conv_split(xs, ws, n_img, hw, 1, C, N, taps=1, bias=b.to(DEV))
    assert_close(out, xa @ w.double().t() + b.double(), what='GN apply')
    res = rnd(n_img * hw, N, seed=13)
    out = ops.conv_split(ops.gn_apply_split(x.to(DEV)), ws, n_img, hw, 1, C, N, taps=1, bias=b.to(DEV),
                         residual=res.to(DEV))
    assert_close(out, x.double() @ w.double().t() + b.double() + res.double(), what='plain + residual')


def test_conv_split_rejects_what_it_does_not_serve():
    from text2human_amd import _lib
    x, w = rnd(2 * 96, 64, seed=1).to(DEV), rnd(64, 64, seed=2).to(DEV)
    with pytest.raises(_lib.T2HError, match='multiple of 128'):
        ops.conv_split(ops.gn_apply_split(x), ops.split_rows(w), 2, 96, 1, 64, 64, taps=1)
    assert not ops.conv_split_ok(96) and not ops.conv_split_ok(512, 'down')
    ops.split_overflow(reset=True)
    ops.gn_apply_split(x * 1e5)
    assert ops.split_overflow(reset=True)


def test_decode_with_split_convs_vs_reference_golden():
    """Same check as tests/test_gpu_path.py::test_decode_vs_golden, with the decoders' convolutions
    on the split-precision kernel (the default) -- and next to the exact-fp32 path's error."""
    from oracle.make_golden import WEIGHT_SEED, golden_inputs
    from text2human_amd import defaults, options, synthetic
    from text2human_amd.models import SampleFromParsingModel
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'decode_b1.npz'))
    gi = golden_inputs('decoder')
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    sds = synthetic.make_state_dicts(opt, seed=WEIGHT_SEED)
    errs = {}
    for name, env in (('split', '1'), ('fp32', '0')):
        os.environ['T2H_SPLIT_CONV'] = env
        try:
            model = SampleFromParsingModel(opt, state_dicts=sds)
        finally:
            os.environ.pop('T2H_SPLIT_CONV')
        assert ('dec.conv_in.ws' in model.P.t) == (name == 'split')
        zb = ops.nchw_to_nhwc(gi['zb'].to(DEV))
        z = ops.nchw_to_nhwc(gi['z'].to(DEV))
        bot_h = model.bot_decoder_res.decode_res(zb, 1, 64, 32)
        dec, ho, wo = model.decoder.decode(z, 1, 32, 16, bot_h=bot_h)
        out = ops.nhwc_to_nchw(dec, 1, 512, 256)
        errs[name] = float((out[0, :, ::4, ::4].cpu().double() - torch.as_tensor(g['dec_sample']).double()).abs().max())
        assert not ops.split_overflow(reset=True)
    assert errs['split'] < 2e-4 and errs['fp32'] < 2e-4, errs
    assert errs['split'] < 3 * errs['fp32'] + 1e-6, errs
