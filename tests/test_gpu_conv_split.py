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


@pytest.mark.parametrize('mode', ['same', 'up'])
@pytest.mark.parametrize('cin,cout,h,w', [(64, 128, 16, 8), (128, 96, 32, 16), (512, 512, 16, 8), (256, 128, 8, 16)])
def test_conv3x3_split(mode, cin, cout, h, w):
    n_img = 2
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
        if mode == 'up':
            xin = F.interpolate(xin, scale_factor=2.0, mode='nearest')
        ref = F.conv2d(xin, wt.double(), b.double(), 1, 1)
        ho, wo = ref.shape[2:]
        assert ops.conv_split_ok(ho * wo, mode)
        res = rnd(n_img * ho * wo, cout, seed=18)
        kw = dict(bias=b.to(DEV), residual=res.to(DEV), mode=mode,
                  pro=(sc.to(DEV), sh.to(DEV), ops.PRO_SWISH) if use_pro else None)
        out = ops.conv3x3(rows, wp, n_img, h, w, cin, w_split=ws, **kw)
        ref_rows = ref.permute(0, 2, 3, 1).reshape(-1, cout) + res.double()
        assert_close(out, ref_rows, what=f'{mode} pro={use_pro}')
        # and no worse than twice the exact-fp32 kernel's own distance from fp64
        e32 = (ops.conv3x3(rows, wp, n_img, h, w, cin, **kw).cpu().double() - ref_rows).abs().max().item()
        es = (out.cpu().double() - ref_rows).abs().max().item()
        assert es <= 2 * e32 + 1e-6, (es, e32)


def test_conv1x1_split_with_and_without_groupnorm_prologue():
    n_img, hw, C, N = 3, 512, 256, 768
    x, w, b = rnd(n_img * hw, C, seed=8), rnd(N, C, seed=9, scale=0.1), rnd(N, seed=12)
    sc, sh = rnd(n_img, C, seed=10) * 0.5 + 1, rnd(n_img, C, seed=11)
    ws = ops.split_rows(w.to(DEV))
    xa = (x.view(n_img, hw, C) * sc[:, None] + sh[:, None]).double().view(-1, C)
    out = ops.gemm(x.to(DEV), w.to(DEV), bias=b.to(DEV), pro=(sc.to(DEV), sh.to(DEV), hw, ops.PRO_NONE), w_split=ws)
    assert_close(out, xa @ w.double().t() + b.double(), what='GN prologue')
    res = rnd(n_img * hw, N, seed=13)
    out = ops.gemm(x.to(DEV), w.to(DEV), bias=b.to(DEV), residual=res.to(DEV), w_split=ws, rows_per_img=hw)
    assert_close(out, x.double() @ w.double().t() + b.double() + res.double(), what='plain + residual')


def test_conv_split_rejects_what_it_does_not_serve():
    from text2human_amd import _lib
    x, w = rnd(2 * 96, 64, seed=1).to(DEV), rnd(64, 64, seed=2).to(DEV)
    with pytest.raises(_lib.T2HError, match='multiple of 128'):
        ops.gemm(x, w, w_split=ops.split_rows(w), rows_per_img=96)
    assert not ops.conv_split_ok(96) and not ops.conv_split_ok(512, 'down')


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
