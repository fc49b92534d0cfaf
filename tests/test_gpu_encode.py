"""Encode side of the hierarchy (-m gpu): texture-routed codebook
argmin, top / bottom encoders, image -> tokens -> image against the reference-made
golden fixture (tests/golden/encode_b1.npz) and the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.make_golden import golden_inputs
from text2human_amd import defaults, ops, options, synthetic
from text2human_amd.models import create_model

pytestmark = pytest.mark.gpu
DEV = 'cuda'
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _ref_argmin(rows, books, tex):
    """(idx, margin) of the expanded distance inside each row's own codebook, fp64"""
    idx = torch.full((rows.shape[0], ), -1, dtype=torch.long)
    margin = torch.full((rows.shape[0], ), float('inf'), dtype=torch.double)
    for cb in range(books.shape[0]):
        sel = tex == cb
        if sel.any():
            e = books[cb].double()
            d = (rows[sel].double()**2).sum(1, keepdim=True) + (e**2).sum(1) - 2 * rows[sel].double() @ e.t()
            t2 = d.topk(2, dim=1, largest=False)
            idx[sel] = t2.indices[:, 0]
            margin[sel] = t2.values[:, 1] - t2.values[:, 0]
    return idx, margin


@pytest.mark.parametrize('d,n_e,fold', [(256, 1024, False), (1024, 512, True), (256, 37, False)])
def test_vq_argmin_tex_matches_reference_formula_and_is_idempotent(d, n_e, fold):
    B, h, w, nb = 2, 8, 4, 18
    n = B * h * w
    g = torch.Generator().manual_seed(7)
    books = torch.rand(nb, n_e, d, generator=g) * 2 - 1
    tex = torch.randint(0, nb, (n, ), generator=g)
    tex[5] = 17
    if fold:
        zmap = _rnd(B, d // 4, 2 * h, 2 * w, seed=8)                      # NCHW latent
        rows = F.unfold(zmap, (2, 2), stride=2).permute(0, 2, 1).reshape(n, d)
        z_dev = zmap.permute(0, 2, 3, 1).reshape(-1, d // 4).contiguous().to(DEV)
        got = ops.vq_argmin_tex(z_dev, books.to(DEV), tex.to(DEV), fold_hw=(h, w)).cpu()
    else:
        rows = _rnd(n, d, seed=9)
        got = ops.vq_argmin_tex(rows.to(DEV), books.to(DEV), tex.to(DEV)).cpu()
    ref, margin = _ref_argmin(rows, books, tex)
    mine = got[tex, torch.arange(n)]
    assert (margin[mine != ref] < 1e-3).all()
    assert (mine != ref).float().mean() < 0.02
    off = torch.ones(nb, n, dtype=torch.bool)
    off[tex, torch.arange(n)] = False
    assert (got[off] == -1).all()
    # quantising codebook entries returns their own indices (distance 0 is the unique minimum)
    pick = torch.randint(0, n_e, (n, ), generator=g)
    entries = books[tex, pick]                                             # [n, d]
    if fold:
        emap = F.fold(entries.view(B, h * w, d).permute(0, 2, 1), (2 * h, 2 * w), kernel_size=2, stride=2)
        e_dev = emap.permute(0, 2, 3, 1).reshape(-1, d // 4).contiguous().to(DEV)
        again = ops.vq_argmin_tex(e_dev, books.to(DEV), tex.to(DEV), fold_hw=(h, w)).cpu()
    else:
        again = ops.vq_argmin_tex(entries.contiguous().to(DEV), books.to(DEV), tex.to(DEV)).cpu()
    assert torch.equal(again[tex, torch.arange(n)], pick)


@pytest.fixture(scope='module')
def model_and_sds():
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    opt['model_type'] = 'VQGANTextureAwareSpatialHierarchyInferenceModel'
    sds = synthetic.make_state_dicts(opt, seed=1234, encode=True)
    from text2human_amd.models import VQGANTextureAwareSpatialHierarchyInferenceModel as M
    return M(opt, state_dicts=sds), sds


def test_encode_golden_from_the_reference_modules(model_and_sds):
    model, _ = model_and_sds
    g = np.load(os.path.join(GOLD, 'encode_b1.npz'))
    gi = golden_inputs('encode')
    model.feed_data(dict(image=gi['image'], texture_mask=gi['texture_mask']))
    top = torch.stack([t.view(1, 32, 16) for t in model.top_indices_list]).cpu().numpy()
    bot = torch.stack(model.gt_indices_list).cpu().numpy()
    bad_t = (top != g['top_indices']).any(0).reshape(-1)
    bad_b = (bot != g['bot_indices']).any(0).reshape(-1)
    assert (g['top_margin'][bad_t] < 1e-3).all() and bad_t.mean() < 0.02
    assert (g['bot_margin'][bad_b] < 1e-3).all() and bad_b.mean() < 0.02
    if not bad_t.any():
        err = (model.quant_t[0, ::8, ::2, ::2].cpu() - torch.from_numpy(g['quant_t_sample'])).abs().max().item()
        assert err < 2e-4, err
    if not (bad_t.any() or bad_b.any()):
        rec = model.index_to_image(model.gt_indices_list, model.texture_mask)
        err = (rec[0, :, ::4, ::4].cpu() - torch.from_numpy(g['rec_sample'])).abs().max().item()
        assert err < 5e-4, err


def test_reconstruct_matches_oracle_b2(model_and_sds):
    from oracle import torch_ref as R
    model, sds = model_and_sds
    g = torch.Generator().manual_seed(21)
    img = torch.rand(2, 3, 512, 256, generator=g) * 2 - 1
    mask = synthetic.parsing_batch(2, seed=77)['texture_mask']
    with torch.no_grad():
        ref_rec, inter = R.reconstruct(img, mask, sds)
    model.feed_data(dict(image=img, texture_mask=mask))
    top = torch.stack([t.view(2, 32, 16) for t in model.top_indices_list]).cpu()
    bot = torch.stack(model.gt_indices_list).cpu()
    assert (top != torch.stack(inter['top_indices'])).float().mean() < 0.005
    assert (bot != torch.stack(inter['bot_indices'])).float().mean() < 0.005
    if torch.equal(top, torch.stack(inter['top_indices'])) and torch.equal(bot, torch.stack(inter['bot_indices'])):
        rec = model.index_to_image(model.gt_indices_list, model.texture_mask).cpu()
        assert (rec - ref_rec).abs().max().item() < 5e-4
    # create_model wiring
    assert type(model).__name__ == 'VQGANTextureAwareSpatialHierarchyInferenceModel'
    assert callable(create_model)
