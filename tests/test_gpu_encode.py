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

from parity_util import odev, osds  # noqa: E402

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


def _account_level(z_hip_rows, z_ref_rows, books, tex, lists_hip, lists_ref, lat_tol):
    """Every differing index of one level accounted for as a codebook near-tie on THAT row's measured latent error
    (parity_util.vq_mismatch_accounting with the row's own texture codebook).  -> (n differing rows, latent max err)."""
    from parity_util import vq_mismatch_accounting
    n = tex.numel()
    ar = torch.arange(n)
    mine, ref = lists_hip.reshape(18, n)[tex, ar], lists_ref.reshape(18, n)[tex, ar]
    # off-texture entries are -1 in both
    off = torch.ones(18, n, dtype=torch.bool)
    off[tex, ar] = False
    assert (lists_hip.reshape(18, n)[off] == -1).all() and (lists_ref.reshape(18, n)[off] == -1).all()
    lat = (z_hip_rows - z_ref_rows).abs().max().item()
    assert lat < lat_tol, f'latent max abs err {lat}'
    bad = (mine != ref).nonzero().flatten().tolist()
    for row in bad:
        acc = vq_mismatch_accounting(z_hip_rows[row:row + 1], z_ref_rows[row:row + 1], books[int(tex[row])],
                                     mine[row:row + 1], ref[row:row + 1])
        assert len(acc) == 1 and acc[0]['explained'], (row, acc)
    return len(bad), lat


def _levels(model, image, mask, sds):
    """latent rows of both levels, HIP and oracle (fp32 CPU), in the row order the index lists use"""
    from oracle import torch_ref as R
    b = image.shape[0]
    with torch.no_grad():
        zt, zb = (t.cpu() for t in R.encode_latents(odev(image), osds(sds)))   # (parity_util.ORACLE_DEV)
    zt_ref = zt.permute(0, 2, 3, 1).reshape(-1, zt.shape[1])
    zb_ref = F.unfold(zb, (2, 2), stride=2).permute(0, 2, 1).reshape(-1, zb.shape[1] * 4)
    zt_hip = model._top_latent_rows.cpu()
    h, w = model._bot_latent_hw
    zb_map = model._bot_latent_rows.cpu().view(b, h, w, -1).permute(0, 3, 1, 2)
    zb_hip = F.unfold(zb_map, (2, 2), stride=2).permute(0, 2, 1).reshape(-1, zb_map.shape[1] * 4)
    tex = mask[:, 0, ::16, ::16].reshape(-1).long()
    tb = torch.stack([sds['top_quantize'][f'embedding_list.{i}.weight'] for i in range(18)])
    bb = torch.stack([sds['bot_quantize'][f'embedding_list.{i}.weight'] for i in range(18)])
    return (zt_hip, zt_ref, tb), (zb_hip, zb_ref, bb), tex


def test_encode_golden_from_the_reference_modules(model_and_sds):
    """B=1 against the fixture made by the UNMODIFIED reference (oracle/make_golden.py): every differing top / bottom
    index is accounted for as a near-tie of its row; then the REFERENCE's indices are injected and the latent after
    top_post_quant_conv and the reconstruction are compared UNCONDITIONALLY (hierarchy_inference_model.py:170-209)."""
    from oracle import torch_ref as R
    model, sds = model_and_sds
    g = np.load(os.path.join(GOLD, 'encode_b1.npz'))
    gi = golden_inputs('encode')
    model.feed_data(dict(image=gi['image'], texture_mask=gi['texture_mask']))
    top = torch.stack([t.view(1, 32, 16) for t in model.top_indices_list]).cpu()
    bot = torch.stack(model.gt_indices_list).cpu()
    g_top, g_bot = torch.from_numpy(g['top_indices']).long(), torch.from_numpy(g['bot_indices']).long()
    # the oracle's indices ARE the reference's on this input (pins the latents used for the accounting below)
    with torch.no_grad():
        _, inter = R.reconstruct(odev(gi['image']), odev(gi['texture_mask']), osds(sds))
    inter = odev(inter, 'cpu')
    assert torch.equal(torch.stack(inter['top_indices']), g_top) and torch.equal(torch.stack(inter['bot_indices']), g_bot)
    (zt_h, zt_r, tb), (zb_h, zb_r, bb), tex = _levels(model, gi['image'], gi['texture_mask'], sds)
    n_t, lat_t = _account_level(zt_h, zt_r, tb, tex, top, g_top, 2e-4)
    n_b, lat_b = _account_level(zb_h, zb_r, bb, tex, bot, g_bot, 2e-4)
    assert n_t <= 5 and n_b <= 5, (n_t, n_b)
    # the golden's own latent samples agree with what the HIP encoders produced
    got = zt_h.view(32, 16, -1).permute(2, 0, 1)[::8, ::2, ::2]
    assert (got - torch.from_numpy(g['top_latent_sample'])).abs().max().item() < 2e-4
    # ---- downstream, on the reference's indices, unconditionally
    quant_t = model.quant_from_top_indices(g_top.reshape(18, -1), gi['texture_mask'], (1, 32, 16))
    err = (quant_t[0, ::8, ::2, ::2].cpu() - torch.from_numpy(g['quant_t_sample'])).abs().max().item()
    assert err < 2e-4, err
    rec = model.index_to_image([t for t in g_bot.to(DEV)], model.texture_mask)
    err = (rec[0, :, ::4, ::4].cpu() - torch.from_numpy(g['rec_sample'])).abs().max().item()
    assert err < 5e-4, err
    mom = g['rec_moments']
    assert abs(rec.double().mean().item() - mom[0]) < 1e-4
    print(f'encode golden: top {n_t} / bottom {n_b} differing rows (all near-ties), latent err {lat_t:.1e} / {lat_b:.1e}')


def test_reconstruct_matches_oracle_b2(model_and_sds):
    """B=2 random images against the oracle: per-row accounting of both levels, then the oracle's indices injected
    and quant_t / the reconstruction compared unconditionally."""
    from oracle import torch_ref as R
    model, sds = model_and_sds
    g = torch.Generator().manual_seed(21)
    img = torch.rand(2, 3, 512, 256, generator=g) * 2 - 1
    mask = synthetic.parsing_batch(2, seed=77)['texture_mask']
    with torch.no_grad():
        ref_rec, inter = R.reconstruct(odev(img), odev(mask), osds(sds))
    ref_rec, inter = ref_rec.cpu(), odev(inter, 'cpu')
    model.feed_data(dict(image=img, texture_mask=mask))
    top = torch.stack([t.view(2, 32, 16) for t in model.top_indices_list]).cpu()
    bot = torch.stack(model.gt_indices_list).cpu()
    r_top, r_bot = torch.stack(inter['top_indices']), torch.stack(inter['bot_indices'])
    (zt_h, zt_r, tb), (zb_h, zb_r, bb), tex = _levels(model, img, mask, sds)
    n_t, _ = _account_level(zt_h, zt_r, tb, tex, top, r_top, 2e-4)
    n_b, _ = _account_level(zb_h, zb_r, bb, tex, bot, r_bot, 2e-4)
    assert n_t <= 10 and n_b <= 10, (n_t, n_b)
    quant_t = model.quant_from_top_indices(r_top.reshape(18, -1), mask, (2, 32, 16))
    assert (quant_t.cpu() - inter['quant_t']).abs().max().item() < 2e-4
    rec = model.index_to_image([t for t in r_bot.to(DEV)], model.texture_mask).cpu()
    assert (rec - ref_rec).abs().max().item() < 5e-4
    # create_model wiring
    assert type(model).__name__ == 'VQGANTextureAwareSpatialHierarchyInferenceModel'
    assert callable(create_model)
