"""Oracle (oracle/torch_ref.py) vs the golden vectors that oracle/make_golden.py
produced by running the unmodified reference.  Runs anywhere (CPU, no reference
tree needed): this is what keeps the oracle pinned on the GPU box."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from oracle.make_golden import (TRANSFORMER_HEADS, TRANSFORMER_ROWS, WEIGHT_SEED,
                                golden_inputs)
from text2human_amd import defaults, options, synthetic

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def opt():
    return options.dict_to_nonedict(defaults.sample_from_pose())


@pytest.fixture(scope='module')
def sds(opt):
    return synthetic.make_state_dicts(opt, seed=WEIGHT_SEED)


def _close(a, b, tol):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    err = (a - b).abs().max().item()
    assert err <= tol, f'max abs err {err} > {tol}'


def test_decode_golden(sds):
    g = np.load(os.path.join(GOLD, 'decode_b1.npz'))
    gi = golden_inputs('decoder')
    with torch.no_grad():
        bh = R.decoder_res(gi['zb'], sds['bot_decoder_res'])
        out = R.decoder(gi['z'], sds['decoder'], bot_h=bh)
    _close(bh[0, ::16, ::4, ::4], g['bot_h_sample'], 1e-5)
    _close(out[0, :, ::4, ::4], g['dec_sample'], 1e-5)


def test_transformer_golden(sds):
    g = np.load(os.path.join(GOLD, 'transformer_b1.npz'))
    gi = golden_inputs('transformer')
    with torch.no_grad():
        logits = R.transformer_logits(gi['idx'], gi['seg'], gi['tex'], sds['sampler'])
    for j, h in enumerate(TRANSFORMER_HEADS):
        _close(logits[h][0, TRANSFORMER_ROWS], g['logits'][j], 2e-5)


def test_index_pred_golden(sds):
    g = np.load(os.path.join(GOLD, 'index_pred_b2.npz'))
    gi = golden_inputs('unet')
    with torch.no_grad():
        feat = R.unet(gi['x'], sds['guidance_encoder'])[4]
        hl = R.multihead_fcn(feat, sds['index_decoder'])
    _close(feat[:, ::8, ::4, ::4], g['feat_sample'], 1e-5)
    am = torch.stack([l.argmax(1) for l in hl]).numpy()
    bad = am != g['argmax']
    assert (g['margin'][bad] < 1e-5).all()


def test_e2e_golden(sds):
    g = np.load(os.path.join(GOLD, 'e2e_parsing_b2_steps5.npz'))
    batch = synthetic.parsing_batch(2, seed=2021)
    torch.manual_seed(2021)
    with torch.no_grad():
        img, inter = R.sample_from_parsing(batch['segm'], batch['texture_mask'], sds,
                                           sample_steps=5, noise=R.TorchNoise('cpu'))
    assert np.array_equal(inter['segm_tokens'].numpy(), g['segm_tokens'])
    assert np.array_equal(torch.stack(inter['top_indices']).numpy(), g['top_indices'])
    assert np.array_equal(torch.stack(inter['bot_idx']).numpy(), g['bot_indices'])
    _close(img[:, :, ::4, ::4], g['img_sample'], 1e-5)
    assert np.array_equal(R.to_uint8(img).permute(0, 3, 1, 2)[:, :, ::4, ::4].numpy(),
                          g['img_u8_sample'])


def test_pose_golden(sds, opt):
    g = np.load(os.path.join(GOLD, 'pose_b2_128x64.npz'))
    pb = synthetic.pose_batch(2, seed=2021)
    pose = pb['densepose'][:, :, :128, :64].contiguous()
    with torch.no_grad():
        emb = R.shape_attr_embedding(pb['shape_attr'], sds['shape_embedder'],
                                     opt['shape_attr_class_num'])
        segm, logits = R.parsing_from_pose(pose, pb['shape_attr'], sds['shape_embedder'],
                                           sds['shape_encoder'], sds['shape_decoder'],
                                           opt['shape_attr_class_num'])
    _close(emb, g['attr_embedding'], 1e-6)
    _close(logits[:, :, ::8, ::8], g['logits_sample'], 1e-5)
    bad = segm[:, 0].numpy() != g['segm']
    assert (g['margin'][bad] < 1e-5).all()
