"""Stage-level parity (-m gpu) of the HIP path, through the model classes that
mirror the reference's API, against (a) the golden vectors produced by the
unmodified reference (tests/golden, oracle/make_golden.py) and (b) the oracle
(oracle/torch_ref.py) on the same seeded inputs.

Tolerances (fp32 path, exact-fp32 MFMA): images / activations 2e-4 abs on
O(1) values (the reference's natural bound is 1/255 = 3.9e-3); token / index
outputs bit-exact, a mismatch is accepted only where the reference's own
decision margin is a float near-tie (margin stored in the golden file).
"""
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from oracle.make_golden import (TRANSFORMER_HEADS, TRANSFORMER_ROWS, WEIGHT_SEED,
                                golden_inputs)
from text2human_amd import defaults, ops, options, synthetic
from text2human_amd.models import create_model

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
DEV = 'cuda'


@pytest.fixture(scope='module')
def opt():
    return options.dict_to_nonedict(defaults.sample_from_pose())


@pytest.fixture(scope='module')
def sds(opt):
    return synthetic.make_state_dicts(opt, seed=WEIGHT_SEED)


@pytest.fixture(scope='module')
def ckpt_opt(opt, tmp_path_factory):
    d = tmp_path_factory.mktemp('ckpt')
    return synthetic.write_checkpoints(opt, str(d), seed=WEIGHT_SEED)


@pytest.fixture(scope='module')
def model(ckpt_opt):
    # through the real checkpoint files + create_model(opt): the drop-in boundary
    o = dict(ckpt_opt)
    o['model_type'] = 'SampleFromParsingModel'
    return create_model(options.dict_to_nonedict(o))


@pytest.fixture(scope='module')
def pose_model(ckpt_opt):
    return create_model(options.dict_to_nonedict(dict(ckpt_opt)))


def _err(a, b):
    return (torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max().item()


def test_native_library_is_loaded(model):
    from text2human_amd import _lib
    maps = open('/proc/self/maps').read()
    assert 'libt2h_hip.so' in maps and _lib.load().t2h_version() >= 100


def test_decode_vs_golden(model):
    g = np.load(os.path.join(GOLD, 'decode_b1.npz'))
    gi = golden_inputs('decoder')
    zb = ops.nchw_to_nhwc(gi['zb'].to(DEV))
    z = ops.nchw_to_nhwc(gi['z'].to(DEV))
    bot_h = model.bot_decoder_res.decode_res(zb, 1, 64, 32)
    dec, ho, wo = model.decoder.decode(z, 1, 32, 16, bot_h=bot_h)
    assert (ho, wo) == (512, 256)
    bh = ops.nhwc_to_nchw(bot_h, 1, 64, 32)
    out = ops.nhwc_to_nchw(dec, 1, 512, 256)
    assert _err(bh[0, ::16, ::4, ::4], g['bot_h_sample']) < 2e-4
    assert _err(out[0, :, ::4, ::4], g['dec_sample']) < 2e-4


def test_transformer_vs_golden(model):
    g = np.load(os.path.join(GOLD, 'transformer_b1.npz'))
    gi = golden_inputs('transformer')
    logits = model.sampler_fn.logits(gi['idx'].to(DEV), gi['seg'].to(DEV), gi['tex'].to(DEV),
                                     heads=set(TRANSFORMER_HEADS))
    for j, h in enumerate(TRANSFORMER_HEADS):
        assert _err(logits[h][0, TRANSFORMER_ROWS], g['logits'][j]) < 2e-4


def test_transformer_batch_vs_oracle(model, sds):
    gen = torch.Generator().manual_seed(5)
    idx = torch.randint(0, 18433, (3, 512), generator=gen)
    seg = torch.randint(0, 1024, (3, 512), generator=gen)
    tex = torch.randint(0, 18, (3, 512), generator=gen)
    with torch.no_grad():
        ref = R.transformer_hidden(idx, seg, tex, sds['sampler'])
    hid = model.sampler_fn.hidden(idx.to(DEV), seg.to(DEV), tex.to(DEV))
    P = model.P
    got = ops.layernorm(hid, P['tf.ln_f.g'], P['tf.ln_f.b']).view(3, 512, 512)
    assert _err(got, ref) < 2e-4


def test_index_prediction_vs_golden(model):
    g = np.load(os.path.join(GOLD, 'index_pred_b2.npz'))
    gi = golden_inputs('unet')
    x = gi['x'].to(DEV)
    feat, _, _ = model.index_pred_guidance_encoder.forward(ops.nchw_to_nhwc(x), 2, 32, 16)
    f = ops.nhwc_to_nchw(feat, 2, 32, 16)
    assert _err(f[:, ::8, ::4, ::4], g['feat_sample']) < 2e-4
    # per-texture routing: run once per head with a constant mask and compare that head
    for h in (0, 3, 17):
        mask = torch.full((2, 1, 512, 256), float(h), device=DEV)
        lists = model.bot_index_prediction(x, mask)
        got = lists[h].cpu().numpy()
        bad = got != g['argmax'][h]
        assert (g['margin'][h][bad] < 1e-4).all(), f'head {h}: {bad.sum()} non-tie mismatches'
        assert all((lists[k] == -1).all() for k in range(18) if k != h)


def test_segm_tokens_vs_golden(model):
    g = np.load(os.path.join(GOLD, 'e2e_parsing_b2_steps5.npz'))
    batch = synthetic.parsing_batch(2, seed=2021)
    tok = model.get_quantized_segm(batch['segm'].to(DEV))
    assert tok.shape == (2, 32, 16) and tok.dtype == torch.int64
    assert np.array_equal(tok.view(2, -1).cpu().numpy(), g['segm_tokens'])


def test_end_to_end_vs_golden_with_reference_noise(model):
    """The golden run consumed torch's CPU generator (seed 2021); replaying the
    same draws through the noise hook must reproduce the reference's tokens,
    bottom indices and image."""
    g = np.load(os.path.join(GOLD, 'e2e_parsing_b2_steps5.npz'))
    batch = synthetic.parsing_batch(2, seed=2021)
    model.feed_data(batch)
    torch.manual_seed(2021)
    model.noise = R.TorchNoise('cpu')  # draws on the CPU stream, uploaded by the engine
    try:
        top = model.sample_fn(temp=1, sample_steps=5)
    finally:
        model.noise = None
    got_top = torch.stack(top).cpu().numpy()
    assert np.array_equal(got_top, g['top_indices']), \
        f'{(got_top != g["top_indices"]).sum()} sampled tokens differ'
    img, u8, inter = model.decode_indices(top, want_u8=True, return_inter=True)
    bot = inter[0]['bot_lists'].view(18, 2, 32, 16).cpu().numpy()
    assert np.array_equal(bot, g['bot_indices'])
    assert _err(img[:, :, ::4, ::4], g['img_sample']) < 2e-4
    d = np.abs(u8.permute(0, 3, 1, 2)[:, :, ::4, ::4].cpu().numpy().astype(int) - g['img_u8_sample'].astype(int))
    assert d.max() <= 1 and (d != 0).mean() < 0.01


def test_sampler_matches_oracle_on_device_rng(model, sds):
    """Same GPU, same seed, torch's device generator: the HIP sampler and the
    oracle (run with eager PyTorch-ROCm on the GPU) must unmask the same tokens
    with the same values -- this is the reference's RNG contract (rand per step,
    one full exponential_ draw per ACTIVE head).  Teacher-forced on the oracle's trajectory (every one of the 1024
    decisions compared on identical inputs; a differing one must be a float near-tie by parity_util.account: the
    oracle's own preference for its token is below twice the logit difference of the two implementations, itself
    inside the activation tolerance), then free-running."""
    from parity_util import account, forced_run, oracle_run
    batch = synthetic.parsing_batch(2, seed=7)
    model.feed_data(batch)
    sd_dev = {k: v.to(DEV) for k, v in sds['sampler'].items()}
    ref, trace, rng_state = oracle_run(model.segm_tokens, batch['texture_mask'], sd_dev, 6, 123)
    mism, _ = forced_run(model, trace, 6, 123, compact=True)
    rows = account(model, sd_dev, batch['texture_mask'], trace, rng_state, mism, 6)
    assert len(mism) <= 1 and all(r['explained'] for r in rows), rows
    torch.manual_seed(123)
    torch.cuda.manual_seed_all(123)
    got = model.sample_fn(temp=1, sample_steps=6)
    ref_t, got_t = torch.stack(ref).cpu(), torch.stack(got).cpu()
    diff = (ref_t != got_t).sum().item()
    if not mism:
        assert diff == 0, f'{diff} of {ref_t.numel()} entries differ'
    else:  # the accounted near-tie decides one token differently; the trajectories part there
        print(f'one accounted near-tie ({rows}); free-running entries differing after it: {diff}')


def test_categorical_sample_equals_exponential_race_on_gpu():
    torch.manual_seed(9)
    logits = (torch.randn(512, 1024) * 3).to(DEV)
    torch.cuda.manual_seed_all(77)
    ref = torch.distributions.Categorical(logits=logits).sample()
    torch.cuda.manual_seed_all(77)
    expo = torch.empty(512, 1024, device=DEV).exponential_(1.0)
    assert torch.equal(R.categorical_argmax(logits, expo), ref)


def test_pose_front_end_vs_golden(pose_model, opt):
    model = pose_model
    g = np.load(os.path.join(GOLD, 'pose_b2_128x64.npz'))
    pb = synthetic.pose_batch(2, seed=2021)
    pb['densepose'] = pb['densepose'][:, :, :128, :64].contiguous()
    model.feed_data(pb)
    emb = model._attr_embedding(model.shape_attr)
    assert _err(emb, g['attr_embedding']) < 1e-5
    model.generate_parsing_map()
    lg = ops.nhwc_to_nchw(model.seg_logits_rows, 2, 128, 64)
    assert _err(lg[:, :, ::8, ::8], g['logits_sample']) < 2e-4
    bad = model.segm[:, 0].cpu().numpy() != g['segm']
    assert (g['margin'][bad] < 1e-4).all()
    model.generate_texture_map()
    ref_mask = R.texture_map(model.segm.cpu(), pb['upper_fused_attr'], pb['lower_fused_attr'],
                             pb['outer_fused_attr'])
    assert torch.equal(model.texture_mask.cpu(), ref_mask)


def test_api_surface_and_png_output(model, tmp_path):
    batch = synthetic.parsing_batch(2, seed=11)
    model.sample_steps = 3
    try:
        model.feed_data(batch)
        img = model.sample_and_refine()
        assert img.shape == (1, 3, 512, 256) and 0.0 <= img.min() and img.max() <= 1.0
        model.inference([batch], str(tmp_path))
    finally:
        model.sample_steps = 256
    from PIL import Image
    for name in batch['img_name']:
        im = Image.open(tmp_path / name)
        assert im.size == (256, 512) and im.mode == 'RGB'
    assert model.shape == (32, 16) and model.mask_id == 18432 and model.batch_size == 2
    with pytest.raises(ValueError):
        create_model({'model_type': 'NoSuchModel'})


def test_strict_checkpoint_validation(opt, sds):
    from text2human_amd import weights
    bad = dict(sds['sampler'])
    bad.pop('ln_f.weight')
    with pytest.raises(RuntimeError, match='Missing key'):
        weights.check_state_dict(bad, synthetic.module_schemas(opt)['sampler'], 'sampler')


def test_graft_entry_smoke_runs():
    """the driver's round-end smoke check (one small invocation of the hot path on cuda:0 against the oracle)"""
    import __graft_entry__
    __graft_entry__.smoke()
