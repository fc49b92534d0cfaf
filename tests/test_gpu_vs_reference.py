"""The HIP path against the UNMODIFIED reference on the same GPU (-m gpu).

`oracle/make_ref.py` compiles the reference's own modules of the path (from the sources where they lie in the build
container) into byte code under the git-ignored `oracle/_ref/`, which travels to the GPU box; `oracle/ref_shim.py`
loads it with the mmcv / torchvision stubs.  Here the reference's `SampleFromParsingModel` / `SampleFromPoseModel`
(models/sample_model.py:343-498) run as eager PyTorch-ROCm fp32 on cuda:0 -- the exact program a user of the reference
runs -- next to `text2human_amd.models`' classes: same `.pth` files, same batch, `set_random_seed(2021)` before each
(the reference contract: torch's global device generator).  Free-running: tokens must be EQUAL; images within 2e-4.
(The teacher-forced per-decision accounting against the oracle port is tests/test_gpu_bench_parity.py; the port is
pinned against the reference on the CPU by tests/test_oracle_vs_reference.py.)"""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from oracle import ref_shim
from text2human_amd import defaults, models, options, synthetic

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_shim.available(), reason='oracle/_ref (byte code of the reference) absent: '
                                                                  'run `python oracle/make_ref.py` in the build container')]
DEV = 'cuda:0'


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _to_dev(batch):
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}


@pytest.fixture(scope='module')
def ns():
    return ref_shim.load_reference(DEV)


@pytest.mark.parametrize('steps,B', [(256, 2), (40, 4)])
def test_sample_from_parsing_equals_the_reference_on_this_gpu(ns, tmp_path, steps, B):
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    opt['sample_steps'] = steps
    o = synthetic.write_checkpoints(opt, str(tmp_path / 'ckpt'), seed=1234)
    batch = _to_dev(synthetic.parsing_batch(B, seed=2021))
    ref = _quiet(ns.sample_model.SampleFromParsingModel, o)
    ours = models.create_model(o)
    with torch.no_grad():
        ns.util.set_random_seed(2021)
        ref.feed_data(batch)
        ref_top = ref.sample_fn(temp=1, sample_steps=steps)
        ref_state = torch.cuda.get_rng_state(0)
    options.set_random_seed(2021)
    ours.feed_data(batch)
    top = ours.sample_fn(temp=1, sample_steps=steps)
    assert torch.equal(ours.segm_tokens.view(B, -1), ref.segm_tokens.view(B, -1))
    a, b = torch.stack(top), torch.stack(ref_top)
    assert (b >= 0).sum().item() == B * 512
    assert torch.equal(a, b), f'{int((a != b).sum())} of {B * 512} sampled tokens differ from the reference'
    # the generator stands where the reference's does: the NEXT draw is the same
    assert torch.equal(torch.cuda.get_rng_state(0), ref_state)
    # refine + decode through the public entry: sample_and_refine re-samples, then writes {save_dir}/{name}
    ref_shim.saved_images.clear()
    with torch.no_grad():
        ns.util.set_random_seed(2021)
        ref.sample_and_refine(str(tmp_path), batch['img_name'])
    ref_imgs = torch.cat([t for t, _ in ref_shim.saved_images], 0)
    assert [os.path.basename(p) for _, p in ref_shim.saved_images] == list(batch['img_name'])
    out_dir = tmp_path / 'ours'
    out_dir.mkdir()
    options.set_random_seed(2021)
    ours.sample_and_refine(str(out_dir), batch['img_name'])
    from PIL import Image
    for i, name in enumerate(batch['img_name']):
        got = np.asarray(Image.open(out_dir / name))
        want = ref_imgs[i].mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
        assert got.shape == want.shape == (512, 256, 3)
        d = np.abs(got.astype(np.int16) - want.astype(np.int16))
        assert d.max() <= 1 and (d != 0).mean() < 5e-3, (name, int(d.max()), float((d != 0).mean()))
    # and the float image of the first sample (the no-argument form of the reference, sample_model.py:246-247)
    options.set_random_seed(2021)
    img0 = ours.sample_and_refine()
    assert tuple(img0.shape) == (1, 3, 512, 256)
    assert (img0.cpu() - ref_imgs[:1]).abs().max().item() < 2e-4


def test_sample_from_pose_equals_the_reference_on_this_gpu(ns, tmp_path):
    """pose front end + tokenizer + sampler on the reference's own call order (sample_model.py:412-441)."""
    from parity_util import balanced_pose_state_dicts
    steps, B = 24, 2
    opt = options.dict_to_nonedict(defaults.sample_from_pose())
    opt['sample_steps'] = steps
    sds = balanced_pose_state_dicts(synthetic.make_state_dicts(opt, seed=1234), opt)
    o = synthetic.write_checkpoints(opt, str(tmp_path / 'ckpt'), seed=1234, state_dicts=sds)
    batch = _to_dev(synthetic.pose_batch(B, seed=2021))
    ref = _quiet(ns.sample_model.SampleFromPoseModel, o)
    ours = models.create_model(o)
    with torch.no_grad():
        ns.util.set_random_seed(2021)
        ref.feed_data(batch)
        ref.generate_parsing_map()
        ref.generate_quantized_segm()
        ref.generate_texture_map()
        ref_top = ref.sample_fn(temp=1, sample_steps=steps)
    options.set_random_seed(2021)
    ours.feed_data(batch)
    ours.generate_parsing_map()
    ours.generate_quantized_segm()
    ours.generate_texture_map()
    top = ours.sample_fn(temp=1, sample_steps=steps)
    seg_diff = (ours.segm.view(-1) != ref.segm.view(-1)).float().mean().item()
    assert seg_diff < 1e-3, seg_diff  # argmax near-ties of two fp32 implementations (accounted in test_gpu_configs.py)
    if seg_diff == 0:
        assert torch.equal(ours.segm_tokens.view(B, -1), ref.segm_tokens.view(B, -1))
        assert torch.equal(ours.texture_mask, ref.texture_mask)
        assert torch.equal(torch.stack(top), torch.stack(ref_top))
    else:  # a differing parsing pixel may move a token: feed the reference's state, the rest must be exact
        ours.segm, ours.segm_tokens, ours.texture_mask = ref.segm, ref.segm_tokens, ref.texture_mask
        options.set_random_seed(2021)
        with torch.no_grad():
            ns.util.set_random_seed(2021)
            ref_top = ref.sample_fn(temp=1, sample_steps=steps)
        options.set_random_seed(2021)
        top = ours.sample_fn(temp=1, sample_steps=steps)
        assert torch.equal(torch.stack(top), torch.stack(ref_top))
