"""the fine-grained methods the reference's UI drives
(`ui_demo.py:100-167`): feed_pose_data -> feed_shape_attributes -> generate_parsing_map ->
generate_quantized_segm -> palette_result -> (user edits the parsing) -> model.segm = ... ->
generate_quantized_segm -> feed_texture_attributes -> generate_texture_map ->
sample_and_refine() -> [1,3,512,256]; replayed here in that order on SampleFromPoseModel
against the oracle (models/sample_model.py:431-498)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from text2human_amd import defaults, options, synthetic
from text2human_amd.models import SampleFromPoseModel

from parity_util import odev, osds  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_ui_demo_call_order_vs_oracle():
    opt = options.dict_to_nonedict(defaults.sample_from_pose())
    sds = synthetic.make_state_dicts(opt, seed=1234)
    model = SampleFromPoseModel(opt, state_dicts=sds)
    model.sample_steps = 4
    pb = synthetic.pose_batch(1, seed=5)

    # open_densepose (ui_demo.py:69-98): CPU float32 [1,1,512,256]
    model.feed_pose_data(pb['densepose'])
    assert model.batch_size == 1 and model.pose.is_cuda
    # generate_parsing (ui_demo.py:100-131)
    model.feed_shape_attributes(torch.LongTensor(pb['shape_attr'][0].tolist()).unsqueeze(0))
    model.generate_parsing_map()
    model.generate_quantized_segm()
    od = osds(sds)   # (the oracle's convolutional stages: parity_util.ORACLE_DEV)
    with torch.no_grad():
        ref_segm, _ = R.parsing_from_pose(odev(pb['densepose']), odev(pb['shape_attr']), od['shape_embedder'],
                                          od['shape_encoder'], od['shape_decoder'],
                                          opt['shape_attr_class_num'])
    ref_segm = ref_segm.cpu()
    assert model.segm.shape == (1, 1, 512, 256) and model.segm.dtype == torch.int64
    agree = (model.segm.cpu() == ref_segm).float().mean().item()
    assert agree > 0.9999, f'parsing agrees on {agree:.6f} of the pixels'  # float near-ties only
    colored = model.palette_result(model.segm[0].cpu())
    assert colored.shape == (512, 256, 3) and colored.dtype == np.uint8
    pal = np.array(model.palette, dtype=np.uint8)
    assert np.array_equal(colored, pal[model.segm[0, 0].cpu().numpy()])

    # generate_human (ui_demo.py:133-167): the (edited) parsing comes back as a numpy-born
    # int64 map; use the oracle's so both sides continue from the same parsing
    seg_map = ref_segm[0, 0].numpy().astype(np.int64)
    model.segm = torch.from_numpy(seg_map).unsqueeze(0).unsqueeze(0).to(model.device)
    model.generate_quantized_segm()
    texture_attributes = torch.LongTensor([int(pb['upper_fused_attr'][0]), int(pb['lower_fused_attr'][0]),
                                           int(pb['outer_fused_attr'][0])])
    model.feed_texture_attributes(texture_attributes)
    assert model.upper_fused_attr.shape == (1, )
    model.generate_texture_map()
    ref_mask = R.texture_map(ref_segm, pb['upper_fused_attr'], pb['lower_fused_attr'], pb['outer_fused_attr'])
    assert torch.equal(model.texture_mask.cpu(), ref_mask)

    model.noise = R.SeededNoise(11, 'cpu')
    try:
        result = model.sample_and_refine()
    finally:
        model.noise = None
    assert result.shape == (1, 3, 512, 256) and result.dtype == torch.float32
    with torch.no_grad():   # R.sample_from_parsing, its sampler on the CPU like the explicit noise it is given
        tok = R.segm_tokens(odev(ref_segm.float()), od['segm_encoder'], od['segm_quant_conv'],
                            od['segm_quantizer']['embedding.weight']).view(1, -1).cpu()
        top = R.sample_fn(tok, ref_mask, sds['sampler'], 4, noise=R.SeededNoise(11, 'cpu'))
        ref_img, _ = R.refine_and_decode(odev(top), odev(ref_mask), od)
    assert torch.equal(model.segm_tokens.cpu(), tok)
    assert (result.cpu() - ref_img.cpu()).abs().max().item() < 2e-4
    # what the UI does with it (ui_demo.py:162-167)
    out = np.asarray((result.permute(0, 2, 3, 1).detach().cpu().numpy() * 255)[0], dtype=np.uint8)
    assert out.shape == (512, 256, 3)
