"""Each model class builds from ITS OWN reference YAML key set: the hierarchy model
from configs/index_pred_net.yml's keys (top_vae_path, bot_vae_path only), the sampler training-forward
model from configs/sampler.yml's (img_ae_path, segm_ae_path, img_* keys).  CPU: the loaders and their
strict validation; GPU (-m gpu): the classes constructed from the written .pth files."""
import pytest
import torch

from text2human_amd import defaults, options, synthetic, weights


def test_index_pred_net_yaml_keys_load_only_the_two_vaes(tmp_path):
    opt = options.dict_to_nonedict(defaults.index_pred_net())
    for absent in ('segm_token_path', 'pretrained_index_network', 'pretrained_sampler', 'bot_codebook_spatial_size'):
        assert opt[absent] is None
    o, sds = synthetic.write_hierarchy_checkpoints(opt, str(tmp_path))
    got = weights.load_hierarchy_checkpoints(o)
    assert set(got) == {'top_encoder', 'decoder', 'top_quantize', 'top_quant_conv', 'top_post_quant_conv',
                        'bot_encoder', 'bot_decoder_res', 'bot_quantize', 'bot_quant_conv', 'bot_post_quant_conv'}
    assert all(torch.equal(got[m][k], sds[m][k]) for m in got for k in got[m])
    # strict: a missing parameter in the file is an error, like load_state_dict(strict=True)
    bad = torch.load(o['bot_vae_path'], weights_only=False)
    bad['bot_encoder'].pop('conv_in.weight')
    torch.save(bad, o['bot_vae_path'])
    with pytest.raises(RuntimeError, match='Missing key'):
        weights.load_hierarchy_checkpoints(o)


def test_sampler_yaml_keys_load_image_vae_tokenizer_and_sampler(tmp_path):
    opt = options.dict_to_nonedict(defaults.sampler())
    assert opt['top_vae_path'] is None and opt['bot_vae_path'] is None and opt['top_ch'] is None
    o, sds = synthetic.write_transformer_checkpoints(opt, str(tmp_path))
    got = weights.load_transformer_checkpoints(o)
    assert set(got) == {'top_encoder', 'top_quantize', 'top_quant_conv', 'segm_encoder', 'segm_quantizer',
                        'segm_quant_conv', 'sampler'}
    assert torch.equal(got['top_encoder']['conv_in.weight'], sds['img_encoder']['conv_in.weight'])
    o2 = type(o)(o)
    o2['pretrained_sampler'] = None
    with pytest.raises(KeyError, match='pretrained_sampler'):
        weights.load_transformer_checkpoints(o2)


@pytest.mark.gpu
def test_models_construct_from_their_own_yaml_and_files(tmp_path):
    from text2human_amd.models import TransformerTextureAwareModel, VQGANTextureAwareSpatialHierarchyInferenceModel
    o, _ = synthetic.write_hierarchy_checkpoints(options.dict_to_nonedict(defaults.index_pred_net()),
                                                 str(tmp_path / 'h'))
    m = VQGANTextureAwareSpatialHierarchyInferenceModel(o)
    img = torch.rand(1, 3, 512, 256) * 2 - 1
    mask = synthetic.parsing_batch(1, seed=3)['texture_mask']
    m.feed_data(dict(image=img, texture_mask=mask))
    assert len(m.gt_indices_list) == 18 and m.spatial == 2
    o, _ = synthetic.write_transformer_checkpoints(options.dict_to_nonedict(defaults.sampler()), str(tmp_path / 't'))
    t = TransformerTextureAwareModel(o)
    assert t.mask_id == 18432 and t.shape == (32, 16)
