"""Default hyper-parameters of the two sampling configurations.

The reference ships them as YAML (configs/sample_from_parsing.yml:15-93,
configs/sample_from_pose.yml:91-107); users of the drop-in path keep using
their own YAML files.  These dicts exist so tests / bench.py / smoke() can run
on a box that has neither the reference nor its configs; `write_yaml` emits a
file that `options.parse` (and the reference's own parser) accept.
"""
from collections import OrderedDict

import yaml


def sample_from_parsing(ckpt_dir='./pretrained_models'):
    o = OrderedDict()
    o['name'] = 'sample_from_parsing'
    o['use_tb_logger'] = True
    o['set_CUDA_VISIBLE_DEVICES'] = None
    o['gpu_ids'] = [0]
    # dataset keys (ignored by the sampling scripts beyond directory names)
    o['batch_size'] = 4
    o['num_workers'] = 4
    o['test_img_dir'] = './datasets/test_images'
    o['segm_dir'] = './datasets/segm'
    o['pose_dir'] = './datasets/densepose'
    o['test_ann_file'] = './datasets/texture_ann/test'
    o['downsample_factor'] = 2
    o['model_type'] = 'SampleFromParsingModel'
    o['embed_dim'] = 256
    o['n_embed'] = 1024
    o['codebook_spatial_size'] = 2
    # bottom-level VQVAE
    o.update(bot_n_embed=512, bot_codebook_spatial_size=2, bot_double_z=False,
             bot_z_channels=256, bot_resolution=512, bot_in_channels=3,
             bot_out_ch=3, bot_ch=128, bot_ch_mult=[1, 1, 2, 4],
             bot_num_res_blocks=2, bot_attn_resolutions=[64], bot_dropout=0.0,
             bot_vae_path=f'{ckpt_dir}/vqvae_bottom.pth')
    # top-level VQGAN
    o.update(top_double_z=False, top_z_channels=256, top_resolution=512,
             top_in_channels=3, top_out_ch=3, top_ch=128,
             top_ch_mult=[1, 1, 2, 2, 4], top_num_res_blocks=2,
             top_attn_resolutions=[32], top_dropout=0.0,
             top_vae_path=f'{ckpt_dir}/vqvae_top.pth')
    # index-prediction UNet + 18-head FCN
    o.update(index_pred_encoder_in_channels=256, index_pred_fc_in_channels=64,
             index_pred_fc_in_index=4, index_pred_fc_channels=64,
             index_pred_fc_num_convs=1, index_pred_fc_concat_input=False,
             index_pred_fc_dropout_ratio=0.1, index_pred_fc_num_classes=512,
             index_pred_fc_align_corners=False,
             pretrained_index_network=f'{ckpt_dir}/index_pred_net.pth')
    # parsing-map tokenizer
    o.update(segm_double_z=False, segm_z_channels=32, segm_resolution=512,
             segm_in_channels=24, segm_out_ch=24, segm_ch=64,
             segm_ch_mult=[1, 1, 2, 2, 4], segm_num_res_blocks=1,
             segm_attn_resolutions=[16], segm_dropout=0.0,
             segm_num_segm_classes=24, segm_n_embed=1024, segm_embed_dim=32,
             segm_token_path=f'{ckpt_dir}/parsing_token.pth')
    # index sampler
    o.update(codebook_size=18432, segm_codebook_size=1024,
             texture_codebook_size=18, bert_n_emb=512, bert_n_layers=24,
             bert_n_head=8, block_size=512, latent_shape=[32, 16],
             embd_pdrop=0.0, resid_pdrop=0.0, attn_pdrop=0.0, num_head=18,
             pretrained_sampler=f'{ckpt_dir}/sampler.pth')
    o['manual_seed'] = 2021
    o['sample_steps'] = 256
    return o


def sample_from_pose(ckpt_dir='./pretrained_models'):
    o = sample_from_parsing(ckpt_dir)
    seed, steps = o.pop('manual_seed'), o.pop('sample_steps')
    o['name'] = 'sample_from_pose'
    o['model_type'] = 'SampleFromPoseModel'
    o.update(shape_embedder_dim=8, shape_embedder_out_dim=128,
             shape_attr_class_num=[2, 4, 6, 5, 4, 3, 5, 5, 3, 2, 2, 2, 2, 2, 2],
             shape_encoder_in_channels=1, shape_fc_in_channels=64,
             shape_fc_in_index=4, shape_fc_channels=64, shape_fc_num_convs=1,
             shape_fc_concat_input=False, shape_fc_dropout_ratio=0.1,
             shape_fc_num_classes=24, shape_fc_align_corners=False,
             pretrained_parsing_gen=f'{ckpt_dir}/parsing_gen.pth')
    o['manual_seed'] = seed
    o['sample_steps'] = steps
    return o


def index_pred_net(ckpt_dir='./pretrained_models'):
    """Network keys of the reference's configs/index_pred_net.yml (model_type
    VQGANTextureAwareSpatialHierarchyInferenceModel): ONLY the two VAE checkpoints."""
    o = OrderedDict()
    o['name'] = 'index_prediction_network'
    o['model_type'] = 'VQGANTextureAwareSpatialHierarchyInferenceModel'
    o.update(embed_dim=256, n_embed=1024, codebook_spatial_size=2)
    o.update(bot_n_embed=512, bot_double_z=False, bot_z_channels=256, bot_resolution=512, bot_in_channels=3,
             bot_out_ch=3, bot_ch=128, bot_ch_mult=[1, 1, 2, 4], bot_num_res_blocks=2, bot_attn_resolutions=[64],
             bot_dropout=0.0, bot_vae_path=f'{ckpt_dir}/vqvae_bottom.pth')
    o.update(top_double_z=False, top_z_channels=256, top_resolution=512, top_in_channels=3, top_out_ch=3,
             top_ch=128, top_ch_mult=[1, 1, 2, 2, 4], top_num_res_blocks=2, top_attn_resolutions=[32],
             top_dropout=0.0, top_vae_path=f'{ckpt_dir}/vqvae_top.pth')
    o.update(encoder_in_channels=256, fc_in_channels=64, fc_in_index=4, fc_channels=64, fc_num_convs=1,
             fc_concat_input=False, fc_dropout_ratio=0.1, fc_num_classes=512, fc_align_corners=False)
    o['manual_seed'] = 2021
    return o


def sampler(ckpt_dir='./pretrained_models'):
    """Network keys of the reference's configs/sampler.yml (model_type TransformerTextureAwareModel):
    img_ae_path / segm_ae_path + img_* / segm_* architecture keys.  `pretrained_sampler` (not in the
    reference YAML, which TRAINS the sampler) names the checkpoint train_sampler.py wrote, for the
    forward-only class of this package."""
    o = OrderedDict()
    o['name'] = 'sampler'
    o['model_type'] = 'TransformerTextureAwareModel'
    o.update(img_ae_path=f'{ckpt_dir}/vqvae_top.pth', segm_ae_path=f'{ckpt_dir}/parsing_token.pth')
    o.update(img_embed_dim=256, img_n_embed=1024, img_double_z=False, img_z_channels=256, img_resolution=512,
             img_in_channels=3, img_out_ch=3, img_ch=128, img_ch_mult=[1, 1, 2, 2, 4], img_num_res_blocks=2,
             img_attn_resolutions=[32], img_dropout=0.0)
    o.update(segm_double_z=False, segm_z_channels=32, segm_resolution=512, segm_in_channels=24, segm_out_ch=24,
             segm_ch=64, segm_ch_mult=[1, 1, 2, 2, 4], segm_num_res_blocks=1, segm_attn_resolutions=[16],
             segm_dropout=0.0, segm_num_segm_classes=24, segm_n_embed=1024, segm_embed_dim=32)
    o.update(codebook_size=18432, segm_codebook_size=1024, texture_codebook_size=18, bert_n_emb=512,
             bert_n_layers=24, bert_n_head=8, block_size=512, latent_shape=[32, 16], embd_pdrop=0.0,
             resid_pdrop=0.0, attn_pdrop=0.0, num_head=18, loss_type='reweighted_elbo', mask_schedule='random',
             sample_steps=256, pretrained_sampler=f'{ckpt_dir}/sampler.pth')
    o['manual_seed'] = 2021
    return o


def write_yaml(opt, path):
    with open(path, 'w') as f:
        yaml.safe_dump(dict(opt), f, sort_keys=False)
    return path
