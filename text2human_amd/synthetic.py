"""Synthetic checkpoints and inputs in the reference's exact on-disk layout.

No pretrained Text2Human weights or DeepFashion data exist offline, so every
test and benchmark runs on *synthetic* `.pth` files whose file names, top-level
keys and `state_dict` parameter names/shapes are the ones the reference loads
with ``strict=True`` (models/sample_model.py:124-181,397-410).
The schema is rebuilt here from the YAML hyper-parameters alone (this module
must work on the GPU box, where /root/reference does not exist);
``tests/test_schema.py`` pins it against ``tests/golden/state_dict_schema.json``
which was dumped from the reference's own constructors.

Values: seeded torch-CPU draws, same scale as the reference's default inits
(conv/linear U(+-1/sqrt(fan_in)), codebooks U(+-1/n_e) as vqgan_arch.py:36,
169,358) but with *non-trivial* norm affines / BN running stats so that a
kernel that forgets a scale, shift or running statistic fails parity.
"""
import math
import os
from collections import OrderedDict

import torch

# --------------------------------------------------------------------------
# schema builders: OrderedDict name -> (shape, kind)
# kinds: conv_w, lin_w, bias(fan_in), norm_w, norm_b, bn_mean, bn_var, bn_cnt,
#        emb, pos, codebook(n_e)
# --------------------------------------------------------------------------


def _conv(sd, name, cout, cin, k, bias=True):
    fan_in = cin * k * k
    sd[f'{name}.weight'] = ((cout, cin, k, k), ('uniform', 1.0 / math.sqrt(fan_in)))
    if bias:
        sd[f'{name}.bias'] = ((cout, ), ('uniform', 1.0 / math.sqrt(fan_in)))


def _linear(sd, name, cout, cin, bias=True):
    sd[f'{name}.weight'] = ((cout, cin), ('uniform', 1.0 / math.sqrt(cin)))
    if bias:
        sd[f'{name}.bias'] = ((cout, ), ('uniform', 1.0 / math.sqrt(cin)))


def _norm(sd, name, c):
    sd[f'{name}.weight'] = ((c, ), ('norm_w', None))
    sd[f'{name}.bias'] = ((c, ), ('norm_b', None))


def _resblock(sd, name, cin, cout):
    # ResnetBlock, models/archs/vqgan_arch.py:557-595 (temb_channels=0 here)
    _norm(sd, f'{name}.norm1', cin)
    _conv(sd, f'{name}.conv1', cout, cin, 3)
    _norm(sd, f'{name}.norm2', cout)
    _conv(sd, f'{name}.conv2', cout, cout, 3)
    if cin != cout:
        _conv(sd, f'{name}.nin_shortcut', cout, cin, 1)


def _attnblock(sd, name, c):
    # AttnBlock, models/archs/vqgan_arch.py:620-634
    _norm(sd, f'{name}.norm', c)
    for p in ('q', 'k', 'v', 'proj_out'):
        _conv(sd, f'{name}.{p}', c, c, 1)


def decoder_schema(ch, ch_mult, num_res_blocks, attn_resolutions, resolution,
                   z_channels, out_ch):
    """Decoder.__init__, models/archs/vqgan_arch.py:924-998."""
    sd = OrderedDict()
    nres = len(ch_mult)
    block_in = ch * ch_mult[nres - 1]
    curr_res = resolution // 2**(nres - 1)
    _conv(sd, 'conv_in', block_in, z_channels, 3)
    _resblock(sd, 'mid.block_1', block_in, block_in)
    _attnblock(sd, 'mid.attn_1', block_in)
    _resblock(sd, 'mid.block_2', block_in, block_in)
    levels = {}
    for i_level in reversed(range(nres)):
        lv = OrderedDict()
        block_out = ch * ch_mult[i_level]
        n_attn = 0
        for i_block in range(num_res_blocks + 1):
            _resblock(lv, f'block.{i_block}', block_in, block_out)
            block_in = block_out
            if curr_res in attn_resolutions:
                n_attn += 1
        for i in range(n_attn):
            _attnblock(lv, f'attn.{i}', block_in)
        if i_level != 0:
            _conv(lv, 'upsample.conv', block_in, block_in, 3)
            curr_res *= 2
        levels[i_level] = lv
    for i_level in range(nres):  # nn.ModuleList order after insert(0, ...)
        for k, v in levels[i_level].items():
            sd[f'up.{i_level}.{k}'] = v
    _norm(sd, 'norm_out', block_in)
    _conv(sd, 'conv_out', out_ch, block_in, 3)
    return sd


def decoder_res_schema(ch, ch_mult, z_channels):
    """DecoderRes.__init__, models/archs/vqgan_arch.py:1092-1134."""
    sd = OrderedDict()
    block_in = ch * ch_mult[-1]
    _conv(sd, 'conv_in', block_in, z_channels, 3)
    _resblock(sd, 'mid.block_1', block_in, block_in)
    _attnblock(sd, 'mid.attn_1', block_in)
    _resblock(sd, 'mid.block_2', block_in, block_in)
    return sd


def encoder_schema(ch, ch_mult, num_res_blocks, attn_resolutions, in_channels,
                   resolution, z_channels, double_z):
    """Encoder.__init__, models/archs/vqgan_arch.py:820-890."""
    sd = OrderedDict()
    _conv(sd, 'conv_in', ch, in_channels, 3)
    curr_res = resolution
    in_ch_mult = (1, ) + tuple(ch_mult)
    block_in = ch
    for i_level in range(len(ch_mult)):
        block_in = ch * in_ch_mult[i_level]
        block_out = ch * ch_mult[i_level]
        n_attn = 0
        for i_block in range(num_res_blocks):
            _resblock(sd, f'down.{i_level}.block.{i_block}', block_in, block_out)
            block_in = block_out
            if curr_res in attn_resolutions:
                n_attn += 1
        for i in range(n_attn):
            _attnblock(sd, f'down.{i_level}.attn.{i}', block_in)
        if i_level != len(ch_mult) - 1:
            _conv(sd, f'down.{i_level}.downsample.conv', block_in, block_in, 3)
            curr_res //= 2
    _resblock(sd, 'mid.block_1', block_in, block_in)
    _attnblock(sd, 'mid.attn_1', block_in)
    _resblock(sd, 'mid.block_2', block_in, block_in)
    _norm(sd, 'norm_out', block_in)
    _conv(sd, 'conv_out', 2 * z_channels if double_z else z_channels, block_in, 3)
    return sd


def codebook_list_schema(n_e, e_dim, n_books=18, spread=None):
    """spread None = the constructors' U(+-1/n_e) (vqgan_arch.py:169,358)."""
    sd = OrderedDict()
    for i in range(n_books):
        sd[f'embedding_list.{i}.weight'] = ((n_e, e_dim), ('uniform', spread or 1.0 / n_e))
    return sd


def _convmodule(sd, name, cout, cin, k):
    # mmcv ConvModule with norm_cfg=BN: conv without bias, then `bn`
    _conv(sd, f'{name}.conv', cout, cin, k, bias=False)
    sd[f'{name}.bn.weight'] = ((cout, ), ('norm_w', None))
    sd[f'{name}.bn.bias'] = ((cout, ), ('norm_b', None))
    sd[f'{name}.bn.running_mean'] = ((cout, ), ('bn_mean', None))
    sd[f'{name}.bn.running_var'] = ((cout, ), ('bn_var', None))
    sd[f'{name}.bn.num_batches_tracked'] = ((), ('bn_cnt', None))


def unet_schema(in_channels, base=64, num_stages=5, attr_embedding=0):
    """UNet / ShapeUNet constructors, models/archs/unet_arch.py:372-468,558-655.
    ModuleList registration order: per stage i the decoder block i-1 is created
    before encoder stage i, but state_dict() walks `encoder` then `decoder`."""
    enc, dec = OrderedDict(), OrderedDict()
    cin = in_channels
    for i in range(num_stages):
        cout = base * 2**i
        blk = 1 if i != 0 else 0  # a MaxPool2d sits at index 0 for stages >= 1
        _convmodule(enc, f'encoder.{i}.{blk}.convs.0', cout, cin + attr_embedding, 3)
        _convmodule(enc, f'encoder.{i}.{blk}.convs.1', cout, cout, 3)
        cin = cout
        if i != 0:
            skip = base * 2**(i - 1)
            d = i - 1
            _convmodule(dec, f'decoder.{d}.conv_block.convs.0', skip, 2 * skip, 3)
            _convmodule(dec, f'decoder.{d}.conv_block.convs.1', skip, skip, 3)
            _convmodule(dec, f'decoder.{d}.upsample.interp_upsample.1', skip, cout, 1)
    sd = OrderedDict()
    sd.update(enc)
    sd.update(dec)
    return sd


def multihead_fcn_schema(in_channels, channels, num_classes, num_head=18):
    """MultiHeadFCNHead.__init__ (num_convs=1, concat_input=False),
    models/archs/fcn_arch.py:240-331."""
    sd = OrderedDict()
    for h in range(num_head):
        sd[f'conv_seg_head_list.{h}.weight'] = ((num_classes, channels, 1, 1),
                                                ('uniform', 1.0 / math.sqrt(channels)))
        sd[f'conv_seg_head_list.{h}.bias'] = ((num_classes, ),
                                              ('uniform', 1.0 / math.sqrt(channels)))
    for h in range(num_head):
        _convmodule(sd, f'convs_list.{h}.0', channels, in_channels, 3)
    return sd


def fcn_head_schema(in_channels, channels, num_classes):
    """FCNHead.__init__ (num_convs=1, concat_input=False), fcn_arch.py:159-216."""
    sd = OrderedDict()
    _conv(sd, 'conv_seg', num_classes, channels, 1)
    _convmodule(sd, 'convs.0', channels, in_channels, 3)
    return sd


def transformer_schema(codebook_size, segm_codebook_size, texture_codebook_size,
                       n_emb, n_layers, block_size, num_head):
    """TransformerMultiHead.__init__, models/archs/transformer_arch.py:187-235."""
    sd = OrderedDict()
    sd['pos_emb'] = ((1, block_size, n_emb), ('normal', 0.02))
    sd['start_tok'] = ((1, 1, n_emb), ('normal', 0.02))
    sd['tok_emb.weight'] = ((codebook_size + 1, n_emb), ('normal', 1.0))
    sd['segm_emb.weight'] = ((segm_codebook_size, n_emb), ('normal', 1.0))
    sd['texture_emb.weight'] = ((texture_codebook_size, n_emb), ('normal', 1.0))
    for i in range(n_layers):
        p = f'blocks.{i}'
        _norm(sd, f'{p}.ln1', n_emb)
        _norm(sd, f'{p}.ln2', n_emb)
        for nm in ('key', 'query', 'value', 'proj'):
            _linear(sd, f'{p}.attn.{nm}', n_emb, n_emb)
        _linear(sd, f'{p}.mlp.0', 4 * n_emb, n_emb)
        _linear(sd, f'{p}.mlp.2', n_emb, 4 * n_emb)
    _norm(sd, 'ln_f', n_emb)
    for h in range(num_head):
        _linear(sd, f'head_list.{h}', codebook_size // num_head, n_emb, bias=False)
    return sd


def shape_embedder_schema(dim, out_dim, cls_num_list):
    """ShapeAttrEmbedding.__init__, shape_attr_embedding_arch.py:8-21."""
    sd = OrderedDict()
    for i, c in enumerate(cls_num_list):
        _linear(sd, f'attr_{i}.0', dim, c)
        _linear(sd, f'attr_{i}.2', dim, dim)
    _linear(sd, 'fusion.0', out_dim, dim * len(cls_num_list))
    _linear(sd, 'fusion.2', out_dim, out_dim)
    return sd


# --------------------------------------------------------------------------
# seeded fill
# --------------------------------------------------------------------------


def fill(schema, seed):
    g = torch.Generator(device='cpu')
    g.manual_seed(int(seed))
    out = OrderedDict()
    for name, (shape, (kind, arg)) in schema.items():
        if kind == 'uniform':
            t = (torch.rand(shape, generator=g) * 2 - 1) * arg
        elif kind == 'normal':
            t = torch.randn(shape, generator=g) * arg
        elif kind == 'norm_w':
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif kind == 'norm_b':
            t = 0.1 * torch.randn(shape, generator=g)
        elif kind == 'bn_mean':
            t = 0.1 * torch.randn(shape, generator=g)
        elif kind == 'bn_var':
            t = 1.0 + 0.2 * torch.rand(shape, generator=g)
        elif kind == 'bn_cnt':
            t = torch.tensor(100, dtype=torch.long)
        else:
            raise KeyError(kind)
        out[name] = t
    return out


def module_schemas(opt, encode=False):
    """All state_dict schemas the sampling path consumes, keyed like App. B.

    encode=True adds the encode side of the hierarchy (top / bottom
    Encoder + quant_conv, hierarchy_inference_model.py:28-37,63-87) and draws the texture
    codebooks with a trained-like spread U(+-1) so that the L2-argmin is decided far above
    fp32 rounding (the segm codebook below does the same)."""
    s = OrderedDict()
    cb_spread = 1.0 if encode else None
    s['decoder'] = decoder_schema(opt['top_ch'], opt['top_ch_mult'],
                                  opt['top_num_res_blocks'],
                                  opt['top_attn_resolutions'],
                                  opt['top_resolution'], opt['top_z_channels'],
                                  opt['top_out_ch'])
    s['top_quantize'] = codebook_list_schema(1024, opt['embed_dim'], spread=cb_spread)
    s['top_post_quant_conv'] = OrderedDict()
    _conv(s['top_post_quant_conv'], '', opt['top_z_channels'], opt['embed_dim'], 1)
    s['bot_decoder_res'] = decoder_res_schema(opt['bot_ch'], opt['bot_ch_mult'],
                                              opt['bot_z_channels'])
    sp = opt['bot_codebook_spatial_size']
    s['bot_quantize'] = codebook_list_schema(opt['bot_n_embed'],
                                             opt['embed_dim'] * sp * sp, spread=cb_spread)
    s['bot_post_quant_conv'] = OrderedDict()
    _conv(s['bot_post_quant_conv'], '', opt['bot_z_channels'], opt['embed_dim'], 1)
    s['segm_encoder'] = encoder_schema(opt['segm_ch'], opt['segm_ch_mult'],
                                       opt['segm_num_res_blocks'],
                                       opt['segm_attn_resolutions'],
                                       opt['segm_in_channels'],
                                       opt['segm_resolution'],
                                       opt['segm_z_channels'],
                                       opt['segm_double_z'])
    s['segm_quantizer'] = OrderedDict(
        # U(+-1) instead of the constructor's U(+-1/n_e): a spread like a trained
        # codebook, so the argmin is decided far above fp32 rounding (with the
        # init-time spread ~1e-3 the best/second-best gap is ~1 ulp of |z|^2).
        [('embedding.weight', ((opt['segm_n_embed'], opt['segm_embed_dim']),
                               ('uniform', 1.0)))])
    s['segm_quant_conv'] = OrderedDict()
    _conv(s['segm_quant_conv'], '', opt['segm_embed_dim'], opt['segm_z_channels'], 1)
    s['guidance_encoder'] = unet_schema(opt['index_pred_encoder_in_channels'])
    s['index_decoder'] = multihead_fcn_schema(opt['index_pred_fc_in_channels'],
                                              opt['index_pred_fc_channels'],
                                              opt['index_pred_fc_num_classes'])
    s['sampler'] = transformer_schema(opt['codebook_size'],
                                      opt['segm_codebook_size'],
                                      opt['texture_codebook_size'],
                                      opt['bert_n_emb'], opt['bert_n_layers'],
                                      opt['block_size'], opt['num_head'])
    if opt.get('shape_attr_class_num') is not None:
        s['shape_embedder'] = shape_embedder_schema(opt['shape_embedder_dim'],
                                                    opt['shape_embedder_out_dim'],
                                                    opt['shape_attr_class_num'])
        s['shape_encoder'] = unet_schema(opt['shape_encoder_in_channels'],
                                         attr_embedding=opt['shape_embedder_out_dim'])
        s['shape_decoder'] = fcn_head_schema(opt['shape_fc_in_channels'],
                                             opt['shape_fc_channels'],
                                             opt['shape_fc_num_classes'])
    one_by_one = ['top_post_quant_conv', 'bot_post_quant_conv', 'segm_quant_conv']
    if encode:
        s['top_encoder'] = encoder_schema(opt['top_ch'], opt['top_ch_mult'], opt['top_num_res_blocks'],
                                          opt['top_attn_resolutions'], opt['top_in_channels'],
                                          opt['top_resolution'], opt['top_z_channels'], opt['top_double_z'])
        s['bot_encoder'] = encoder_schema(opt['bot_ch'], opt['bot_ch_mult'], opt['bot_num_res_blocks'],
                                          opt['bot_attn_resolutions'], opt['bot_in_channels'],
                                          opt['bot_resolution'], opt['bot_z_channels'], opt['bot_double_z'])
        for k, zc in (('top_quant_conv', opt['top_z_channels']), ('bot_quant_conv', opt['bot_z_channels'])):
            s[k] = OrderedDict()
            _conv(s[k], '', opt['embed_dim'], zc, 1)
            one_by_one.append(k)
    for k in one_by_one:
        s[k] = OrderedDict((n.lstrip('.'), v) for n, v in s[k].items())
    return s


def _one_by_one(cout, cin):
    s = OrderedDict()
    _conv(s, '', cout, cin, 1)
    return OrderedDict((n.lstrip('.'), v) for n, v in s.items())


def hierarchy_schemas(opt):
    """state_dict schemas of VQGANTextureAwareSpatialHierarchyInferenceModel from the key set of
    the reference's configs/index_pred_net.yml (hierarchy_inference_model.py:28-129): the two VAEs only."""
    s = OrderedDict()
    sp = opt['codebook_spatial_size'] or opt['bot_codebook_spatial_size']
    s['top_encoder'] = encoder_schema(opt['top_ch'], opt['top_ch_mult'], opt['top_num_res_blocks'],
                                      opt['top_attn_resolutions'], opt['top_in_channels'], opt['top_resolution'],
                                      opt['top_z_channels'], opt['top_double_z'])
    s['decoder'] = decoder_schema(opt['top_ch'], opt['top_ch_mult'], opt['top_num_res_blocks'],
                                  opt['top_attn_resolutions'], opt['top_resolution'], opt['top_z_channels'],
                                  opt['top_out_ch'])
    s['top_quantize'] = codebook_list_schema(opt['n_embed'], opt['embed_dim'], spread=1.0)
    s['top_quant_conv'] = _one_by_one(opt['embed_dim'], opt['top_z_channels'])
    s['top_post_quant_conv'] = _one_by_one(opt['top_z_channels'], opt['embed_dim'])
    s['bot_encoder'] = encoder_schema(opt['bot_ch'], opt['bot_ch_mult'], opt['bot_num_res_blocks'],
                                      opt['bot_attn_resolutions'], opt['bot_in_channels'], opt['bot_resolution'],
                                      opt['bot_z_channels'], opt['bot_double_z'])
    s['bot_decoder_res'] = decoder_res_schema(opt['bot_ch'], opt['bot_ch_mult'], opt['bot_z_channels'])
    s['bot_quantize'] = codebook_list_schema(opt['bot_n_embed'], opt['embed_dim'] * sp * sp, spread=1.0)
    s['bot_quant_conv'] = _one_by_one(opt['embed_dim'], opt['bot_z_channels'])
    s['bot_post_quant_conv'] = _one_by_one(opt['bot_z_channels'], opt['embed_dim'])
    return s


def transformer_model_schemas(opt):
    """state_dict schemas of TransformerTextureAwareModel from the key set of the reference's
    configs/sampler.yml (transformer_model.py:27-99): image VAE (img_*), parsing tokenizer (segm_*),
    the sampler."""
    s = OrderedDict()
    s['img_encoder'] = encoder_schema(opt['img_ch'], opt['img_ch_mult'], opt['img_num_res_blocks'],
                                      opt['img_attn_resolutions'], opt['img_in_channels'], opt['img_resolution'],
                                      opt['img_z_channels'], opt['img_double_z'])
    s['img_decoder'] = decoder_schema(opt['img_ch'], opt['img_ch_mult'], opt['img_num_res_blocks'],
                                      opt['img_attn_resolutions'], opt['img_resolution'], opt['img_z_channels'],
                                      opt['img_out_ch'])
    s['img_quantizer'] = codebook_list_schema(opt['img_n_embed'], opt['img_embed_dim'], spread=1.0)
    s['img_quant_conv'] = _one_by_one(opt['img_embed_dim'], opt['img_z_channels'])
    s['img_post_quant_conv'] = _one_by_one(opt['img_z_channels'], opt['img_embed_dim'])
    s['segm_encoder'] = encoder_schema(opt['segm_ch'], opt['segm_ch_mult'], opt['segm_num_res_blocks'],
                                       opt['segm_attn_resolutions'], opt['segm_in_channels'], opt['segm_resolution'],
                                       opt['segm_z_channels'], opt['segm_double_z'])
    s['segm_quantizer'] = OrderedDict([('embedding.weight', ((opt['segm_n_embed'], opt['segm_embed_dim']),
                                                            ('uniform', 1.0)))])
    s['segm_quant_conv'] = _one_by_one(opt['segm_embed_dim'], opt['segm_z_channels'])
    s['sampler'] = transformer_schema(opt['codebook_size'], opt['segm_codebook_size'], opt['texture_codebook_size'],
                                      opt['bert_n_emb'], opt['bert_n_layers'], opt['block_size'], opt['num_head'])
    return s


_SEEDS = dict(decoder=11, top_quantize=12, top_post_quant_conv=13,
              bot_decoder_res=21, bot_quantize=22, bot_post_quant_conv=23,
              segm_encoder=31, segm_quantizer=32, segm_quant_conv=33,
              guidance_encoder=41, index_decoder=42, sampler=51,
              shape_embedder=61, shape_encoder=62, shape_decoder=63,
              top_encoder=71, top_quant_conv=72, bot_encoder=73, bot_quant_conv=74,
              img_encoder=71, img_decoder=11, img_quantizer=12, img_quant_conv=72, img_post_quant_conv=13)


def write_hierarchy_checkpoints(opt, out_dir, seed=1234):
    """vqvae_top.pth / vqvae_bottom.pth as the reference's train_vqvae_* scripts leave them, for a
    configs/index_pred_net.yml-style `opt`; returns the opt copy pointing at them."""
    os.makedirs(out_dir, exist_ok=True)
    sds = {n: fill(sc, seed * 1000 + _SEEDS[n]) for n, sc in hierarchy_schemas(opt).items()}
    files = {
        'top_vae_path': ('vqvae_top.pth', dict(encoder=sds['top_encoder'], decoder=sds['decoder'],
                                               quantize=sds['top_quantize'], quant_conv=sds['top_quant_conv'],
                                               post_quant_conv=sds['top_post_quant_conv'])),
        'bot_vae_path': ('vqvae_bottom.pth', dict(bot_encoder=sds['bot_encoder'],
                                                  bot_decoder_res=sds['bot_decoder_res'], decoder=sds['decoder'],
                                                  bot_quantize=sds['bot_quantize'],
                                                  bot_quant_conv=sds['bot_quant_conv'],
                                                  bot_post_quant_conv=sds['bot_post_quant_conv'])),
    }
    new_opt = type(opt)(opt)
    for key, (fname, payload) in files.items():
        torch.save(payload, os.path.join(out_dir, fname))
        new_opt[key] = os.path.join(out_dir, fname)
    return new_opt, sds


def write_transformer_checkpoints(opt, out_dir, seed=1234):
    """vqvae_top.pth / parsing_token.pth / sampler.pth for a configs/sampler.yml-style `opt`."""
    os.makedirs(out_dir, exist_ok=True)
    sds = {n: fill(sc, seed * 1000 + _SEEDS[n]) for n, sc in transformer_model_schemas(opt).items()}
    files = {
        'img_ae_path': ('vqvae_top.pth', dict(encoder=sds['img_encoder'], decoder=sds['img_decoder'],
                                              quantize=sds['img_quantizer'], quant_conv=sds['img_quant_conv'],
                                              post_quant_conv=sds['img_post_quant_conv'])),
        'segm_ae_path': ('parsing_token.pth', dict(encoder=sds['segm_encoder'], quantize=sds['segm_quantizer'],
                                                   quant_conv=sds['segm_quant_conv'])),
        'pretrained_sampler': ('sampler.pth', sds['sampler']),
    }
    new_opt = type(opt)(opt)
    for key, (fname, payload) in files.items():
        torch.save(payload, os.path.join(out_dir, fname))
        new_opt[key] = os.path.join(out_dir, fname)
    return new_opt, sds


def make_state_dicts(opt, seed=1234, head_scale=1.0, argmax_scale=1.0, encode=False):
    """Dict module-name -> state_dict (fp32 CPU tensors).

    head_scale / argmax_scale multiply the sampler heads / index-prediction
    heads to get peaked logits ("peaked-logits variant")."""
    sds = OrderedDict()
    for name, schema in module_schemas(opt, encode=encode).items():
        sds[name] = fill(schema, seed * 1000 + _SEEDS[name])
    if head_scale != 1.0:
        for k in sds['sampler']:
            if k.startswith('head_list.'):
                sds['sampler'][k] = sds['sampler'][k] * head_scale
    if argmax_scale != 1.0:
        for k in sds['index_decoder']:
            if k.startswith('conv_seg_head_list.'):
                sds['index_decoder'][k] = sds['index_decoder'][k] * argmax_scale
    return sds


def write_checkpoints(opt, out_dir, seed=1234, state_dicts=None, **kw):
    """Writes the 5 (+1 for pose) `.pth` files in the reference layout
    (SURVEY.md section 5 "Checkpoint / resume") and returns an opt copy whose
    *_path keys point at them.  `state_dicts`: write these instead of make_state_dicts(opt, seed, **kw)."""
    os.makedirs(out_dir, exist_ok=True)
    sds = state_dicts if state_dicts is not None else make_state_dicts(opt, seed, **kw)
    files = {
        'top_vae_path': ('vqvae_top.pth', dict(decoder=sds['decoder'],
                                               quantize=sds['top_quantize'],
                                               post_quant_conv=sds['top_post_quant_conv'])),
        'bot_vae_path': ('vqvae_bottom.pth', dict(bot_decoder_res=sds['bot_decoder_res'],
                                                  decoder=sds['decoder'],
                                                  bot_quantize=sds['bot_quantize'],
                                                  bot_post_quant_conv=sds['bot_post_quant_conv'])),
        'segm_token_path': ('parsing_token.pth', dict(encoder=sds['segm_encoder'],
                                                      quantize=sds['segm_quantizer'],
                                                      quant_conv=sds['segm_quant_conv'])),
        'pretrained_index_network': ('index_pred_net.pth', dict(
            guidance_encoder=sds['guidance_encoder'],
            index_decoder=sds['index_decoder'])),
        'pretrained_sampler': ('sampler.pth', sds['sampler']),
    }
    if 'top_encoder' in sds:  # encode side: the real checkpoints carry these keys too
        files['top_vae_path'][1].update(encoder=sds['top_encoder'], quant_conv=sds['top_quant_conv'])
        files['bot_vae_path'][1].update(bot_encoder=sds['bot_encoder'], bot_quant_conv=sds['bot_quant_conv'])
    if 'shape_embedder' in sds:
        files['pretrained_parsing_gen'] = ('parsing_gen.pth', dict(
            embedder=sds['shape_embedder'], encoder=sds['shape_encoder'],
            decoder=sds['shape_decoder']))
    new_opt = type(opt)(opt)
    for key, (fname, payload) in files.items():
        path = os.path.join(out_dir, fname)
        torch.save(payload, path)
        new_opt[key] = path
    return new_opt


# --------------------------------------------------------------------------
# synthetic inputs (SURVEY.md 8(d) "Synthetic inputs")
# --------------------------------------------------------------------------

UPPER_CLS = (1, 4)
LOWER_CLS = (3, 5, 21)
OUTER_CLS = (2, )


def texture_mask_from_segm(segm, upper, lower, outer):
    """Mask rule of data/segm_attr_dataset.py:140-151 / sample_model.py:443-467:
    0 = common codebook, attr+1 = texture codebook; attr 17 = "none"."""
    mask = torch.zeros_like(segm)
    for b in range(segm.shape[0]):
        for attr, classes in ((upper[b], UPPER_CLS), (lower[b], LOWER_CLS),
                              (outer[b], OUTER_CLS)):
            if int(attr) != 17:
                for c in classes:
                    mask[b][segm[b] == c] = float(int(attr) + 1)
    return mask


def parsing_batch(batch, seed=2021, height=512, width=256, cell=16):
    g = torch.Generator(device='cpu')
    g.manual_seed(int(seed))
    coarse = torch.randint(0, 24, (batch, 1, height // cell, width // cell), generator=g)
    segm = coarse.repeat_interleave(cell, 2).repeat_interleave(cell, 3).float()
    attrs = torch.randint(0, 18, (3, batch), generator=g)
    mask = texture_mask_from_segm(segm, attrs[0], attrs[1], attrs[2])
    return dict(segm=segm, texture_mask=mask,
                img_name=[f'{i}.png' for i in range(batch)])


POSE_CLS_NUM = (2, 4, 6, 5, 4, 3, 5, 5, 3, 2, 2, 2, 2, 2, 2)


def pose_batch(batch, seed=2021, height=512, width=256, cls_num=POSE_CLS_NUM):
    g = torch.Generator(device='cpu')
    g.manual_seed(int(seed))
    dp = torch.randint(0, 25, (batch, 1, height // 16, width // 16), generator=g)
    dp = dp.repeat_interleave(16, 2).repeat_interleave(16, 3).float() / 12.0 - 1.0
    shape_attr = torch.stack(
        [torch.randint(0, c, (batch, ), generator=g) for c in cls_num], 1)
    attrs = torch.randint(0, 18, (3, batch), generator=g)
    return dict(densepose=dp, shape_attr=shape_attr, upper_fused_attr=attrs[0],
                lower_fused_attr=attrs[1], outer_fused_attr=attrs[2],
                img_name=[f'{i}.png' for i in range(batch)])


def write_dataset_tree(root, n=3, seed=7, height=1024, width=512):
    """A tiny DeepFashion-MultiModal style tree (full-resolution PNGs, the three fused
    texture annotation files and the shape annotation file) for the dataset / entry-point
    tests.  Returns the option keys the sampling YAMLs use for it."""
    from PIL import Image
    import numpy as np
    rng = np.random.default_rng(seed)
    dirs = {k: os.path.join(root, k) for k in ('images', 'segm', 'densepose', 'texture_ann', 'shape_ann')}
    for d in dirs.values():
        os.makedirs(d, exist_ok=True)
    names = [f'MEN-Tees_{i:03d}-id_{seed:04d}_{i}_front.jpg' for i in range(n)]
    rows = {'upper': [], 'lower': [], 'outer': []}
    shape_rows = []
    for i, name in enumerate(names):
        stem = name[:-4]
        img = rng.integers(0, 256, (height, width, 3), dtype=np.uint8)
        Image.fromarray(img).save(os.path.join(dirs['images'], name.replace('.jpg', '.png')))
        Image.fromarray(img).save(os.path.join(dirs['images'], name), quality=95)
        cells = rng.integers(0, 24, (height // 32, width // 32), dtype=np.uint8)
        segm = np.kron(cells, np.ones((32, 32), dtype=np.uint8))
        Image.fromarray(segm).save(os.path.join(dirs['segm'], f'{stem}_segm.png'))
        dp = np.zeros((height, width, 3), dtype=np.uint8)
        dp[:, :, 2] = np.kron(rng.integers(0, 25, (height // 16, width // 16), dtype=np.uint8),
                              np.ones((16, 16), dtype=np.uint8))
        dp[:, :, :2] = rng.integers(0, 256, (height, width, 2), dtype=np.uint8)
        Image.fromarray(dp).save(os.path.join(dirs['densepose'], f'{stem}_densepose.png'))
        for part in rows:
            rows[part].append(f'{name} {int(rng.integers(0, 18))}')
        shape_rows.append(name + ' ' + ' '.join(str(int(rng.integers(0, c))) for c in POSE_CLS_NUM))
    for part, lines in rows.items():
        with open(os.path.join(dirs['texture_ann'], f'{part}_fused.txt'), 'w') as f:
            f.write('\n'.join(lines) + '\n')
    shape_path = os.path.join(dirs['shape_ann'], 'test_ann_file.txt')
    with open(shape_path, 'w') as f:
        f.write('\n'.join(shape_rows) + '\n')
    return dict(test_img_dir=dirs['images'], segm_dir=dirs['segm'], pose_dir=dirs['densepose'],
                test_ann_file=dirs['texture_ann'], texture_ann_file=dirs['texture_ann'],
                shape_ann_path=shape_path, names=names)
