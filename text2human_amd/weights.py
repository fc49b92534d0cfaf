"""Checkpoint loading + one-time repack into kernel-friendly HBM layouts.

Accepts the reference's `.pth` files unchanged (file names from the YAML, dict
keys and parameter names of models/sample_model.py:124-181,397-410) and
validates them the way `load_state_dict(strict=True)` would: a missing /
unexpected key or a shape mismatch raises RuntimeError.

Repacks (device resident, done once at model construction; fp32, plus the two-fp16-plane "split
row" copies of the sampler Linears and of the decoders' convolutions that the split-precision
kernels multiply -- `pack_transformer`, `add_split_conv_weights`; the x8 copies of the sampler Linears
are packed by engine.SamplerNet.calibrate_x8 with each matrix's own power-of-two scale):
  * 3x3 conv  [Cout,Cin,3,3] -> [Cout, 9*Cin']  K order [tap(dy,dx)][cin],
    Cin' = Cin rounded up to a multiple of 32 (zero columns);
  * 1x1 conv  [Cout,Cin,1,1] -> [Cout, Cin];
  * mmcv ConvModule (conv no-bias + BN eval + ReLU): BN folded into the conv
    weight rows and a bias;
  * AttnBlock q/k/v and transformer query/key/value -> one [3C, C] matrix;
  * 18 codebooks / 18 sampler heads / 18 index heads stacked into one tensor.
"""
import os

import torch

from . import synthetic


def _round32(c):
    return (c + 31) // 32 * 32


def check_state_dict(sd, schema, what):
    """strict=True semantics of nn.Module.load_state_dict."""
    missing = [k for k in schema if k not in sd]
    unexpected = [k for k in sd if k not in schema]
    errs = []
    if missing:
        errs.append('Missing key(s) in state_dict: ' + ', '.join(f'"{k}"' for k in missing[:8]))
    if unexpected:
        errs.append('Unexpected key(s) in state_dict: ' + ', '.join(f'"{k}"' for k in unexpected[:8]))
    for k, (shape, _) in schema.items():
        if k in sd and tuple(sd[k].shape) != tuple(shape):
            errs.append(f'size mismatch for {k}: copying a param with shape {tuple(sd[k].shape)} '
                        f'from checkpoint, the shape in current model is {tuple(shape)}.')
    if errs:
        raise RuntimeError(f'Error(s) in loading state_dict for {what}:\n\t' + '\n\t'.join(errs))


def pack_conv3x3(w, cin_pad=None):
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3
    cp = cin_pad or _round32(cin)
    out = torch.zeros((cout, 9, cp), dtype=torch.float32)
    out[:, :, :cin] = w.float().permute(0, 2, 3, 1).reshape(cout, 9, cin)
    return out.reshape(cout, 9 * cp).contiguous()


def pack_conv1x1(w):
    return w.float().reshape(w.shape[0], w.shape[1]).contiguous()


def fold_bn(sd, name, eps=1e-5):
    """ConvModule `name` -> (weight [Cout,Cin,k,k] scaled, bias [Cout])."""
    w = sd[f'{name}.conv.weight'].double()
    g, b = sd[f'{name}.bn.weight'].double(), sd[f'{name}.bn.bias'].double()
    mu, var = sd[f'{name}.bn.running_mean'].double(), sd[f'{name}.bn.running_var'].double()
    s = g / torch.sqrt(var + eps)
    return (w * s.view(-1, 1, 1, 1)).float(), (b - mu * s).float()


class Params:
    """Bag of device tensors addressed by name."""

    def __init__(self, device):
        self.device = device
        self.t = {}

    def put(self, name, tensor):
        self.t[name] = tensor.to(self.device, torch.float32).contiguous()

    def __getitem__(self, name):
        return self.t[name]

    def __contains__(self, name):
        return name in self.t

    def nbytes(self):
        return sum(v.numel() * v.element_size() for v in self.t.values())


def _pack_resblock(P, sd, src, dst):
    for n in ('norm1', 'norm2'):
        P.put(f'{dst}.{n}.g', sd[f'{src}.{n}.weight'])
        P.put(f'{dst}.{n}.b', sd[f'{src}.{n}.bias'])
    for n in ('conv1', 'conv2'):
        P.put(f'{dst}.{n}.w', pack_conv3x3(sd[f'{src}.{n}.weight']))
        P.put(f'{dst}.{n}.b', sd[f'{src}.{n}.bias'])
    if f'{src}.nin_shortcut.weight' in sd:
        P.put(f'{dst}.nin.w', pack_conv1x1(sd[f'{src}.nin_shortcut.weight']))
        P.put(f'{dst}.nin.b', sd[f'{src}.nin_shortcut.bias'])


def _pack_attnblock(P, sd, src, dst):
    P.put(f'{dst}.norm.g', sd[f'{src}.norm.weight'])
    P.put(f'{dst}.norm.b', sd[f'{src}.norm.bias'])
    P.put(f'{dst}.qkv.w', torch.cat([pack_conv1x1(sd[f'{src}.{n}.weight']) for n in 'qkv'], 0))
    P.put(f'{dst}.qkv.b', torch.cat([sd[f'{src}.{n}.bias'] for n in 'qkv'], 0))
    P.put(f'{dst}.proj.w', pack_conv1x1(sd[f'{src}.proj_out.weight']))
    P.put(f'{dst}.proj.b', sd[f'{src}.proj_out.bias'])


def _conv_in_out(P, sd, src, dst):
    P.put(f'{dst}.w', pack_conv3x3(sd[f'{src}.weight']))
    P.put(f'{dst}.b', sd[f'{src}.bias'])


def _count(sd, prefix):
    p = prefix + '.'
    return len({int(k[len(p):].split('.')[0]) for k in sd if k.startswith(p)})


def pack_vqgan(P, sd, dst):
    """Encoder / Decoder / DecoderRes state_dict -> Params under `dst.`;
    returns a structure description the engine walks."""
    _conv_in_out(P, sd, 'conv_in', f'{dst}.conv_in')
    for blk in ('mid.block_1', 'mid.block_2'):
        _pack_resblock(P, sd, blk, f'{dst}.{blk}')
    _pack_attnblock(P, sd, 'mid.attn_1', f'{dst}.mid.attn_1')
    desc = dict(levels=[])
    for kind in ('down', 'up'):
        n = _count(sd, kind)
        for lv in range(n):
            nb = _count(sd, f'{kind}.{lv}.block')
            blocks = []
            for b in range(nb):
                _pack_resblock(P, sd, f'{kind}.{lv}.block.{b}', f'{dst}.{kind}.{lv}.block.{b}')
                has_attn = f'{kind}.{lv}.attn.{b}.norm.weight' in sd
                if has_attn:
                    _pack_attnblock(P, sd, f'{kind}.{lv}.attn.{b}', f'{dst}.{kind}.{lv}.attn.{b}')
                cin = sd[f'{kind}.{lv}.block.{b}.conv1.weight'].shape[1]
                cout = sd[f'{kind}.{lv}.block.{b}.conv1.weight'].shape[0]
                blocks.append(dict(cin=cin, cout=cout, attn=has_attn))
            res = 'downsample' if kind == 'down' else 'upsample'
            has_res = f'{kind}.{lv}.{res}.conv.weight' in sd
            if has_res:
                _conv_in_out(P, sd, f'{kind}.{lv}.{res}.conv', f'{dst}.{kind}.{lv}.{res}')
            desc['levels'].append(dict(kind=kind, level=lv, blocks=blocks, resample=has_res))
    if 'norm_out.weight' in sd:
        P.put(f'{dst}.norm_out.g', sd['norm_out.weight'])
        P.put(f'{dst}.norm_out.b', sd['norm_out.bias'])
        _conv_in_out(P, sd, 'conv_out', f'{dst}.conv_out')
    desc['cin'] = sd['conv_in.weight'].shape[1]
    desc['c0'] = sd['conv_in.weight'].shape[0]
    desc['cmid'] = sd['mid.block_1.conv1.weight'].shape[0]
    desc['cout'] = sd['conv_out.weight'].shape[0] if 'conv_out.weight' in sd else None
    return desc


def add_split_conv_weights(P, dst):
    """Split rows (2 x fp16 planes, t2h_conv_split_f32) of every packed conv / 1x1 matrix under
    `dst.` whose shape the kernel serves (K % 32 == 0, Cout % 8 == 0), stored next to the fp32
    matrix as `<name>.ws`.  Split on the device (bit-identical to ops.pack_split_rows_host); a
    weight outside fp16's range raises."""
    from . import engine, ops
    ops.split_overflow(reset=True)
    for k in [k for k in P.t if k.startswith(dst + '.') and k.endswith('.w')]:
        w = P.t[k]
        if w.dim() == 2 and w.shape[1] % 32 == 0 and w.shape[0] % 8 == 0:
            P.t[k + 's'] = ops.split_rows(w)
    engine.check_split_overflow(f'weights of {dst}')


def pack_transformer(P, sd, dst='tf'):
    n_layers = _count(sd, 'blocks')
    P.put(f'{dst}.tok_emb', sd['tok_emb.weight'])
    P.put(f'{dst}.pos_emb', sd['pos_emb'][0])
    P.put(f'{dst}.segm_emb', sd['segm_emb.weight'])
    P.put(f'{dst}.tex_emb', sd['texture_emb.weight'])
    for i in range(n_layers):
        s, d = f'blocks.{i}', f'{dst}.{i}'
        for ln in ('ln1', 'ln2'):
            P.put(f'{d}.{ln}.g', sd[f'{s}.{ln}.weight'])
            P.put(f'{d}.{ln}.b', sd[f'{s}.{ln}.bias'])
        P.put(f'{d}.qkv.w', torch.cat([sd[f'{s}.attn.{n}.weight'] for n in ('query', 'key', 'value')], 0))
        P.put(f'{d}.qkv.b', torch.cat([sd[f'{s}.attn.{n}.bias'] for n in ('query', 'key', 'value')], 0))
        # split rows (2 x fp16 planes) of the four Linears for the split-precision GEMM
        from . import ops as _ops
        for lin, key in (('qkv', None), ('proj', f'{s}.attn.proj.weight'), ('fc1', f'{s}.mlp.0.weight'),
                         ('fc2', f'{s}.mlp.2.weight')):
            w = P[f'{d}.qkv.w'].cpu() if key is None else sd[key]
            P.t[f'{d}.{lin}.w_split'] = _ops.pack_split_rows_host(w).to(P.device)
        P.put(f'{d}.proj.w', sd[f'{s}.attn.proj.weight'])
        P.put(f'{d}.proj.b', sd[f'{s}.attn.proj.bias'])
        P.put(f'{d}.fc1.w', sd[f'{s}.mlp.0.weight'])
        P.put(f'{d}.fc1.b', sd[f'{s}.mlp.0.bias'])
        P.put(f'{d}.fc2.w', sd[f'{s}.mlp.2.weight'])
        P.put(f'{d}.fc2.b', sd[f'{s}.mlp.2.bias'])
    P.put(f'{dst}.ln_f.g', sd['ln_f.weight'])
    P.put(f'{dst}.ln_f.b', sd['ln_f.bias'])
    n_heads = _count(sd, 'head_list')
    P.put(f'{dst}.heads', torch.stack([sd[f'head_list.{h}.weight'] for h in range(n_heads)], 0))
    return dict(n_layers=n_layers, n_heads=n_heads, C=sd['ln_f.weight'].shape[0])


def _pack_convmodule(P, sd, src, dst, cin_pad=None):
    w, b = fold_bn(sd, src)
    if w.shape[2] == 3:
        P.put(f'{dst}.w', pack_conv3x3(w, cin_pad))
    else:
        P.put(f'{dst}.w', pack_conv1x1(w))
    P.put(f'{dst}.b', b)


def pack_unet(P, sd, dst, attr_channels=0):
    """UNet / ShapeUNet.  For ShapeUNet the first conv of every encoder stage
    sees [x, attr] channels; the weight is split into the image part (packed
    for the implicit GEMM) and the attribute part kept as [Cout, 9, A] for the
    per-image tap-bias (the attribute map is spatially constant)."""
    n = _count(sd, 'encoder')
    stages = []
    for i in range(n):
        blk = 1 if i != 0 else 0
        s0 = f'encoder.{i}.{blk}.convs.0'
        w, b = fold_bn(sd, s0)
        cin_img = w.shape[1] - attr_channels
        P.put(f'{dst}.enc.{i}.0.w', pack_conv3x3(w[:, :cin_img]))
        P.put(f'{dst}.enc.{i}.0.b', b)
        if attr_channels:
            P.put(f'{dst}.enc.{i}.0.wattr',
                  w[:, cin_img:].permute(0, 2, 3, 1).reshape(w.shape[0], 9, attr_channels))
        _pack_convmodule(P, sd, f'encoder.{i}.{blk}.convs.1', f'{dst}.enc.{i}.1')
        stages.append(dict(cin=cin_img, cout=w.shape[0]))
        if i != 0:
            d = i - 1
            _pack_convmodule(P, sd, f'decoder.{d}.conv_block.convs.0', f'{dst}.dec.{d}.0')
            _pack_convmodule(P, sd, f'decoder.{d}.conv_block.convs.1', f'{dst}.dec.{d}.1')
            _pack_convmodule(P, sd, f'decoder.{d}.upsample.interp_upsample.1', f'{dst}.dec.{d}.up')
    return dict(stages=stages)


def pack_multihead_fcn(P, sd, dst):
    nh = _count(sd, 'convs_list')
    ws, bs = [], []
    for h in range(nh):
        w, b = fold_bn(sd, f'convs_list.{h}.0')
        ws.append(pack_conv3x3(w))
        bs.append(b)
    P.put(f'{dst}.conv.w', torch.cat(ws, 0))   # [nh*64, 9*64]: all heads' 3x3 convs as one GEMM
    P.put(f'{dst}.conv.b', torch.cat(bs, 0))
    P.put(f'{dst}.seg.w', torch.stack([pack_conv1x1(sd[f'conv_seg_head_list.{h}.weight']) for h in range(nh)], 0))
    P.put(f'{dst}.seg.b', torch.stack([sd[f'conv_seg_head_list.{h}.bias'] for h in range(nh)], 0))
    return dict(n_heads=nh, cf=ws[0].shape[0], n_class=sd['conv_seg_head_list.0.weight'].shape[0])


def pack_fcn_head(P, sd, dst):
    _pack_convmodule(P, sd, 'convs.0', f'{dst}.conv')
    P.put(f'{dst}.seg.w', pack_conv1x1(sd['conv_seg.weight']))
    P.put(f'{dst}.seg.b', sd['conv_seg.bias'])
    return dict(n_class=sd['conv_seg.weight'].shape[0])


def pack_shape_embedder(P, sd, dst, cls_num_list):
    """ShapeAttrEmbedding -> tensors of t2h_shape_attr_embed_f32."""
    n = len(cls_num_list)
    dim = sd['attr_0.2.weight'].shape[0]
    out_dim = sd['fusion.2.weight'].shape[0]
    offs, acc = [], 0
    for c in cls_num_list:
        offs.append(acc)
        acc += c
    P.put(f'{dst}.w0t', torch.cat([sd[f'attr_{i}.0.weight'].t() for i in range(n)], 0))
    P.put(f'{dst}.b0', torch.stack([sd[f'attr_{i}.0.bias'] for i in range(n)], 0))
    P.put(f'{dst}.w1', torch.stack([sd[f'attr_{i}.2.weight'] for i in range(n)], 0))
    P.put(f'{dst}.b1', torch.stack([sd[f'attr_{i}.2.bias'] for i in range(n)], 0))
    P.put(f'{dst}.f0', sd['fusion.0.weight'])
    P.put(f'{dst}.fb0', sd['fusion.0.bias'])
    P.put(f'{dst}.f1', sd['fusion.2.weight'])
    P.put(f'{dst}.fb1', sd['fusion.2.bias'])
    cls_off = torch.tensor(offs, dtype=torch.int32, device=P.device)
    emb = dict(cls_off=cls_off, dim=dim, out_dim=out_dim, cls_num=list(cls_num_list))
    for k in ('w0t', 'b0', 'w1', 'b1', 'f0', 'fb0', 'f1', 'fb1'):
        emb[k] = P[f'{dst}.{k}']
    return emb


def stack_codebooks(sd):
    n = _count(sd, 'embedding_list')
    return torch.stack([sd[f'embedding_list.{i}.weight'] for i in range(n)], 0)


def _load(path, what):
    if not path:
        raise KeyError(f'the option file names no checkpoint for {what}')
    return torch.load(path, map_location='cpu', weights_only=False)


def load_hierarchy_checkpoints(opt):
    """What VQGANTextureAwareSpatialHierarchyInferenceModel loads (hierarchy_inference_model.py:
    131-161) from the key set of the reference's configs/index_pred_net.yml: top_vae_path and
    bot_vae_path ONLY (the bottom checkpoint's `decoder` overrides the top one's, :155), every
    state_dict validated strictly.  Module names as engine / models use them."""
    sc = synthetic.hierarchy_schemas(opt)
    top, bot = _load(opt['top_vae_path'], 'top_vae_path'), _load(opt['bot_vae_path'], 'bot_vae_path')
    check_state_dict(top['decoder'], sc['decoder'], 'Decoder (top_vae_path)')  # loaded first, then overridden
    sds = dict(top_encoder=top['encoder'], decoder=bot['decoder'], top_quantize=top['quantize'],
               top_quant_conv=top['quant_conv'], top_post_quant_conv=top['post_quant_conv'],
               bot_encoder=bot['bot_encoder'], bot_decoder_res=bot['bot_decoder_res'],
               bot_quantize=bot['bot_quantize'], bot_quant_conv=bot['bot_quant_conv'],
               bot_post_quant_conv=bot['bot_post_quant_conv'])
    for name, sd in sds.items():
        check_state_dict(sd, sc[name], name)
    return sds


def load_transformer_checkpoints(opt):
    """What TransformerTextureAwareModel loads (transformer_model.py:107-139) from the key set of the
    reference's configs/sampler.yml: img_ae_path {encoder, decoder, quantize, quant_conv,
    post_quant_conv} and segm_ae_path {encoder, quantize, quant_conv}, strictly validated; plus the
    sampler itself from `pretrained_sampler` (the reference trains it; this package's class is
    forward-only).  Returned under the module names the model packs (top_* = the image VAE)."""
    sc = synthetic.transformer_model_schemas(opt)
    img, seg = _load(opt['img_ae_path'], 'img_ae_path'), _load(opt['segm_ae_path'], 'segm_ae_path')
    smp = _load(opt['pretrained_sampler'], 'pretrained_sampler (the checkpoint train_sampler.py wrote)')
    for name, sd in (('img_encoder', img['encoder']), ('img_decoder', img['decoder']),
                     ('img_quantizer', img['quantize']), ('img_quant_conv', img['quant_conv']),
                     ('img_post_quant_conv', img['post_quant_conv']), ('segm_encoder', seg['encoder']),
                     ('segm_quantizer', seg['quantize']), ('segm_quant_conv', seg['quant_conv']), ('sampler', smp)):
        check_state_dict(sd, sc[name], name)
    return dict(top_encoder=img['encoder'], top_quantize=img['quantize'], top_quant_conv=img['quant_conv'],
                segm_encoder=seg['encoder'], segm_quantizer=seg['quantize'], segm_quant_conv=seg['quant_conv'],
                sampler=smp)


def load_checkpoints(opt, map_location='cpu', encode=False):
    """Reads the `.pth` files named by the YAML exactly like
    BaseSampleModel.load_* (models/sample_model.py:124-181,397-410), including
    the fact that the bottom checkpoint's `decoder` overrides the top one's.
    Returns dict module-name -> state_dict and validates every one strictly."""
    schemas = synthetic.module_schemas(opt, encode=encode)
    top = torch.load(opt['top_vae_path'], map_location=map_location, weights_only=False)
    bot = torch.load(opt['bot_vae_path'], map_location=map_location, weights_only=False)
    seg = torch.load(opt['segm_token_path'], map_location=map_location, weights_only=False)
    ipn = torch.load(opt['pretrained_index_network'], map_location=map_location, weights_only=False)
    smp = torch.load(opt['pretrained_sampler'], map_location=map_location, weights_only=False)
    sds = dict(
        decoder=bot['decoder'], top_quantize=top['quantize'],
        top_post_quant_conv=top['post_quant_conv'], bot_decoder_res=bot['bot_decoder_res'],
        bot_quantize=bot['bot_quantize'], bot_post_quant_conv=bot['bot_post_quant_conv'],
        segm_encoder=seg['encoder'], segm_quantizer=seg['quantize'],
        segm_quant_conv=seg['quant_conv'], guidance_encoder=ipn['guidance_encoder'],
        index_decoder=ipn['index_decoder'], sampler=smp)
    check_state_dict(top['decoder'], schemas['decoder'], 'Decoder')  # loaded first, then overridden
    if encode:  # hierarchy_inference_model.py:126-161
        sds.update(top_encoder=top['encoder'], top_quant_conv=top['quant_conv'],
                   bot_encoder=bot['bot_encoder'], bot_quant_conv=bot['bot_quant_conv'])
    if opt.get('pretrained_parsing_gen') and 'shape_embedder' in schemas:
        pg = torch.load(opt['pretrained_parsing_gen'], map_location=map_location, weights_only=False)
        sds.update(shape_embedder=pg['embedder'], shape_encoder=pg['encoder'],
                   shape_decoder=pg['decoder'])
    for name, sd in sds.items():
        check_state_dict(sd, schemas[name], name)
    return sds
