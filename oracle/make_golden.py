"""TEST INFRASTRUCTURE.  Generates tests/golden/*.npz by executing the
UNMODIFIED reference modules (through oracle/ref_shim.py) on CPU.

    python -m oracle.make_golden            # needs /root/reference

The reference publishes no golden vectors; these files
are what pins parity on the GPU box, where the reference tree does not exist.
Weights are the seeded synthetic checkpoints of text2human_amd/synthetic.py
(seed 1234), inputs are seeded too, so every consumer can regenerate the same
inputs and compare against the stored reference outputs.  Large outputs are
stored as strided samples plus moments to keep the repository small.
"""
import contextlib
import io
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import ref_shim
from text2human_amd import defaults, options, synthetic

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'tests', 'golden')
WEIGHT_SEED = 1234


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def moments(t):
    t = t.double()
    return np.array([t.mean().item(), t.abs().mean().item(), t.std().item()])


def golden_inputs(name):
    """Seeded inputs shared by make_golden and the tests."""
    g = torch.Generator().manual_seed({'decoder': 100, 'transformer': 101,
                                       'unet': 102, 'encode': 103}[name])
    if name == 'decoder':
        return dict(z=torch.randn(1, 256, 32, 16, generator=g) * 0.05,
                    zb=torch.randn(1, 256, 64, 32, generator=g) * 0.05)
    if name == 'transformer':
        return dict(idx=torch.randint(0, 18433, (1, 512), generator=g),
                    seg=torch.randint(0, 1024, (1, 512), generator=g),
                    tex=torch.randint(0, 18, (1, 512), generator=g))
    if name == 'unet':
        return dict(x=torch.randn(2, 256, 32, 16, generator=g) * 0.1)
    if name == 'encode':
        return dict(image=torch.rand(1, 3, 512, 256, generator=g) * 2 - 1,
                    texture_mask=synthetic.parsing_batch(1, seed=2021)['texture_mask'])
    raise KeyError(name)


TRANSFORMER_ROWS = [0, 1, 77, 300, 511]
TRANSFORMER_HEADS = [0, 5, 11, 17]


def _routed_margin(zf, tex, books):
    """per row: (second best - best) expanded L2 distance inside the row's own codebook"""
    out = torch.zeros(zf.shape[0])
    for cb in range(18):
        sel = tex == cb
        if sel.any():
            e = books[f'embedding_list.{cb}.weight']
            d = (zf[sel]**2).sum(1, keepdim=True) + (e**2).sum(1) - 2 * zf[sel] @ e.t()
            t2 = d.topk(2, dim=1, largest=False).values
            out[sel] = t2[:, 1] - t2[:, 0]
    return out


def make_encode_golden(ns, opt):
    """Encode side: reference Encoder / quantizer modules,
    encode-mode synthetic weights (codebooks U(+-1)), one image."""
    V = ns.vqgan_arch
    sds = synthetic.make_state_dicts(opt, seed=WEIGHT_SEED, encode=True)
    mk = lambda cm, ar, key: _load(quiet(V.Encoder, ch=128, num_res_blocks=2, attn_resolutions=ar, ch_mult=cm,
                                         in_channels=3, resolution=512, z_channels=256, double_z=False,
                                         dropout=0.0).eval(), sds[key])
    top_enc, bot_enc = mk([1, 1, 2, 2, 4], [32], 'top_encoder'), mk([1, 1, 2, 4], [64], 'bot_encoder')
    top_q = _load(quiet(V.VectorQuantizerTexture, 1024, 256, beta=0.25).eval(), sds['top_quantize'])
    bot_q = _load(quiet(V.VectorQuantizerSpatialTextureAware, 512, 256, beta=0.25, spatial_size=2).eval(),
                  sds['bot_quantize'])
    dec = _load(quiet(V.Decoder, in_channels=3, resolution=512, z_channels=256, ch=128, out_ch=3,
                      num_res_blocks=2, attn_resolutions=[32], ch_mult=[1, 1, 2, 2, 4], dropout=0.0).eval(),
                sds['decoder'])
    res = _load(quiet(V.DecoderRes, in_channels=3, resolution=512, z_channels=256, ch=128, num_res_blocks=2,
                      ch_mult=[1, 1, 2, 4], dropout=0.0).eval(), sds['bot_decoder_res'])
    conv = lambda sd: (lambda x: F.conv2d(x, sd['weight'], sd['bias']))
    gi = golden_inputs('encode')
    img, mask = gi['image'], gi['texture_mask']
    h = conv(sds['top_quant_conv'])(top_enc(img))
    zq, _, (_, _, top_idx) = top_q(h, mask)
    quant_t = conv(sds['top_post_quant_conv'])(zq)
    hb = conv(sds['bot_quant_conv'])(bot_enc(img))
    zqb, _, (_, _, bot_idx) = bot_q(hb, mask)
    rec = dec(quant_t, bot_h=res(conv(sds['bot_post_quant_conv'])(zqb)))
    tex = F.interpolate(mask, (32, 16), mode='nearest').view(-1)
    top_margin = _routed_margin(h.permute(0, 2, 3, 1).reshape(-1, 256), tex, sds['top_quantize'])
    patches = F.unfold(hb, (2, 2), stride=2).permute(0, 2, 1).reshape(-1, 1024)
    bot_margin = _routed_margin(patches, tex, sds['bot_quantize'])
    np.savez_compressed(
        os.path.join(OUT, 'encode_b1.npz'),
        top_indices=torch.stack(top_idx).numpy().astype(np.int16),
        bot_indices=torch.stack(bot_idx).numpy().astype(np.int16),
        top_margin=top_margin.numpy(), bot_margin=bot_margin.numpy(),
        top_latent_sample=h[0, ::8, ::2, ::2].numpy(), bot_latent_sample=hb[0, ::8, ::4, ::4].numpy(),
        quant_t_sample=quant_t[0, ::8, ::2, ::2].numpy(),
        rec_sample=rec[0, :, ::4, ::4].numpy(), rec_moments=moments(rec))


def _load(module, sd):
    module.load_state_dict(sd, strict=True)
    return module


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = ref_shim.load_reference('cpu')
    opt = options.dict_to_nonedict(defaults.sample_from_pose())
    torch.set_grad_enabled(False)
    make_encode_golden(ns, opt)
    if os.environ.get('T2H_GOLDEN_ONLY') == 'encode':
        return
    sds = synthetic.make_state_dicts(opt, seed=WEIGHT_SEED)
    V = ns.vqgan_arch
    torch.set_grad_enabled(False)

    # ---- decode (rows D-2, D-3)
    dec = quiet(V.Decoder, in_channels=3, resolution=512, z_channels=256, ch=128,
                out_ch=3, num_res_blocks=2, attn_resolutions=[32],
                ch_mult=[1, 1, 2, 2, 4], dropout=0.0).eval()
    dec.load_state_dict(sds['decoder'], strict=True)
    res = quiet(V.DecoderRes, in_channels=3, resolution=512, z_channels=256,
                ch=128, num_res_blocks=2, ch_mult=[1, 1, 2, 4], dropout=0.0).eval()
    res.load_state_dict(sds['bot_decoder_res'], strict=True)
    gi = golden_inputs('decoder')
    bh = res(gi['zb'])
    out = dec(gi['z'], bot_h=bh.clone())
    np.savez_compressed(
        os.path.join(OUT, 'decode_b1.npz'),
        bot_h_sample=bh[0, ::16, ::4, ::4].numpy(), bot_h_moments=moments(bh),
        dec_sample=out[0, :, ::4, ::4].numpy(), dec_moments=moments(out))

    # ---- transformer (row S-2)
    T = ns.transformer_arch.TransformerMultiHead(
        codebook_size=18432, segm_codebook_size=1024, texture_codebook_size=18,
        bert_n_emb=512, bert_n_layers=24, bert_n_head=8, block_size=512,
        latent_shape=[32, 16], embd_pdrop=0., resid_pdrop=0., attn_pdrop=0.,
        num_head=18).eval()
    T.load_state_dict(sds['sampler'], strict=True)
    gi = golden_inputs('transformer')
    logits = T(gi['idx'], gi['seg'], gi['tex'])
    np.savez_compressed(
        os.path.join(OUT, 'transformer_b1.npz'),
        rows=np.array(TRANSFORMER_ROWS), heads=np.array(TRANSFORMER_HEADS),
        logits=np.stack([logits[h][0, TRANSFORMER_ROWS].numpy()
                         for h in TRANSFORMER_HEADS]),
        moments=np.stack([moments(l) for l in logits]))

    # ---- index-prediction UNet + heads (row R-3)
    U = ns.unet_arch.UNet(in_channels=256).eval()
    U.load_state_dict(sds['guidance_encoder'], strict=True)
    H = ns.fcn_arch.MultiHeadFCNHead(
        in_channels=64, in_index=4, channels=64, num_convs=1, concat_input=False,
        dropout_ratio=0.1, num_classes=512, align_corners=False, num_head=18).eval()
    H.load_state_dict(sds['index_decoder'], strict=True)
    gi = golden_inputs('unet')
    feats = U(gi['x'])
    hl = H(feats)
    am = torch.stack([l.argmax(1) for l in hl])  # [18, 2, 32, 16]
    top2 = torch.stack([l.topk(2, dim=1).values for l in hl])
    np.savez_compressed(
        os.path.join(OUT, 'index_pred_b2.npz'),
        feat_sample=feats[4][:, ::8, ::4, ::4].numpy(), feat_moments=moments(feats[4]),
        argmax=am.numpy().astype(np.int16),
        margin=(top2[:, :, 0] - top2[:, :, 1]).numpy(),
        logits_sample=hl[3][0, :, 5, 7].numpy())

    # ---- end to end (rows T-*, S-*, R-*, D-*): reference model object, B=2, 5 steps
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        o = synthetic.write_checkpoints(opt, d, seed=WEIGHT_SEED)
        o['model_type'] = 'SampleFromParsingModel'
        model = quiet(ns.sample_model.SampleFromParsingModel, o)
        model.sample_steps = 5
        batch = synthetic.parsing_batch(2, seed=2021)
        ns.util.set_random_seed(2021)
        model.feed_data(batch)
        top = model.sample_fn(temp=1, sample_steps=5)
        ns.util.set_random_seed(2021)
        ref_shim.saved_images.clear()
        model.sample_and_refine('/nonexistent', batch['img_name'])
        imgs = torch.cat([t for t, _ in ref_shim.saved_images], 0)
        # intermediates: recompute stage by stage with the reference modules
        bot_idx = []
        for i in range(2):
            si = [t[i:i + 1] for t in top]
            tm = model.texture_mask[i:i + 1]
            tq = model.top_post_quant_conv(
                model.top_quantize.get_codebook_entry(si, tm, (1, 32, 16, 256)))
            bot_idx.append(torch.stack(model.bot_index_prediction(tq, tm)))
        bot_idx = torch.stack(bot_idx, 1)  # [18, 2, 1, 32, 16]
        np.savez_compressed(
            os.path.join(OUT, 'e2e_parsing_b2_steps5.npz'),
            segm_tokens=model.segm_tokens.numpy().astype(np.int16),
            top_indices=torch.stack(top).numpy().astype(np.int16),
            bot_indices=bot_idx.view(18, 2, 32, 16).numpy().astype(np.int16),
            img_sample=imgs[:, :, ::4, ::4].numpy(), img_moments=moments(imgs),
            img_u8_sample=imgs.mul(255).add(0.5).clamp(0, 255).to(torch.uint8)[:, :, ::4, ::4].numpy())

        # ---- pose front-end (rows P-1..P-3) on a 128x64 crop (fully convolutional)
        E = ns.shape_attr_embedding_arch.ShapeAttrEmbedding(
            dim=8, out_dim=128, cls_num_list=opt['shape_attr_class_num']).eval()
        E.load_state_dict(sds['shape_embedder'], strict=True)
        SU = ns.unet_arch.ShapeUNet(in_channels=1).eval()
        SU.load_state_dict(sds['shape_encoder'], strict=True)
        FH = ns.fcn_arch.FCNHead(in_channels=64, in_index=4, channels=64,
                                 num_convs=1, concat_input=False, dropout_ratio=0.1,
                                 num_classes=24, align_corners=False).eval()
        FH.load_state_dict(sds['shape_decoder'], strict=True)
        pb = synthetic.pose_batch(2, seed=2021)
        pose = pb['densepose'][:, :, :128, :64].contiguous()
        emb = E(pb['shape_attr'])
        lg = FH(SU(pose, emb))
        t2 = lg.topk(2, dim=1).values
        np.savez_compressed(
            os.path.join(OUT, 'pose_b2_128x64.npz'),
            attr_embedding=emb.numpy(), segm=lg.argmax(1).numpy().astype(np.int8),
            margin=(t2[:, 0] - t2[:, 1]).numpy(),
            logits_sample=lg[:, :, ::8, ::8].numpy(), logits_moments=moments(lg))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
