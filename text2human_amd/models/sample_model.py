"""MI355X-native SampleFromParsingModel / SampleFromPoseModel.

Same public surface as the reference's models/sample_model.py (constructor
from the YAML `opt`, `feed_data`, `inference`, `sample_and_refine`,
`sample_fn`, `get_quantized_segm`, `bot_index_prediction`, the pose helpers and
the attributes callers read: .segm, .segm_tokens, .texture_mask, .device,
.batch_size, .shape, .mask_id, .sample_steps), same checkpoint files -- but
every tensor op runs in hand-written HIP kernels (libt2h_hip.so) through
text2human_amd.engine, batched over images, with no PyTorch compute fallback.
"""
import logging
import os

import numpy as np
import torch

from .. import _lib, engine, ops, weights
from ..ops import ACT_RELU

logger = logging.getLogger('base')

# images decoded per pass: the decode's activations take ~0.6 GB / image at 512x256 (2.4 GB at 1024x512, where a pass
# takes a quarter of this); T2H_DECODE_CHUNK overrides (a box with less free HBM, or larger passes on an idle 288 GB)
DECODE_CHUNK = max(1, int(os.environ.get('T2H_DECODE_CHUNK', '8')))


class BaseSampleModel():
    """Base Model (reference: models/sample_model.py:21-340)."""

    def __init__(self, opt, state_dicts=None):
        self.opt = opt
        if not torch.cuda.is_available():
            raise RuntimeError('text2human_amd needs a ROCm GPU (MI355X); there is no CPU path')
        _lib.load()  # fail loudly if the HIP library is missing
        self.device = torch.device('cuda', torch.cuda.current_device())
        sds = state_dicts if state_dicts is not None else weights.load_checkpoints(opt)
        self._pack(sds)
        self.shape = tuple(opt['latent_shape'])
        self.mask_id = opt['codebook_size']
        self.sample_steps = opt['sample_steps']
        self.noise = None  # None -> torch's global GPU generator, like the reference
        self.batch_size = 0

    # ------------------------------------------------------------ weights
    def _pack(self, sds):
        P = weights.Params(self.device)
        self.P = P
        d_dec = weights.pack_vqgan(P, sds['decoder'], 'dec')
        d_res = weights.pack_vqgan(P, sds['bot_decoder_res'], 'res')
        d_enc = weights.pack_vqgan(P, sds['segm_encoder'], 'senc')
        # T2H_SPLIT_CONV=0 keeps the decoders' convolutions on the exact-fp32 matrix instructions;
        # default: split-precision (2 x fp16 planes, three products) like the sampler's Linears.
        # The tokenizer encoder, the index-prediction UNet and the parsing generator stay exact
        # fp32: their outputs are argmin / argmax decisions that must match the reference bit for bit.
        self.split_conv = os.environ.get('T2H_SPLIT_CONV', '1') != '0'
        if self.split_conv:
            weights.add_split_conv_weights(P, 'dec')
            weights.add_split_conv_weights(P, 'res')
        self.decoder = engine.VQGANStack(P, 'dec', d_dec)
        self.bot_decoder_res = engine.VQGANStack(P, 'res', d_res)
        self.decoder.flash_attn = self.bot_decoder_res.flash_attn = True
        self.segm_encoder = engine.VQGANStack(P, 'senc', d_enc)
        self.segm_cin_pad = P['senc.conv_in.w'].shape[1] // 9
        P.put('top.books', weights.stack_codebooks(sds['top_quantize']))
        P.put('bot.books', weights.stack_codebooks(sds['bot_quantize']))
        P.put('segm.book', sds['segm_quantizer']['embedding.weight'])
        for nm, key in (('top.pq', 'top_post_quant_conv'), ('bot.pq', 'bot_post_quant_conv'),
                        ('segm.qc', 'segm_quant_conv')):
            P.put(f'{nm}.w', weights.pack_conv1x1(sds[key]['weight']))
            P.put(f'{nm}.b', sds[key]['bias'])
        d_unet = weights.pack_unet(P, sds['guidance_encoder'], 'ipu')
        self.index_pred_guidance_encoder = engine.UNetStack(P, 'ipu', d_unet)
        self.ipd = weights.pack_multihead_fcn(P, sds['index_decoder'], 'ipd')
        d_tf = weights.pack_transformer(P, sds['sampler'], 'tf')
        self._tf_desc = d_tf
        # T2H_SPLIT_GEMM=0 selects the exact-fp32 MFMA GEMMs for the sampler's Linears;
        # default: split-precision (2 x fp16 planes, three products) on the fp16 matrix
        # cores -- same fp32-class accuracy (tests/test_gpu_split.py), higher throughput
        # (T2H_SPLIT_MHA=0 keeps the attention on the exact-fp32 kernel)
        split = os.environ.get('T2H_SPLIT_GEMM', '1') != '0'
        split_mha = os.environ.get('T2H_SPLIT_MHA', '1') != '0'
        self.sampler_fn = engine.SamplerNet(P, d_tf, self.opt['bert_n_head'], 'tf', split=split,
                                            split_mha=split_mha)
        # x8 operands (engine.SamplerNet): weights packed and the per-tensor scales of the 8-bit planes fixed HERE,
        # from the checkpoint alone -- nothing fed to the model later changes how a value is rounded
        self.sampler_fn.ensure_x8()

    # ------------------------------------------------------------ helpers
    def _texture_tokens(self, texture_mask):
        """F.interpolate(mask, (32,16), 'nearest') -> source pixel (16i,16j)
        (models/sample_model.py:187-188,264-266)."""
        b, _, hh, ww = texture_mask.shape
        sy, sx = hh // self.shape[0], ww // self.shape[1]
        return texture_mask[:, 0, ::sy, ::sx].reshape(b, -1).long().contiguous()

    # ------------------------------------------------------------ stage T
    @torch.no_grad()
    def get_quantized_segm(self, segm):
        """models/sample_model.py:330-340 -> int64 [B, 32, 16]."""
        P = self.P
        b, _, hh, ww = segm.shape
        x = ops.onehot_nhwc(segm.to(self.device, torch.float32).reshape(-1),
                            self.opt['segm_num_segm_classes'], self.segm_cin_pad)
        z, h, w = self.segm_encoder.encode(x, b, hh, ww)
        z = ops.gemm(z, P['segm.qc.w'], bias=P['segm.qc.b'])
        return ops.vq_l2_argmin(z, P['segm.book']).view(b, h, w)

    # ------------------------------------------------------------ stage S
    @torch.no_grad()
    def sample_fn(self, temp=1.0, sample_steps=None):
        """models/sample_model.py:256-328 -> list of 18 int64 [B, 512]."""
        sample_steps = sample_steps or self.sample_steps
        tex_tok = self._texture_tokens(self.texture_mask)
        # The reference computes ANY checkpoint in fp32 (transformer_arch.py:91-99).  The split-precision kernels
        # cover |x| < 65504; an activation outside raises SplitOverflowError at the end of the run -- after
        # build_schedule has advanced the generator.  So: remember the generator, and on overflow restore it and
        # run THIS call on the exact-fp32 kernels (same schedule, same draws, the reference's tokens).
        gen = torch.cuda.default_generators[self.device.index]
        state = gen.get_state()
        net = self.sampler_fn
        x8_was = net.x8
        try:
            for _ in range(3):  # x8 planes -> fp16 planes -> exact fp32, each at most once
                try:
                    out = engine.sample_tokens(net, self.segm_tokens.contiguous(), tex_tok, sample_steps, self.mask_id,
                                               temp=temp, noise=self.noise)
                    break
                except engine.X8RangeError as e:
                    # an activation beyond 14x its calibration maximum: the 8-bit planes saturated, the fp16 planes are
                    # fine -- THIS call is re-run on the fp16-plane kernels (about 15 % slower) from the same generator
                    # state; the next call starts on x8 again (the result of a call never depends on an earlier one)
                    if not _overflow_fallback('index sampler (x8 range)', 'T2H_X8', e, 'fp16-plane'):
                        raise
                    net.x8 = False
                    gen.set_state(state)
                except engine.SplitOverflowError as e:
                    if net is not self.sampler_fn or not _overflow_fallback('index sampler', 'T2H_SPLIT_GEMM', e):
                        raise
                    gen.set_state(state)
                    net = self._exact_sampler()
        finally:
            self.sampler_fn.x8 = x8_was
        b = self.batch_size
        return [out[i].view(b, -1) for i in range(out.shape[0])]

    def _exact_sampler(self):
        """The same transformer on the exact-fp32 matrix instructions (built on first use; shares the weights)."""
        if getattr(self, '_sampler_exact', None) is None:
            self._sampler_exact = engine.SamplerNet(self.P, self._tf_desc, self.opt['bert_n_head'], 'tf', split=False)
        return self._sampler_exact

    # ------------------------------------------------------------ stage R
    def _top_quant_rows(self, top_lists, tex_tok):
        """R-1/R-2: texture-routed gather + top_post_quant_conv -> rows [B*512, 256]."""
        P = self.P
        zq = ops.codebook_gather_tex(top_lists, tex_tok.reshape(-1), P['top.books'])
        return ops.gemm(zq, P['top.pq.w'], bias=P['top.pq.b'])

    def _bot_indices(self, top_quant_rows, tex_tok, b):
        """R-3 batched: UNet -> all 18 head convs as one GEMM -> routed 1x1 + argmax."""
        P = self.P
        h, w = self.shape
        feat, _, _ = self.index_pred_guidance_encoder.forward(top_quant_rows, b, h, w)
        hc = ops.conv3x3(feat, P['ipd.conv.w'], b, h, w, feat.shape[1], bias=P['ipd.conv.b'],
                         act=ACT_RELU)
        return ops.routed_head_argmax(hc, P['ipd.seg.w'], P['ipd.seg.b'], tex_tok.reshape(-1),
                                      self.ipd['n_heads'], self.ipd['cf'], self.ipd['n_class'])

    @torch.no_grad()
    def bot_index_prediction(self, feature_top, texture_mask):
        """models/sample_model.py:183-213.  feature_top f32 [B,256,32,16] (NCHW,
        as the reference passes it) -> list of 18 int64 [B,32,16]."""
        b = feature_top.shape[0]
        tex_tok = self._texture_tokens(texture_mask.to(self.device))
        rows = ops.nchw_to_nhwc(feature_top.to(self.device, torch.float32))
        lists = self._bot_indices(rows, tex_tok, b)
        return [lists[i].view(b, self.shape[0], self.shape[1]) for i in range(lists.shape[0])]

    # ------------------------------------------------------------ stage D
    def _decode(self, top_lists, tex_tok, b, want_u8=False, return_inter=False, upscale=False):
        """sample_and_refine body after sample_fn (models/sample_model.py:220-246),
        batched.  top_lists int64 [18, b*512]."""
        P = self.P
        h, w = self.shape
        top_quant = self._top_quant_rows(top_lists, tex_tok)
        bot_lists = self._bot_indices(top_quant, tex_tok, b)
        quant_bot = ops.codebook_gather_fold(bot_lists, tex_tok.reshape(-1), P['bot.books'], b, h, w)
        quant_bot = ops.gemm(quant_bot, P['bot.pq.w'], bias=P['bot.pq.b'])
        bot_h = self.bot_decoder_res.decode_res(quant_bot, b, 2 * h, 2 * w, upscale=upscale)
        dec, ho, wo = self.decoder.decode(top_quant, b, h, w, bot_h=bot_h, upscale=upscale)
        img, u8 = ops.image_epilogue(dec, b, ho, wo, want_u8=want_u8)
        if return_inter:
            return img, u8, dict(top_quant=top_quant, bot_lists=bot_lists, bot_h=bot_h, dec=dec)
        return img, u8

    @torch.no_grad()
    def decode_indices(self, top_indices_list, want_u8=False, return_inter=False, upscale=False):
        """Batched refine + decode of sampled top indices (list of 18 [B,512]).
        upscale=True: 1024x512 output -- both quantised latents are nearest-x2
        upsampled before the (fully convolutional) decoders, the interpretation
        of BASELINE.json configs[4] given in SURVEY.md 8(d)."""
        try:
            return self._decode_indices(top_indices_list, want_u8, return_inter, upscale)
        except engine.SplitOverflowError as e:
            # (decode draws no random numbers: simply once more, convolutions on the exact-fp32 kernels)
            if not _overflow_fallback('VQGAN refine / decode', 'T2H_SPLIT_CONV', e):
                raise
            keep = self.decoder.use_split, self.bot_decoder_res.use_split
            self.decoder.use_split = self.bot_decoder_res.use_split = False
            try:
                return self._decode_indices(top_indices_list, want_u8, return_inter, upscale)
            finally:
                self.decoder.use_split, self.bot_decoder_res.use_split = keep

    def _decode_indices(self, top_indices_list, want_u8, return_inter, upscale):
        b = self.batch_size
        tex_tok = self._texture_tokens(self.texture_mask)
        top = torch.stack([t.reshape(-1) for t in top_indices_list]).contiguous()
        imgs, u8s, inters = [], [], []
        t_len = self.shape[0] * self.shape[1]
        chunk = max(1, DECODE_CHUNK // (4 if upscale else 1))
        for s in range(0, b, chunk):
            e = min(b, s + chunk)
            res = self._decode(top[:, s * t_len:e * t_len].contiguous(), tex_tok[s:e], e - s,
                               want_u8=want_u8, return_inter=return_inter, upscale=upscale)
            imgs.append(res[0])
            u8s.append(res[1])
            if return_inter:
                inters.append(res[2])
        if self.split_conv and self.decoder.use_split:  # split-precision convolutions: loud on fp16-range overflow
            engine.check_split_overflow('VQGAN refine / decode', knob='T2H_SPLIT_CONV')
        img = torch.cat(imgs, 0) if len(imgs) > 1 else imgs[0]
        u8 = (torch.cat(u8s, 0) if len(u8s) > 1 else u8s[0]) if want_u8 else None
        if return_inter:
            return img, u8, inters
        return img, u8

    @torch.no_grad()
    def sample_and_refine(self, save_dir=None, img_name=None):
        """models/sample_model.py:215-254.  With both arguments None returns the
        first sample as f32 [1,3,512,256] in [0,1] (what ui_demo.py:162 uses);
        otherwise writes {save_dir}/{img_name[i]} PNGs."""
        sampled_top_indices_list = self.sample_fn(temp=1, sample_steps=self.sample_steps)
        want_files = not (save_dir is None and img_name is None)
        if not want_files:
            keep_b, keep_mask = self.batch_size, self.texture_mask
            self.batch_size, self.texture_mask = 1, self.texture_mask[:1]
            try:
                img, _ = self.decode_indices([t[:1] for t in sampled_top_indices_list])
            finally:
                self.batch_size, self.texture_mask = keep_b, keep_mask
            return img
        _, u8 = self.decode_indices(sampled_top_indices_list, want_u8=True)
        save_u8_images(u8, save_dir, img_name)

    def inference(self, data_loader, save_dir):
        for _, data in enumerate(data_loader):
            img_name = data['img_name']
            self.feed_data(data)
            self.sample_and_refine(save_dir, img_name)


_warned = set()


def _overflow_fallback(stage, knob, err, to='exact-fp32'):
    """True: re-run the stage on the `to` kernels (default; warns once per stage).  T2H_OVERFLOW_FALLBACK=0:
    the SplitOverflowError propagates, as before."""
    if os.environ.get('T2H_OVERFLOW_FALLBACK', '1') == '0':
        return False
    if stage not in _warned:
        _warned.add(stage)
        import warnings
        warnings.warn(f'text2human_amd: {err}  Re-running the {stage} on the {to} kernels; '
                      f'set {knob}=0 to start there, T2H_OVERFLOW_FALLBACK=0 to raise instead.')
    return True


def save_u8_images(u8, save_dir, img_name):
    """torchvision.utils.save_image(dec, path, nrow=1, padding=4) of a single
    image == its uint8 HWC PNG (models/sample_model.py:250-254)."""
    from PIL import Image
    arr = u8.cpu().numpy()
    for i in range(arr.shape[0]):
        Image.fromarray(arr[i]).save(os.path.join(save_dir, img_name[i]))


class SampleFromParsingModel(BaseSampleModel):
    """SampleFromParsing model (models/sample_model.py:343-360)."""

    def feed_data(self, data):
        self.segm = data['segm'].to(self.device)
        self.texture_mask = data['texture_mask'].to(self.device)
        self.batch_size = self.segm.size(0)
        self.segm_tokens = self.get_quantized_segm(self.segm)
        self.segm_tokens = self.segm_tokens.view(self.batch_size, -1)


class SampleFromPoseModel(BaseSampleModel):
    """SampleFromPose model (models/sample_model.py:363-498)."""

    def __init__(self, opt, state_dicts=None):
        super().__init__(opt, state_dicts)
        self.palette = [[0, 0, 0], [255, 250, 250], [220, 220, 220], [250, 235, 215],
                        [255, 250, 205], [211, 211, 211], [70, 130, 180], [127, 255, 212],
                        [0, 100, 0], [50, 205, 50], [255, 255, 0], [245, 222, 179],
                        [255, 140, 0], [255, 0, 0], [16, 78, 139], [144, 238, 144],
                        [50, 205, 174], [50, 155, 250], [160, 140, 88], [213, 140, 88],
                        [90, 140, 90], [185, 210, 205], [130, 165, 180], [225, 141, 151]]

    def _pack(self, sds):
        super()._pack(sds)
        P = self.P
        self.shape_emb = weights.pack_shape_embedder(P, sds['shape_embedder'], 'semb',
                                                     self.opt['shape_attr_class_num'])
        d_su = weights.pack_unet(P, sds['shape_encoder'], 'sunet',
                                 attr_channels=self.opt['shape_embedder_out_dim'])
        self.shape_parsing_encoder = engine.UNetStack(P, 'sunet', d_su)
        self.shape_head = weights.pack_fcn_head(P, sds['shape_decoder'], 'shead')

    def feed_data(self, data):
        self.pose = data['densepose'].to(self.device)
        self.batch_size = self.pose.size(0)
        self.shape_attr = data['shape_attr'].to(self.device)
        self.upper_fused_attr = data['upper_fused_attr'].to(self.device)
        self.lower_fused_attr = data['lower_fused_attr'].to(self.device)
        self.outer_fused_attr = data['outer_fused_attr'].to(self.device)

    def inference(self, data_loader, save_dir):
        for _, data in enumerate(data_loader):
            img_name = data['img_name']
            self.feed_data(data)
            self.generate_parsing_map()
            self.generate_quantized_segm()
            self.generate_texture_map()
            self.sample_and_refine(save_dir, img_name)

    def _attr_embedding(self, shape_attr):
        """ShapeAttrEmbedding.forward (shape_attr_embedding_arch.py:23-35)."""
        return ops.shape_attr_embed(shape_attr.long().contiguous(), self.shape_emb)

    @torch.no_grad()
    def generate_parsing_map(self):
        """models/sample_model.py:431-437 -> self.segm int64 [B,1,H,W]."""
        P = self.P
        b, _, hh, ww = self.pose.shape
        attr = self._attr_embedding(self.shape_attr)
        cin_pad = P['sunet.enc.0.0.w'].shape[1] // 9
        x = ops.nchw_to_nhwc(self.pose.to(torch.float32), cpad=cin_pad)
        feat, h, w = self.shape_parsing_encoder.forward(x, b, hh, ww, attr=attr)
        y = ops.conv3x3(feat, P['shead.conv.w'], b, h, w, feat.shape[1], bias=P['shead.conv.b'],
                        act=ACT_RELU)
        logits = ops.gemm(y, P['shead.seg.w'], bias=P['shead.seg.b'])
        self.seg_logits_rows = logits
        self.segm = ops.argmax_rows(logits).view(b, 1, h, w)

    def generate_quantized_segm(self):
        self.segm_tokens = self.get_quantized_segm(self.segm)
        self.segm_tokens = self.segm_tokens.view(self.batch_size, -1)

    def generate_texture_map(self):
        """models/sample_model.py:443-467."""
        self.texture_mask = ops.texture_map(self.segm.contiguous(),
                                            self.upper_fused_attr.long().contiguous(),
                                            self.lower_fused_attr.long().contiguous(),
                                            self.outer_fused_attr.long().contiguous())

    def feed_pose_data(self, pose_img):
        self.pose = pose_img.to(self.device)
        self.batch_size = self.pose.size(0)

    def feed_shape_attributes(self, shape_attr):
        self.shape_attr = shape_attr.to(self.device)

    def feed_texture_attributes(self, texture_attr):
        self.upper_fused_attr = texture_attr[0].unsqueeze(0).to(self.device)
        self.lower_fused_attr = texture_attr[1].unsqueeze(0).to(self.device)
        self.outer_fused_attr = texture_attr[2].unsqueeze(0).to(self.device)

    def palette_result(self, result):
        seg = result[0]
        palette = np.array(self.palette)
        color_seg = np.zeros((seg.shape[0], seg.shape[1], 3), dtype=np.uint8)
        for label, color in enumerate(palette):
            color_seg[seg == label, :] = color
        return color_seg
