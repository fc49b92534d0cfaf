"""Encode side of the hierarchy: image -> top / bottom tokens ->
image.  Mirrors the inference subset of the reference's
VQGANTextureAwareSpatialHierarchyInferenceModel (models/hierarchy_inference_model.py):
top_encode (:170-176), feed_data (:178-185), bot_encode (:187-192), get_gt_indices
(:194-197), index_to_image (:198-209).  Training (optimize_parameters, losses) is out of
scope."""
import torch

from .. import _lib, engine, ops, weights


class VQGANTextureAwareSpatialHierarchyInferenceModel():

    def __init__(self, opt, state_dicts=None):
        self.opt = opt
        if not torch.cuda.is_available():
            raise RuntimeError('text2human_amd needs a ROCm GPU (MI355X); there is no CPU path')
        _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.is_train = False
        sds = state_dicts if state_dicts is not None else weights.load_hierarchy_checkpoints(opt)
        P = weights.Params(self.device)
        self.P = P
        self.top_encoder = engine.VQGANStack(P, 'tenc', weights.pack_vqgan(P, sds['top_encoder'], 'tenc'))
        self.bot_encoder = engine.VQGANStack(P, 'benc', weights.pack_vqgan(P, sds['bot_encoder'], 'benc'))
        self.decoder = engine.VQGANStack(P, 'dec', weights.pack_vqgan(P, sds['decoder'], 'dec'))
        self.bot_decoder_res = engine.VQGANStack(P, 'res', weights.pack_vqgan(P, sds['bot_decoder_res'], 'res'))
        self.decoder.flash_attn = self.bot_decoder_res.flash_attn = True
        self.cin_pad = P['tenc.conv_in.w'].shape[1] // 9
        P.put('top.books', weights.stack_codebooks(sds['top_quantize']))
        P.put('bot.books', weights.stack_codebooks(sds['bot_quantize']))
        for nm, key in (('top.qc', 'top_quant_conv'), ('top.pq', 'top_post_quant_conv'),
                        ('bot.qc', 'bot_quant_conv'), ('bot.pq', 'bot_post_quant_conv')):
            P.put(f'{nm}.w', weights.pack_conv1x1(sds[key]['weight']))
            P.put(f'{nm}.b', sds[key]['bias'])
        self.spatial = opt['codebook_spatial_size'] or opt['bot_codebook_spatial_size']

    # ------------------------------------------------------------ helpers
    def _tex_tokens(self, mask, h, w):
        """F.interpolate(mask, (h, w), 'nearest') (vqgan_arch.py:228,391-395)."""
        b, _, hh, ww = mask.shape
        return mask[:, 0, ::hh // h, ::ww // w].reshape(-1).long().contiguous()

    def _image_rows(self, x):
        return ops.nchw_to_nhwc(x.to(self.device, torch.float32), cpad=self.cin_pad)

    # ------------------------------------------------------------ reference surface
    @torch.no_grad()
    def top_encode(self, x, mask):
        """-> (quant, quant): f32 [B, 256, 32, 16] after top_post_quant_conv (:170-176)."""
        P = self.P
        b, _, hh, ww = x.shape
        z, h, w = self.top_encoder.encode(self._image_rows(x), b, hh, ww)
        z = ops.gemm(z, P['top.qc.w'], bias=P['top.qc.b'])
        tex = self._tex_tokens(mask.to(self.device), h, w)
        self._top_latent_rows = z  # [B*h*w, 256] what the codebooks see (parity tests account near-ties on it)
        self.top_indices_list = ops.vq_argmin_tex(z, P['top.books'], tex)
        quant = self.quant_from_top_indices(self.top_indices_list, mask, (b, h, w))
        return quant, quant

    @torch.no_grad()
    def quant_from_top_indices(self, index_lists, mask, bhw):
        """18 x i64 [B*h*w] (or [18, B*h*w]) top indices -> quant_t f32 [B, 256, h, w] (texture-routed codebook
        lookup + top_post_quant_conv, vqgan_arch.py:289-309 + hierarchy_inference_model.py:174-175); also what
        index_to_image decodes from.  Lets a caller continue from GIVEN top indices (tests: the reference's)."""
        P = self.P
        b, h, w = bhw
        tex = self._tex_tokens(mask.to(self.device), h, w)
        lists = index_lists if torch.is_tensor(index_lists) else torch.stack([t.reshape(-1) for t in index_lists])
        lists = lists.reshape(lists.shape[0], -1).to(self.device, torch.long).contiguous()
        zq = ops.codebook_gather_tex(lists, tex, P['top.books'])
        self._quant_t_rows = ops.gemm(zq, P['top.pq.w'], bias=P['top.pq.b'])
        return ops.nhwc_to_nchw(self._quant_t_rows, b, h, w)

    @torch.no_grad()
    def bot_encode(self, x, mask):
        """-> list of 18 int64 [B, 32, 16] (:187-192)."""
        P = self.P
        b, _, hh, ww = x.shape
        z, h, w = self.bot_encoder.encode(self._image_rows(x), b, hh, ww)
        z = ops.gemm(z, P['bot.qc.w'], bias=P['bot.qc.b'])
        ph, pw = h // self.spatial, w // self.spatial
        tex = self._tex_tokens(mask.to(self.device), ph, pw)
        self._bot_latent_rows, self._bot_latent_hw = z, (h, w)  # NHWC rows [B*h*w, 256] before the 2x2 patch fold
        lists = ops.vq_argmin_tex(z, P['bot.books'], tex, fold_hw=(ph, pw))
        return [lists[i].view(b, ph, pw) for i in range(lists.shape[0])]

    def feed_data(self, data):
        self.image = data['image'].to(self.device)
        self.texture_mask = data['texture_mask'].float().to(self.device)
        self.get_gt_indices()
        self.texture_tokens = self._tex_tokens(self.texture_mask, 32, 16).view(self.image.size(0), -1)

    def get_gt_indices(self):
        self.quant_t, self.feature_t = self.top_encode(self.image, self.texture_mask)
        self.gt_indices_list = self.bot_encode(self.image, self.texture_mask)

    @torch.no_grad()
    def index_to_image(self, index_bottom_list, texture_mask):
        """-> dec f32 [B, 3, 512, 256] in [-1, 1] scale (:198-209); uses self.quant_t."""
        P = self.P
        b, h, w = index_bottom_list[0].shape
        tex = self._tex_tokens(texture_mask.to(self.device), h, w)
        lists = torch.stack([t.reshape(-1) for t in index_bottom_list]).to(self.device).contiguous()
        qb = ops.codebook_gather_fold(lists, tex, P['bot.books'], b, h, w)
        qb = ops.gemm(qb, P['bot.pq.w'], bias=P['bot.pq.b'])
        bot_h = self.bot_decoder_res.decode_res(qb, b, 2 * h, 2 * w)
        dec, ho, wo = self.decoder.decode(self._quant_t_rows, b, h, w, bot_h=bot_h)
        return ops.nhwc_to_nchw(dec, b, ho, wo, C=3)
