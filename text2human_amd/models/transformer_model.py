"""Sampler training-time FORWARD: the inference-side subset of the
reference's TransformerTextureAwareModel (models/transformer_model.py) -- image / parsing
tokenisation (get_quantized_img :152-170, get_quantized_segm :305-316), feed_data (:276-288),
sample_time 'uniform' (:203-207), q_sample (:212-231) and _train_loss (:233-274).  No backward
pass, optimiser or logging: it exists to check this package's kernels against losses of
checkpoints produced by the reference's train_sampler.py."""
import math

import torch

from .. import _lib, engine, ops, weights


class TransformerTextureAwareModel():

    def __init__(self, opt, state_dicts=None):
        self.opt = opt
        if not torch.cuda.is_available():
            raise RuntimeError('text2human_amd needs a ROCm GPU (MI355X); there is no CPU path')
        _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.is_train = False
        sds = state_dicts if state_dicts is not None else weights.load_transformer_checkpoints(opt)
        P = weights.Params(self.device)
        self.P = P
        self.img_encoder = engine.VQGANStack(P, 'ienc', weights.pack_vqgan(P, sds['top_encoder'], 'ienc'))
        self.segm_encoder = engine.VQGANStack(P, 'senc', weights.pack_vqgan(P, sds['segm_encoder'], 'senc'))
        self.img_cin_pad = P['ienc.conv_in.w'].shape[1] // 9
        self.segm_cin_pad = P['senc.conv_in.w'].shape[1] // 9
        P.put('img.books', weights.stack_codebooks(sds['top_quantize']))
        P.put('segm.book', sds['segm_quantizer']['embedding.weight'])
        for nm, key in (('img.qc', 'top_quant_conv'), ('segm.qc', 'segm_quant_conv')):
            P.put(f'{nm}.w', weights.pack_conv1x1(sds[key]['weight']))
            P.put(f'{nm}.b', sds[key]['bias'])
        d_tf = weights.pack_transformer(P, sds['sampler'], 'tf')
        self._denoise_fn = engine.SamplerNet(P, d_tf, opt['bert_n_head'], 'tf', split=True, x8=False)  # (loss values: fp16 planes)
        self.shape = tuple(opt['latent_shape'])
        self.num_timesteps = 1000  # transformer_model.py:99
        self.mask_id = opt['codebook_size']
        self.loss_type = opt['loss_type'] or 'reweighted_elbo'
        self.mask_schedule = opt['mask_schedule'] or 'random'
        self.n_e = P['img.books'].shape[1]

    # ------------------------------------------------------------ tokenisation
    def _tex_tokens(self, mask):
        b, _, hh, ww = mask.shape
        return mask[:, 0, ::hh // self.shape[0], ::ww // self.shape[1]].reshape(b, -1).long().contiguous()

    @torch.no_grad()
    def get_quantized_img(self, image, texture_mask):
        """-> (continual indices int64 [B, 512] = code + 1024 * texture, 18 x int64 [B, 512])."""
        P = self.P
        b, _, hh, ww = image.shape
        x = ops.nchw_to_nhwc(image.to(self.device, torch.float32), cpad=self.img_cin_pad)
        z, h, w = self.img_encoder.encode(x, b, hh, ww)
        z = ops.gemm(z, P['img.qc.w'], bias=P['img.qc.b'])
        tex = self._tex_tokens(texture_mask.to(self.device)).reshape(-1)
        lists = ops.vq_argmin_tex(z, P['img.books'], tex)                       # [18, B*512]
        own = lists.gather(0, tex.clamp(0, lists.shape[0] - 1)[None])[0]
        cont = torch.where(own >= 0, own + self.n_e * tex, own)
        return cont.view(b, -1), [lists[i].view(b, -1) for i in range(lists.shape[0])]

    @torch.no_grad()
    def get_quantized_segm(self, segm):
        P = self.P
        b, _, hh, ww = segm.shape
        x = ops.onehot_nhwc(segm.to(self.device, torch.float32).reshape(-1), self.opt['segm_num_segm_classes'],
                            self.segm_cin_pad)
        z, h, w = self.segm_encoder.encode(x, b, hh, ww)
        z = ops.gemm(z, P['segm.qc.w'], bias=P['segm.qc.b'])
        return ops.vq_l2_argmin(z, P['segm.book']).view(b, h, w)

    def feed_data(self, data):
        self.image = data['image'].to(self.device)
        self.segm = data['segm'].to(self.device)
        self.texture_mask = data['texture_mask'].to(self.device)
        self.input_indices, self.gt_indices_list = self.get_quantized_img(self.image, self.texture_mask)
        self.texture_tokens = self._tex_tokens(self.texture_mask)
        self.segm_tokens = self.get_quantized_segm(self.segm).view(self.image.size(0), -1)

    # ------------------------------------------------------------ loss (forward only)
    def sample_time(self, b, device, method='uniform'):
        if method != 'uniform':
            raise ValueError(method)
        t = torch.randint(1, self.num_timesteps + 1, (b, ), device=device).long()
        return t, torch.ones_like(t).float() / self.num_timesteps

    @torch.no_grad()
    def q_sample(self, x_0, x_0_gt_list, t, u=None):
        """-> (x_t, targets with -1 at unmasked positions, mask); `u` defaults to the
        reference's torch.rand_like draw on the global generator."""
        u = torch.rand_like(x_0.float()) if u is None else u.to(self.device, torch.float32)
        x_t, mask = ops.q_sample(x_0, u, t, self.num_timesteps, self.mask_id)
        keep = mask.bool()
        return x_t, [torch.where(keep, g, torch.full_like(g, -1)) for g in x_0_gt_list], keep

    @torch.no_grad()
    def _train_loss(self, x_0, x_0_gt_list, t=None, u=None):
        """-> (loss.mean(), vb_loss.mean()) like transformer_model.py:233-274; the per-sample
        summed cross entropy is kept in self.cross_entropy_loss."""
        P = self.P
        b, T = x_0.shape
        if t is None:
            t, pt = self.sample_time(b, x_0.device, 'uniform')
        else:
            t = t.to(self.device).long()
            pt = torch.ones_like(t).float() / self.num_timesteps
        if self.mask_schedule != 'random':
            raise NotImplementedError
        u = torch.rand_like(x_0.float()) if u is None else u.to(self.device, torch.float32)
        x_t, mask = ops.q_sample(x_0, u, t, self.num_timesteps, self.mask_id)
        hidden = self._denoise_fn.hidden(x_t, self.segm_tokens.contiguous(), self.texture_tokens)
        gt = torch.stack([g.reshape(-1) for g in x_0_gt_list]).contiguous()
        _, ce = ops.masked_ce_heads(hidden, P['tf.ln_f.g'], P['tf.ln_f.b'], P['tf.heads'],
                                    self.texture_tokens.reshape(-1), mask.reshape(-1), gt, b, T)
        self.cross_entropy_loss = ce
        numel = T
        vb_loss = ce / t / pt / (math.log(2) * numel)
        if self.loss_type == 'elbo':
            loss = vb_loss
        elif self.loss_type == 'mlm':
            denom = mask.float().sum(1)
            denom[denom == 0] = 1
            loss = ce / denom
        elif self.loss_type == 'reweighted_elbo':
            loss = (1 - (t / self.num_timesteps)) * ce / (math.log(2) * numel)
        else:
            raise ValueError(self.loss_type)
        return loss.mean(), vb_loss.mean()
