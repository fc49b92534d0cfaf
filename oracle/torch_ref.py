"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A plain-PyTorch fp32 *functional* restatement of the Text2Human sampling hot
path, written against ``state_dict`` dictionaries in
the reference's checkpoint layout.  Only ``tests/``, ``bench.py``'s
``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import it, and only as
the checker.  The product (``text2human_amd``) never routes through it.

Pinning: the reference has no tests or golden vectors of its own (SURVEY.md
section 4), so this restatement is pinned by *executing the unmodified
reference modules* in the build container (``oracle/ref_shim.py``):
``tests/test_oracle_vs_reference.py`` (skipped where /root/reference is absent)
checks every function below against the corresponding reference module, and
``oracle/make_golden.py`` stores reference outputs under ``tests/golden/``
which ``tests/test_oracle_golden.py`` re-checks anywhere.

Every function cites the reference file:line it restates (paths relative to
/root/reference).  Layout is the reference's NCHW / [B,T,C]; dtype fp32;
indices int64.
"""
import math

import torch
import torch.nn.functional as F


def _sub(sd, prefix):
    """View of the entries of `sd` under `prefix.` with the prefix removed."""
    p = prefix + '.'
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


# ---------------------------------------------------------------- VQGAN blocks


def swish(x):
    """nonlinearity, models/archs/vqgan_arch.py:510-512."""
    return x * torch.sigmoid(x)


def group_norm(x, sd, name):
    """Normalize = GroupNorm(32, C, eps=1e-6, affine), vqgan_arch.py:515-517."""
    return F.group_norm(x, 32, sd[f'{name}.weight'], sd[f'{name}.bias'], 1e-6)


def conv(x, sd, name, stride=1, padding=0):
    return F.conv2d(x, sd[f'{name}.weight'], sd.get(f'{name}.bias'), stride, padding)


def resnet_block(x, sd, name):
    """ResnetBlock.forward with temb=None, dropout p=0, vqgan_arch.py:597-617."""
    h = conv(swish(group_norm(x, sd, f'{name}.norm1')), sd, f'{name}.conv1', 1, 1)
    h = conv(swish(group_norm(h, sd, f'{name}.norm2')), sd, f'{name}.conv2', 1, 1)
    if f'{name}.nin_shortcut.weight' in sd:
        x = conv(x, sd, f'{name}.nin_shortcut')
    return x + h


def attn_block(x, sd, name):
    """AttnBlock.forward, vqgan_arch.py:636-661: single-head spatial attention,
    scale C^-0.5, softmax over keys, residual."""
    b, c, hh, ww = x.shape
    n = hh * ww
    hn = group_norm(x, sd, f'{name}.norm')
    q = conv(hn, sd, f'{name}.q').reshape(b, c, n)
    k = conv(hn, sd, f'{name}.k').reshape(b, c, n)
    v = conv(hn, sd, f'{name}.v').reshape(b, c, n)
    w = torch.bmm(q.transpose(1, 2), k) * (int(c)**(-0.5))  # [b, query, key]
    w = torch.softmax(w, dim=2)
    o = torch.bmm(v, w.transpose(1, 2)).reshape(b, c, hh, ww)
    return x + conv(o, sd, f'{name}.proj_out')


def upsample(x, sd, name):
    """Upsample.forward: nearest x2 then 3x3 pad 1, vqgan_arch.py:529-534."""
    x = F.interpolate(x, scale_factor=2.0, mode='nearest')
    return conv(x, sd, f'{name}.conv', 1, 1)


def downsample(x, sd, name):
    """Downsample.forward: zero-pad right/bottom by 1, 3x3 stride 2,
    vqgan_arch.py:546-551."""
    x = F.pad(x, (0, 1, 0, 1), mode='constant', value=0)
    return conv(x, sd, f'{name}.conv', 2, 0)


def _count(sd, prefix):
    idx = set()
    p = prefix + '.'
    for k in sd:
        if k.startswith(p):
            idx.add(int(k[len(p):].split('.')[0]))
    return len(idx)


def encoder(x, sd):
    """Encoder.forward, vqgan_arch.py:892-919 (structure read off the
    state_dict: levels = down.*, blocks = down.i.block.*, optional attn)."""
    h = conv(x, sd, 'conv_in', 1, 1)
    n_levels = _count(sd, 'down')
    for lv in range(n_levels):
        for blk in range(_count(sd, f'down.{lv}.block')):
            h = resnet_block(h, sd, f'down.{lv}.block.{blk}')
            if f'down.{lv}.attn.{blk}.norm.weight' in sd:
                h = attn_block(h, sd, f'down.{lv}.attn.{blk}')
        if f'down.{lv}.downsample.conv.weight' in sd:
            h = downsample(h, sd, f'down.{lv}.downsample')
    h = resnet_block(h, sd, 'mid.block_1')
    h = attn_block(h, sd, 'mid.attn_1')
    h = resnet_block(h, sd, 'mid.block_2')
    return conv(swish(group_norm(h, sd, 'norm_out')), sd, 'conv_out', 1, 1)


def decoder(z, sd, bot_h=None):
    """Decoder.forward, vqgan_arch.py:1000-1033.  The bottom-level residual is
    added right after the level-4 upsample (:1023-1024)."""
    h = conv(z, sd, 'conv_in', 1, 1)
    h = resnet_block(h, sd, 'mid.block_1')
    h = attn_block(h, sd, 'mid.attn_1')
    h = resnet_block(h, sd, 'mid.block_2')
    n_levels = _count(sd, 'up')
    for lv in reversed(range(n_levels)):
        for blk in range(_count(sd, f'up.{lv}.block')):
            h = resnet_block(h, sd, f'up.{lv}.block.{blk}')
            if f'up.{lv}.attn.{blk}.norm.weight' in sd:
                h = attn_block(h, sd, f'up.{lv}.attn.{blk}')
        if lv != 0:
            h = upsample(h, sd, f'up.{lv}.upsample')
        if lv == 4 and bot_h is not None:
            h = h + bot_h
    return conv(swish(group_norm(h, sd, 'norm_out')), sd, 'conv_out', 1, 1)


def decoder_res(z, sd):
    """DecoderRes.forward, vqgan_arch.py:1136-1151."""
    h = conv(z, sd, 'conv_in', 1, 1)
    h = resnet_block(h, sd, 'mid.block_1')
    h = attn_block(h, sd, 'mid.attn_1')
    return resnet_block(h, sd, 'mid.block_2')


# ---------------------------------------------------------------- quantizers


def vq_l2_argmin(z_flat, codebook):
    """VectorQuantizer.forward distance + argmin, vqgan_arch.py:88-92:
    d = sum z^2 + sum e^2 - 2 z.e^T (expanded form), first minimum wins."""
    d = (z_flat**2).sum(1, keepdim=True) + (codebook**2).sum(1) \
        - 2 * z_flat @ codebook.t()
    return torch.argmin(d, dim=1)


def segm_tokens(segm, enc_sd, quant_conv_sd, codebook, num_classes=24):
    """BaseSampleModel.get_quantized_segm, models/sample_model.py:330-340.
    segm: f32/i64 [B,1,H,W] class ids -> i64 [B, H/16, W/16]."""
    one_hot = F.one_hot(segm.squeeze(1).long(), num_classes).permute(0, 3, 1, 2).float()
    z = encoder(one_hot, enc_sd)
    z = F.conv2d(z, quant_conv_sd['weight'], quant_conv_sd['bias'])
    b, c, h, w = z.shape
    idx = vq_l2_argmin(z.permute(0, 2, 3, 1).reshape(-1, c), codebook)
    return idx.view(b, h, w)


def texture_tokens(texture_mask, shape=(32, 16)):
    """F.interpolate(mask, (32,16), 'nearest') -> source pixel (16i,16j),
    sample_model.py:187-188,264-266; vqgan_arch.py:291-292,465-466."""
    return F.interpolate(texture_mask, shape, mode='nearest').view(
        texture_mask.shape[0], -1).long()


def top_codebook_entry(indices_list, texture_mask, books_sd, shape_hw=(32, 16)):
    """VectorQuantizerTexture.get_codebook_entry, vqgan_arch.py:289-309.
    indices_list: 18 x i64 [B, T]; returns f32 [B, 256, h, w]."""
    tex = texture_tokens(texture_mask, shape_hw).view(-1)
    e_dim = books_sd['embedding_list.0.weight'].shape[1]
    zq = torch.zeros(tex.numel(), e_dim, device=tex.device)
    for cb in range(18):
        sel = tex == cb
        if sel.any():
            zq[sel] = books_sd[f'embedding_list.{cb}.weight'][
                indices_list[cb].reshape(-1)[sel]]
    b = texture_mask.shape[0]
    return zq.view(b, shape_hw[0], shape_hw[1], e_dim).permute(0, 3, 1, 2).contiguous()


def bot_codebook_entry(indices_list, texture_mask, books_sd, shape_hw=(32, 16),
                       spatial=2):
    """VectorQuantizerSpatialTextureAware.get_codebook_entry,
    vqgan_arch.py:463-486: gather 1024-d entries = [c, kh, kw] patches and
    F.fold them (k=2, s=2) -> f32 [B, 256, 2h, 2w]."""
    tex = texture_tokens(texture_mask, shape_hw).view(-1)
    e_dim = books_sd['embedding_list.0.weight'].shape[1]
    zq = torch.zeros(tex.numel(), e_dim, device=tex.device)
    for cb in range(18):
        sel = tex == cb
        if sel.any():
            zq[sel] = books_sd[f'embedding_list.{cb}.weight'][
                indices_list[cb].reshape(-1)[sel]]
    b = texture_mask.shape[0]
    h, w = shape_hw
    return F.fold(zq.view(b, h * w, e_dim).permute(0, 2, 1),
                  (h * spatial, w * spatial), kernel_size=spatial, stride=spatial)


# ---------------------------------------------------------------- encode side
# (image -> tokens; hierarchy_inference_model.py:170-197)


def texture_vq_forward(z, texture_mask, books_sd):
    """VectorQuantizerTexture.forward, vqgan_arch.py:212-287: every latent pixel is
    quantised with the codebook of its (nearest-resized) texture id.
    z f32 [B, C, h, w] -> (z_q f32 [B, C, h, w], 18 x i64 [B, h, w], -1 off-texture)."""
    b, c, h, w = z.shape
    tex = F.interpolate(texture_mask, (h, w), mode='nearest').view(-1)
    zf = z.permute(0, 2, 3, 1).reshape(-1, c)
    zq = torch.zeros_like(zf)
    idx_lists = []
    for cb in range(18):
        idx = torch.full((tex.numel(), ), -1, dtype=torch.long, device=tex.device)
        sel = tex == cb
        if sel.any():
            book = books_sd[f'embedding_list.{cb}.weight']
            found = vq_l2_argmin(zf[sel], book)
            zq[sel] = book[found]
            idx[sel] = found
        idx_lists.append(idx.view(b, h, w))
    return zq.view(b, h, w, c).permute(0, 3, 1, 2).contiguous(), idx_lists


def spatial_texture_vq_forward(z, texture_mask, books_sd, spatial=2):
    """VectorQuantizerSpatialTextureAware.forward, vqgan_arch.py:375-461: the latent is cut
    into spatial x spatial patches (F.unfold, entry layout [c, kh, kw]) and every patch is
    quantised with the codebook of its texture id.
    z f32 [B, C, H, W] -> (z_q f32 [B, C, H, W], 18 x i64 [B, H/s, W/s])."""
    b, c, hh, ww = z.shape
    h, w = hh // spatial, ww // spatial
    tex = F.interpolate(texture_mask, (h, w), mode='nearest').view(-1)
    patches = F.unfold(z, (spatial, spatial), stride=spatial).permute(0, 2, 1)  # [B, h*w, C*s*s]
    pf = patches.reshape(-1, c * spatial * spatial)
    zq = torch.zeros_like(pf)
    idx_lists = []
    for cb in range(18):
        idx = torch.full((tex.numel(), ), -1, dtype=torch.long, device=tex.device)
        sel = tex == cb
        if sel.any():
            book = books_sd[f'embedding_list.{cb}.weight']
            found = vq_l2_argmin(pf[sel], book)
            zq[sel] = book[found]
            idx[sel] = found
        idx_lists.append(idx.view(b, h, w))
    zq = F.fold(zq.view(patches.shape).permute(0, 2, 1), (hh, ww), kernel_size=spatial, stride=spatial)
    return zq, idx_lists


def encode_latents(image, sds):
    """The tensors the two texture-aware quantizers see (hierarchy_inference_model.py:171-172,188-189):
    (top f32 [B,256,32,16], bottom f32 [B,256,64,32]) = Encoder -> 1x1 quant_conv of each level."""
    zt = F.conv2d(encoder(image, sds['top_encoder']), sds['top_quant_conv']['weight'], sds['top_quant_conv']['bias'])
    zb = F.conv2d(encoder(image, sds['bot_encoder']), sds['bot_quant_conv']['weight'], sds['bot_quant_conv']['bias'])
    return zt, zb


def top_encode(image, texture_mask, sds):
    """top_encode, hierarchy_inference_model.py:170-176: Encoder -> 1x1 quant_conv ->
    texture-routed VQ -> 1x1 post_quant_conv.  Returns (quant_t [B,256,32,16], idx lists)."""
    hcur = encoder(image, sds['top_encoder'])
    hcur = F.conv2d(hcur, sds['top_quant_conv']['weight'], sds['top_quant_conv']['bias'])
    zq, idx_lists = texture_vq_forward(hcur, texture_mask, sds['top_quantize'])
    quant = F.conv2d(zq, sds['top_post_quant_conv']['weight'], sds['top_post_quant_conv']['bias'])
    return quant, idx_lists


def bot_encode(image, texture_mask, sds, spatial=2):
    """bot_encode, hierarchy_inference_model.py:187-192: the 18 bottom index maps."""
    hcur = encoder(image, sds['bot_encoder'])
    hcur = F.conv2d(hcur, sds['bot_quant_conv']['weight'], sds['bot_quant_conv']['bias'])
    _, idx_lists = spatial_texture_vq_forward(hcur, texture_mask, sds['bot_quantize'], spatial)
    return idx_lists


def index_to_image(quant_t, bot_idx_lists, texture_mask, sds):
    """index_to_image, hierarchy_inference_model.py:198-209."""
    b, h, w = bot_idx_lists[0].shape
    qb = bot_codebook_entry([i.view(b, -1) for i in bot_idx_lists], texture_mask, sds['bot_quantize'], (h, w))
    qb = F.conv2d(qb, sds['bot_post_quant_conv']['weight'], sds['bot_post_quant_conv']['bias'])
    return decoder(quant_t, sds['decoder'], bot_h=decoder_res(qb, sds['bot_decoder_res']))


def reconstruct(image, texture_mask, sds):
    """get_gt_indices + index_to_image: image -> (top, bottom) tokens -> image."""
    quant_t, top_idx = top_encode(image, texture_mask, sds)
    bot_idx = bot_encode(image, texture_mask, sds)
    return index_to_image(quant_t, bot_idx, texture_mask, sds), dict(
        quant_t=quant_t, top_indices=top_idx, bot_indices=bot_idx)


# ---------------------------------------------------------------- sampler training forward
# (models/transformer_model.py:186-274, forward only)


def q_sample(x_0, x_0_gt_list, t, u, num_timesteps, mask_id):
    """TransformerTextureAwareModel.q_sample, transformer_model.py:212-231, with the uniform
    draw `u` (= torch.rand_like(x_t.float())) passed in.  Returns (x_t, gt lists with -1 at
    unmasked positions, mask)."""
    mask = u < (t.float().unsqueeze(-1) / num_timesteps)
    x_t = x_0.clone()
    x_t[mask] = mask_id
    out = []
    for gt in x_0_gt_list:
        g = gt.clone()
        g[~mask] = -1
        out.append(g)
    return x_t, out, mask


def train_loss(x_0, x_0_gt_list, segm_tok, tex_tok, sd, t, u, num_timesteps=256, mask_id=18432,
               loss_type='reweighted_elbo', n_head=8):
    """TransformerTextureAwareModel._train_loss, transformer_model.py:233-274 (uniform time
    sampling: pt = 1 / num_timesteps): returns (loss.mean(), vb_loss.mean()) and the per-sample
    summed cross entropy."""
    import math
    x_t, gt_ignore, mask = q_sample(x_0, x_0_gt_list, t, u, num_timesteps, mask_id)
    logits = transformer_logits(x_t, segm_tok, tex_tok, sd, n_head=n_head)
    ce = 0
    for lg, gt in zip(logits, gt_ignore):
        ce = ce + F.cross_entropy(lg.permute(0, 2, 1), gt, ignore_index=-1, reduction='none').sum(1)
    pt = torch.ones_like(t).float() / num_timesteps
    numel = x_0.shape[1:].numel()
    vb = ce / t / pt / (math.log(2) * numel)
    if loss_type == 'elbo':
        loss = vb
    elif loss_type == 'mlm':
        denom = mask.float().sum(1)
        denom[denom == 0] = 1
        loss = ce / denom
    elif loss_type == 'reweighted_elbo':
        loss = (1 - (t / num_timesteps)) * ce / (math.log(2) * numel)
    else:
        raise ValueError(loss_type)
    return loss.mean(), vb.mean(), ce


# ---------------------------------------------------------------- transformer


def transformer_block(x, sd, p, n_head):
    """Block.forward + CausalSelfAttention.forward (causal=False),
    models/archs/transformer_arch.py:37-71,91-99."""
    b, t, c = x.shape
    hd = c // n_head
    h = F.layer_norm(x, (c, ), sd[f'{p}.ln1.weight'], sd[f'{p}.ln1.bias'], 1e-5)

    def heads(nm):
        y = F.linear(h, sd[f'{p}.attn.{nm}.weight'], sd[f'{p}.attn.{nm}.bias'])
        return y.view(b, t, n_head, hd).transpose(1, 2)

    q, k, v = heads('query'), heads('key'), heads('value')
    att = torch.softmax((q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hd)), dim=-1)
    y = (att @ v).transpose(1, 2).contiguous().view(b, t, c)
    x = x + F.linear(y, sd[f'{p}.attn.proj.weight'], sd[f'{p}.attn.proj.bias'])
    h = F.layer_norm(x, (c, ), sd[f'{p}.ln2.weight'], sd[f'{p}.ln2.bias'], 1e-5)
    h = F.gelu(F.linear(h, sd[f'{p}.mlp.0.weight'], sd[f'{p}.mlp.0.bias']))
    return x + F.linear(h, sd[f'{p}.mlp.2.weight'], sd[f'{p}.mlp.2.bias'])


def transformer_hidden(idx, segm_tok, tex_tok, sd, n_head=8):
    """TransformerMultiHead.forward up to ln_f, transformer_arch.py:249-270."""
    t = idx.shape[1]
    x = sd['tok_emb.weight'][idx] + sd['pos_emb'][:, :t, :] \
        + sd['segm_emb.weight'][segm_tok] + sd['texture_emb.weight'][tex_tok]
    n_layers = _count(sd, 'blocks')
    for i in range(n_layers):
        x = transformer_block(x, sd, f'blocks.{i}', n_head)
    c = x.shape[-1]
    return F.layer_norm(x, (c, ), sd['ln_f.weight'], sd['ln_f.bias'], 1e-5)


def transformer_logits(idx, segm_tok, tex_tok, sd, n_head=8, heads=None):
    """18 bias-free heads, transformer_arch.py:271.  Returns list (None for
    heads not requested)."""
    x = transformer_hidden(idx, segm_tok, tex_tok, sd, n_head)
    n = _count(sd, 'head_list')
    return [F.linear(x, sd[f'head_list.{i}.weight'])
            if heads is None or i in heads else None for i in range(n)]


class TorchNoise:
    """Noise source that consumes torch's *global* generator exactly like the
    reference: rand([B,T]) per step (sample_model.py:286) and, per active head,
    a full [B*T, 1024] Exp(1) draw (Categorical.sample -> multinomial(1) ->
    exponential_, sample_model.py:305-306)."""

    def __init__(self, device='cpu'):
        self.device = device

    def uniform(self, step, shape):
        return torch.rand(shape, device=self.device)

    def exponential(self, step, head, shape):
        return torch.empty(shape, device=self.device).exponential_(1.0)


class SeededNoise:
    """Counter-based noise (seeded per (step, head) on the CPU generator) so a
    CPU oracle and a GPU implementation can consume bit-identical draws."""

    def __init__(self, seed, device='cpu'):
        self.seed, self.device = int(seed), device

    def _gen(self, a, b):
        g = torch.Generator(device='cpu')
        g.manual_seed(self.seed * 1000003 + a * 1009 + b)
        return g

    def uniform(self, step, shape):
        return torch.rand(shape, generator=self._gen(step, 999)).to(self.device)

    def exponential(self, step, head, shape):
        return torch.empty(shape).exponential_(
            1.0, generator=self._gen(step, head)).to(self.device)


def categorical_argmax(logits, expo):
    """Categorical(logits=l).sample() == argmax(softmax(l) / Exp(1)-noise):
    torch.distributions normalises logits (l - logsumexp), `probs` is their
    softmax, and multinomial(n=1) takes argmax(probs / q), q ~ Exp(1)
    [ATen fast path, verified against Categorical.sample() under one seed in
    tests/test_oracle_vs_reference.py]."""
    lp = logits - logits.logsumexp(-1, keepdim=True)
    probs = torch.softmax(lp, dim=-1)
    return torch.argmax(probs / expo, dim=-1)


def sample_fn(segm_tok, texture_mask, sd, sample_steps=256, temp=1.0,
              mask_id=18432, shape=(32, 16), n_head=8, noise=None,
              logits_fn=None, trace=None):
    """BaseSampleModel.sample_fn, models/sample_model.py:256-328.

    Returns list of 18 i64 [B, T] (-1 where the token is not of that texture).
    `noise` supplies the draws (TorchNoise reproduces the reference's use of
    the global generator); `logits_fn(x_t, heads)` may replace the transformer
    (tests inject logits); `trace` (list) receives per-step dicts."""
    noise = noise or TorchNoise(segm_tok.device)
    b = segm_tok.shape[0]
    t_len = shape[0] * shape[1]
    dev = segm_tok.device
    x_t = torch.full((b, t_len), mask_id, dtype=torch.long, device=dev)
    unmasked = torch.zeros_like(x_t, dtype=torch.bool)
    tex_tok = texture_tokens(texture_mask, shape).to(dev)
    tex_flat = tex_tok.view(-1)
    out = [torch.full((b * t_len, ), -1, dtype=torch.long, device=dev)
           for _ in range(18)]
    for t in range(sample_steps, 0, -1):
        changes = noise.uniform(t, (b, t_len)) < 1.0 / float(t)
        changes = torch.bitwise_xor(changes, torch.bitwise_and(changes, unmasked))
        unmasked = torch.bitwise_or(unmasked, changes)
        ch_flat = changes.view(-1)
        active = [cb for cb in range(18)
                  if bool((tex_flat[ch_flat] == cb).sum() > 0)]
        if logits_fn is None:
            logits_list = transformer_logits(x_t, segm_tok, tex_tok, sd, n_head,
                                             heads=set(active))
        else:
            logits_list = logits_fn(x_t, set(active))
        x_flat = x_t.view(-1).clone()
        for cb in active:
            lg = logits_list[cb] / temp
            expo = noise.exponential(t, cb, (b * t_len, lg.shape[-1]))
            x0 = categorical_argmax(lg.reshape(b * t_len, -1), expo)
            sel = torch.bitwise_and(ch_flat, tex_flat == cb)
            x_flat[sel] = x0[sel] + 1024 * cb
            out[cb][sel] = x0[sel]
        x_t = x_flat.view(b, t_len)
        if trace is not None:
            trace.append(dict(t=t, changes=changes.clone(), active=active,
                              x_t=x_t.clone()))
    return [o.view(b, t_len) for o in out]


# ---------------------------------------------------------------- UNet / FCN


def conv_module(x, sd, name, padding):
    """mmcv ConvModule in eval: ReLU(BN_running(conv_nobias(x)))
    (BN eps 1e-5)."""
    y = F.conv2d(x, sd[f'{name}.conv.weight'], None, 1, padding)
    y = F.batch_norm(y, sd[f'{name}.bn.running_mean'], sd[f'{name}.bn.running_var'],
                     sd[f'{name}.bn.weight'], sd[f'{name}.bn.bias'], False, 0.0, 1e-5)
    return F.relu(y)


def unet(x, sd, attr=None):
    """UNet.forward / ShapeUNet.forward, models/archs/unet_arch.py:470-481,
    657-674.  Returns dec_outs = [bottleneck, ..., full-res].  For ShapeUNet
    the attribute vector is broadcast-concatenated in front of every encoder
    stage, i.e. before that stage's MaxPool (:660-667)."""
    n_stages = _count(sd, 'encoder')
    enc_outs = []
    for i in range(n_stages):
        if attr is not None:
            bb, cc = attr.shape
            x = torch.cat([x, attr.view(bb, cc, 1, 1).expand(bb, cc, x.shape[2], x.shape[3])], 1)
        blk = 0
        if i != 0:
            x = F.max_pool2d(x, 2)
            blk = 1
        x = conv_module(x, sd, f'encoder.{i}.{blk}.convs.0', 1)
        x = conv_module(x, sd, f'encoder.{i}.{blk}.convs.1', 1)
        enc_outs.append(x)
    dec_outs = [x]
    for i in reversed(range(n_stages - 1)):
        # UpConvBlock.forward, unet_arch.py:99-107; InterpConv :243-314
        up = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
        up = conv_module(up, sd, f'decoder.{i}.upsample.interp_upsample.1', 0)
        x = torch.cat([enc_outs[i], up], 1)
        x = conv_module(x, sd, f'decoder.{i}.conv_block.convs.0', 1)
        x = conv_module(x, sd, f'decoder.{i}.conv_block.convs.1', 1)
        dec_outs.append(x)
    return dec_outs


def multihead_fcn(feat, sd, num_head=18):
    """MultiHeadFCNHead.forward (num_convs=1, concat_input=False, dropout =
    eval no-op), models/archs/fcn_arch.py:333-348.  feat = dec_outs[in_index]."""
    outs = []
    for h in range(num_head):
        y = conv_module(feat, sd, f'convs_list.{h}.0', 1)
        outs.append(F.conv2d(y, sd[f'conv_seg_head_list.{h}.weight'],
                             sd[f'conv_seg_head_list.{h}.bias']))
    return outs


def fcn_head(feat, sd):
    """FCNHead.forward, fcn_arch.py:218-225."""
    y = conv_module(feat, sd, 'convs.0', 1)
    return F.conv2d(y, sd['conv_seg.weight'], sd['conv_seg.bias'])


def bot_index_prediction(feature_top, texture_mask, unet_sd, head_sd, in_index=4):
    """BaseSampleModel.bot_index_prediction, sample_model.py:183-213, batched:
    feature_top [B,256,32,16] -> 18 x i64 [B,32,16] (-1 off-texture)."""
    b = feature_top.shape[0]
    tex = texture_tokens(texture_mask, (32, 16)).view(-1)
    out = [torch.full((b * 512, ), -1, dtype=torch.long, device=tex.device) for _ in range(18)]
    logits = multihead_fcn(unet(feature_top, unet_sd)[in_index], head_sd)
    for cb in range(18):
        roi = tex == cb
        if roi.any():
            pred = logits[cb].argmax(dim=1).view(-1)
            out[cb][roi] = pred[roi]
    return [o.view(b, 32, 16) for o in out]


# ---------------------------------------------------------------- pose front-end


def shape_attr_embedding(attr, sd, cls_num_list):
    """ShapeAttrEmbedding.forward, shape_attr_embedding_arch.py:23-35
    (LeakyReLU slope 0.01)."""
    parts = []
    for i, n in enumerate(cls_num_list):
        x = F.one_hot(attr[:, i], n).float()
        x = F.leaky_relu(F.linear(x, sd[f'attr_{i}.0.weight'], sd[f'attr_{i}.0.bias']))
        parts.append(F.linear(x, sd[f'attr_{i}.2.weight'], sd[f'attr_{i}.2.bias']))
    x = torch.cat(parts, 1)
    x = F.leaky_relu(F.linear(x, sd['fusion.0.weight'], sd['fusion.0.bias']))
    return F.linear(x, sd['fusion.2.weight'], sd['fusion.2.bias'])


def parsing_from_pose(pose, shape_attr, emb_sd, unet_sd, head_sd, cls_num_list,
                      in_index=4):
    """SampleFromPoseModel.generate_parsing_map, sample_model.py:431-437."""
    attr = shape_attr_embedding(shape_attr, emb_sd, cls_num_list)
    logits = fcn_head(unet(pose, unet_sd, attr)[in_index], head_sd)
    return logits.argmax(dim=1).unsqueeze(1), logits


def texture_map(segm, upper, lower, outer):
    """SampleFromPoseModel.generate_texture_map, sample_model.py:443-467."""
    mask = torch.zeros_like(segm)
    for b in range(segm.shape[0]):
        for attr, classes in ((upper[b], (1, 4)), (lower[b], (3, 5, 21)),
                              (outer[b], (2, ))):
            if int(attr) != 17:
                for c in classes:
                    mask[b][segm[b] == c] = int(attr) + 1
    return mask.to(torch.float32)


# ---------------------------------------------------------------- whole path


def refine_and_decode(top_indices, texture_mask, sds):
    """Body of BaseSampleModel.sample_and_refine after sample_fn,
    sample_model.py:220-246, batched over samples (every op is per-sample, so
    batching is exact).  Returns (image f32 [B,3,512,256] in [0,1], dict of
    intermediates)."""
    pq = sds['top_post_quant_conv']
    top_quant = top_codebook_entry(top_indices, texture_mask, sds['top_quantize'])
    top_quant = F.conv2d(top_quant, pq['weight'], pq['bias'])
    bot_idx = bot_index_prediction(top_quant, texture_mask, sds['guidance_encoder'],
                                   sds['index_decoder'])
    bq = sds['bot_post_quant_conv']
    quant_bot = bot_codebook_entry(bot_idx, texture_mask, sds['bot_quantize'])
    quant_bot = F.conv2d(quant_bot, bq['weight'], bq['bias'])
    bot_h = decoder_res(quant_bot, sds['bot_decoder_res'])
    dec = decoder(top_quant, sds['decoder'], bot_h=bot_h)
    img = ((dec + 1) / 2).clamp(0, 1)
    return img, dict(top_quant=top_quant, bot_idx=bot_idx, quant_bot=quant_bot,
                     bot_h=bot_h, dec=dec)


def to_uint8(img):
    """torchvision 0.8.2 save_image quantisation: mul(255).add(0.5).clamp(0,255)
    .to(uint8), HWC ([3p-memory])."""
    return img.mul(255).add(0.5).clamp(0, 255).permute(0, 2, 3, 1).to(torch.uint8)


def sample_from_parsing(segm, texture_mask, sds, sample_steps=256, noise=None):
    """SampleFromParsingModel.feed_data + sample_and_refine,
    sample_model.py:347-360,215-254."""
    tok = segm_tokens(segm, sds['segm_encoder'], sds['segm_quant_conv'],
                      sds['segm_quantizer']['embedding.weight'])
    tok = tok.view(segm.shape[0], -1)
    top = sample_fn(tok, texture_mask, sds['sampler'], sample_steps, noise=noise)
    img, inter = refine_and_decode(top, texture_mask, sds)
    inter.update(segm_tokens=tok, top_indices=top)
    return img, inter
