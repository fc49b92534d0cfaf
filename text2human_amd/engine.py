"""Host-side composition of the HIP kernels into the stages of the sampling
path: segm tokenizer (T), index sampler (S),
feed-forward index refinement (R), hierarchical decode (D), pose front-end (P).

Everything here is batched over images (the reference decodes one image at a
time, models/sample_model.py:220; every op is per-sample so batching is exact)
and keeps activations as NHWC pixel rows [B*H*W, C] / token rows [B*T, C] in
HBM.  PyTorch only allocates tensors and provides the stream.
"""
import os

import torch

from . import _lib, ops
from .ops import ACT_GELU, ACT_NONE, ACT_RELU, PRO_NONE, PRO_SWISH


# ---------------------------------------------------------------- VQGAN stacks


class VQGANStack:
    """Encoder / Decoder / DecoderRes forward over packed params `P` (see
    weights.pack_vqgan) under prefix `name`."""

    def __init__(self, P, name, desc):
        self.P, self.name, self.desc = P, name, desc

    def _ws(self, key, hw, mode='same'):
        """Split-row weights of `key` if the stack was packed with them (weights.add_split_conv_weights)
        and t2h_conv_split_f32 serves the shape, else None (-> exact-fp32 kernel)."""
        ws = self.P.t.get(key + 's')
        return ws if ws is not None and ops.conv_split_ok(hw, mode) else None

    def _gn(self, x, pfx, n_img, hw):
        P = self.P
        return ops.groupnorm_tables(x, P[f'{pfx}.g'], P[f'{pfx}.b'], n_img, hw)

    def _norm_conv3x3(self, x, norm, conv, n_img, h, w, mode='same', residual=None):
        """conv(swish(GroupNorm(x))) (+ residual): GroupNorm + swish fused into the conv's operand
        staging (exact-fp32 kernel), or -- split path -- applied by ONE elementwise pass that also
        writes the two fp16 planes the split-precision conv reads (norm None: plain conv)."""
        P = self.P
        cin = P[f'{conv}.w'].shape[1] // 9
        cout = P[f'{conv}.w'].shape[0]
        hw_out = h * w * (4 if mode == 'up' else 1)
        ws = self._ws(f'{conv}.w', hw_out, mode)
        pro = self._gn(x, norm, n_img, h * w) if norm is not None else None
        if ws is not None:
            xs = ops.gn_apply_split(x, *(pro or (None, None)), rows_per_img=h * w,
                                    act=PRO_SWISH if pro is not None else PRO_NONE)
            return ops.conv_split(xs, ws, n_img, h, w, cin, cout, bias=P[f'{conv}.b'], residual=residual, mode=mode,
                                  gn_stats=cout <= 1024)
        return ops.conv3x3(x, P[f'{conv}.w'], n_img, h, w, cin, bias=P[f'{conv}.b'], mode=mode, residual=residual,
                           pro=(pro[0], pro[1], PRO_SWISH) if pro is not None else None)

    def _conv1x1(self, x, key, n_rows_img, pro=None, residual=None):
        """1x1 convolution of pixel rows (pro = GroupNorm tables applied without activation)."""
        P = self.P
        ws = self._ws(f'{key}.w', n_rows_img)
        if ws is not None:
            xs = ops.gn_apply_split(x, *(pro or (None, None)), rows_per_img=n_rows_img)
            cout, cin = P[f'{key}.w'].shape
            return ops.conv_split(xs, ws, x.shape[0] // n_rows_img, n_rows_img, 1, cin, cout, taps=1,
                                  bias=P[f'{key}.b'], residual=residual, gn_stats=cout <= 1024)
        return ops.gemm(x, P[f'{key}.w'], bias=P[f'{key}.b'], residual=residual,
                        pro=(pro[0], pro[1], n_rows_img, PRO_NONE) if pro is not None else None)

    def resblock(self, x, pfx, n_img, h, w):
        """ResnetBlock.forward (vqgan_arch.py:597-617): GroupNorm + swish go with the following conv's
        operand, bias + skip into its epilogue."""
        t = self._norm_conv3x3(x, f'{pfx}.norm1', f'{pfx}.conv1', n_img, h, w)
        skip = x
        if f'{pfx}.nin.w' in self.P:
            skip = self._conv1x1(x, f'{pfx}.nin', h * w)
        return self._norm_conv3x3(t, f'{pfx}.norm2', f'{pfx}.conv2', n_img, h, w, residual=skip)

    def attnblock(self, x, pfx, n_img, n):
        """AttnBlock.forward (vqgan_arch.py:636-661)."""
        c = x.shape[1]
        qkv = self._conv1x1(x, f'{pfx}.qkv', n, pro=self._gn(x, f'{pfx}.norm', n_img, n))
        q3 = qkv.view(n_img, n, 3 * c)
        s = torch.empty((n_img, n, n), device=x.device, dtype=torch.float32)
        ops.bgemm(q3[:, :, :c], q3[:, :, c:2 * c], s, alpha=float(int(c)**(-0.5)))
        ops.softmax_rows_(s)
        o = torch.empty((n_img, n, c), device=x.device, dtype=torch.float32)
        ops.bgemm(s, q3[:, :, 2 * c:], o, b_trans=True)
        return self._conv1x1(o.view(n_img * n, c), f'{pfx}.proj', n, residual=x)

    def _mid(self, h, n_img, hh, ww):
        nm = self.name
        h = self.resblock(h, f'{nm}.mid.block_1', n_img, hh, ww)
        h = self.attnblock(h, f'{nm}.mid.attn_1', n_img, hh * ww)
        return self.resblock(h, f'{nm}.mid.block_2', n_img, hh, ww)

    def _conv(self, x, pfx, n_img, h, w, mode='same', residual=None):
        """3x3 convolution without a norm in front (conv_in, Upsample / Downsample convs)."""
        if mode == 'down':
            P = self.P
            return ops.conv3x3(x, P[f'{pfx}.w'], n_img, h, w, P[f'{pfx}.w'].shape[1] // 9, bias=P[f'{pfx}.b'],
                               mode=mode, residual=residual)
        return self._norm_conv3x3(x, None, pfx, n_img, h, w, mode=mode, residual=residual)

    def encode(self, x, n_img, h, w):
        """Encoder.forward (vqgan_arch.py:892-919); x rows [n_img*h*w, Cin_pad]."""
        nm = self.name
        t = self._conv(x, f'{nm}.conv_in', n_img, h, w)
        for lv in self.desc['levels']:
            assert lv['kind'] == 'down'
            for b, blk in enumerate(lv['blocks']):
                t = self.resblock(t, f"{nm}.down.{lv['level']}.block.{b}", n_img, h, w)
                if blk['attn']:
                    t = self.attnblock(t, f"{nm}.down.{lv['level']}.attn.{b}", n_img, h * w)
            if lv['resample']:
                t = self._conv(t, f"{nm}.down.{lv['level']}.downsample", n_img, h, w, mode='down')
                h, w = h // 2, w // 2
        t = self._mid(t, n_img, h, w)
        return self._norm_conv3x3(t, f'{nm}.norm_out', f'{nm}.conv_out', n_img, h, w), h, w

    def decode(self, z, n_img, h, w, bot_h=None, upscale=False):
        """Decoder.forward (vqgan_arch.py:1000-1033).  `bot_h` is added in the
        epilogue of the level-4 Upsample conv (:1021-1024).  upscale=True feeds
        the nearest-x2 upsampled latent (fused into conv_in's operand load): the
        1024x512 interpretation of SURVEY.md 8(d), config 5."""
        nm = self.name
        t = self._conv(z, f'{nm}.conv_in', n_img, h, w, mode='up' if upscale else 'same')
        if upscale:
            h, w = 2 * h, 2 * w
        t = self._mid(t, n_img, h, w)
        levels = [lv for lv in self.desc['levels'] if lv['kind'] == 'up']
        for lv in sorted(levels, key=lambda d: -d['level']):
            for b, blk in enumerate(lv['blocks']):
                t = self.resblock(t, f"{nm}.up.{lv['level']}.block.{b}", n_img, h, w)
                if blk['attn']:
                    t = self.attnblock(t, f"{nm}.up.{lv['level']}.attn.{b}", n_img, h * w)
            if lv['resample']:
                res = bot_h if (lv['level'] == 4 and bot_h is not None) else None
                t = self._conv(t, f"{nm}.up.{lv['level']}.upsample", n_img, h, w, mode='up',
                               residual=res)
                h, w = 2 * h, 2 * w
            elif lv['level'] == 4 and bot_h is not None:
                t = t + bot_h
        return self._norm_conv3x3(t, f'{nm}.norm_out', f'{nm}.conv_out', n_img, h, w), h, w

    def decode_res(self, z, n_img, h, w, upscale=False):
        """DecoderRes.forward (vqgan_arch.py:1136-1151)."""
        t = self._conv(z, f'{self.name}.conv_in', n_img, h, w, mode='up' if upscale else 'same')
        if upscale:
            h, w = 2 * h, 2 * w
        return self._mid(t, n_img, h, w)


# ---------------------------------------------------------------- transformer


class SamplerNet:
    """TransformerMultiHead.forward (models/archs/transformer_arch.py:249-273)
    up to (not including) ln_f; ln_f + the routed head live in the sampling
    tail kernel.  24 x [LN, QKV GEMM, flash MHA, proj GEMM(+res), LN, fc1
    GEMM(+GELU), fc2 GEMM(+res)]."""

    def __init__(self, P, desc, n_head, name='tf', split=False, split_mha=True, n_streams=1):
        self.P, self.desc, self.n_head, self.name = P, desc, n_head, name
        self.split = split
        # n_streams > 1 (split path): the batch is cut into that many independent slices
        # whose 24-layer kernel chains run on separate HIP streams, so one slice's launch
        # latency / first-tile fill / epilogue tail overlaps the other's main loops
        self.n_streams = n_streams
        self._streams = None
        self._deferred = None
        # split_mha (with split): attention on the fp16 matrix cores too -- the q|k|v
        # projection writes q, k as split rows and v as transposed planes, no fp32 qkv
        self.split_mha = split_mha
        # fold_ln (T2H_FOLD_LN=1, with split + split_mha; off by default): no LayerNorm launches inside the
        # stack.  proj / fc2 (the writers of the residual stream) also emit split(x) and per-row partial
        # moments; q|k|v / fc1 run on split(x) with gamma folded into their weights and apply (mean, rstd)
        # in the epilogue (t2h_gemm_split_args.ln_part_*).  Layer 0's ln1 (input from the embedding) stays
        # a kernel.  Measured at B=8: the 46 LayerNorm launches it removes per step (5.1 us each) are paid
        # back almost entirely by the four GEMMs (+1.5 .. +5 us each): 816 vs 818-826 ms per batch.
        self.fold_ln = (split and split_mha and os.environ.get('T2H_FOLD_LN', '0') == '1'
                        and f'{name}.0.fc1.wf_split' in P)
        self._buf = {}

    def _buffers(self, M, C, dev):
        key = (M, C, str(dev))
        if key not in self._buf:
            e = lambda n: torch.empty((M, n), device=dev, dtype=torch.float32)
            self._buf = {key: dict(x=e(C), h=e(C), qkv=e(3 * C), y=e(C), u=e(4 * C),
                                   h_split=ops.split_rows_empty(M, C, dev), y_split=ops.split_rows_empty(M, C, dev),
                                   u_split=ops.split_rows_empty(M, 4 * C, dev),
                                   qk_split=ops.split_rows_empty(M, 3 * C, dev),
                                   x_split=ops.split_rows_empty(M, C, dev), ln_part=ops.ln_partials_empty(M, C, dev),
                                   vt=ops.vt_empty(M // 512 if M % 512 == 0 else 1, self.n_head, 512, dev))}
        return self._buf[key]

    # The LAST layer's row-wise tail (proj + residual, LayerNorm, fc1 + GELU, fc2 + residual) only
    # matters for the rows whose logits are sampled this step -- ~16 of 4096 at B=8.  With
    # defer_tail=True, hidden() stops after the last layer's attention; finish_tail() evaluates the
    # tail on the compacted changed rows once the host knows how many there are (the Linears, the
    # LayerNorm and the residual adds are row-wise, so those rows come out as in the full evaluation).
    TRIM_MAX_ROWS = 256

    def hidden(self, idx, segm_tok, tex_tok, defer_tail=False):
        P, nm = self.P, self.name
        B, T = idx.shape
        C = self.desc['C']
        buf = self._buffers(B * T, C, idx.device)
        x, h, qkv, y, u = buf['x'], buf['h'], buf['qkv'], buf['y'], buf['u']
        ops.embed_sum4(idx, segm_tok, tex_tok, P[f'{nm}.tok_emb'], P[f'{nm}.pos_emb'],
                       P[f'{nm}.segm_emb'], P[f'{nm}.tex_emb'], out=x)
        if self.split:
            # Split-precision path: the four Linears run on the fp16 matrix cores with
            # 2 x fp16 planes per operand (fp32-class accuracy, gemm_split.hip).  The
            # producers write split rows directly: LayerNorm -> h, attention -> y,
            # fc1's GELU epilogue -> u; the residual stream x and q|k|v stay fp32.
            M = B * T
            hs, ys, us = buf['h_split'], buf['y_split'], buf['u_split']
            vt = buf['vt']
            if self.split_mha and tuple(vt.shape) != (B, self.n_head, 3, C // self.n_head, T):
                vt = buf['vt'] = ops.vt_empty(B, self.n_head, T, idx.device, C // self.n_head)
            qks = buf['qk_split']
            ns = self.n_streams if (self.split_mha and B % max(self.n_streams, 1) == 0) else 1
            # batch slices: (rows lo:hi, batch size); all buffers are row-major over B*T rows
            Bs = B // ns
            sl = [(j * Bs * T, (j + 1) * Bs * T, j * Bs, (j + 1) * Bs) for j in range(ns)]

            fold = self.fold_ln and ns == 1
            xsp, part = buf['x_split'], buf['ln_part']
            L = self.desc['n_layers']

            def layer(i, lo, hi, b0, b1, tail=True):
                p = f'{nm}.{i}'
                m, xs = hi - lo, x[lo:hi]
                if fold and i > 0:  # split(x) and its row moments came with the previous layer's fc2
                    ops.gemm_split(xsp, P[f'{p}.qkv.wf_split'], m, 3 * C, C, out_split=qks, bias=P[f'{p}.qkv.bf'],
                                   vt=vt, vt_col0=2 * C, vt_T=T, vt_hd=C // self.n_head,
                                   ln_in=(part, P[f'{p}.qkv.cs']))
                    ops.mha_split(qks, 3 * C, vt, B, T, self.n_head, out_split=ys)
                else:
                    ops.layernorm_split(xs, P[f'{p}.ln1.g'], P[f'{p}.ln1.b'], hs[lo:hi])
                    if self.split_mha:
                        ops.gemm_split(hs[lo:hi], P[f'{p}.qkv.w_split'], m, 3 * C, C, out_split=qks[lo:hi],
                                       bias=P[f'{p}.qkv.b'], vt=vt[b0:b1], vt_col0=2 * C, vt_T=T,
                                       vt_hd=C // self.n_head)
                        ops.mha_split(qks[lo:hi], 3 * C, vt[b0:b1], b1 - b0, T, self.n_head, out_split=ys[lo:hi])
                    else:
                        ops.gemm_split(hs[lo:hi], P[f'{p}.qkv.w_split'], m, 3 * C, C, out=qkv[lo:hi],
                                       bias=P[f'{p}.qkv.b'])
                        ops.mha_noncausal_split(qkv[lo:hi], b1 - b0, T, self.n_head, ys[lo:hi])
                if not tail:
                    return
                if fold:
                    ops.gemm_split(ys, P[f'{p}.proj.w_split'], m, C, C, out=x, bias=P[f'{p}.proj.b'], residual=x,
                                   out_split=xsp, ln_part_out=part)
                    ops.gemm_split(xsp, P[f'{p}.fc1.wf_split'], m, 4 * C, C, out_split=us, bias=P[f'{p}.fc1.bf'],
                                   act=ACT_GELU, ln_in=(part, P[f'{p}.fc1.cs']))
                    nxt = i + 1 < L  # the last layer's output only feeds ln_f in the sampling tail
                    ops.gemm_split(us, P[f'{p}.fc2.w_split'], m, C, 4 * C, out=x, bias=P[f'{p}.fc2.b'], residual=x,
                                   out_split=xsp if nxt else None, ln_part_out=part if nxt else None)
                    return
                ops.gemm_split(ys[lo:hi], P[f'{p}.proj.w_split'], m, C, C, out=xs, bias=P[f'{p}.proj.b'],
                               residual=xs)
                ops.layernorm_split(xs, P[f'{p}.ln2.g'], P[f'{p}.ln2.b'], hs[lo:hi])
                ops.gemm_split(hs[lo:hi], P[f'{p}.fc1.w_split'], m, 4 * C, C, out_split=us[lo:hi],
                               bias=P[f'{p}.fc1.b'], act=ACT_GELU)
                ops.gemm_split(us[lo:hi], P[f'{p}.fc2.w_split'], m, C, 4 * C, out=xs, bias=P[f'{p}.fc2.b'],
                               residual=xs)

            self._deferred = None
            if ns == 1:
                for i in range(L - 1):
                    layer(i, *sl[0])
                layer(L - 1, *sl[0], tail=not defer_tail)
                if defer_tail:
                    self._deferred = (x, ys, M, C)
                return x
            if self._streams is None or len(self._streams) != ns:
                self._streams = [torch.cuda.Stream(device=idx.device) for _ in range(ns)]
            main = torch.cuda.current_stream()
            for st in self._streams:
                st.wait_stream(main)
            for i in range(self.desc['n_layers']):
                for j, st in enumerate(self._streams):
                    with torch.cuda.stream(st):
                        layer(i, *sl[j])
            for st in self._streams:
                main.wait_stream(st)
            return x
        for i in range(self.desc['n_layers']):
            p = f'{nm}.{i}'
            ops.layernorm(x, P[f'{p}.ln1.g'], P[f'{p}.ln1.b'], out=h)
            ops.gemm(h, P[f'{p}.qkv.w'], out=qkv, bias=P[f'{p}.qkv.b'])
            ops.mha_noncausal(qkv, B, T, self.n_head, out=y)
            ops.gemm(y, P[f'{p}.proj.w'], out=x, bias=P[f'{p}.proj.b'], residual=x)
            ops.layernorm(x, P[f'{p}.ln2.g'], P[f'{p}.ln2.b'], out=h)
            ops.gemm(h, P[f'{p}.fc1.w'], out=u, bias=P[f'{p}.fc1.b'], act=ACT_GELU)
            ops.gemm(u, P[f'{p}.fc2.w'], out=x, bias=P[f'{p}.fc2.b'], residual=x)
        return x

    def finish_tail(self, rows, n_rows):
        """After hidden(..., defer_tail=True): the last layer's row-wise tail.  Returns (hidden, compact):
        compact=True -> hidden[i] is row rows[i] (only the first n_rows rows of the list were evaluated)."""
        x, ys, M, C = self._deferred
        self._deferred = None
        P = self.P
        p = f"{self.name}.{self.desc['n_layers'] - 1}"
        if 0 < n_rows <= self.TRIM_MAX_ROWS:
            m = int(n_rows)
            xc, yc = ops.gather_rows(x, rows, m), ops.gather_rows(ys, rows, m)
            hc, uc = ops.split_rows_empty(m, C, x.device), ops.split_rows_empty(m, 4 * C, x.device)
            compact = True
        else:
            m, xc, yc, compact = M, x, ys, False
            buf = self._buffers(M, C, x.device)
            hc, uc = buf['h_split'], buf['u_split']
        if self.fold_ln:
            pc = ops.ln_partials_empty(m, C, x.device) if compact else self._buffers(M, C, x.device)['ln_part']
            ops.gemm_split(yc, P[f'{p}.proj.w_split'], m, C, C, out=xc, bias=P[f'{p}.proj.b'], residual=xc,
                           out_split=hc, ln_part_out=pc)
            ops.gemm_split(hc, P[f'{p}.fc1.wf_split'], m, 4 * C, C, out_split=uc, bias=P[f'{p}.fc1.bf'],
                           act=ACT_GELU, ln_in=(pc, P[f'{p}.fc1.cs']))
        else:
            ops.gemm_split(yc, P[f'{p}.proj.w_split'], m, C, C, out=xc, bias=P[f'{p}.proj.b'], residual=xc)
            ops.layernorm_split(xc, P[f'{p}.ln2.g'], P[f'{p}.ln2.b'], hc)
            ops.gemm_split(hc, P[f'{p}.fc1.w_split'], m, 4 * C, C, out_split=uc, bias=P[f'{p}.fc1.b'],
                           act=ACT_GELU)
        ops.gemm_split(uc, P[f'{p}.fc2.w_split'], m, C, 4 * C, out=xc, bias=P[f'{p}.fc2.b'], residual=xc)
        return xc, compact

    def logits(self, idx, segm_tok, tex_tok, heads=None):
        """Full [B*T, 1024] logits per head (tests / API parity only; the
        sampling loop never materialises them)."""
        P, nm = self.P, self.name
        x = self.hidden(idx, segm_tok, tex_tok)
        xf = ops.layernorm(x, P[f'{nm}.ln_f.g'], P[f'{nm}.ln_f.b'])
        B, T = idx.shape
        out = []
        for hd in range(self.desc['n_heads']):
            if heads is not None and hd not in heads:
                out.append(None)
                continue
            out.append(ops.gemm(xf, P[f'{nm}.heads'][hd]).view(B, T, -1))
        return out


class TorchDeviceNoise:
    """Default noise source: consumes torch's global generator of the GPU
    exactly like the reference does (rand per step; one full [B*T, 1024]
    exponential_ per ACTIVE head, the draw Categorical.sample() makes inside
    multinomial) so identical seeds give identical tokens."""

    def __init__(self, device):
        self.device = device

    def uniform(self, step, shape):
        return torch.rand(shape, device=self.device)

    def exponential(self, step, head, shape):
        return torch.empty(shape, device=self.device).exponential_(1.0)

    def reserve_exponential(self, heads, shape):
        """The same consumption of the generator WITHOUT the draws: advances torch's device generator by
        what one full `exponential_` of `shape` per head (ascending) would take and returns
        (seed, {head: offset before its draw}); t2h_sample_heads then computes the few elements it
        needs of those tensors itself (bit-identical, tests/test_gpu_kernels.py)."""
        dev = torch.device(self.device)
        index = dev.index if dev.index is not None else torch.cuda.current_device()  # (initialises CUDA)
        gen = torch.cuda.default_generators[index]
        _, inc = ops.torch_draw_geometry(shape[0] * shape[1], index)
        off = gen.get_offset()
        offsets = {}
        for h in sorted(heads):
            offsets[h] = off
            off += inc
        gen.set_offset(off)
        return gen.initial_seed(), offsets


class SplitOverflowError(_lib.T2HError):
    """An activation of the split-precision sampler left fp16's range."""


def check_split_overflow(what='sampler'):
    """Raises if a split-row producer flagged |x| >= 65504 since the last check (the fp16
    planes would hold inf / NaN).  There is deliberately no silent fallback: the caller reruns
    with T2H_SPLIT_GEMM=0 (exact-fp32 matrix instructions)."""
    if ops.split_overflow(reset=True):
        raise SplitOverflowError(
            f'{what}: an activation reached |x| >= 65504, outside the range of the 2 x fp16 split '
            'representation (include/t2h_hip.h); the result is invalid.  Rerun with T2H_SPLIT_GEMM=0 '
            '(exact-fp32 kernels).')


def sample_tokens(net, segm_tok, tex_tok, sample_steps, mask_id, temp=1.0, noise=None,
                  n_books=18, step_hook=None):
    """BaseSampleModel.sample_fn (models/sample_model.py:256-328) on device.

    Per step: one tiny kernel does the mask algebra and counts the changed
    tokens per texture head; ONE host read of those 18 counters decides which
    heads draw noise (the reference's data-dependent `if`, :301-302, which
    gates RNG consumption) -- it is issued before the transformer launches so
    the host wait overlaps the GPU work; then the texture-routed sampling tail
    runs once per active head.  Returns int64 [18, B*T] (-1 off-texture).

    step_hook(t, x_t, out) (tests only) runs after every step and may overwrite x_t in
    place, e.g. to teacher-force the reference's trajectory."""
    P, nm = net.P, net.name
    B, T = segm_tok.shape
    dev = segm_tok.device
    # the reference indexes texture_emb / head_list with these ids and raises on a bad one
    # (transformer_arch.py:262, sample_model.py:300-317); here they index device arrays
    lo, hi = int(tex_tok.min()), int(tex_tok.max())
    if lo < 0 or hi >= n_books:
        raise _lib.T2HError(f'texture ids must lie in [0, {n_books}), got [{lo}, {hi}]')
    noise = noise or TorchDeviceNoise(dev)
    n = B * T
    x_t = torch.full((B, T), mask_id, dtype=torch.int64, device=dev)
    unmasked = torch.zeros(n, dtype=torch.uint8, device=dev)
    changes = torch.zeros(n, dtype=torch.uint8, device=dev)
    out = torch.full((n_books, n), -1, dtype=torch.int64, device=dev)
    counts = torch.zeros(n_books + 1, dtype=torch.int32, device=dev)  # per head + total
    counts_host = torch.zeros(n_books + 1, dtype=torch.int32).pin_memory()
    rows = torch.empty(n, dtype=torch.int32, device=dev)               # compact list of changed rows
    tex_flat = tex_tok.reshape(-1).contiguous()
    n_class = P[f'{nm}.heads'].shape[1]
    ev = torch.cuda.Event()
    for t in range(sample_steps, 0, -1):
        rnd = noise.uniform(t, (B, T)).to(dev, torch.float32).contiguous()
        counts.zero_()
        ops.unmask_step(rnd, t, unmasked, changes, tex_flat, counts, rows, n_books)
        counts_host.copy_(counts, non_blocking=True)
        ev.record()
        defer = bool(getattr(net, 'split', False) and getattr(net, 'split_mha', False) and net.n_streams == 1
                     and os.environ.get('T2H_TRIM_LAST_LAYER', '1') != '0')
        hidden = net.hidden(x_t, segm_tok, tex_tok, defer_tail=defer)
        ev.synchronize()
        active = torch.nonzero(counts_host[:n_books]).flatten().tolist()
        if not active:
            continue
        compact = False
        if defer and net._deferred is not None:
            hidden, compact = net.finish_tail(rows, int(counts_host[n_books]))
        # the reference's draws: one full [n, 1024] tensor per active head in ascending head order.
        # On torch's device generator they are not materialised: the generator is advanced as if,
        # and the sampling tail computes the elements of those tensors it needs (the changed rows).
        # Other noise sources (tests replaying CPU draws) hand over explicit tensors.
        if hasattr(noise, 'reserve_exponential') and os.environ.get('T2H_PHILOX_TAIL', '1') != '0':
            expo, philox = {}, noise.reserve_exponential(active, (n, n_class))
        else:
            expo = {cb: noise.exponential(t, cb, (n, n_class)).to(dev, torch.float32).contiguous() for cb in active}
            philox = None
        ops.sample_heads(hidden, P[f'{nm}.ln_f.g'], P[f'{nm}.ln_f.b'], P[f'{nm}.heads'], expo, rows,
                         int(counts_host[n_books]), tex_flat, temp, x_t, out, philox=philox, hidden_compact=compact)
        if step_hook is not None:
            step_hook(t, x_t, out)
    if net.split:
        check_split_overflow('index sampler')
    return out


# ---------------------------------------------------------------- UNet + heads


class UNetStack:
    """UNet.forward / ShapeUNet.forward (models/archs/unet_arch.py:470-481,
    657-674) with BN folded; returns the full-resolution decoder output
    (dec_outs[4], the only one the heads read, in_index=4)."""

    def __init__(self, P, name, desc):
        self.P, self.name, self.desc = P, name, desc

    def _cm(self, x, pfx, n_img, h, w, **kw):
        P = self.P
        cin = P[f'{pfx}.w'].shape[1] // 9
        return ops.conv3x3(x, P[f'{pfx}.w'], n_img, h, w, cin, bias=P[f'{pfx}.b'], act=ACT_RELU, **kw)

    def _attr_bias_map(self, attr, pfx, n_img, h, w):
        """Per-pixel bias of the spatially constant attribute channels: sum over
        the taps that fall inside the image of (W_attr[tap] @ attr[b])."""
        P = self.P
        wattr = P[f'{pfx}.wattr']  # [Cout, 9, A]
        cout = wattr.shape[0]
        tapc = ops.gemm(attr, wattr.view(cout * 9, -1))  # [B, Cout*9] = [B, Cout, 9]
        return ops.tap_bias_map(tapc, n_img, h, w, cout)

    def forward(self, x, n_img, h, w, attr=None):
        nm = self.name
        n_st = len(self.desc['stages'])
        skips = []
        for i in range(n_st):
            if i != 0:
                x = ops.maxpool2(x, n_img, h, w)
                h, w = h // 2, w // 2
            if attr is not None:
                bm = self._attr_bias_map(attr, f'{nm}.enc.{i}.0', n_img, h, w)
                x = self._cm(x, f'{nm}.enc.{i}.0', n_img, h, w, residual=bm, res_pre=True)
            else:
                x = self._cm(x, f'{nm}.enc.{i}.0', n_img, h, w)
            x = self._cm(x, f'{nm}.enc.{i}.1', n_img, h, w)
            skips.append((x, h, w))
        for d in reversed(range(n_st - 1)):
            skip, sh_, sw_ = skips[d]
            up = ops.bilinear_up2(x, n_img, h, w)
            h, w = 2 * h, 2 * w
            P = self.P
            cs = skip.shape[1]
            cat = torch.empty((n_img * h * w, 2 * cs), device=x.device, dtype=torch.float32)
            cat[:, :cs].copy_(skip)
            ops.gemm(up, P[f'{nm}.dec.{d}.up.w'], out=cat[:, cs:], bias=P[f'{nm}.dec.{d}.up.b'],
                     act=ACT_RELU)
            x = self._cm(cat, f'{nm}.dec.{d}.0', n_img, h, w)
            x = self._cm(x, f'{nm}.dec.{d}.1', n_img, h, w)
        return x, h, w
