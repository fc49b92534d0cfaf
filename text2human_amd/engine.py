"""Host-side composition of the HIP kernels into the stages of the sampling
path: segm tokenizer (T), index sampler (S),
feed-forward index refinement (R), hierarchical decode (D), pose front-end (P).

Everything here is batched over images (the reference decodes one image at a
time, models/sample_model.py:220; every op is per-sample so batching is exact)
and keeps activations as NHWC pixel rows [B*H*W, C] / token rows [B*T, C] in
HBM.  PyTorch only allocates tensors and provides the stream.
"""
import gc
import os
import weakref

import numpy as np
import torch

from . import _lib, ops, schedule
from .ops import ACT_GELU, ACT_NONE, ACT_RELU, PRO_NONE, PRO_SWISH


# ---------------------------------------------------------------- VQGAN stacks


class VQGANStack:
    """Encoder / Decoder / DecoderRes forward over packed params `P` (see
    weights.pack_vqgan) under prefix `name`."""

    def __init__(self, P, name, desc):
        self.P, self.name, self.desc = P, name, desc
        self.use_split = True  # False: exact-fp32 convolutions even if split-row weights were packed (A/B, bench parity)
        # the decoders' AttnBlocks (N = 512 .. 8192 positions) MAY run flash-style when the N x N score tensor is too
        # large to allocate (attnblock); the encoders' (tokenizer: an argmin decision pinned to golden tokens) always
        # keep the materialised bmm -> softmax -> bmm form
        self.flash_attn = False

    def _ws(self, key, hw, mode='same'):
        """Split-row weights of `key` if the stack was packed with them (weights.add_split_conv_weights)
        and t2h_conv_split_f32 serves the shape, else None (-> exact-fp32 kernel)."""
        ws = self.P.t.get(key + 's') if self.use_split else None
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
        if (ws is not None and x.stride(0) * h * w * 4 < 2**31
                and ops.conv_halo_ok(n_img, h * (2 if mode == 'up' else 1), w * (2 if mode == 'up' else 1), cin, cout, mode)):
            # large levels: GroupNorm apply + swish + split happen while the 18 x 18-pixel halo of a 16 x 16 tile is
            # staged (csrc/conv_halo.hip) -- no elementwise pass, every input pixel read 1.27x instead of 9x
            return ops.conv_halo(x, ws, n_img, h, w, cin, cout, bias=P[f'{conv}.b'], residual=residual, mode=mode,
                                 pro=pro, gn_stats=cout <= 1024)
        if ws is not None:
            xs = ops.gn_apply_split(x, *(pro or (None, None)), rows_per_img=h * w,
                                    act=PRO_SWISH if pro is not None else PRO_NONE)
            return ops.conv_split(xs, ws, n_img, h, w, cin, cout, bias=P[f'{conv}.b'], residual=residual, mode=mode,
                                  gn_stats=cout <= 1024)
        if residual is None and ops.conv3x3_small_ok(cin, cout, mode):  # conv_out: 3 output channels
            return ops.conv3x3_small(x, P[f'{conv}.w'], n_img, h, w, cin, bias=P[f'{conv}.b'],
                                     pro=(pro[0], pro[1], PRO_SWISH) if pro is not None else None)
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
        # The materialised form (bmm -> softmax -> bmm) is the default: it is FASTER than the flash-style kernel at
        # every decoder shape (671 vs 845 us at N = 2048 / B = 8, 2501 vs 2981 us for two images at N = 8192,
        # profiles/r03_spatial_attention_bench.log -- both run at 80-110 TF of the exact-fp32 rate and the score
        # tensor's round trip is served by the Infinity Cache).  The flash-style kernel (no N x N tensor) takes over
        # only where the score tensor would not be reasonable to allocate (T2H_ATTN_SCORE_MB, default 2048 MB:
        # 268 MB per image at N = 8192).
        score_mb = n_img * n * n * 4 / 2**20
        if (self.flash_attn and ops.spatial_attention_ok(n, c)
                and score_mb > float(os.environ.get('T2H_ATTN_SCORE_MB', '2048'))):
            return self._conv1x1(ops.spatial_attention(qkv, n_img, n, c), f'{pfx}.proj', n, residual=x)
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
                raise _lib.T2HError('decoder level 4 has no Upsample conv to carry bot_h (vqgan_arch.py:1021-1024)')
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

    def __init__(self, P, desc, n_head, name='tf', split=False, split_mha=True, x8=None):
        self.P, self.desc, self.n_head, self.name = P, desc, n_head, name
        self.split = split
        self._deferred = None
        # x8 (default with the split path; T2H_X8=0 opts out): the four Linears read "x8" operands -- fp16 plane +
        # two e4m3 planes (csrc/common.h) -- and run both cross terms of a K tile in ONE 8-bit matrix instruction
        # (a third fewer matrix-pipe cycles; tools/cross_term_emulation.py: the hidden state moves by 4e-5 against a
        # tolerance of 2e-4).  Needs per-tensor power-of-two scales: weights from their own maximum, activations from
        # a calibration evaluation that depends on the CHECKPOINT ALONE (calibrate_x8: fixed synthetic states, run at
        # load) -- the scales, and with them every rounding, are a function of the weights, never of what the model
        # object has been fed before.  A value beyond a scale's range raises bit 1 of the overflow word and the caller
        # re-runs THAT call on the fp16 planes (X8RangeError).
        self.x8 = bool(split and split_mha and os.environ.get('T2H_X8', '1') != '0') if x8 is None else bool(x8)
        self._x8 = None  # {'w': {(layer, lin): scale}, 'a': {(layer, role): scale}, 'act_max': ...} once calibrated
        # split_mha (with split): attention on the fp16 matrix cores too -- the q|k|v
        # projection writes q, k as split rows and v as transposed planes, no fp32 qkv
        self.split_mha = split_mha
        self._buf = {}
        self._graphs = {}       # captured sampling rounds (RoundGraph), see sample_tokens
        self.last_stats = None  # schedule.stats of the last sample_tokens run
        self.last_launch_mode = None  # 'graph' (one replay per round) / 'eager' (individual launches)

    def _buffers(self, M, C, dev):
        key = (M, C, str(dev))
        if key not in self._buf:
            self._graphs = {}  # captured rounds hold pointers into the buffers replaced below
            e = lambda n: torch.empty((M, n), device=dev, dtype=torch.float32)
            R = self.TRIM_MAX_ROWS
            self._buf = {key: dict(x=e(C), h=e(C), qkv=e(3 * C), y=e(C), u=e(4 * C),
                                   h_split=ops.split_rows_empty(M, C, dev), y_split=ops.split_rows_empty(M, C, dev),
                                   u_split=ops.split_rows_empty(M, 4 * C, dev),
                                   qk_split=ops.split_rows_empty(M, 3 * C, dev),
                                   vt=ops.vt_empty(M // 512 if M % 512 == 0 else 1, self.n_head, 512, dev),
                                   # the last layer's tail on the changed rows (finish_tail)
                                   xc=torch.empty((R, C), device=dev, dtype=torch.float32),
                                   yc=ops.split_rows_empty(R, C, dev), hc=ops.split_rows_empty(R, C, dev),
                                   uc=ops.split_rows_empty(R, 4 * C, dev))}
        return self._buf[key]

    # The LAST layer's row-wise tail (proj + residual, LayerNorm, fc1 + GELU, fc2 + residual) only
    # matters for the rows whose logits are sampled this round -- ~16 of 4096 at B=8.  With
    # defer_tail=True, hidden() stops after the last layer's attention; finish_tail() evaluates the
    # tail on the compacted changed rows.  The Linears, the LayerNorm and the residual adds are
    # row-wise, so those rows equal the full evaluation's up to the summation order of the few-rows
    # GEMM kernel (K split over 8 waves): tolerance-level equality, not bitwise
    # (tests/test_gpu_edge_cases.py compares the tokens of both forms on the bench configuration).
    TRIM_MAX_ROWS = 256

    def hidden(self, idx, segm_tok, tex_tok, defer_tail=False, active=None):
        """active = k < B: only the first k samples of the batch, on the first k * T rows of the SAME buffers
        (samples never interact: attention is per sample) -- the samples of a compact schedule that have no step
        left sit at the end of the batch (schedule.leave_order) and are not evaluated."""
        P, nm = self.P, self.name
        B, T = idx.shape
        C = self.desc['C']
        full = buf = self._buffers(B * T, C, idx.device)
        if self.split and self.split_mha and tuple(full['vt'].shape) != (B, self.n_head, 2, C // self.n_head, T):
            self._graphs = {}  # (captured rounds hold a pointer to the buffer replaced here)
            full['vt'] = ops.vt_empty(B, self.n_head, T, idx.device, C // self.n_head)
        if active is not None and 0 < active < B:
            k = int(active)
            buf = {name: (v[:k] if name == 'vt' else v[:k * T]) for name, v in full.items()
                   if name not in ('xc', 'yc', 'hc', 'uc')}
            idx, segm_tok, tex_tok, B = idx[:k], segm_tok[:k], tex_tok[:k], k
        x, h, qkv, y, u = buf['x'], buf['h'], buf['qkv'], buf['y'], buf['u']
        ops.embed_sum4(idx, segm_tok, tex_tok, P[f'{nm}.tok_emb'], P[f'{nm}.pos_emb'],
                       P[f'{nm}.segm_emb'], P[f'{nm}.tex_emb'], out=x)
        L = self.desc['n_layers']
        self._deferred = None
        if self.split:
            # Split-precision path: the four Linears run on the fp16 matrix cores with
            # 2 x fp16 planes per operand (fp32-class accuracy, gemm_split.hip).  The
            # producers write split rows directly: LayerNorm -> h, attention -> y,
            # fc1's GELU epilogue -> u; the residual stream x stays fp32.
            M = B * T
            hs, ys, us, qks = buf['h_split'], buf['y_split'], buf['u_split'], buf['qk_split']
            vt = buf['vt']
            hd = C // self.n_head
            if self.x8 and self.split_mha and self._x8 is None:
                # (bare SamplerNets of tests / tools; the models calibrate when the checkpoint is packed.  It runs in
                # buffers of its own and may switch x8 off for a checkpoint that leaves fp16's range)
                self.ensure_x8()
            if self.x8 and self.split_mha:
                sw, sa = self._x8['w'], self._x8['a']
                for i in range(L):
                    p = f'{nm}.{i}'
                    ops.layernorm_x8(x, P[f'{p}.ln1.g'], P[f'{p}.ln1.b'], hs, sa[i, 'h1'])
                    ops.gemm_split(hs, P[f'{p}.qkv.w_x8'], M, 3 * C, C, out_split=qks, bias=P[f'{p}.qkv.b'],
                                   vt=vt, vt_col0=2 * C, vt_T=T, vt_hd=hd, x8=(sa[i, 'h1'], sw[i, 'qkv']))
                    ops.mha_split_x8(qks, 3 * C, vt, B, T, self.n_head, ys, sa[i, 'y'])
                    if i == L - 1 and defer_tail:
                        self._deferred = (full['x'], full['y_split'], M, C)
                        return x
                    ops.gemm_split(ys, P[f'{p}.proj.w_x8'], M, C, C, out=x, bias=P[f'{p}.proj.b'], residual=x,
                                   x8=(sa[i, 'y'], sw[i, 'proj']))
                    ops.layernorm_x8(x, P[f'{p}.ln2.g'], P[f'{p}.ln2.b'], hs, sa[i, 'h2'])
                    ops.gemm_split(hs, P[f'{p}.fc1.w_x8'], M, 4 * C, C, out_split=us, bias=P[f'{p}.fc1.b'],
                                   act=ACT_GELU, x8=(sa[i, 'h2'], sw[i, 'fc1']), out_x8_scale=sa[i, 'u'])
                    ops.gemm_split(us, P[f'{p}.fc2.w_x8'], M, C, 4 * C, out=x, bias=P[f'{p}.fc2.b'], residual=x,
                                   x8=(sa[i, 'u'], sw[i, 'fc2']))
                return x
            for i in range(L):
                p = f'{nm}.{i}'
                ops.layernorm_split(x, P[f'{p}.ln1.g'], P[f'{p}.ln1.b'], hs)
                if self.split_mha:
                    ops.gemm_split(hs, P[f'{p}.qkv.w_split'], M, 3 * C, C, out_split=qks, bias=P[f'{p}.qkv.b'],
                                   vt=vt, vt_col0=2 * C, vt_T=T, vt_hd=hd)
                    ops.mha_split(qks, 3 * C, vt, B, T, self.n_head, out_split=ys)
                else:
                    ops.gemm_split(hs, P[f'{p}.qkv.w_split'], M, 3 * C, C, out=qkv, bias=P[f'{p}.qkv.b'])
                    ops.mha_noncausal_split(qkv, B, T, self.n_head, ys)
                if i == L - 1 and defer_tail:
                    # (the WHOLE batch's buffers: finish_tail addresses rows by their index in the batch; only
                    # the first M rows -- the active samples -- were evaluated, and only they are listed)
                    self._deferred = (full['x'], full['y_split'], M, C)
                    return x
                ops.gemm_split(ys, P[f'{p}.proj.w_split'], M, C, C, out=x, bias=P[f'{p}.proj.b'], residual=x)
                ops.layernorm_split(x, P[f'{p}.ln2.g'], P[f'{p}.ln2.b'], hs)
                ops.gemm_split(hs, P[f'{p}.fc1.w_split'], M, 4 * C, C, out_split=us, bias=P[f'{p}.fc1.b'],
                               act=ACT_GELU)
                ops.gemm_split(us, P[f'{p}.fc2.w_split'], M, C, 4 * C, out=x, bias=P[f'{p}.fc2.b'], residual=x)
            return x
        for i in range(L):
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
        x, ys, M, C = self._deferred   # (x, ys: the whole batch's buffers; M: the rows that were evaluated)
        self._deferred = None
        P = self.P
        p = f"{self.name}.{self.desc['n_layers'] - 1}"
        buf = self._buffers(x.shape[0], C, x.device)
        if 0 < n_rows <= self.TRIM_MAX_ROWS:
            m = int(n_rows)
            xc, yc = ops.gather_rows(x, rows, m, out=buf['xc'][:m]), ops.gather_rows(ys, rows, m, out=buf['yc'][:m])
            hc, uc = buf['hc'][:m], buf['uc'][:m]
            compact = True
        else:
            m, xc, yc, compact = M, x, ys, False
            hc, uc = buf['h_split'], buf['u_split']
        if self.x8 and self._x8 is not None:
            i = self.desc['n_layers'] - 1
            sw, sa = self._x8['w'], self._x8['a']
            ops.gemm_split(yc, P[f'{p}.proj.w_x8'], m, C, C, out=xc, bias=P[f'{p}.proj.b'], residual=xc,
                           x8=(sa[i, 'y'], sw[i, 'proj']))
            ops.layernorm_x8(xc, P[f'{p}.ln2.g'], P[f'{p}.ln2.b'], hc, sa[i, 'h2'])
            ops.gemm_split(hc, P[f'{p}.fc1.w_x8'], m, 4 * C, C, out_split=uc, bias=P[f'{p}.fc1.b'], act=ACT_GELU,
                           x8=(sa[i, 'h2'], sw[i, 'fc1']), out_x8_scale=sa[i, 'u'])
            ops.gemm_split(uc, P[f'{p}.fc2.w_x8'], m, C, 4 * C, out=xc, bias=P[f'{p}.fc2.b'], residual=xc,
                           x8=(sa[i, 'u'], sw[i, 'fc2']))
            return xc, compact
        ops.gemm_split(yc, P[f'{p}.proj.w_split'], m, C, C, out=xc, bias=P[f'{p}.proj.b'], residual=xc)
        ops.layernorm_split(xc, P[f'{p}.ln2.g'], P[f'{p}.ln2.b'], hc)
        ops.gemm_split(hc, P[f'{p}.fc1.w_split'], m, 4 * C, C, out_split=uc, bias=P[f'{p}.fc1.b'], act=ACT_GELU)
        ops.gemm_split(uc, P[f'{p}.fc2.w_split'], m, C, 4 * C, out=xc, bias=P[f'{p}.fc2.b'], residual=xc)
        return xc, compact

    def ensure_x8(self):
        """x8 weights packed and activation scales calibrated (once per net; a no-op on the other paths)."""
        if self.x8 and self.split and self.split_mha and self._x8 is None:
            self.calibrate_x8()
        return self

    # The calibration states: X8_CAL_SAMPLES synthetic samples over the checkpoint's own vocabulary, drawn from a
    # fixed-seed MT19937 stream (numpy's frozen legacy generator: the same numbers on every version / box).
    X8_CAL_SEED = 0x78382D63
    X8_CAL_MASKED = (1.0, 0.0, 0.5, 0.25)  # masked fraction per sample: the state every run starts from, a finished one, two in between

    def x8_calibration_states(self):
        """(idx, segm_tok, tex_tok) int64 [4, T] on the CPU -- a function of the checkpoint's SHAPES only."""
        P, nm = self.P, self.name
        T, V = P[f'{nm}.pos_emb'].shape[0], P[f'{nm}.tok_emb'].shape[0]
        n_seg, n_tex, n_class = P[f'{nm}.segm_emb'].shape[0], P[f'{nm}.tex_emb'].shape[0], P[f'{nm}.heads'].shape[1]
        rs = np.random.RandomState(self.X8_CAL_SEED)
        B = len(self.X8_CAL_MASKED)
        tex = rs.randint(0, n_tex, size=(B, T)).astype(np.int64)
        tok = np.minimum(rs.randint(0, n_class, size=(B, T)).astype(np.int64) + n_class * tex, V - 2)
        u = rs.random_sample((B, T))
        idx = np.where(u < np.asarray(self.X8_CAL_MASKED)[:, None], V - 1, tok)  # V - 1: the mask id
        seg = rs.randint(0, n_seg, size=(B, T)).astype(np.int64)
        return torch.from_numpy(idx), torch.from_numpy(seg), torch.from_numpy(tex)

    def calibrate_x8(self):
        """One-time set-up of the x8 path, at load (init-time plumbing, not on the sampling path), a function of the
        CHECKPOINT ALONE: (1) the four Linears' weights as x8 rows, each matrix with the power-of-two scale that puts
        its own maximum into [128, 256); (2) the activation scales: ONE evaluation, on the fp16-plane kernels, of four
        fixed synthetic states (x8_calibration_states) in buffers of its own; the maximum of every producer's output
        (LayerNorm 1 / 2, attention, GELU) per layer is read from the fp16 plane by t2h_split_rows_absmax and scaled
        into [16, 32) -- 14x headroom before the 8-bit planes saturate, full 4-bit precision down to 2^-10 of the
        maximum.  Nothing the model is fed later changes a scale: same (checkpoint, input, seed) -> same roundings,
        in any process, on any rank, after any history (tests/test_gpu_x8.py)."""
        P, nm, C, L = self.P, self.name, self.desc['C'], self.desc['n_layers']
        dev = P.device
        lins, roles = ('qkv', 'proj', 'fc1', 'fc2'), ('h1', 'y', 'h2', 'u')
        bits = torch.zeros(2 * 4 * L, dtype=torch.int32, device=dev)  # [weights L x 4 | activations L x 4] maxima (fp32 bits)
        slot_w = lambda i, lin: i * 4 + lins.index(lin)
        slot_a = lambda i, role: 4 * L + i * 4 + roles.index(role)
        for i in range(L):
            for lin in lins:
                ops.absmax_f32(P[f'{nm}.{i}.{lin}.w'], bits[slot_w(i, lin):])
        idx, segm_tok, tex_tok = (t.to(dev) for t in self.x8_calibration_states())
        B, T = idx.shape
        M, hd = B * T, C // self.n_head
        x = torch.empty((M, C), device=dev, dtype=torch.float32)
        hs, ys = ops.split_rows_empty(M, C, dev), ops.split_rows_empty(M, C, dev)
        us, qks = ops.split_rows_empty(M, 4 * C, dev), ops.split_rows_empty(M, 3 * C, dev)
        vt = ops.vt_empty(B, self.n_head, T, dev, hd)
        ops.split_overflow(reset=True)
        ops.embed_sum4(idx, segm_tok, tex_tok, P[f'{nm}.tok_emb'], P[f'{nm}.pos_emb'], P[f'{nm}.segm_emb'],
                       P[f'{nm}.tex_emb'], out=x)
        for i in range(L):
            p = f'{nm}.{i}'
            ops.layernorm_split(x, P[f'{p}.ln1.g'], P[f'{p}.ln1.b'], hs)
            ops.split_rows_absmax(hs, M, C, bits[slot_a(i, 'h1'):])
            ops.gemm_split(hs, P[f'{p}.qkv.w_split'], M, 3 * C, C, out_split=qks, bias=P[f'{p}.qkv.b'], vt=vt,
                           vt_col0=2 * C, vt_T=T, vt_hd=hd)
            ops.mha_split(qks, 3 * C, vt, B, T, self.n_head, out_split=ys)
            ops.split_rows_absmax(ys, M, C, bits[slot_a(i, 'y'):])
            ops.gemm_split(ys, P[f'{p}.proj.w_split'], M, C, C, out=x, bias=P[f'{p}.proj.b'], residual=x)
            ops.layernorm_split(x, P[f'{p}.ln2.g'], P[f'{p}.ln2.b'], hs)
            ops.split_rows_absmax(hs, M, C, bits[slot_a(i, 'h2'):])
            ops.gemm_split(hs, P[f'{p}.fc1.w_split'], M, 4 * C, C, out_split=us, bias=P[f'{p}.fc1.b'], act=ACT_GELU)
            ops.split_rows_absmax(us, M, 4 * C, bits[slot_a(i, 'u'):])
            ops.gemm_split(us, P[f'{p}.fc2.w_split'], M, C, 4 * C, out=x, bias=P[f'{p}.fc2.b'], residual=x)
        try:
            check_split_overflow('index sampler (x8 calibration)')
        except SplitOverflowError:
            # the checkpoint's activations leave fp16's range on the calibration states: no x8 scales exist for it; the
            # fp16-plane path will flag the same overflow in a run and fall back to exact fp32 (a property of the
            # checkpoint, like everything else decided here)
            import warnings
            warnings.warn('text2human_amd: x8 calibration met activations beyond fp16\'s range; this checkpoint runs '
                          'without the x8 operands')
            self.x8 = False
            return
        mx_bits = bits.cpu().numpy().view(np.float32).astype(np.float64)
        sw, sa, mx = {}, {}, {}
        for i in range(L):
            for lin in lins:
                sw[i, lin] = ops.x8_scale_for(mx_bits[slot_w(i, lin)], 256.0)
                if f'{nm}.{i}.{lin}.w_x8' not in P.t:
                    P.t[f'{nm}.{i}.{lin}.w_x8'] = ops.split_rows_x8(P[f'{nm}.{i}.{lin}.w'], sw[i, lin])
            for role in roles:
                mx[i, role] = float(mx_bits[slot_a(i, role)])
                sa[i, role] = ops.x8_scale_for(mx[i, role])
        if ops.split_overflow(reset=True):
            raise _lib.T2HError('x8 weight packing overflowed')  # (cannot happen: every scale is derived from its matrix's maximum)
        self._x8 = {'w': sw, 'a': sa, 'act_max': mx}

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
    """Default noise source: torch's global generator of the GPU, consumed exactly like the
    reference does (rand per step; one full [B*T, 1024] exponential_ per ACTIVE head, the draw
    Categorical.sample() makes inside multinomial) so identical seeds give identical tokens.

    Normally no draw is materialised: build_schedule reproduces the `rand` draws on the device
    (t2h_unmask_schedule), the sampling tail computes the elements of the exponential_ tensors it
    needs, and the generator is advanced to where the reference's would stand.  That emulation of
    ATen's Philox kernels is verified against the installed torch at first use (`emulation_ok`);
    if it ever disagrees (another torch / ROCm build) the draws below are made for real."""

    _checked = {}

    def __init__(self, device):
        self.device = device

    def uniform(self, step, shape):
        return torch.rand(shape, device=self.device)

    def exponential(self, step, head, shape):
        return torch.empty(shape, device=self.device).exponential_(1.0)

    def generator(self):
        dev = torch.device(self.device)
        index = dev.index if dev.index is not None else torch.cuda.current_device()  # (initialises CUDA)
        return torch.cuda.default_generators[index], index

    def emulation_ok(self, n, n_class):
        """True iff t2h_philox_uniform_f32 / t2h_philox_exponential_f32 reproduce this torch build's
        `rand(n)` / `empty(n * n_class).exponential_()` bit for bit (generator state restored)."""
        gen, index = self.generator()
        key = (index, int(n), int(n_class))
        if key not in self._checked:
            state = gen.get_state()
            try:
                seed, off = gen.initial_seed(), gen.get_offset()
                want_u = torch.rand(n, device=self.device)
                off_e = gen.get_offset()
                want_e = torch.empty(n * n_class, device=self.device).exponential_(1.0)
                got_u = ops.philox_uniform(seed, off, n, self.device)
                got_e = ops.philox_exponential(seed, off_e, n * n_class, self.device)
                ok = bool(torch.equal(want_u, got_u)) and bool(torch.equal(want_e, got_e))
            finally:
                gen.set_state(state)
            if not ok:
                import warnings
                warnings.warn('text2human_amd: the in-kernel reproduction of torch\'s Philox draws does not match '
                              'this torch / ROCm build; falling back to explicit torch draws (slower, same tokens)')
            self._checked[key] = ok
        return self._checked[key]


class SplitOverflowError(_lib.T2HError):
    """An activation of the split-precision sampler left fp16's range."""


class X8RangeError(SplitOverflowError):
    """An activation of the x8 path left the range its calibrated scale gives the 8-bit planes (the fp16 planes are
    fine): the caller re-runs on the fp16-plane kernels."""


def check_split_overflow(what='sampler', knob='T2H_SPLIT_GEMM'):
    """Raises if a split-row producer flagged |x| >= 65504 since the last check (the fp16
    planes would hold inf / NaN) -- SplitOverflowError, the caller reruns with the exact-fp32 kernels -- or a value
    outside an x8 tensor's 8-bit range -- X8RangeError, the caller reruns on the fp16 planes."""
    bits = ops.split_overflow_bits(reset=True)
    if bits & 1 == 0 and bits & 2:
        raise X8RangeError(f'{what}: an activation exceeded 14x its calibration maximum, outside the range of the x8 '
                           'format\'s 8-bit planes (include/t2h_hip.h); the result is invalid.  Rerun with T2H_X8=0 '
                           '(fp16 planes).')
    if bits:
        raise SplitOverflowError(
            f'{what}: an activation reached |x| >= 65504, outside the range of the 2 x fp16 split '
            f'representation (include/t2h_hip.h); the result is invalid.  Rerun with {knob}=0 '
            '(exact-fp32 kernels).')


class SampleSchedule:
    """Device-resident schedule of one sample_tokens run (see schedule.py): rows of round r =
    rows[start[r]:start[r + 1]], each with its own noise reference."""

    def __init__(self, rows, start, round_steps, noise_kind, seed=None, offsets=None, expo_rows=None, slots=None,
                 host=None, perm=None, rng_rows=None, active=None):
        self.rows, self.start, self.round_steps = rows, start, round_steps
        self.host = host  # (row order, per-row offsets[, rng rows]) as numpy, for the padded tables of the graph path
        # finished samples leave the batch (schedule.leave_order): perm[new position] = original sample, rows /
        # round_steps are in the NEW order, rng_rows = the original row of every listed row (its noise), active[r] =
        # samples still running in round r (a prefix of the reordered batch)
        self.perm, self.rng_rows, self.active = perm, rng_rows, active
        self.noise_kind, self.seed, self.offsets, self.expo_rows, self.slots = noise_kind, seed, offsets, expo_rows, slots
        self.n_rounds = len(start) - 1
        self.max_rows = int(max(int(start[r + 1] - start[r]) for r in range(self.n_rounds))) if self.n_rounds else 0

    def row_noise(self, lo, hi):
        if self.noise_kind == 'philox':
            return ('philox', self.seed, self.offsets[lo:hi], self.rng_rows[lo:hi] if self.rng_rows is not None else None)
        return ('explicit', self.expo_rows, self.slots[lo:hi])


def build_schedule(tex_tok, sample_steps, n_books, n_class, noise, compact=True, shrink=False):
    """The unmasking schedule + RNG bookkeeping of one sample_fn call (schedule.py), consuming
    `noise` exactly as the reference's loop would (models/sample_model.py:279-306).  shrink (compact rounds on the
    device generator only): the samples are reordered so that finished ones leave the batch (SampleSchedule.perm)."""
    B, T = tex_tok.shape
    dev = tex_tok.device
    n = B * T
    tex_flat = tex_tok.reshape(-1).contiguous()
    tex_host = tex_flat.cpu().numpy()
    # the reference indexes texture_emb / head_list with these ids and raises on a bad one
    # (transformer_arch.py:262, sample_model.py:300-317); here they index device arrays -- checked on the
    # host copy the schedule needs anyway (no extra device read)
    if tex_host.size and (int(tex_host.min()) < 0 or int(tex_host.max()) >= n_books):
        raise _lib.T2HError(f'texture ids must lie in [0, {n_books}), got [{int(tex_host.min())}, {int(tex_host.max())}]')
    if isinstance(noise, TorchDeviceNoise) and noise.emulation_ok(n, n_class):
        # one launch reproduces every `rand` draw; one host read fetches the whole schedule
        gen, _ = noise.generator()
        seed, off0 = gen.initial_seed(), gen.get_offset()
        step_dev, mask_dev, rand_inc, expo_inc = ops.unmask_schedule(seed, off0, tex_flat, sample_steps, n_books, n_class)
        step_of_row = step_dev.cpu().numpy()
        head_mask = mask_dev.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        _, expo_off, final = schedule.draw_offsets(head_mask, sample_steps, off0, rand_inc, expo_inc, n_books)
        gen.set_offset(final)  # where the reference's generator stands after its loop
        perm = rng_rows = rng_dev = active = None
        if shrink and compact and B > 1:
            perm, _ = schedule.leave_order(step_of_row, B, T)
            if (perm == np.arange(B)).all():
                perm = None
        if perm is not None:
            orig_row = (perm[:, None] * T + np.arange(T)[None, :]).reshape(-1)  # original row of every reordered row
            step_of_row, tex_host = step_of_row[orig_row], tex_host[orig_row]
        order, start, round_steps = schedule.group_rounds(step_of_row, B, T, compact)
        offs = expo_off[step_of_row[order], tex_host[order]]
        assert (offs >= 0).all()
        if shrink and compact:
            active = (round_steps > 0).sum(1).astype(np.int64)
            assert all((round_steps[r, :active[r]] > 0).all() for r in range(len(active)))  # a prefix of the batch
        if perm is not None:
            rng_rows = orig_row[order].astype(np.int32)
            rng_dev = torch.from_numpy(rng_rows).to(dev)
        return SampleSchedule(torch.from_numpy(order.astype(np.int32)).to(dev), start, round_steps, 'philox', seed=seed,
                              offsets=torch.from_numpy(offs).to(dev), host=(order, offs, rng_rows), perm=perm,
                              rng_rows=rng_dev, active=active)
    # explicit draws (tests replaying CPU noise; the emulation fallback): the reference's own loop
    # order, with the rows each head needs copied out of its full draw
    unmasked = torch.zeros(n, dtype=torch.uint8, device=dev)
    changes = torch.zeros(n, dtype=torch.uint8, device=dev)
    counts = torch.zeros(n_books + 1, dtype=torch.int32, device=dev)
    rows = torch.empty(n, dtype=torch.int32, device=dev)
    step_of_row = np.zeros(n, dtype=np.int32)
    slot_of_row = np.zeros(n, dtype=np.int32)
    expo_rows = torch.empty((n, n_class), dtype=torch.float32, device=dev)
    fill = 0
    for t in range(sample_steps, 0, -1):
        rnd = noise.uniform(t, (B, T)).to(dev, torch.float32).contiguous()
        counts.zero_()
        ops.unmask_step(rnd, t, unmasked, changes, tex_flat, counts, rows, n_books)
        c = counts.cpu().numpy()
        if c[n_books] == 0:
            continue
        rows_t = np.sort(rows[:int(c[n_books])].cpu().numpy())
        step_of_row[rows_t] = t
        for h in np.nonzero(c[:n_books])[0]:  # ascending: the reference's draw order
            e = noise.exponential(t, int(h), (n, n_class)).to(dev, torch.float32).contiguous()
            rh = rows_t[tex_host[rows_t] == h]
            ops.gather_rows(e, torch.from_numpy(rh.astype(np.int32)).to(dev), len(rh), out=expo_rows[fill:fill + len(rh)])
            slot_of_row[rh] = np.arange(fill, fill + len(rh), dtype=np.int32)
            fill += len(rh)
    assert fill == n, (fill, n)
    order, start, round_steps = schedule.group_rounds(step_of_row, B, T, compact)
    return SampleSchedule(torch.from_numpy(order.astype(np.int32)).to(dev), start, round_steps, 'explicit',
                          expo_rows=expo_rows, slots=torch.from_numpy(slot_of_row[order]).to(dev))


class RoundGraph:
    """One sampling round of sample_tokens as a captured hipGraph (torch.cuda.CUDAGraph is the stream /
    graph plumbing; every node is a kernel of libt2h_hip.so).  A round is the same launch sequence with
    the same arguments every time: t2h_schedule_advance moves the round's rows / generator offsets from
    the run's padded tables into fixed staging buffers, the 24 layers and the sampling tail work on
    persistent buffers, the seed is read from device memory.  Captured once per (batch, padded rows per
    round -- a power of two --, temperature) for every count of running samples, before the first run
    (prepare), replayed for every round of every run."""

    def __init__(self, net, B, T, steps, maxr, n_books, n_class, temp, mask_id, dev):
        i64 = lambda *s: torch.empty(s, dtype=torch.int64, device=dev)
        i32 = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        # (a weak reference: net._graphs owns this object -- a strong one would make a cycle that only the cyclic
        # collector frees, at a moment of ITS choosing, e.g. in the middle of another graph's capture)
        self._net = weakref.ref(net)
        self.maxr, self.temp, self.mask_id = maxr, float(temp), mask_id
        self.x_t, self.out, self.segm, self.tex = i64(B, T), i64(n_books, B * T), i64(B, T), i64(B, T)
        self.rows_tbl, self.offs_tbl, self.rng_tbl = i32(steps, maxr), i64(steps, maxr), i32(steps, maxr)
        self.cur_rows, self.cur_offs, self.cur_rng = i32(maxr), i64(maxr), i32(maxr)
        self.round_ctr, self.seed = i32(1), i64(1)
        self.logits_ws = torch.empty((maxr, n_class), dtype=torch.float32, device=dev)
        self.stream = torch.cuda.Stream(device=dev)
        self.B = B
        self.graphs = {}  # samples still running (a prefix of the batch, schedule.leave_order) -> captured round
        self.pool = torch.cuda.graph_pool_handle()  # one memory pool for all of them (a capture allocates nothing big)

    def prepare(self, counts):
        """Captures the round for every count of running samples in `counts` (all of 1..B when finished samples
        leave the batch), once, before the first run: which counts a run meets depends on its seed, and a capture
        in the middle of a later run would be an eager round + a device-wide synchronisation inside the sampling
        loop.  One eager round on a valid dummy state first (it
        sizes every lazily allocated workspace); a capture itself executes nothing.  Call on self.stream."""
        self.x_t.fill_(self.mask_id)
        self.out.fill_(-1)
        for t in (self.segm, self.tex, self.rows_tbl, self.offs_tbl, self.rng_tbl, self.round_ctr, self.seed):
            t.zero_()
        self.body(self.B)
        self.stream.synchronize()
        for k in sorted(counts, reverse=True):
            if k not in self.graphs:
                self.capture(k)

    def body(self, k):
        """One round on the first k samples of the batch (the others have no step left)."""
        net = self._net()
        P, nm = net.P, net.name
        ops.schedule_advance(self.rows_tbl, self.offs_tbl, self.rng_tbl, self.round_ctr, self.cur_rows, self.cur_offs,
                             self.cur_rng, self.maxr)
        net.hidden(self.x_t, self.segm, self.tex, defer_tail=True, active=k)
        hidden, compact = net.finish_tail(self.cur_rows, self.maxr)
        ops.sample_heads(hidden, P[f'{nm}.ln_f.g'], P[f'{nm}.ln_f.b'], P[f'{nm}.heads'], {}, self.cur_rows, self.maxr,
                         self.tex.view(-1), self.temp, self.x_t, self.out,
                         row_noise=('philox', self.seed, self.cur_offs, self.cur_rng), hidden_compact=compact,
                         logits_ws=self.logits_ws)

    def capture(self, k):
        g = torch.cuda.CUDAGraph()
        # thread_local: another thread of this process -- RCCL's watchdog, a data loader -- may call into
        # the HIP runtime while this thread captures.  No cyclic garbage collection inside the capture:
        # a collected tensor / graph / event would be released through the runtime in the middle of it.
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(g, pool=self.pool, stream=self.stream, capture_error_mode='thread_local'):
                self.body(k)
        finally:
            if gc_was_on:
                gc.enable()
        self.graphs[k] = g

    def run(self, sched, segm_tok, tex_tok):
        """All rounds of one run on this graph's stream; returns `out` (valid once the caller's stream
        has waited, which this does)."""
        order, offs, rng_rows = sched.host
        tables = schedule.RoundTables(order, offs, sched.start, self.maxr, per_row32=rng_rows if rng_rows is not None else order)
        R, rows_tbl, offs_tbl = tables.n_rounds, tables.rows_tbl, tables.val_tbl
        active = sched.active if sched.active is not None else np.full(R, self.B, dtype=np.int64)
        caller = torch.cuda.current_stream()
        self.stream.wait_stream(caller)
        with torch.cuda.stream(self.stream):
            ops.split_overflow(reset=True)  # (this stream's flag; allocated here, before any capture)
            counts = set(range(1, self.B + 1)) if sched.active is not None else {self.B}
            if not counts <= set(self.graphs):
                self.prepare(counts)
                ops.split_overflow(reset=True)
            self.rows_tbl[:R].copy_(torch.from_numpy(rows_tbl), non_blocking=False)
            self.offs_tbl[:R].copy_(torch.from_numpy(offs_tbl), non_blocking=False)
            self.rng_tbl[:R].copy_(torch.from_numpy(tables.aux32_tbl), non_blocking=False)
            self.segm.copy_(segm_tok)
            self.tex.copy_(tex_tok)
            self.x_t.fill_(self.mask_id)
            self.out.fill_(-1)
            self.round_ctr.zero_()
            self.seed.fill_(schedule.as_int64(sched.seed))  # (a uint64 seed >= 2^63 in its two's-complement form)
            for r in range(R):
                k = int(active[r])
                self.graphs[k].replay()
            check_split_overflow('index sampler')
        caller.wait_stream(self.stream)
        return self.out


def sample_tokens(net, segm_tok, tex_tok, sample_steps, mask_id, temp=1.0, noise=None,
                  n_books=18, step_hook=None, round_hook=None, compact=None):
    """BaseSampleModel.sample_fn (models/sample_model.py:256-328) on device.

    The unmasking schedule and every generator offset are computed up front (build_schedule: they
    do not depend on the transformer), so the loop below never reads anything back from the GPU.
    compact=True (default; T2H_COMPACT_ROUNDS=0 or a step_hook turn it off): every sample advances
    through its own active steps -- a (sample, step) pair that changes no token is never evaluated,
    its logits would not be read (sample_model.py:300-317) -- which takes ~13.5 % fewer transformer
    evaluations than the reference's synchronous loop and gives the same tokens up to fp32 summation-order
    ties (a round's row list differs between the two schedules, and the last layer's tail picks its GEMM
    tile by the row count; on the bench configuration the tokens of both schedules are identical,
    tests/test_gpu_edge_cases.py).
    compact=False: one round per step that changes a token, all samples at that step.
    Returns int64 [18, B*T] (-1 off-texture).

    Test hooks, called after each round and allowed to overwrite x_t in place (teacher forcing):
    round_hook(r, steps, x_t, out) with steps[b] = the step sample b just took (0 = idle);
    step_hook(t, x_t, out) (compact=False only; steps that change no token are skipped)."""
    P, nm = net.P, net.name
    B, T = segm_tok.shape
    dev = segm_tok.device
    split = bool(getattr(net, 'split', False))
    if split:
        ops.split_overflow(reset=True)  # a flag left by an earlier stage is not this run's
    if compact is None:
        compact = step_hook is None and os.environ.get('T2H_COMPACT_ROUNDS', '1') != '0'
    if compact and step_hook is not None:
        raise ValueError('step_hook needs compact=False (samples are at different steps in a compact round)')
    noise = noise or TorchDeviceNoise(dev)
    n = B * T
    n_class = P[f'{nm}.heads'].shape[1]
    tex_flat = tex_tok.reshape(-1).contiguous()
    # finished samples leave the batch (T2H_SHRINK_BATCH=0 opts out; hooks see the batch in its own order)
    shrink = (compact and step_hook is None and round_hook is None and os.environ.get('T2H_SHRINK_BATCH', '1') != '0')
    if split and getattr(net, 'x8', False):
        net.ensure_x8()  # (a no-op after the model's load-time calibration; bare SamplerNets of tests / tools)
        ops.split_overflow(reset=True)
    sched = build_schedule(tex_tok, sample_steps, n_books, n_class, noise, compact, shrink)
    defer = bool(split and getattr(net, 'split_mha', False) and os.environ.get('T2H_TRIM_LAST_LAYER', '1') != '0')
    # (only the deferred-tail form of hidden() runs on the prefix of running samples; every other form evaluates the
    # whole batch each round, and the counts say so)
    net.last_stats = schedule.stats(sched.round_steps, sample_steps, sched.active if defer else None)
    if sched.perm is not None:
        perm_t = torch.from_numpy(sched.perm).to(dev)
        segm_tok, tex_tok = segm_tok[perm_t].contiguous(), tex_tok[perm_t].contiguous()
        tex_flat = tex_tok.reshape(-1)

    def in_batch_order(out):  # [n_books, n] in the schedule's sample order -> the caller's
        if sched.perm is None:
            return out
        inv = torch.from_numpy(np.argsort(sched.perm)).to(dev)
        return out.view(out.shape[0], B, T)[:, inv].reshape(out.shape[0], n).contiguous()

    # Default (T2H_GRAPH=0 opts out): every round is ONE replay of a captured launch sequence (RoundGraph)
    # instead of ~180 launches from this thread.  The GPU work is the same; what changes is the host side --
    # with one process per GPU, eight Python threads issuing ~60 k launches/s each is the scaling risk of
    # SURVEY.md 8(e).  Test hooks, explicit noise sources and the per-launch event sampling of bench.py's
    # profiling leg need individual launches and take the loop below.
    net.last_launch_mode = 'eager'
    if (os.environ.get('T2H_GRAPH', '1') != '0' and defer and sched.noise_kind == 'philox' and step_hook is None
            and round_hook is None and 0 < sched.max_rows <= net.TRIM_MAX_ROWS and ops.gemm_profile_active() is False):
        net.last_launch_mode = 'graph'
        # padded rows per round: a power of two >= 16 (at most five graph sets per batch size, whatever the seeds)
        maxr = min(net.TRIM_MAX_ROWS, max(16, 1 << (int(sched.max_rows) - 1).bit_length()))
        net._buffers(n, net.desc['C'], dev)  # (a change of batch size drops the graphs of the old buffers)
        key = (B, T, sample_steps, maxr, float(temp), int(mask_id), n_books, bool(getattr(net, 'x8', False)))
        if key not in net._graphs:
            net._graphs[key] = RoundGraph(net, B, T, sample_steps, maxr, n_books, n_class, temp, mask_id, dev)
        return in_batch_order(net._graphs[key].run(sched, segm_tok, tex_tok).clone())
    x_t = torch.full((B, T), mask_id, dtype=torch.int64, device=dev)
    out = torch.full((n_books, n), -1, dtype=torch.int64, device=dev)
    logits_ws = torch.empty((max(sched.max_rows, 1), n_class), dtype=torch.float32, device=dev)
    for r in range(sched.n_rounds):
        lo, hi = int(sched.start[r]), int(sched.start[r + 1])
        k = int(sched.active[r]) if sched.active is not None else None
        hidden = (net.hidden(x_t, segm_tok, tex_tok, defer_tail=True, active=k) if defer
                  else net.hidden(x_t, segm_tok, tex_tok))
        rows_r = sched.rows[lo:hi]
        hidden_compact = False
        if defer:
            hidden, hidden_compact = net.finish_tail(rows_r, hi - lo)
        ops.sample_heads(hidden, P[f'{nm}.ln_f.g'], P[f'{nm}.ln_f.b'], P[f'{nm}.heads'], {}, rows_r, hi - lo, tex_flat,
                         temp, x_t, out, row_noise=sched.row_noise(lo, hi), hidden_compact=hidden_compact,
                         logits_ws=logits_ws)
        if round_hook is not None:
            round_hook(r, sched.round_steps[r], x_t, out)
        if step_hook is not None:
            step_hook(int(sched.round_steps[r].max()), x_t, out)
    if split:
        check_split_overflow('index sampler')
    return in_batch_order(out)


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
