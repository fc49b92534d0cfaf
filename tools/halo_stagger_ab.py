"""A/B of the halo convolution's staggered start (round 6; t2h_conv_halo_set_stagger): the decoders' level-0 layer
(8 x 512 x 256 pixels, 128 -> 128 channels, GroupNorm tables + residual) and the whole refine + decode stage, with 0
(off) / 2 / 4 populations, interleaved blocks, HIP-event times; outputs compared bit for bit.  GPU only.

    python tools/halo_stagger_ab.py [batch=8] [upscale=0]
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import _lib, defaults, ops, options, synthetic  # noqa: E402
from text2human_amd.models import SampleFromParsingModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
up = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
lib = _lib.load()
POPS = (0, 2, 4, 8)


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, r


# ---- one level-0 layer
g = torch.Generator().manual_seed(0)
h, w, c = (1024, 512, 128) if up else (512, 256, 128)
n_img = min(B, 2) if up else B
x = torch.randn(n_img * h * w, c, generator=g).cuda()
wt = (torch.randn(c, c, 3, 3, generator=g) * 0.05)
from text2human_amd import weights  # noqa: E402
ws = ops.split_rows(weights.pack_conv3x3(wt).cuda())
bias = torch.randn(c, generator=g).cuda()
res = torch.randn(n_img * h * w, c, generator=g).cuda()
sc, sh = (torch.rand(n_img, c, generator=g) + 0.5).cuda(), (torch.randn(n_img, c, generator=g) * 0.1).cuda()
layer = lambda: ops.conv_halo(x, ws, n_img, h, w, c, c, bias=bias, residual=res, pro=(sc, sh), gn_stats=True)
times, outs = {p: [] for p in POPS}, {}
for blk in range(6):
    for p in (POPS if blk % 2 == 0 else POPS[::-1]):
        ops._HALO_STAGGER = p
        lib.t2h_conv_halo_set_stagger(p)
        if blk == 0:
            timed(layer, 2)
        t, o = timed(layer, 10)
        times[p].append(t)
        outs[p] = o
for p in POPS:
    print(f'level-0 layer ({n_img} x {h} x {w}, {c} -> {c}) stagger {p}: median {statistics.median(times[p]):7.1f} us  [min {min(times[p]):.1f}, max {max(times[p]):.1f}]'
          f' | equal to stagger 0: {torch.equal(outs[p], outs[0])}', flush=True)

# ---- the stage
opt = options.dict_to_nonedict(defaults.sample_from_parsing())
sds = synthetic.make_state_dicts(opt, seed=1234)
batch = synthetic.parsing_batch(B, seed=2021)
model = SampleFromParsingModel(opt, state_dicts=sds)
model.texture_mask = batch['texture_mask'].to(model.device)
model.batch_size = B
tex = model._texture_tokens(model.texture_mask)
idx = torch.randint(0, 1024, (B, 512), generator=torch.Generator().manual_seed(3)).cuda()
top = [torch.where(tex == hd, idx, torch.full_like(idx, -1)) for hd in range(18)]
stage = lambda: model.decode_indices(top, want_u8=True, upscale=up)[0]
times, outs = {p: [] for p in POPS}, {}
for blk in range(4):
    for p in (POPS if blk % 2 == 0 else POPS[::-1]):
        ops._HALO_STAGGER = p
        lib.t2h_conv_halo_set_stagger(p)
        if blk == 0:
            timed(stage, 1)
        t, o = timed(stage, 3)
        times[p].append(t / 1e3 / B)
        outs[p] = o
for p in POPS:
    print(f'refine + decode, B={B}{" 1024x512" if up else ""}, stagger {p}: median {statistics.median(times[p]):.3f} ms/image  [min {min(times[p]):.3f}, max {max(times[p]):.3f}]'
          f' | images equal to stagger 0: {torch.equal(outs[p], outs[0])}', flush=True)
