"""Refine + hierarchical decode of B images (default 8) with the decoders' convolutions on the
split-precision kernel vs the exact-fp32 kernel: HIP-event time per image, image difference.  GPU only.

    python tools/decode_bench.py [batch=8] [upscale=0]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import defaults, options, synthetic  # noqa: E402
from text2human_amd.models import SampleFromParsingModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
up = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
opt = options.dict_to_nonedict(defaults.sample_from_parsing())
sds = synthetic.make_state_dicts(opt, seed=1234)
batch = synthetic.parsing_batch(B, seed=2021)
g = torch.Generator().manual_seed(3)
top = None
imgs = {}
from text2human_amd import _lib  # noqa: E402
for name, env, tile in (('split', '1', 0), ('s128', '1', 128), ('s256', '1', 256), ('fp32', '0', 0)):
    os.environ['T2H_SPLIT_CONV'] = env
    _lib.load().t2h_conv_split_force_tile(tile)
    model = SampleFromParsingModel(opt, state_dicts=sds)
    model.feed_data(batch)
    if top is None:
        tex = model._texture_tokens(model.texture_mask)
        idx = torch.randint(0, 1024, (B, 512), generator=g).cuda()
        top = [torch.where(tex == h, idx, torch.full_like(idx, -1)) for h in range(18)]
    for _ in range(2):
        img, _ = model.decode_indices(top, want_u8=True, upscale=up)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        img, _ = model.decode_indices(top, want_u8=True, upscale=up)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    imgs[name] = img
    gf = (2380.0 if up else 562.88) + 2.19
    print(f'{name:5s}: {ms:7.2f} ms per batch of {B} = {ms / B:6.2f} ms/image = {gf * B / ms:6.1f} TFLOP/s (fp32-equivalent)')
_lib.load().t2h_conv_split_force_tile(0)
print(f'max |s256 - s128|: {(imgs["s256"] - imgs["s128"]).abs().max().item():.2e}')
print(f'max |split - fp32| over the images: {(imgs["split"] - imgs["fp32"]).abs().max().item():.2e}')
