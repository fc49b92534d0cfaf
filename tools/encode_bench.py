"""Encode side (image -> top / bottom tokens -> image) on one MI355X: parity counters
against the reference-made golden fixture and throughput at batch 8.  GPU only."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2human_amd import defaults, options, synthetic  # noqa: E402
from text2human_amd.models import VQGANTextureAwareSpatialHierarchyInferenceModel as M  # noqa: E402

opt = options.dict_to_nonedict(defaults.sample_from_parsing())
sds = synthetic.make_state_dicts(opt, seed=1234, encode=True)
model = M(opt, state_dicts=sds)
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'encode_b1.npz'))
# the seeded inputs of oracle/make_golden.py golden_inputs('encode') (tools must not import oracle/)
gen0 = torch.Generator().manual_seed(103)
gi = dict(image=torch.rand(1, 3, 512, 256, generator=gen0) * 2 - 1,
          texture_mask=synthetic.parsing_batch(1, seed=2021)['texture_mask'])
model.feed_data(dict(image=gi['image'], texture_mask=gi['texture_mask']))
top = torch.stack([t.view(1, 32, 16) for t in model.top_indices_list]).cpu().numpy()
bot = torch.stack(model.gt_indices_list).cpu().numpy()
rec = model.index_to_image(model.gt_indices_list, model.texture_mask)
print('top index mismatches', int((top != g['top_indices']).any(0).sum()), '/ 512 | bottom',
      int((bot != g['bot_indices']).any(0).sum()), '/ 512 | quant_t err',
      (model.quant_t[0, ::8, ::2, ::2].cpu() - torch.from_numpy(g['quant_t_sample'])).abs().max().item(),
      '| reconstruction err', (rec[0, :, ::4, ::4].cpu() - torch.from_numpy(g['rec_sample'])).abs().max().item())
B = 8
gen = torch.Generator().manual_seed(1)
img = (torch.rand(B, 3, 512, 256, generator=gen) * 2 - 1).cuda()
mask = synthetic.parsing_batch(B, seed=2021)['texture_mask'].cuda()
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.feed_data(dict(image=img, texture_mask=mask))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    model.index_to_image(model.gt_indices_list, mask)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f'B={B}: encode (top + bottom) {1e3 * (t1 - t0):.1f} ms = {B / (t1 - t0):.1f} images/s | '
      f'decode {1e3 * (t2 - t1):.1f} ms | round trip {B / (t2 - t0):.1f} images/s')
