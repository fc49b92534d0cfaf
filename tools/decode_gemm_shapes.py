"""Which exact-fp32 GEMM / conv launches does one refine + decode of 8 images make?  (shapes of every t2h_gemm_f32 call,
with the Python frame that made it).  GPU only; diagnostic."""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2human_amd import defaults, ops, options, synthetic  # noqa: E402
from text2human_amd.models import SampleFromParsingModel  # noqa: E402

opt = options.dict_to_nonedict(defaults.sample_from_parsing())
model = SampleFromParsingModel(opt, state_dicts=synthetic.make_state_dicts(opt, seed=1234))
batch = synthetic.parsing_batch(8, seed=2021)
model.texture_mask = batch['texture_mask'].to(model.device)
model.batch_size = 8
tex = model._texture_tokens(model.texture_mask)
idx = torch.randint(0, 1024, (8, 512), generator=torch.Generator().manual_seed(3)).cuda()
top = [torch.where(tex == h, idx, torch.full_like(idx, -1)) for h in range(18)]
model.decode_indices(top, want_u8=True)
seen = collections.Counter()
real = ops._launch_gemm


def spy(g, what):
    fr = [f for f in traceback.extract_stack()[:-1] if f.filename.endswith(('engine.py', 'sample_model.py'))][-1]
    seen[(what, g.M, g.N, g.K, g.a_mode, g.batch, g.ksplit, f'{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}')] += 1
    return real(g, what)


ops._launch_gemm = spy
model.decode_indices(top, want_u8=True)
torch.cuda.synchronize()
for k, n in sorted(seen.items(), key=lambda kv: -kv[0][1] * kv[0][2] * kv[0][3]):
    print(n, k)
