"""What is a kernel's time worth?  Times the B = 8 sampler (256 steps, graph replay) with the LayerNorm launches, the
attention launches, or the proj GEMMs simply SKIPPED (results are garbage; the launches that remain are unchanged):
if the chip were limited by each kernel's own duration, skipping a kernel that takes x % of the step would save x %;
under a power limit that averages over many kernels, skipping a low-power kernel saves less than its share (the
kernels after it lose the headroom it left them), skipping a high-power one saves more.  GPU only.

    python tools/what_if_skip.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import defaults, ops, options, synthetic  # noqa: E402
from text2human_amd.models import SampleFromParsingModel  # noqa: E402

opt = options.dict_to_nonedict(defaults.sample_from_parsing())
model = SampleFromParsingModel(opt, state_dicts=synthetic.make_state_dicts(opt, seed=1234))
model.feed_data({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synthetic.parsing_batch(8, seed=2021).items()})
real = dict(ln=ops.layernorm_split, mha=ops.mha_split, gemm=ops.gemm_split)


def run(skip):
    ops.layernorm_split = (lambda x, g, b, out, eps=1e-5: out) if 'ln' in skip else real['ln']
    ops.mha_split = (lambda *a, **k: k.get('out_split')) if 'mha' in skip else real['mha']
    if 'proj' in skip:  # the N = K = 512 GEMMs with a residual
        ops.gemm_split = lambda a, w, M, N, K, **k: (k.get('out') if (N == 512 and K == 512 and M > 256) else real['gemm'](a, w, M, N, K, **k))
    else:
        ops.gemm_split = real['gemm']
    model.sampler_fn._graphs = {}
    ts = []
    for i in range(3):
        options.set_random_seed(2021)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            model.sample_fn(temp=1, sample_steps=256)
        except Exception as e:  # (garbage activations may trip the overflow check: the timing is still valid)
            if 'overflow' not in str(e).lower() and '65504' not in str(e):
                raise
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1000.0 * min(ts[1:])


base = run(())
print(f'sampler, nothing skipped: {base:7.1f} ms')
for skip, share in ((('ln', ), 'layernorm 7.1 % of the step by kernel time'), (('mha', ), 'attention 13.8 %'),
                    (('proj', ), 'proj GEMM ~7 %'), (('ln', 'mha'), 'both')):
    t = run(skip)
    print(f'  skipping {"+".join(skip):7s}: {t:7.1f} ms  ({100 * (base - t) / base:5.1f} % saved; {share})')
print(f'sampler, nothing skipped (again): {run(()):7.1f} ms')
