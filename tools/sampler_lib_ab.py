"""The B = 8 sampler (256 steps, seed 2021, graph replay: what bench.py times as its sampler stage) on ONE build of the
library per process, for A/B runs of kernel variants built by tools/build_ln_xcd.sh:

    T2H_AB_LIB=tools/_tb/libt2h_<variant>.so python tools/sampler_lib_ab.py [batch=8]

prints the best and median of 4 timed runs and a checksum of the sampled tokens (variants that only move work
between XCDs or change a store's cache policy must give the same tokens).  GPU only."""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2human_amd import _lib  # noqa: E402
if os.environ.get('T2H_AB_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['T2H_AB_LIB'])
from text2human_amd import defaults, options, synthetic  # noqa: E402
from text2human_amd.models import SampleFromParsingModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
opt = options.dict_to_nonedict(defaults.sample_from_parsing())
model = SampleFromParsingModel(opt, state_dicts=synthetic.make_state_dicts(opt, seed=1234))
model.feed_data({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synthetic.parsing_batch(B, seed=2021).items()})
ts, tok = [], None
for i in range(5):
    options.set_random_seed(2021)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tok = model.sample_fn(temp=1, sample_steps=256)
    torch.cuda.synchronize()
    if i:
        ts.append(1000.0 * (time.perf_counter() - t0))
cs = int(sum(int((t.long() * (j + 1)).sum()) for j, t in enumerate(tok)))
print(f'{os.environ.get("T2H_AB_LIB", "product library"):36s} B={B}: sampler best {min(ts):7.1f} ms, median {statistics.median(ts):7.1f} ms; token checksum {cs}')
