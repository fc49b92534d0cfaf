"""Per-step pieces of the sampling loop outside the 24 transformer layers (B=8): embedding sum,
unmask step, RNG draws, the sampling tail in its one-launch and two-launch forms.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import ops  # noqa: E402

DEV = 'cuda'
B, T, C, K, H = 8, 512, 512, 1024, 18
n = B * T
g = torch.Generator().manual_seed(0)
hidden = (torch.randn(n, C, generator=g) * 2).to(DEV)
gam, bet = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
w = (torch.randn(H, K, C, generator=g) * 0.05).to(DEV)
tex = torch.randint(0, H, (n, ), generator=g).to(DEV)
rows = torch.randperm(n, generator=g)[:16].to(torch.int32).to(DEV)
active = sorted(set(tex[rows.long()].tolist()))
expo = {h: torch.empty(n, K, device=DEV).exponential_(1.0) for h in active}
x_t = torch.zeros(n, dtype=torch.int64, device=DEV)
out = torch.full((H, n), -1, dtype=torch.int64, device=DEV)


def timeit(fn, iters=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print(f'{len(active)} active heads, 16 changed rows')
print(f'sample_heads one launch : {timeit(lambda: ops.sample_heads(hidden, gam, bet, w, expo, rows, 16, tex, 1.0, x_t, out, split=False)):6.1f} us')
print(f'sample_heads two launch : {timeit(lambda: ops.sample_heads(hidden, gam, bet, w, expo, rows, 16, tex, 1.0, x_t, out)):6.1f} us')
print(f'exponential_ [4096,1024]: {timeit(lambda: torch.empty(n, K, device=DEV).exponential_(1.0)):6.1f} us per active head')
print(f'rand [8,512]            : {timeit(lambda: torch.rand(B, T, device=DEV)):6.1f} us')
