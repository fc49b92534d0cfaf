"""Does the split GEMM slow down when its weights are not cache-resident?  Cycles
through `ncopy` separate copies of W (24 layers x 12.6 MB of split weights exceed the
256 MB Infinity Cache together with the activations) and optionally of A.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import ops  # noqa: E402


def run(m, n, k, ncopy_w, ncopy_a, iters=96):
    g = torch.Generator().manual_seed(0)
    a = [ops.split_rows((torch.randn(m, k, generator=g) * 1.3).cuda()) for _ in range(ncopy_a)]
    w = [ops.split_rows((torch.randn(n, k, generator=g) * 0.05).cuda()) for _ in range(ncopy_w)]
    out = torch.empty(m, n, device='cuda')
    for i in range(8):
        ops.gemm_split(a[i % ncopy_a], w[i % ncopy_w], m, n, k, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        ops.gemm_split(a[i % ncopy_a], w[i % ncopy_w], m, n, k, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, (m, n, k) in {'qkv': (4096, 1536, 512), 'proj': (4096, 512, 512), 'fc1': (4096, 2048, 512),
                        'fc2': (4096, 512, 2048)}.items():
    print(f'{name:5s} warm W, warm A {run(m, n, k, 1, 1):6.1f} us | 96 Ws (cold W) {run(m, n, k, 96, 1):6.1f} us | '
          f'cold W + 8 As {run(m, n, k, 96, 8):6.1f} us')
