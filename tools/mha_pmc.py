"""The product attention kernel alone (B = 8: mha_split_pipe_kernel<2>, x8 output) for rocprofv3 --pmc passes
(tools/run_pmc_mha.sh).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T, H, C = 512, 8, 512
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * T, 3 * C, generator=g) * 1.2).cuda()
qk_s = ops.split_rows(qkv)
vt = ops.vt_empty(B, H, T, 'cuda')
vt.copy_((torch.randn(vt.shape, generator=g) * 0.5).half().view(torch.int16))
ys = ops.split_rows_empty(B * T, C, 'cuda')
for _ in range(12):
    ops.mha_split_x8(qk_s, 3 * C, vt, B, T, H, ys, 8.0)
torch.cuda.synchronize()
