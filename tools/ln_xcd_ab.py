"""Does the LayerNorm find its input in the L2 of the XCD that wrote it?  (Round 6.)  The proj / fc2 GEMM's tile mapping
gives XCD x the x-th eighth of the row blocks; the LayerNorm that follows reads those rows from workgroups spread
round-robin over all XCDs (FETCH_SIZE = the whole tensor, profiles/r06_pmc_summary.md).  Since round 6 the LayerNorm's
workgroup -> rows mapping follows the GEMM's (T2H_LN_XCD, the default); tools/build_ln_xcd.sh builds the round-robin
twin (libt2h_lnrr.so) and other variants of csrc/norm.hip.  One library per process:

    T2H_AB_LIB=tools/_tb/libt2h_lnrr.so python tools/ln_xcd_ab.py [batch=8]

times a chain of [proj GEMM (x8 operands, residual in place) -> LayerNorm (x8 out)] x 48, HIP events.  GPU only."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2human_amd import _lib  # noqa: E402
if os.environ.get('T2H_AB_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['T2H_AB_LIB'])
from text2human_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M, C = 512 * B, 512
g = torch.Generator().manual_seed(0)
x = torch.randn(M, C, generator=g).cuda()
w = (torch.randn(C, C, generator=g) * 0.02).cuda()
bias, gam, bet = torch.zeros(C).cuda(), torch.ones(C).cuda(), torch.zeros(C).cuda()
sa, sw = ops.x8_scale_for(8.0), ops.x8_scale_for(float(w.abs().max()), 256.0)
w8 = ops.split_rows_x8(w, sw)
h8 = ops.split_rows_empty(M, C, 'cuda')
ops.layernorm_x8(x, gam, bet, h8, sa)


def chain(n):
    for _ in range(n):
        ops.gemm_split(h8, w8, M, C, C, out=x, bias=bias, residual=x, x8=(sa, sw))
        ops.layernorm_x8(x, gam, bet, h8, sa)


chain(4)
torch.cuda.synchronize()
ts = []
for _ in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    chain(48)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 48 * 1e3)
print(f'{os.environ.get("T2H_AB_LIB", "product library"):34s} B={B}: GEMM + LayerNorm pair median {statistics.median(ts):6.2f} us [min {min(ts):.2f}, max {max(ts):.2f}]; '
      f'overflow word {ops.split_overflow(reset=True)}; checksum {float(h8.view(torch.int16).float().abs().sum()):.6e}')
