"""Workload for rocprofv3 --pmc passes on the split-precision GEMM: the four sampler
shapes at B=8, `iters` launches each with the config in argv[1] (-1 = auto).  GPU only.

    rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d out -o p -- python tools/gemm_split_pmc.py 0
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import _lib, ops  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else -1
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    M = 4096
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    for n, k in ((1536, 512), (512, 512), (2048, 512), (512, 2048)):
        a_s = ops.split_rows(torch.randn(M, k, generator=g).cuda())
        w_s = ops.split_rows((torch.randn(n, k, generator=g) * 0.05).cuda())
        out = torch.empty(M, n, device='cuda')
        lib.t2h_gemm_split_force_config(cfg)
        for _ in range(iters):
            ops.gemm_split(a_s, w_s, M, n, k, out=out)
        lib.t2h_gemm_split_force_config(-1)
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
