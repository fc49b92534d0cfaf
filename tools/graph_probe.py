"""Probe: host enqueue time of one transformer forward vs its GPU time, and the same
forward replayed from a captured hipGraph with 1 / 2 / 4 batch-slice streams.  GPU only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import defaults, engine, options, synthetic, weights  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
opt = options.dict_to_nonedict(defaults.sample_from_parsing())
sd = synthetic.make_state_dicts(opt, seed=1234)['sampler']
P = weights.Params('cuda')
desc = weights.pack_transformer(P, sd, 'tf')
g = torch.Generator().manual_seed(0)
idx = torch.randint(0, 18433, (B, 512), generator=g).cuda()
seg = torch.randint(0, 1024, (B, 512), generator=g).cuda()
tex = torch.randint(0, 18, (B, 512), generator=g).cuda()
for ns in (1, 2, 4):
    net = engine.SamplerNet(P, desc, 8, 'tf', split=True, split_mha=True, n_streams=ns)
    for _ in range(3):
        net.hidden(idx, seg, tex)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        net.hidden(idx, seg, tex)
    t_host = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    t_gpu = (time.perf_counter() - t0) / 10
    line = f'streams {ns}: eager host enqueue {1e3 * t_host:.2f} ms, wall {1e3 * t_gpu:.2f} ms'
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            net.hidden(idx, seg, tex)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = net.hidden(idx, seg, tex)
        ref = net.hidden(idx, seg, tex).clone()
        graph.replay()
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        t0 = time.perf_counter()
        for _ in range(20):
            graph.replay()
        torch.cuda.synchronize()
        line += f' | graph replay {1e3 * (time.perf_counter() - t0) / 20:.2f} ms (bitwise same: {same})'
    except Exception as e:  # noqa: BLE001
        line += f' | graph capture failed: {type(e).__name__}: {str(e)[:120]}'
    print(line, flush=True)
