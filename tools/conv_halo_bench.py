"""The decoders' 3x3 convolutions, per shape of one decode of B images: t2h_gn_apply_split_f32 + t2h_conv_split_f32 (the
two-kernel path) against t2h_conv_halo_f32 (GroupNorm apply + swish + split folded into the halo staging), HIP-event
time per call; then the whole refine + decode stage with T2H_HALO_CONV = 0 / 1 / 2.  GPU only.

    python tools/conv_halo_bench.py [batch=8] [upscale=0]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2human_amd import _lib, defaults, engine, ops, options, synthetic  # noqa: E402
from text2human_amd.models import SampleFromParsingModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
up = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
opt = options.dict_to_nonedict(defaults.sample_from_parsing())
sds = synthetic.make_state_dicts(opt, seed=1234)
batch = synthetic.parsing_batch(B, seed=2021)
g = torch.Generator().manual_seed(3)


def make_model():
    model = SampleFromParsingModel(opt, state_dicts=sds)
    model.feed_data(batch)
    return model


def timed(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


os.environ['T2H_HALO_CONV'] = '0'
model = make_model()
tex = model._texture_tokens(model.texture_mask)
idx = torch.randint(0, 1024, (B, 512), generator=g).cuda()
top = [torch.where(tex == h, idx, torch.full_like(idx, -1)) for h in range(18)]

# ---- the shapes of one decode
shapes = {}
real = engine.VQGANStack._norm_conv3x3


def spy(self, x, norm, conv, n_img, h, w, mode='same', residual=None):
    P = self.P
    cin, cout = P[f'{conv}.w'].shape[1] // 9, P[f'{conv}.w'].shape[0]
    if self._ws(f'{conv}.w', h * w * (4 if mode == 'up' else 1), mode) is not None:
        key = (n_img, h, w, cin, cout, mode, norm is not None, residual is not None)
        shapes[key] = shapes.get(key, 0) + 1
    return real(self, x, norm, conv, n_img, h, w, mode=mode, residual=residual)


engine.VQGANStack._norm_conv3x3 = spy
model.decode_indices(top, want_u8=True, upscale=up)
engine.VQGANStack._norm_conv3x3 = real
torch.cuda.synchronize()

print(f'# B = {B}, upscale = {up}: 3x3 convolutions on the split path, per shape (us per call)')
print(f'{"n  h x w  cin->cout mode pro res":40s} {"calls":>5s} {"apply":>8s} {"conv":>8s} {"old":>8s} {"halo":>8s} {"old/halo":>8s} '
      f'{"halo TF":>8s} {"max|d|":>9s}')
tot_old = tot_new = tot_best = 0.0
os.environ['T2H_HALO_CONV'] = '2'
ONLY = int(os.environ.get('CONV_HALO_BENCH_SHAPES', '0'))  # > 0: the N largest shapes only, no stage timing (PMC passes)
for key, calls in sorted(shapes.items(), key=lambda kv: -kv[0][1] * kv[0][2] * kv[0][3] * kv[0][4])[:ONLY or None]:
    n_img, h, w, cin, cout, mode, pro, res = key
    ho, wo = (2 * h, 2 * w) if mode == 'up' else (h, w)
    x = torch.randn(n_img * h * w, cin, device='cuda')
    wt = torch.randn(cout, 9 * cin, device='cuda') * 0.05
    ws = ops.split_rows(wt)
    bias = torch.randn(cout, device='cuda')
    r = torch.randn(n_img * ho * wo, cout, device='cuda') if res else None
    sc = torch.rand(n_img, cin, device='cuda') + 0.5 if pro else None
    sh = torch.randn(n_img, cin, device='cuda') * 0.3 if pro else None
    xs = ops.split_rows_empty(n_img * h * w, cin, x.device)
    out_o = torch.empty(n_img * ho * wo, cout, device='cuda')
    out_n = torch.empty_like(out_o)

    def f_apply():
        ops.gn_apply_split(x, sc, sh, rows_per_img=h * w, act=ops.PRO_SWISH if pro else ops.PRO_NONE, out=xs)

    def f_conv():
        ops.conv_split(xs, ws, n_img, h, w, cin, cout, out=out_o, bias=bias, residual=r, mode=mode, gn_stats=True)

    def f_halo():
        ops.conv_halo(x, ws, n_img, h, w, cin, cout, out=out_n, bias=bias, residual=r, mode=mode,
                      pro=(sc, sh) if pro else None, gn_stats=True)

    ta, tc = timed(f_apply) * 1e3, timed(f_conv) * 1e3
    if not ops.conv_halo_ok(n_img, ho, wo, cin, cout, mode):
        print(f'{n_img} {h}x{w} {cin}->{cout} {mode} {int(pro)} {int(res)}'.ljust(40) + f' {calls:5d} {ta:8.1f} {tc:8.1f} {ta + tc:8.1f}   (not served)')
        tot_old += calls * (ta + tc); tot_new += calls * (ta + tc); tot_best += calls * (ta + tc)
        continue
    lib = _lib.load()
    lib.t2h_conv_halo_force_variant(0)
    th0 = timed(f_halo) * 1e3
    lib.t2h_conv_halo_force_variant(1)
    th = timed(f_halo) * 1e3
    tf = 2.0 * n_img * ho * wo * cout * 9 * cin / th * 1e-6
    d = (out_n - out_o).abs().max().item()
    tot_old += calls * (ta + tc); tot_new += calls * th; tot_best += calls * min(ta + tc, th)
    print(f'{n_img} {h}x{w} {cin}->{cout} {mode} {int(pro)} {int(res)}'.ljust(40)
          + f' {calls:5d} {ta:8.1f} {tc:8.1f} {ta + tc:8.1f} {th:8.1f} {(ta + tc) / th:8.2f} {tf:8.1f} {d:9.2e}   (variant 0: {th0:8.1f})')
print(f'sum over the decode: two-kernel path {tot_old / 1e3:.2f} ms, halo everywhere {tot_new / 1e3:.2f} ms, '
      f'the faster of the two per shape {tot_best / 1e3:.2f} ms')

# ---- the stage
if ONLY:
    sys.exit(0)
imgs = {}
for knob in ('0', '1', '2'):
    os.environ['T2H_HALO_CONV'] = knob
    m = make_model()
    ms = timed(lambda: imgs.__setitem__(knob, m.decode_indices(top, want_u8=True, upscale=up)[0]), iters=3, warm=2)
    print(f'T2H_HALO_CONV={knob}: {ms:7.2f} ms per batch of {B} = {ms / B:6.3f} ms/image')
print(f'max |halo(1) - two-kernel| over the images: {(imgs["1"] - imgs["0"]).abs().max().item():.2e}; '
      f'(2): {(imgs["2"] - imgs["0"]).abs().max().item():.2e}')
assert not ops.split_overflow(reset=True)
