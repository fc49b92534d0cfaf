"""Split-precision path (-m gpu): two-fp16-plane operands (x = h + l / 2048),
three-product GEMM / attention on the fp16 matrix cores, split-row producers.
Accuracy bar = the exact-fp32 path's (summation-order-level error against an fp64
reference)."""
import pytest
import torch
import torch.nn.functional as F

from text2human_amd import _lib, engine, ops, synthetic, weights

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _unsplit(s, rows, c):
    return ops.unsplit_rows_host(s, rows, c)


def test_split_rows_carry_22_bits_and_match_the_host_pack():
    x = torch.cat([_rnd(300, 512, seed=1) * 3, _rnd(300, 512, seed=2) * 1e-3, _rnd(8, 512, seed=3) * 1e4,
                   _rnd(8, 512, seed=4) * 1e-6])
    s = ops.split_rows(x.to(DEV))
    assert torch.equal(s.cpu(), ops.pack_split_rows_host(x).view_as(s.cpu()))
    back = _unsplit(s.cpu(), x.shape[0], 512)
    # two 11-bit planes; below 2^-14 the hi plane is subnormal and the bound is absolute
    assert ((back - x).abs() <= x.abs() * 2.0**-21 + 2.0**-36).all()


@pytest.mark.parametrize('cfg', [0, 1, 2, 3, 5, 6, 8, 9, 10])
@pytest.mark.parametrize('M,N,K', [(4096, 512, 512), (1024, 1536, 512), (512, 512, 2048), (130, 96, 64),
                                   (40, 32, 32)])
def test_gemm_split(cfg, M, N, K):
    if cfg == 6 and K % 64:
        pytest.skip('the in-block K split needs K % 64 == 0')
    a, w, b, r = _rnd(M, K, seed=4) * 1.3, _rnd(N, K, seed=5, scale=0.08), _rnd(N, seed=6), _rnd(M, N, seed=7)
    ref = a.double() @ w.double().t() + b.double()
    a_s, w_s = ops.split_rows(a.to(DEV)), ops.pack_split_rows_host(w).to(DEV)
    lib = _lib.load()
    lib.t2h_gemm_split_force_config(cfg)
    try:
        out = torch.empty(M, N, device=DEV)
        ops.gemm_split(a_s, w_s, M, N, K, out=out, bias=b.to(DEV), residual=r.to(DEV))
        err = (out.cpu().double() - (ref + r.double())).abs()
        assert (err <= 2e-5 + 2e-5 * ref.abs()).all(), err.max().item()
        if N % 32 == 0:
            o_s = ops.split_rows_empty(M, N, DEV)
            ops.gemm_split(a_s, w_s, M, N, K, out_split=o_s, bias=b.to(DEV), act=ops.ACT_GELU)
            got = _unsplit(o_s.cpu(), M, N).double()
            err = (got - F.gelu(ref)).abs()
            assert (err <= 2e-5 + 2e-5 * ref.abs()).all(), err.max().item()
    finally:
        lib.t2h_gemm_split_force_config(-1)


def test_split_producers_are_bitwise_the_split_of_the_fp32_result():
    x, g, b = _rnd(777, 512, seed=8) * 2 + 0.1, _rnd(512, seed=9) * 0.1 + 1, _rnd(512, seed=10) * 0.1
    ln = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV))
    ln_s = ops.layernorm_split(x.to(DEV), g.to(DEV), b.to(DEV), ops.split_rows_empty(777, 512, DEV))
    assert torch.equal(ln_s, ops.split_rows(ln))
    qkv = (_rnd(2 * 512, 1536, seed=11) * 1.2).to(DEV)
    y = ops.mha_noncausal(qkv, 2, 512, 8)
    y_s = ops.mha_noncausal_split(qkv, 2, 512, 8, ops.split_rows_empty(1024, 512, DEV))
    assert torch.equal(y_s, ops.split_rows(y))


def test_sampler_net_split_matches_oracle_and_fp32_path():
    from oracle import torch_ref as R
    sd = synthetic.fill(synthetic.transformer_schema(18432, 1024, 18, 512, 4, 512, 18), seed=12)
    P = weights.Params(DEV)
    desc = weights.pack_transformer(P, sd, 'tf')
    gen = torch.Generator().manual_seed(13)
    idx = torch.randint(0, 18433, (2, 512), generator=gen)
    seg = torch.randint(0, 1024, (2, 512), generator=gen)
    tex = torch.randint(0, 18, (2, 512), generator=gen)
    args = (idx.to(DEV), seg.to(DEV), tex.to(DEV))
    a = engine.SamplerNet(P, desc, 8, 'tf', split=False).hidden(*args).clone().cpu()
    b = engine.SamplerNet(P, desc, 8, 'tf', split=True, x8=False).hidden(*args).clone().cpu()
    with torch.no_grad():
        ref = R.transformer_hidden(idx, seg, tex, sd)
    ln = lambda t: F.layer_norm(t.view(2, 512, 512), (512, ), sd['ln_f.weight'], sd['ln_f.bias'], 1e-5)
    ea, eb = (ln(a) - ref).abs().max().item(), (ln(b) - ref).abs().max().item()
    assert ea < 1e-4 and eb < 1e-4, (ea, eb)
    assert eb < 3 * ea + 1e-6, f'split path error {eb:.2e} vs fp32 path {ea:.2e}'


def _pack_vt_host(v, B, T, H):
    """fp32 v [B*T, H*64] -> Vt [B][H][2][64][T] (int16 view) in the kernel's key order"""
    vt = v.view(B, T, H, 64).permute(0, 2, 3, 1).contiguous()          # [B, H, 64, T]
    pl = torch.stack(ops.split_planes_host(vt)).permute(1, 2, 0, 3, 4).contiguous()  # [B, H, 2, 64, T]
    out = torch.empty_like(pl)
    out[..., ops.vt_key_positions(T)] = pl
    return out.view(torch.int16)


@pytest.mark.parametrize('cfg', [-1, 0, 2, 3, 6, 8, 10])
def test_gemm_split_routes_value_columns_to_transposed_planes(cfg):
    B, T, H, C = 2, 512, 8, 512
    M = B * T
    a, w, bias = _rnd(M, C, seed=20) * 1.1, _rnd(3 * C, C, seed=21, scale=0.06), _rnd(3 * C, seed=22)
    a_s, w_s = ops.split_rows(a.to(DEV)), ops.pack_split_rows_host(w).to(DEV)
    lib = _lib.load()
    lib.t2h_gemm_split_force_config(cfg)
    try:
        full = torch.empty(M, 3 * C, device=DEV)
        ops.gemm_split(a_s, w_s, M, 3 * C, C, out=full, bias=bias.to(DEV))
        qk_s = ops.split_rows_empty(M, 3 * C, DEV)
        qk_s.zero_()
        vt = ops.vt_empty(B, H, T, DEV)
        ops.gemm_split(a_s, w_s, M, 3 * C, C, out_split=qk_s, bias=bias.to(DEV), vt=vt, vt_col0=2 * C, vt_T=T)
    finally:
        lib.t2h_gemm_split_force_config(-1)
    full = full.cpu()
    want = ops.pack_split_rows_host(full)                      # [M, 3C/32, 2, 32]
    got = qk_s.cpu()
    assert torch.equal(got[:, :2 * C // 32], want[:, :2 * C // 32])
    assert (got[:, 2 * C // 32:] == 0).all(), 'value columns must not be written as split rows'
    assert torch.equal(vt.cpu(), _pack_vt_host(full[:, 2 * C:].contiguous(), B, T, H))


@pytest.mark.parametrize('form,B', [(2, 2), (1, 2), (1, 3), (0, 16)])
def test_mha_split_matches_fp64_reference_as_well_as_the_fp32_kernel(form, B):
    """both forms of the kernel (t2h_mha_split_force_form: 2 = 128-query workgroups, key halves merged; 1 = 256-query
    workgroups, every wave over all keys; 0 = the dispatcher, which takes the all-keys form at B = 16) against fp64"""
    T, H, C = 512, 8, 512
    qkv = _rnd(B * T, 3 * C, seed=23) * 1.5
    qkv[5, C:C + 64] *= 6.0       # a spiked key: forces the online-softmax rescale
    q, k, v = [t.view(B, T, H, 64).transpose(1, 2).double() for t in qkv.split(C, dim=1)]
    att = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1)
    ref = (att @ v).transpose(1, 2).reshape(B * T, C)
    y32 = ops.mha_noncausal(qkv.to(DEV), B, T, H).cpu().double()
    qk_s = ops.split_rows(qkv.to(DEV))
    vt = _pack_vt_host(qkv[:, 2 * C:].contiguous(), B, T, H).to(DEV)
    y = torch.empty(B * T, C, device=DEV)
    lib = _lib.load()
    lib.t2h_mha_split_force_form(form)
    try:
        ops.mha_split(qk_s, 3 * C, vt, B, T, H, out=y)
        ys = ops.mha_split(qk_s, 3 * C, vt, B, T, H, out_split=ops.split_rows_empty(B * T, C, DEV))
    finally:
        lib.t2h_mha_split_force_form(0)
    assert torch.equal(ys, ops.split_rows(y))
    if form == 0:  # B = 16: the dispatcher's choice is the all-keys form, bit for bit
        lib.t2h_mha_split_force_form(1)
        try:
            y1 = torch.empty_like(y)
            ops.mha_split(qk_s, 3 * C, vt, B, T, H, out=y1)
        finally:
            lib.t2h_mha_split_force_form(0)
        assert torch.equal(y, y1)
    e32, es = (y32 - ref).abs().max().item(), (y.cpu().double() - ref).abs().max().item()
    assert es < 5e-6 + 2 * e32, f'split attention error {es:.2e} vs fp32 kernel {e32:.2e}'


def test_sampler_net_split_mha_on_and_off_agree_with_oracle():
    from oracle import torch_ref as R
    sd = synthetic.fill(synthetic.transformer_schema(18432, 1024, 18, 512, 4, 512, 18), seed=12)
    P = weights.Params(DEV)
    desc = weights.pack_transformer(P, sd, 'tf')
    gen = torch.Generator().manual_seed(14)
    idx = torch.randint(0, 18433, (3, 512), generator=gen)
    seg = torch.randint(0, 1024, (3, 512), generator=gen)
    tex = torch.randint(0, 18, (3, 512), generator=gen)
    args = (idx.to(DEV), seg.to(DEV), tex.to(DEV))
    a = engine.SamplerNet(P, desc, 8, 'tf', split=True, split_mha=False).hidden(*args).clone().cpu()
    b = engine.SamplerNet(P, desc, 8, 'tf', split=True, split_mha=True, x8=False).hidden(*args).clone().cpu()
    with torch.no_grad():
        ref = R.transformer_hidden(idx, seg, tex, sd)
    ln = lambda t: F.layer_norm(t.view(3, 512, 512), (512, ), sd['ln_f.weight'], sd['ln_f.bias'], 1e-5)
    ea, eb = (ln(a) - ref).abs().max().item(), (ln(b) - ref).abs().max().item()
    assert ea < 1e-4 and eb < 1e-4, (ea, eb)
    assert eb < 3 * ea + 1e-6, f'split attention path error {eb:.2e} vs fp32 attention {ea:.2e}'


def test_last_layer_tail_on_the_changed_rows_only_matches_the_full_evaluation():
    """engine.SamplerNet.hidden(defer_tail=True) + finish_tail(rows): the last layer's proj / LayerNorm /
    fc1 / fc2 evaluated for a compact list of rows = those rows of the full forward (row-wise operators;
    the small-M GEMM picks another tile configuration, so equality is to summation order)."""
    sd = synthetic.fill(synthetic.transformer_schema(18432, 1024, 18, 512, 3, 512, 18), seed=12)
    P = weights.Params(DEV)
    desc = weights.pack_transformer(P, sd, 'tf')
    gen = torch.Generator().manual_seed(21)
    idx = torch.randint(0, 18433, (2, 512), generator=gen)
    seg = torch.randint(0, 1024, (2, 512), generator=gen)
    tex = torch.randint(0, 18, (2, 512), generator=gen)
    args = (idx.to(DEV), seg.to(DEV), tex.to(DEV))
    net = engine.SamplerNet(P, desc, 8, 'tf', split=True, x8=False)
    full = net.hidden(*args).clone()
    rows = torch.randperm(1024, generator=gen)[:37].to(torch.int32).to(DEV)
    net.hidden(*args, defer_tail=True)
    got, compact = net.finish_tail(rows, 37)
    assert compact and got.shape == (37, 512)
    assert (got - full[rows.long()]).abs().max().item() < 2e-5
    net.hidden(*args, defer_tail=True)                     # too many rows: the tail runs on all of them
    got, compact = net.finish_tail(rows, 1000)
    assert not compact and torch.equal(got, full)


@pytest.mark.parametrize('M,N,K', [(1, 16, 32), (5, 48, 64), (64, 512, 2048), (17, 2048, 512)])
def test_few_rows_kernel_is_what_small_problems_get(M, N, K):
    """M <= 64 goes to the 16x16x32 kernel (config 9) on its own; every epilogue option it serves."""
    a, w, b, r = _rnd(M, K, seed=50) * 1.1, _rnd(N, K, seed=51, scale=0.07), _rnd(N, seed=52), _rnd(M, N, seed=53)
    ref = a.double() @ w.double().t() + b.double()
    a_s, w_s = ops.split_rows(a.to(DEV)), ops.pack_split_rows_host(w).to(DEV)
    lib = _lib.load()
    out = torch.empty(M, N, device=DEV)
    ops.gemm_split(a_s, w_s, M, N, K, out=out, bias=b.to(DEV), residual=r.to(DEV))
    lib.t2h_gemm_split_force_config(9)
    try:
        forced = torch.empty(M, N, device=DEV)
        ops.gemm_split(a_s, w_s, M, N, K, out=forced, bias=b.to(DEV), residual=r.to(DEV))
    finally:
        lib.t2h_gemm_split_force_config(-1)
    assert torch.equal(out, forced)                                   # automatic choice == config 9
    err = (out.cpu().double() - (ref + r.double())).abs()
    assert (err <= 1e-5 + 1e-5 * ref.abs()).all(), err.max().item()
    if N % 32 == 0:
        o_s = ops.split_rows_empty(M, N, DEV)
        ops.gemm_split(a_s, w_s, M, N, K, out_split=o_s, bias=b.to(DEV), act=ops.ACT_GELU)
        err = (_unsplit(o_s.cpu(), M, N).double() - F.gelu(ref)).abs()
        assert (err <= 1e-5 + 1e-5 * ref.abs()).all(), err.max().item()


@pytest.mark.parametrize('N,K', [(512, 512), (1536, 512), (2048, 512), (512, 2048)])
def test_gemm_split_at_the_b32_shapes_on_the_automatic_dispatch(N, K):
    """M = 16384 = BASELINE.json configs[2] / configs[3]'s 32 images per GPU: the dispatcher gives every
    sampler Linear but proj the 256x128 ping-pong tile there (proj: 128x128), a different choice than at
    the M = 4096 of the headline configuration.  fp64 reference, the sampler's four shapes, with the
    epilogues each of them uses (residual / GELU + split rows / q|k|v routing)."""
    M = 16384
    a, w, b = _rnd(M, K, seed=70) * 1.2, _rnd(N, K, seed=71, scale=0.07), _rnd(N, seed=72)
    ref = a.to(DEV).double() @ w.to(DEV).double().t() + b.to(DEV).double()
    a_s, w_s = ops.split_rows(a.to(DEV)), ops.pack_split_rows_host(w).to(DEV)
    if N == 512:      # proj / fc2: fp32 out + residual, in place like the residual stream
        x = (_rnd(M, N, seed=73)).to(DEV)
        want = ref + x.double()
        ops.gemm_split(a_s, w_s, M, N, K, out=x, bias=b.to(DEV), residual=x)
        err = (x.double() - want).abs()
        assert (err <= 2e-5 + 2e-5 * ref.abs()).all(), err.max().item()
    elif N == 2048:   # fc1: GELU, split rows out
        o_s = ops.split_rows_empty(M, N, DEV)
        ops.gemm_split(a_s, w_s, M, N, K, out_split=o_s, bias=b.to(DEV), act=ops.ACT_GELU)
        err = (_unsplit(o_s, M, N).double() - F.gelu(ref).cpu()).abs()
        assert (err <= 2e-5 + 2e-5 * ref.cpu().abs()).all(), err.max().item()
    else:             # q|k|v: q, k as split rows, v as transposed planes
        B, T, H, C = M // 512, 512, 8, 512
        qk_s, vt = ops.split_rows_empty(M, N, DEV), ops.vt_empty(B, H, T, DEV)
        ops.gemm_split(a_s, w_s, M, N, K, out_split=qk_s, bias=b.to(DEV), vt=vt, vt_col0=2 * C, vt_T=T)
        got = _unsplit(qk_s, M, N)[:, :2 * C].double()
        err = (got - ref[:, :2 * C].cpu()).abs()
        assert (err <= 2e-5 + 2e-5 * ref[:, :2 * C].cpu().abs()).all(), err.max().item()
        full = torch.empty(M, N, device=DEV)
        ops.gemm_split(a_s, w_s, M, N, K, out=full, bias=b.to(DEV))
        assert torch.equal(vt.cpu(), _pack_vt_host(full[:, 2 * C:].contiguous().cpu(), B, T, H))


def test_gelu_epilogue_rational_erf_against_exact_erf():
    """fc1's GELU epilogue uses a rational erf (gemm_split.hip: erf_rational, max abs error 4e-7 on [-4, 4], clamped
    outside) instead of erff.  Bounded here directly: a dense grid of pre-activations over [-9, 9] (fc1's range and
    beyond the clamp) goes through the product epilogue via an identity weight matrix -- the three-product GEMM of a
    22-bit operand with the identity is exact -- and is compared with nn.GELU()'s definition in fp64:
    |gelu_hip - 0.5 x (1 + erf(x / sqrt 2))| <= 0.5 |x| * 4.5e-7 + the 2^-22 rounding of the split-row output."""
    import math
    M, K = 4096, 32
    x = torch.linspace(-9.0, 9.0, M * K, dtype=torch.float64).view(M, K)
    x = _unsplit(ops.pack_split_rows_host(x.float()), M, K)             # the 22-bit values the kernel sees
    a_s, w_s = ops.split_rows(x.to(DEV)), ops.pack_split_rows_host(torch.eye(K)).to(DEV)
    o_s = ops.split_rows_empty(M, K, DEV)
    ops.gemm_split(a_s, w_s, M, K, K, out_split=o_s, act=ops.ACT_GELU)
    o32 = torch.empty(M, K, device=DEV)
    ops.gemm_split(a_s, w_s, M, K, K, out=o32, act=ops.ACT_GELU)
    xd = x.double()
    ref = 0.5 * xd * (1.0 + torch.erf(xd / math.sqrt(2.0)))
    for got in (_unsplit(o_s.cpu(), M, K).double(), o32.cpu().double()):
        err = (got - ref).abs()
        bound = 0.5 * xd.abs() * 4.5e-7 + ref.abs() * 2.0**-21 + 1e-9
        assert (err <= bound).all(), (err - bound).max().item()
    assert (ref - o32.cpu().double()).abs().max().item() > 0            # (not the identity function by accident)
